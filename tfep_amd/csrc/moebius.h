// Moebius map of one d-vector (reference transformers/moebius.py:374-478), shared by the stand-alone kernel
// (transformers.hip) and the fused inverse block kernel (inverse_block.hip) so that both give identical bits.
//
// log|det J| in closed form: with c = N/|x-w|^2 and the reflection R = I - 2 dd^T/|d|^2 (d = x - w) the unit-sphere
// Jacobian is c R, so log|det| = dim*log|c|; the general Jacobian is c R (I - xx^T/|x|^2) + y x^T/|x|^2 whose
// determinant is -(c^(dim-1)/|x|) * xhat . (R y)   (matrix-determinant lemma for a rank-(dim-1) + rank-1 sum).
#pragma once
#include <hip/hip_runtime.h>

namespace tfep {

constexpr int MOEBIUS_MAX_DIM = 8;

// fp64 reciprocal / square root / logarithm for this map's arguments (finite, and positive where it matters), from the
// hardware's approximations + Newton steps / a short series: ~1e-15 relative, a tenth of the instructions of the
// IEEE-correct library routines (the stand-alone kernel spent ~500 vector instructions per 2-vector on two square roots,
// two divisions and a logarithm: 0.86 ms per cfg4-ii layer against 0.2 ms of HBM traffic).
__device__ __forceinline__ double mb_rcp(double a) {
    double r = __builtin_amdgcn_rcp(a);
    r = fma(fma(-a, r, 1.0), r, r);
    return fma(fma(-a, r, 1.0), r, r);
}
__device__ __forceinline__ double mb_sqrt(double a) {         // a >= 0; 0 -> 0
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    return a > 0.0 ? g : 0.0;
}
__device__ __forceinline__ double mb_log(double a) {          // a >= 0; log(0) = -inf, log(inf / NaN) = the argument
    // a = m 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(t), t = (m - 1) / (m + 1), |t| < 0.172: the odd series to
    // t^13 (next term < 2.2e-13 relative to t)
    double m = __builtin_amdgcn_frexp_mant(a);
    int e = __builtin_amdgcn_frexp_exp(a);
    const bool low = m < 0.70710678118654752;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double t = (m - 1.0) * mb_rcp(m + 1.0), t2 = t * t;
    double p = 1.0 / 13.0;
    p = fma(p, t2, 1.0 / 11.0);
    p = fma(p, t2, 1.0 / 9.0);
    p = fma(p, t2, 1.0 / 7.0);
    p = fma(p, t2, 1.0 / 5.0);
    p = fma(p, t2, 1.0 / 3.0);
    p = fma(p * t2, t, t);
    const double out = fma((double)e, 0.69314718055994530942, p + p);
    return a > 0.0 ? (a < INFINITY ? out : a) : (a == 0.0 ? -INFINITY : a);
}

// xv: the point, wv: the (signed: the inverse is the map with -w, moebius.py:142-147) raw parameter vector, rescaled
// in place; yv: the image.  Returns log|det J|.
__device__ __forceinline__ double moebius_vector(const double (&xv)[MOEBIUS_MAX_DIM], double (&wv)[MOEBIUS_MAX_DIM], int dim,
                                                 float max_radius, int unit_sphere, double (&yv)[MOEBIUS_MAX_DIM]) {
    double dv[MOEBIUS_MAX_DIM];
    double wn2 = 0.0, xn2 = 0.0;
#pragma unroll
    for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
        if (i < dim) {
            wn2 += wv[i] * wv[i];
            xn2 += xv[i] * xv[i];
        }
    const double wn = mb_sqrt(wn2);
    double resc = (double)max_radius * mb_rcp(1.0 + wn);       // moebius.py:437-441
    double xn = 1.0;
    if (!unit_sphere) {
        xn = mb_sqrt(xn2);
        resc *= xn;
    }
    const double wns = resc * wn;
    const double numer = (unit_sphere ? 1.0 : xn2) - wns * wns;   // moebius.py:446-449
    double dn2 = 0.0;
#pragma unroll
    for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
        if (i < dim) {
            wv[i] *= resc;
            dv[i] = xv[i] - wv[i];
            dn2 += dv[i] * dv[i];
        }
    const double inv_dn2 = mb_rcp(dn2);
    const double c = numer * inv_dn2;
#pragma unroll
    for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
        if (i < dim) yv[i] = c * dv[i] - wv[i];                // moebius.py:452
    const double log_c = mb_log(fabs(c));
    if (unit_sphere) return dim * log_c;
    double dy = 0.0;   // d . y
    double xy = 0.0;   // x . y
    double xd = 0.0;
#pragma unroll
    for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
        if (i < dim) {
            dy += dv[i] * yv[i];
            xy += xv[i] * yv[i];
            xd += xv[i] * dv[i];
        }
    const double xRy = xy - 2.0 * xd * dy * inv_dn2;           // x . (R y)
    return (dim - 1) * log_c - mb_log(xn2) + mb_log(fabs(xRy));
}

}  // namespace tfep
