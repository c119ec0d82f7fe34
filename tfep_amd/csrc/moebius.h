// Moebius map of one d-vector (reference transformers/moebius.py:374-478), shared by the stand-alone kernel
// (transformers.hip) and the fused inverse block kernel (inverse_block.hip) so that both give identical bits.
//
// log|det J| in closed form: with c = N/|x-w|^2 and the reflection R = I - 2 dd^T/|d|^2 (d = x - w) the unit-sphere
// Jacobian is c R, so log|det| = dim*log|c|; the general Jacobian is c R (I - xx^T/|x|^2) + y x^T/|x|^2 whose
// determinant is -(c^(dim-1)/|x|) * xhat . (R y)   (matrix-determinant lemma for a rank-(dim-1) + rank-1 sum).
#pragma once
#include <hip/hip_runtime.h>

#include "fp64_fast.h"

namespace tfep {

constexpr int MOEBIUS_MAX_DIM = 8;

// (reciprocal, square root and logarithm: fp64_fast.h -- the stand-alone kernel spent ~500 vector instructions per 2-vector
// on two square roots, two divisions and a logarithm of the IEEE-correct kind: 0.86 ms per cfg4-ii layer against 0.2 ms of
// HBM traffic; 0.30 ms now)
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
    const double wn = fast_sqrt64(wn2);
    double resc = (double)max_radius * fast_rcp64(1.0 + wn);       // moebius.py:437-441
    double xn = 1.0;
    if (!unit_sphere) {
        xn = fast_sqrt64(xn2);
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
    const double inv_dn2 = fast_rcp64(dn2);
    const double c = numer * inv_dn2;
#pragma unroll
    for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
        if (i < dim) yv[i] = c * dv[i] - wv[i];                // moebius.py:452
    const double log_c = fast_log64(fabs(c));
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
    return (dim - 1) * log_c - fast_log64(xn2) + fast_log64(fabs(xRy));
}

}  // namespace tfep
