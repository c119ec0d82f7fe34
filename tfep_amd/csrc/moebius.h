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
    const double wn = sqrt(wn2), xn = sqrt(xn2);
    double resc = (double)max_radius / (1.0 + wn);             // moebius.py:437-441
    if (!unit_sphere) resc *= xn;
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
    const double c = numer / dn2;
    double dy = 0.0;   // d . y
    double xy = 0.0;   // x . y
#pragma unroll
    for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
        if (i < dim) {
            yv[i] = c * dv[i] - wv[i];                         // moebius.py:452
            dy += dv[i] * yv[i];
            xy += xv[i] * yv[i];
        }
    if (unit_sphere) return dim * log(fabs(c));
    double xd = 0.0;
#pragma unroll
    for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
        if (i < dim) xd += xv[i] * dv[i];
    const double xRy = xy - 2.0 * xd * dy / dn2;               // x . (R y)
    return (dim - 1) * log(fabs(c)) - 2.0 * log(xn) + log(fabs(xRy));
}

}  // namespace tfep
