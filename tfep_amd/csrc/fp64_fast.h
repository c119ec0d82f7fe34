// fp64 reciprocal / division / square root / logarithm from the hardware's approximations (v_rcp_f64, v_rsq_f64: ~23
// good bits) + Newton steps, for arguments that are finite and -- where it matters -- positive: ~1 ulp, a fifth of the
// instructions of the IEEE-correct sequences the compiler emits for `/`, sqrt() and log().  The element-wise epilogues
// that use them (spline.h, moebius.h) run on waves that issue one vector instruction per ~8 cycles (one wave per SIMD
// beside 400 accumulator registers), so their length is the instruction count.
//   a / 0 and x / inf give NaN instead of inf / 0 here: every denominator of the callers is positive and finite for valid
//   inputs (softmax sums >= 1, bin widths >= min_bin_size, slopes >= min_slope), and a NaN input stays NaN either way.
#pragma once
#include <hip/hip_runtime.h>

namespace tfep {

__device__ __forceinline__ double fast_rcp64(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    return fma(fma(-b, r, 1.0), r, r);
}

// a / b: one Newton step on the reciprocal (46 bits), the quotient, one correction of the quotient with its exact residual
__device__ __forceinline__ double fast_div64(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

__device__ __forceinline__ double fast_sqrt64(double a) {         // 0 -> 0, inf -> inf, negative / NaN -> NaN
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    return (a == 0.0 || a == INFINITY) ? a : g;
}

__device__ __forceinline__ double fast_log64(double a) {          // a >= 0; log(0) = -inf, log(inf / NaN) = the argument
    // a = m 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(t), t = (m - 1) / (m + 1), |t| < 0.172: the odd series to
    // t^13 (next term < 2.2e-13 relative to t)
    double m = __builtin_amdgcn_frexp_mant(a);
    int e = __builtin_amdgcn_frexp_exp(a);
    const bool low = m < 0.70710678118654752;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double t = (m - 1.0) * fast_rcp64(m + 1.0), t2 = t * t;
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

}  // namespace tfep
