// Rational-quadratic neural spline, one element per call, shared by the standalone
// transformer kernels and the fused MADE-output epilogue.
//
// Follows NeuralSplineTransformer (reference transformers/spline.py):
//   _get_parameters  :319-417   softmax widths/heights, softplus slopes, learnable bounds
//   _assign_bins     :567-650   bin = #{knots < x} - 1 with STRICT '>', linear tails
//   forward/inverse  :478-494 / :521-536, log-det :546-564, circular shift :236-238 / :257-259
//
// Numerics: the softmax of widths / heights and everything geometric (knot prefix sums,
// epsilon, the rational quadratic, the log-derivative sum) in fp64; softplus / log in fp32,
// so the result tracks the fp64 reference to ~1e-7 relative instead of inheriting the
// fp32 cancellation of (x - x_k)/w.  The two sentinel knots of the
// reference are algebraically a linear map with the boundary slope; it is evaluated directly.
#pragma once

#include <type_traits>

#include "common.h"
#include "fp64_fast.h"

namespace tfep {

struct SplineFlags {
    int K;
    bool circular, identity, learn_lower, learn_upper;
    float min_bin, min_slope;
    float slope_offset;   // log(exp(1 - min_slope) - 1), spline.py:414
};

// exp(x) for x <= 0 (softmax arguments after subtracting the maximum), fp64, relative error < 1e-11:
// one-constant range reduction + degree-9 Taylor + v_ldexp_f64.  About 4x fewer instructions than the
// library exp(): the 16 softmax exponentials dominate the fused epilogue.
__device__ inline double exp_nonpos(double x) {
    x = fmax(x, -700.0);
    const double n = rint(x * 1.4426950408889634);
    const double r = fma(n, -0.69314718055994530942, x);
    double p = 1.0 / 362880.0;
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

// exp(x) for |x| <= 700 by the same reduction and polynomial (relative error < 1e-11): the scale of a learnable domain
// in the fused epilogue, where the library exp() costs registers the 512-register GEMM wave does not have.
__device__ inline double exp_poly(double x) {
    x = fmin(fmax(x, -700.0), 700.0);
    const double n = rint(x * 1.4426950408889634);
    const double r = fma(n, -0.69314718055994530942, x);
    double p = 1.0 / 362880.0;
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}

__device__ inline float softplus_f(float z) {
    // torch softplus, beta = 1, threshold = 20 (spline.py:415).
    return z > 20.0f ? z : log1pf(expf(z));
}

// Raw-slope accessor position for knot j (0..K) in the per-feature parameter vector
// (spline.py:359-380).  Returns -1 when the raw slope is the constant 0 (identity boundary).
__device__ __host__ inline int spline_slope_param(int j, int K, bool circular, bool identity) {
    if (identity) {
        if (j == 0 || j == K) return -1;
        return 2 * K + j - 1;
    }
    if (circular && j == K) return 2 * K;
    return 2 * K + j;
}

__host__ __device__ inline int spline_n_params(int K, bool circular, bool identity, bool ll, bool lu) {
    int n = 3 * K + 1;          // spline.py:165-182
    if (ll) n += 1;
    if (lu) n += 1;
    if (identity) n -= circular ? 1 : 2;
    return n;
}

// Evaluate one element.
//   w[k], h[k]   raw (pre-softmax) widths / heights, k < K
//   sraw[j]      raw (pre-softplus) slopes of the K+1 knots, already expanded
//   last, last2  parameters P-1 and P-2 (shift / log-scale / domain shift), 0 if unused
//   x0,xf,y0,yf  the registered domain of the feature
// Returns the mapped value; *logd receives log(dy/dx) of the FORWARD map evaluated at the
// point (the caller negates for the inverse, spline.py:562-564).
template <int KMAX, bool INVERSE>
__device__ inline double rq_spline_element(const float (&w)[KMAX], const float (&h)[KMAX],
                                           const float (&sraw)[KMAX + 1], float last, float last2,
                                           const SplineFlags& f, float x0f, float xff, float y0f,
                                           float yff, float vin, double* logd) {
    const int K = f.K;
    const double mb = (double)f.min_bin;

    // ---- domain (spline.py:384-410)
    double x0 = x0f, y0 = y0f;
    double W = (double)xff - (double)x0f - K * mb;
    double H = (double)yff - (double)y0f - K * mb;
    if (f.learn_lower || f.learn_upper) {
        double scale = exp((double)last);
        W *= scale;
        H *= scale;
        if (f.learn_lower && f.learn_upper) {
            x0 += (double)last2;
            y0 += (double)last2;
        } else if (f.learn_lower) {
            x0 = (double)xff - W - K * mb;
            y0 = (double)yff - H - K * mb;
        }
    }

    double v = vin;
    if (f.circular && !INVERSE) {
        // spline.py:236-238 (xf fixed for circular splines)
        v = py_mod(v - x0 + (double)last, (double)xff - x0) + x0;
    }

    // ---- softmax normalisation (spline.py:394-395)
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) {
            mw = fmaxf(mw, w[k]);
            mh = fmaxf(mh, h[k]);
        }
    // exp in fp64: a 1-ulp fp32 softmax moves the knots by ~2e-7 of the domain, which a narrow
    // bin amplifies to > 1e-5 in the log-derivative.
    double ew[KMAX], eh[KMAX];
    double sw = 0.0, sh = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        ew[k] = 0.0;
        eh[k] = 0.0;
        if (k < K) {
            ew[k] = exp_nonpos((double)w[k] - (double)mw);
            eh[k] = exp_nonpos((double)h[k] - (double)mh);
            sw += ew[k];
            sh += eh[k];
        }
    }
    const double iw = W / sw, ih = H / sh;

    // ---- bin search: strict '>' (spline.py:622-625).  v <= first knot -> lower tail.
    double kx = x0, ky = y0;          // lower knot of the current bin
    double bw = 0.0, bh = 0.0;        // width / height of the found bin
    float rs0 = sraw[0], rs1 = sraw[0];
    bool found = false;
    const double ref0 = INVERSE ? y0 : x0;
    const bool lower_tail = !(v > ref0);
    float rs_last = sraw[0];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            const double wk = ew[k] * iw + mb;
            const double hk = eh[k] * ih + mb;
            if (!found) {
                const double upper = INVERSE ? (ky + hk) : (kx + wk);
                if (v > upper) {
                    kx += wk;
                    ky += hk;
                } else {
                    found = true;
                    bw = wk;
                    bh = hk;
                    rs0 = sraw[k];
                    rs1 = sraw[k + 1];
                }
            }
            if (k == K - 1) rs_last = sraw[k + 1];
        }
    }

    double out, ld;
    if (lower_tail || !found) {
        // Linear continuation with the boundary slope (spline.py:599-614; reference
        // NumPy oracle tests/nn/transformers/test_spline.py:83-87).
        const float rs = lower_tail ? sraw[0] : rs_last;
        const double d = (double)(softplus_f(rs + f.slope_offset) + f.min_slope);
        const double bx = lower_tail ? x0 : kx, by = lower_tail ? y0 : ky;
        out = INVERSE ? (bx + (v - by) / d) : (by + d * (v - bx));
        ld = (double)logf((float)d);
    } else {
        const double dk = (double)(softplus_f(rs0 + f.slope_offset) + f.min_slope);
        const double dk1 = (double)(softplus_f(rs1 + f.slope_offset) + f.min_slope);
        const double s = bh / bw;                                  // spline.py:643
        const double t = dk1 + dk - 2.0 * s;
        double eps;
        if (INVERSE) {                                             // spline.py:521-536
            const double ym = v - ky;
            const double a = bh * (s - dk) + ym * t;
            const double b = bh * dk - ym * t;
            const double c = -s * ym;
            eps = 2.0 * c / (-b - sqrt(b * b - 4.0 * a * c));
            out = eps * bw + kx;
        } else {                                                   // spline.py:485-494
            eps = (v - kx) / bw;
            const double e1 = eps * (1.0 - eps);
            out = ky + bh * (s * eps * eps + dk * e1) / (s + t * e1);
        }
        const double e1 = eps * (1.0 - eps);                       // spline.py:556-558
        const double om = 1.0 - eps;
        const double num = s * s * (dk1 * eps * eps + 2.0 * s * e1 + dk * om * om);
        const double den = s + t * e1;
        ld = (double)logf((float)(num / (den * den)));
    }

    if (f.circular && INVERSE) {
        // spline.py:257-259
        out = py_mod(out - x0 - (double)last, (double)xff - x0) + x0;
    }
    *logd = ld;
    return out;
}

// rq_spline_element<KMAX, true> without data-dependent branches: the same arithmetic in the same order (the same bits), the
// bin search and the tail / interior decision as selects.  The inverse block kernels evaluate one element per sample row on a
// wave of 16 rows: with branches every row that falls into another bin took its own copy of everything behind the search
// (the unrolled search duplicates its tail: 25 copies of the solve in the object code of the super-block kernel), and a
// wave walked through as many of them as its rows had distinct bins.
// FAST: the six fp64 divisions and the square root through fp64_fast.h (hardware seed + Newton, ~1 ulp in fp64 -- far below the
// fp32 rounding of the result; every denominator here is positive and finite for valid inputs): ~170 vector instructions less on the
// chain wave of the super-block inverse kernel, which is what that kernel waits for once its loader wave is out of the way.
// QUAD (KMAX = 8, K = 8 only): the caller evaluates the same element on the four lanes of a DPP quad (the 16-row layouts of the
// inverse block kernels: lanes 4 s .. 4 s + 3 hold sample row s; `quad_lane` = lane & 3).  The 16 softmax exponentials -- 18 fp64
// instructions each, the bulk of the evaluation -- are then dealt four to a lane and passed round the quad by DPP (32 moves): the
// same values on every lane, the same sums in the same order.
template <int J>
__device__ __forceinline__ double quad_broadcast(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), J * 0x55, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), J * 0x55, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int KMAX, bool FAST = false, bool QUAD = false>
__device__ __forceinline__ double rq_spline_inverse_selects(const float (&w)[KMAX], const float (&h)[KMAX],
                                                            const float (&sraw)[KMAX + 1], float last, float last2,
                                                            const SplineFlags& f, float x0f, float xff, float y0f, float yff,
                                                            float vin, double* logd, int quad_lane = 0) {
    const int K = f.K;
    const double mb = (double)f.min_bin;
    double x0 = x0f, y0 = y0f;
    double W = (double)xff - (double)x0f - K * mb;
    double H = (double)yff - (double)y0f - K * mb;
    if (f.learn_lower || f.learn_upper) {                                         // wave-uniform
        double scale = exp((double)last);
        W *= scale;
        H *= scale;
        if (f.learn_lower && f.learn_upper) {
            x0 += (double)last2;
            y0 += (double)last2;
        } else if (f.learn_lower) {
            x0 = (double)xff - W - K * mb;
            y0 = (double)yff - H - K * mb;
        }
    }
    const double v = vin;
    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) {                                                              // wave-uniform
            mw = fmaxf(mw, w[k]);
            mh = fmaxf(mh, h[k]);
        }
    double ew[KMAX], eh[KMAX];
    double sw = 0.0, sh = 0.0;
    if constexpr (QUAD && KMAX == 8) {
        // lane p of the quad: exp of widths 2 p, 2 p + 1 and heights 2 p, 2 p + 1
        float aw0 = w[0], aw1 = w[1], ah0 = h[0], ah1 = h[1];
#pragma unroll
        for (int p = 1; p < 4; ++p) {
            aw0 = quad_lane == p ? w[2 * p] : aw0;
            aw1 = quad_lane == p ? w[2 * p + 1] : aw1;
            ah0 = quad_lane == p ? h[2 * p] : ah0;
            ah1 = quad_lane == p ? h[2 * p + 1] : ah1;
        }
        const double e0 = exp_nonpos((double)aw0 - (double)mw), e1 = exp_nonpos((double)aw1 - (double)mw);
        const double e2 = exp_nonpos((double)ah0 - (double)mh), e3 = exp_nonpos((double)ah1 - (double)mh);
        ew[0] = quad_broadcast<0>(e0); ew[1] = quad_broadcast<0>(e1); eh[0] = quad_broadcast<0>(e2); eh[1] = quad_broadcast<0>(e3);
        ew[2] = quad_broadcast<1>(e0); ew[3] = quad_broadcast<1>(e1); eh[2] = quad_broadcast<1>(e2); eh[3] = quad_broadcast<1>(e3);
        ew[4] = quad_broadcast<2>(e0); ew[5] = quad_broadcast<2>(e1); eh[4] = quad_broadcast<2>(e2); eh[5] = quad_broadcast<2>(e3);
        ew[6] = quad_broadcast<3>(e0); ew[7] = quad_broadcast<3>(e1); eh[6] = quad_broadcast<3>(e2); eh[7] = quad_broadcast<3>(e3);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sw += ew[k];
            sh += eh[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            ew[k] = 0.0;
            eh[k] = 0.0;
            if (k < K) {
                ew[k] = exp_nonpos((double)w[k] - (double)mw);
                eh[k] = exp_nonpos((double)h[k] - (double)mh);
                sw += ew[k];
                sh += eh[k];
            }
        }
    }
    auto div = [](double a_, double b_) __attribute__((always_inline)) { return FAST ? fast_div64(a_, b_) : a_ / b_; };
    const double iw = div(W, sw), ih = div(H, sh);
    double kx = x0, ky = y0, bw = 0.0, bh = 0.0;
    float rs0 = sraw[0], rs1 = sraw[0], rs_last = sraw[0];
    bool found = false;
    const bool lower_tail = !(v > y0);
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) {                                                              // wave-uniform
            const double wk = ew[k] * iw + mb;
            const double hk = eh[k] * ih + mb;
            const double upper = ky + hk;
            const bool above = v > upper;
            const bool here = !found && !above, adv = !found && above;
            bw = here ? wk : bw;
            bh = here ? hk : bh;
            rs0 = here ? sraw[k] : rs0;
            rs1 = here ? sraw[k + 1] : rs1;
            kx = adv ? kx + wk : kx;
            ky = adv ? upper : ky;
            found = found || here;
            if (k == K - 1) rs_last = sraw[k + 1];
        }
    const bool tail = lower_tail || !found;
    // tail: the boundary slope takes the place of the bin's first slope -- one softplus serves both forms
    rs0 = tail ? (lower_tail ? sraw[0] : rs_last) : rs0;
    const double dk = (double)(softplus_f(rs0 + f.slope_offset) + f.min_slope);
    const double dk1 = (double)(softplus_f(rs1 + f.slope_offset) + f.min_slope);
    const double bx = lower_tail ? x0 : kx, by = lower_tail ? y0 : ky;
    const double out_t = bx + div(v - by, dk);                                      // spline.py:599-614
    const double s = div(bh, bw);                                                     // spline.py:643
    const double t = dk1 + dk - 2.0 * s;
    const double ym = v - ky;                                                     // spline.py:521-536
    const double a = bh * (s - dk) + ym * t;
    const double b = bh * dk - ym * t;
    const double c = -s * ym;
    const double disc = b * b - 4.0 * a * c;
    const double eps = div(2.0 * c, -b - (FAST ? fast_sqrt64(disc) : sqrt(disc)));
    const double out_i = eps * bw + kx;
    const double e1 = eps * (1.0 - eps);                                          // spline.py:556-558
    const double om = 1.0 - eps;
    const double num = s * s * (dk1 * eps * eps + 2.0 * s * e1 + dk * om * om);
    const double den = s + t * e1;
    const double arg = tail ? dk : div(num, den * den);
    double out = tail ? out_t : out_i;
    if (f.circular) out = py_mod(out - x0 - (double)last, (double)xff - x0) + x0;  // spline.py:257-259 (wave-uniform)
    *logd = (double)logf((float)arg);
    return out;
}

// The parameters of one element, expanded from the P values the conditioner produced for it (spline.py:359-380;
// the same positions as spline_slope_param): K widths, K heights, the K + 1 knot slopes (identity boundary slopes:
// raw slope 0 at knots 0 and K and K - 1 parameters in between; circular: knot K shares the parameter of knot 0),
// then `last` / `last2` = parameters P - 1 / P - 2 (circular shift; log-scale and shift of a learnable domain).
// Every position is a compile-time constant (`get` takes a std::integral_constant), so that the values can stay
// in registers; the flags are wave uniform and become selects.
template <int I, int N, class F>
__device__ __forceinline__ void spline_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        spline_static_for<I + 1, N>(f);
    }
}

// The flags a parameter count pins down, as compile-time constants: the epilogue of the plain layout then carries no
// code for the other ones.  P = 3 K + 1 is the plain / circular layout -- and identity slopes with both bounds learnable
// (IDB), which is an instantiation of its own.
template <int K, int P, bool IDB>
__device__ __forceinline__ SplineFlags spline_flags_of_layout(SplineFlags f) {
    static_assert(!IDB || P == 3 * K + 1, "identity slopes + both bounds learnable: 3 K + 1 parameters");
    if constexpr (IDB) {
        f.identity = true; f.circular = false; f.learn_lower = true; f.learn_upper = true;
        return f;
    }
    if constexpr (P == 3 * K + 1) {
        f.identity = false; f.learn_lower = false; f.learn_upper = false;
    } else if constexpr (P == 3 * K - 1) {
        f.identity = true; f.circular = false; f.learn_lower = false; f.learn_upper = false;
    } else if constexpr (P == 3 * K) {
        f.identity = true;                              // circular, or one learnable bound
    } else if constexpr (P == 3 * K + 2) {
        f.identity = false; f.circular = false;         // one learnable bound
    } else {
        f.identity = false; f.circular = false; f.learn_lower = true; f.learn_upper = true;
    }
    return f;
}

template <int K, int P, class Get>
__device__ __forceinline__ void spline_expand(const SplineFlags& f, Get&& get, float (&w)[K], float (&h)[K],
                                              float (&sraw)[K + 1], float& last, float& last2) {
    static_assert(P >= 3 * K - 1 && P <= 3 * K + 3, "not a parameter count of a K-bin spline");
    spline_static_for<0, K>([&](auto kc) __attribute__((always_inline)) {
        constexpr int k = kc.value;
        w[k] = get(std::integral_constant<int, k>{});
        h[k] = get(std::integral_constant<int, K + k>{});
    });
    spline_static_for<0, K + 1>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = jc.value;
        constexpr int p_plain = 2 * K + j < P ? 2 * K + j : P - 1;         // (clamped: read but not selected)
        constexpr int p_ident = (j == 0 || j == K) ? 0 : 2 * K + j - 1;
        float plain = get(std::integral_constant<int, p_plain>{});
        if (j == K) plain = f.circular ? get(std::integral_constant<int, 2 * K>{}) : plain;
        const float ident = (j == 0 || j == K) ? 0.f : get(std::integral_constant<int, p_ident>{});
        sraw[j] = f.identity ? ident : plain;
    });
    last = get(std::integral_constant<int, P - 1>{});
    last2 = get(std::integral_constant<int, P - 2>{});
}

// Branch-free forward evaluation for the fused epilogue of the split-f16 GEMM: exactly K bins, the K + 1 slopes
// already expanded (spline_expand).  The same arithmetic in the same order as rq_spline_element<K, false>
// (bit-identical results), written as straight-line code with selects: a lone wave on a SIMD has no partner to
// hide the latency of a dependent fp64 chain behind, so the caller evaluates several elements in one basic
// block and lets the scheduler interleave their chains.
template <int K>
__device__ __forceinline__ double rq_spline_forward_full(const float (&w)[K], const float (&h)[K],
                                                        const float (&sraw)[K + 1], float last, float last2,
                                                        const SplineFlags& f, float x0f, float xff, float y0f,
                                                        float yff, float vin, double* logd) {
    const double mb = (double)f.min_bin;
    double x0 = x0f, y0 = y0f;
    double W = (double)xff - (double)x0f - K * mb;
    double H = (double)yff - (double)y0f - K * mb;
    if (f.learn_lower || f.learn_upper) {                                         // wave-uniform branch
        const double scale = exp_poly((double)last);
        W *= scale;
        H *= scale;
        if (f.learn_lower && f.learn_upper) {
            x0 += (double)last2;
            y0 += (double)last2;
        } else if (f.learn_lower) {
            x0 = (double)xff - W - K * mb;
            y0 = (double)yff - H - K * mb;
        }
    }
    double v = vin;
    if (f.circular) v = py_mod(v - x0 + (double)last, (double)xff - x0) + x0;     // wave-uniform branch

    float mw = w[0], mh = h[0];
#pragma unroll
    for (int k = 1; k < K; ++k) {
        mw = fmaxf(mw, w[k]);
        mh = fmaxf(mh, h[k]);
    }
    double ew[K], eh[K];
    double sw = 0.0, sh = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        ew[k] = exp_nonpos((double)w[k] - (double)mw);
        eh[k] = exp_nonpos((double)h[k] - (double)mh);
        sw += ew[k];
        sh += eh[k];
    }
    const double iw = W / sw, ih = H / sh;

    double kx = x0, ky = y0, bw = 0.0, bh = 0.0;
    float rs0 = sraw[K], rs1 = sraw[K];          // never found -> upper tail: boundary slope of the last knot
    bool found = false;
    const bool lower_tail = !(v > x0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double wk = ew[k] * iw + mb;
        const double hk = eh[k] * ih + mb;
        const double upper = kx + wk;
        const bool above = v > upper;
        const bool here = !found && !above;
        const bool adv = !found && above;
        bw = here ? wk : bw;
        bh = here ? hk : bh;
        rs0 = here ? sraw[k] : rs0;
        rs1 = here ? sraw[k + 1] : rs1;
        kx = adv ? upper : kx;
        ky = adv ? ky + hk : ky;
        found = found || here;
    }
    rs0 = lower_tail ? sraw[0] : rs0;
    const bool tail = lower_tail || !found;

    const double dk = (double)(softplus_f(rs0 + f.slope_offset) + f.min_slope);
    const double dk1 = (double)(softplus_f(rs1 + f.slope_offset) + f.min_slope);
    // tail: linear continuation with the boundary slope
    const double bx = lower_tail ? x0 : kx, by = lower_tail ? y0 : ky;
    const double out_t = by + dk * (v - bx);
    // interior
    const double s = bh / bw;
    const double t = dk1 + dk - 2.0 * s;
    const double eps = (v - kx) / bw;
    const double e1 = eps * (1.0 - eps);
    const double out_i = ky + bh * (s * eps * eps + dk * e1) / (s + t * e1);
    const double om = 1.0 - eps;
    const double num = s * s * (dk1 * eps * eps + 2.0 * s * e1 + dk * om * om);
    const double den = s + t * e1;
    const double arg = tail ? dk : num / (den * den);
    *logd = (double)logf((float)arg);
    return tail ? out_t : out_i;
}

// Kernel-side view of a tfep_spline_desc and its (host) validation, shared by the transformer kernels and the
// fused inverse-block kernel.
struct SplineArgs {
    const float *x0, *xf, *y0, *yf;
    SplineFlags f;
    int P;
};


inline int make_spline_args(const tfep_spline_desc* d, SplineArgs* a) {
    TFEP_REQUIRE(d != nullptr, "spline descriptor is NULL");
    TFEP_REQUIRE(d->x0 && d->xf && d->y0 && d->yf, "spline descriptor: x0/xf/y0/yf must be non-NULL");
    TFEP_REQUIRE(d->n_bins >= 1 && d->n_bins <= 32, "spline: n_bins=%d unsupported (1..32)", d->n_bins);
    TFEP_REQUIRE(!(d->circular && (d->learn_lower_bound || d->learn_upper_bound)),
                 "Cannot instantiate a circular spline with learnable limits.");
    TFEP_REQUIRE(d->min_bin_size > 0.f, "The minimum bin size should be positive.");
    TFEP_REQUIRE(d->min_slope > 0.f && d->min_slope < 1.f, "The minimum slope should be between 0 and 1.");
    a->x0 = d->x0;
    a->xf = d->xf;
    a->y0 = d->y0;
    a->yf = d->yf;
    a->f.K = d->n_bins;
    a->f.circular = d->circular != 0;
    a->f.identity = d->identity_boundary_slopes != 0;
    a->f.learn_lower = d->learn_lower_bound != 0;
    a->f.learn_upper = d->learn_upper_bound != 0;
    a->f.min_bin = d->min_bin_size;
    a->f.min_slope = d->min_slope;
    a->f.slope_offset = (float)log(exp(1.0 - (double)d->min_slope) - 1.0);
    a->P = spline_n_params(a->f.K, a->f.circular, a->f.identity, a->f.learn_lower, a->f.learn_upper);
    return TFEP_OK;
}


}  // namespace tfep
