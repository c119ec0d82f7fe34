// Config 5 of BASELINE.json: the EGNN dynamics of the continuous normalizing flow, forward + directional derivative
// (Jacobian-vector product) in one pass, on the matrix cores of gfx950.
//
// Reference: tfep/nn/dynamics/egnn.py (EGNNDynamics.forward :143-194, _EGLayer :222-369), tfep/nn/graph.py (all-pairs
// edges :119-163, distances :222-263, cutoff pruning :266-301, scatter-add segment sum :304-316),
// tfep/nn/embeddings/radial.py (Gaussian / Behler-Parrinello bases :110-130, :161-176, :269-291) and the trace
// estimators of tfep/nn/flows/continuous.py (:285-361).
//
// What the reference does per layer: build the (batch * n * (n-1)) edge list, gather h / x per edge, a 192 -> 64 -> 64
// message MLP, attention, scatter_add of messages and displacements.  What this file does instead:
//   * edges are never materialised: a workgroup owns 16 DESTINATION nodes of one sample and walks all source nodes;
//     one MFMA column group = the 16 edges (source i -> the 16 destinations);
//   * the edge MLPs run TRANSPOSED on v_mfma_f32_16x16x4_f32 (exact fp32): features are matrix rows, edges are columns.
//     In that orientation the accumulator layout of one product (lane = edge column, registers = 4 consecutive
//     features) IS the B-operand layout of the next one, so the whole chain rbf -> W1 -> SiLU -> W2 -> SiLU ->
//     attention -> X1 -> SiLU -> tanh stays in registers: no LDS round trip, no transposes;
//   * the first linear is split by input block: W1 [h_src, h_dest, rbf] = P_i + Q_j + W1c rbf with the node-level
//     products P = W1a h, Q = W1b h + b1 computed once per node (egnn_node_kernel), not once per edge;
//   * the sum over sources (the reference's scatter_add to edges[1]) is a register accumulation -- the destination is
//     the lane -- finished by a fixed-order cross-wave reduction in LDS: no atomics, bit-reproducible;
//   * pruned edges (distance > cutoff, graph.py:297) are masked; a column group with no live edge is skipped entirely;
//   * the tangent (dx -> d velocity) rides along: every product is issued twice with the same weight fragment, the
//     activations carry their derivatives.  e . (J e) is the Hutchinson trace estimate of continuous.py:307-324 (the
//     reference forms (e^T J) . e by a reverse pass -- the same number).
// Weights are re-packed from the reference parameter tensors on every call (tfep_egnn_pack_layer) into "lane-linear"
// images: the 16 bytes lane l needs for 4 consecutive MFMA k-steps of a row tile are contiguous, a wave reads 1 KB per
// ds_read_b128 without bank conflicts.
#include "common.h"
#include <type_traits>

namespace tfep {
namespace {

using f4 = __attribute__((ext_vector_type(4))) float;

#ifndef TFEP_EGNN_FAST_GEOM
#define TFEP_EGNN_FAST_GEOM 1
#endif
constexpr int EDGE_WAVES = 8;            // waves per workgroup of the edge kernel (2 per SIMD)
constexpr int NODE_WAVES = 4;

__host__ __device__ inline int64_t img_floats(int nt) { return (int64_t)nt * nt * 256; }       // (16 nt)^2
// layout of a packed layer (floats)
struct PackedLayout {
    int64_t w1c, w2, x1, b2, d1, wa, x2, mu, gamma, scal, u1a, u1b, u2, c1, c2, w1a, w1b, b1;
    int64_t w1cT, w2T, x1T, w1aT, w1bT, u2T, u1aT, u1bT;        // transposed fp32 images (reverse pass)
    int64_t s_w1c, s_w2, s_x1, s_w1cT, s_w2T, s_x1T, total;     // split-f16 images, forward and transposed
};
// A split-f16 operand image: per (row tile tp, pair of k tiles T) 64 lanes x 8 halves -- the "hi" image (fp16 of the
// scaled weight) followed by the "lo" image (fp16 of the residual); in floats of the packed buffer.
__host__ __device__ inline int64_t split_img_floats(int nt) { return (int64_t)nt * ((nt + 1) / 2) * 64 * 8; }
__host__ __device__ inline PackedLayout packed_layout(int nt) {
    PackedLayout L;
    const int64_t I = img_floats(nt), V = 16 * nt;
    int64_t o = 0;
    L.w1c = o; o += I; L.w2 = o; o += I; L.x1 = o; o += I;
    L.b2 = o; o += V; L.d1 = o; o += V; L.wa = o; o += V; L.x2 = o; o += V; L.mu = o; o += V; L.gamma = o; o += V;
    L.scal = o; o += 4;
    L.u1a = o; o += I; L.u1b = o; o += I; L.u2 = o; o += I; L.c1 = o; o += V; L.c2 = o; o += V;
    L.w1a = o; o += I; L.w1b = o; o += I; L.b1 = o; o += V;
    L.w1cT = o; o += I; L.w2T = o; o += I; L.x1T = o; o += I;
    L.w1aT = o; o += I; L.w1bT = o; o += I; L.u2T = o; o += I; L.u1aT = o; o += I; L.u1bT = o; o += I;
    const int64_t SI = split_img_floats(nt);
    L.s_w1c = o; o += SI; L.s_w2 = o; o += SI; L.s_x1 = o; o += SI;
    L.s_w1cT = o; o += SI; L.s_w2T = o; o += SI; L.s_x1T = o; o += SI;
    L.total = o;
    return L;
}
constexpr int EDGE_CONST_VECS = 6;       // b2, d1, wa, x2, mu, gamma follow the three edge images contiguously

__device__ inline f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// acc[tp] += W[tp][:] x  (and the same for the tangent dx with the SAME weight fragment), W as a lane-linear LDS image
template <int NT, bool TAN>
__device__ inline void chain_gemm(const f4* __restrict__ img, const f4 (&x)[NT], const f4 (&dx)[NT], f4 (&acc)[NT],
                                  f4 (&dacc)[NT], int lane) {
#pragma unroll
    for (int tp = 0; tp < NT; ++tp) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f4 a = img[(tp * NT + t) * 64 + lane];
            acc[tp] = mfma4(a.x, x[t].x, acc[tp]);
            if (TAN) dacc[tp] = mfma4(a.x, dx[t].x, dacc[tp]);
            acc[tp] = mfma4(a.y, x[t].y, acc[tp]);
            if (TAN) dacc[tp] = mfma4(a.y, dx[t].y, dacc[tp]);
            acc[tp] = mfma4(a.z, x[t].z, acc[tp]);
            if (TAN) dacc[tp] = mfma4(a.z, dx[t].z, dacc[tp]);
            acc[tp] = mfma4(a.w, x[t].w, acc[tp]);
            if (TAN) dacc[tp] = mfma4(a.w, dx[t].w, dacc[tp]);
        }
    }
}

using h8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr float SPLIT_X_SCALE = 16.0f;       // activations enter the split products as x * 16: |x| < 4094 fits fp16

// Inline-asm MFMAs on purpose: with the builtin, hipcc (ROCm 7.2) gave destination and SrcC different registers and re-used the
// SrcC registers of the last products two wait states later while those MFMAs were still in flight -- the tangent of the
// split chain came out different from run to run (caught by the bitwise row-independence check of
// tests/test_gpu_continuous.py::test_cfg5_full_size_properties; the value tests against the goldens at 1e-5 passed).
// Accumulating in place there is no separate SrcC to clobber; A / B are read when the instruction issues.  hipcc pads nothing
// around an asm statement: the chain puts the wait states before the first product and after the last itself.

// ---- the forward edge kernel's version of the split product: operands arrive PRE-SCALED (the producers fold the x 16 into
// arithmetic they do anyway), every fp32 -> (hi, lo) pair costs three vector instructions, and the first product of an
// accumulator takes the inline constant 0 as SrcC (no zero fill of 2 x 16 NT registers per product).
using u4 = __attribute__((ext_vector_type(4))) uint32_t;

// (a, b) -> hi = (fp16(a), fp16(b)), lo = (fp16(a - hi_a), fp16(b - hi_b)); v_fma_mix reads the fp16 high half directly
// (op_sel_hi marks src2 as fp16, op_sel picks its upper half), so there is no conversion back and no subtraction.
__device__ inline void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hi), "=&v"(lo) : "v"(a), "v"(b));
}
__device__ inline void mfma16_first(f4& acc, const u4& a, const u4& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}
__device__ inline void mfma16_next(f4& acc, const u4& a, const u4& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int NT>
__device__ inline void split_tiles(const f4 (&x)[NT], u4 (&xh)[(NT + 1) / 2], u4 (&xl)[(NT + 1) / 2]) {
#pragma unroll
    for (int T = 0; T < (NT + 1) / 2; ++T) {
        uint32_t h0, l0, h1, l1, h2 = 0u, l2 = 0u, h3 = 0u, l3 = 0u;
        split_pair(x[2 * T].x, x[2 * T].y, h0, l0);
        split_pair(x[2 * T].z, x[2 * T].w, h1, l1);
        if (2 * T + 1 < NT) {
            const f4& b = x[(2 * T + 1 < NT) ? 2 * T + 1 : 0];
            split_pair(b.x, b.y, h2, l2);
            split_pair(b.z, b.w, h3, l3);
        }
        xh[T] = u4{h0, h1, h2, h3};
        xl[T] = u4{l0, l1, l2, l3};
    }
}
// LDS byte address of a pointer into shared memory (the low 32 bits of a flat address in the LDS aperture are the offset)
__device__ inline uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }
// the hi / lo weight fragments of one (row tile, k-tile pair) group, issued without waiting: the compiler does not know
// these loads (nor count them), lds_wait_pair before the first use is what orders them
template <int OFF_H, int OFF_L>
__device__ inline void lds_read_pair(u4& wh, u4& wl, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                 : "=&v"(wh), "=&v"(wl) : "v"(addr), "i"(OFF_H), "i"(OFF_L));
}
template <int N>
__device__ inline void lds_wait_pair(u4& wh, u4& wl) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(wh), "+v"(wl) : "i"(N));
}
template <int I, int N, class F>
__device__ inline void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// acc = W x, dacc = W dx for pre-scaled x, dx (acc / dacc need no initialisation).  The weight fragments are read one
// group AHEAD of the products that use them (two register sets, counted lgkmcnt): left to the compiler every group was
// "two ds_read_b128, wait, six MFMAs" -- 24 exposed LDS round trips per source with the matrix pipe idle behind each.
template <int NT, bool TAN>
__device__ inline void chain_gemm_split_pre(const u4* __restrict__ img, const f4 (&x)[NT], const f4 (&dx)[NT],
                                            f4 (&acc)[NT], f4 (&dacc)[NT], int lane) {
    constexpr int NT2 = (NT + 1) / 2, G = NT * NT2, LO = NT * NT2 * 64 * 16;
    static_assert(LO + G * 1024 < 65536, "ds_read offset field");
    u4 xh[NT2], xl[NT2], dxh[NT2], dxl[NT2];
    split_tiles<NT>(x, xh, xl);
    if (TAN) split_tiles<NT>(dx, dxh, dxl);
    const uint32_t base = lds_addr(img) + (uint32_t)lane * 16u;
    // the halves come from VALU instructions inside asm statements: keep them ahead of the first product and give the
    // VALU -> MFMA operand hazard its wait states by hand (hipcc pads nothing around asm)
#pragma unroll
    for (int T = 0; T < NT2; ++T) {
        asm volatile("" : "+v"(xh[T]), "+v"(xl[T]));
        if (TAN) asm volatile("" : "+v"(dxh[T]), "+v"(dxl[T]));
    }
    __builtin_amdgcn_sched_barrier(0);
    u4 wh[2], wl[2];
    lds_read_pair<0, LO>(wh[0], wl[0], base);
    static_for<0, G>([&](auto gc) {
        constexpr int g = decltype(gc)::value, tp = g / NT2, T = g % NT2, cur = g & 1;
        if constexpr (g + 1 < G) lds_read_pair<(g + 1) * 1024, LO + (g + 1) * 1024>(wh[cur ^ 1], wl[cur ^ 1], base);
        lds_wait_pair<(g + 1 < G) ? 2 : 0>(wh[cur], wl[cur]);
        if (T == 0) {
            mfma16_first(acc[tp], wh[cur], xh[T]);
            if (TAN) mfma16_first(dacc[tp], wh[cur], dxh[T]);
        } else {
            mfma16_next(acc[tp], wh[cur], xh[T]);
            if (TAN) mfma16_next(dacc[tp], wh[cur], dxh[T]);
        }
        mfma16_next(acc[tp], wl[cur], xh[T]);
        if (TAN) mfma16_next(dacc[tp], wl[cur], dxh[T]);
        mfma16_next(acc[tp], wh[cur], xl[T]);
        if (TAN) mfma16_next(dacc[tp], wh[cur], dxl[T]);
    });
    // leave the matrix pipe's result latency behind before anything reads the accumulators
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// The same product for UNSCALED operands (the reverse-pass kernel: cotangents have no producer to fold the factor into):
// one multiply per element in front of the pre-scaled chain.  acc / dacc need no initialisation; the caller un-scales
// (weights carry a per-matrix power of two, activations SPLIT_X_SCALE).
template <int NT, bool TAN>
__device__ inline void chain_gemm_split(const h8* __restrict__ img, const f4 (&x)[NT], const f4 (&dx)[NT], f4 (&acc)[NT],
                                        f4 (&dacc)[NT], int lane) {
    f4 xs[NT], dxs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xs[t][r] = x[t][r] * SPLIT_X_SCALE;
            if (TAN) dxs[t][r] = dx[t][r] * SPLIT_X_SCALE;
        }
    }
    chain_gemm_split_pre<NT, TAN>(reinterpret_cast<const u4*>(img), xs, dxs, acc, dacc, lane);
}

// exp and 1/x on the transcendental unit (v_exp_f32 / v_rcp_f32, 1 ulp each; the argument scaling x * log2(e) adds
// |x| 6e-8 of relative error): the library expf / IEEE division cost ~10 instructions each, and the edge kernel evaluates
// 64 of each per 16 edges and layer.  Measured against the float64 goldens the velocities keep their 6e-7 relative error.
__device__ inline float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ inline float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// tanh on the same two instructions: 1 - 2 / (1 + e^{2x}) (absolute error ~2e-7: an exp and a reciprocal of 1 ulp each), and the odd
// series to x^5 below |x| = 0.1, where that form cancels (its remainder there: 17 x^7 / 315 < 6e-9).  The library tanhf() is ~40
// instructions, once per edge and layer, on a kernel bound by its vector instructions (tanh'(x) = 1 - tanh^2 stays as it is).
__device__ inline float fast_tanh(float x) {
    const float ax = fminf(fabsf(x), 15.0f);
    const float big = 1.0f - 2.0f * fast_rcp(1.0f + fast_exp(2.0f * ax));
    const float x2 = ax * ax;
    const float small = ax * fmaf(x2, fmaf(x2, 2.0f / 15.0f, -1.0f / 3.0f), 1.0f);
    return copysignf(ax < 0.1f ? small : big, x);
}

// SiLU and its derivative (torch.nn.SiLU: x * sigmoid(x)), elementwise on a tile; dz <- silu'(z) dz, z <- silu(z)
template <int NT, bool TAN>
__device__ inline void silu_tile(f4 (&z)[NT], f4 (&dz)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = z[t][r];
            const float sig = fast_rcp(1.0f + fast_exp(-v));
            const float sv = v * sig;
            z[t][r] = sv;
            if (TAN) dz[t][r] *= fmaf(sv, 1.0f - sig, sig);       // silu' = sig + silu (1 - sig)
        }
    }
}

// The same with the result scaled by XS = 2^LOG2S (the split products take their operands pre-scaled; XS = 1 is the plain
// SiLU at the same instruction count): z <- XS silu(z), dz <- XS silu'(z) dz.  XS sigma = 1 / (1/XS + exp(-z) / XS), the
// 1 / XS inside the exponent is an exact shift of the exp2 argument.
template <int NT, bool TAN, int LOG2S>
__device__ inline void silu_tile_scaled(f4 (&z)[NT], f4 (&dz)[NT]) {
    constexpr float IS = 1.0f / (float)(1 << LOG2S);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = z[t][r];
            const float e = __builtin_amdgcn_exp2f(fmaf(v, -1.44269504088896340736f, -(float)LOG2S));
            const float sig = fast_rcp(e + IS);                  // XS sigma(v)
            const float sv = v * sig;                            // XS silu(v)
            z[t][r] = sv;
            if (TAN) dz[t][r] *= fmaf(sv, fmaf(sig, -IS, 1.0f), sig);       // XS (sigma + silu (1 - sigma))
        }
    }
}

// sum over the four 16-lane groups of a wave (the lanes that hold the same edge column)
__device__ inline float sum_over_q(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// weight re-pack: reference parameter tensors of one _EGLayer -> packed layer
// ---------------------------------------------------------------------------------------------------------------
// Image entry of the block M[row][col] = W[row][col0 + col] (n_rows x n_cols) of a row-major matrix W, or of its transpose
// (`transposed`: the image is of M^T, n_cols x n_rows -- the operand of the reverse pass).
__device__ inline float img_entry(const float* w, int ldw, int col0, int n_rows, int n_cols, int nt, int64_t idx,
                                  bool transposed = false) {
    // idx = ((tp * nt + t) * 64 + l) * 4 + r  ->  entry [16 tp + (l & 15)][16 t + 4 (l >> 4) + r]
    const int r = (int)(idx & 3), l = (int)((idx >> 2) & 63);
    const int tt = (int)(idx >> 8);
    const int t = tt % nt, tp = tt / nt;
    int row = 16 * tp + (l & 15), col = 16 * t + 4 * (l >> 4) + r;
    if (transposed) { const int x = row; row = col; col = x; }
    return (row < n_rows && col < n_cols) ? w[(int64_t)row * ldw + col0 + col] : 0.0f;
}
__device__ inline float vec_entry(const float* v, int n, int64_t i) { return (v != nullptr && i < n) ? v[i] : 0.0f; }

// power-of-two scale that puts the largest |w| of a matrix block into [2^13, 2^14) (fp16 overflows at 65504; the
// products are exact in fp32 whatever the scale)
__global__ void egnn_scale_kernel(tfep_egnn_layer_params p, int nt, float* __restrict__ out) {
    __shared__ float red[4];
    const PackedLayout L = packed_layout(nt);
    const int F = p.F, G = p.G, m = blockIdx.x;
    const float* w = m == 0 ? p.msg0_w : (m == 1 ? p.msg2_w : p.ux0_w);
    const int ldw = m == 0 ? 2 * F + G : F, col0 = m == 0 ? 2 * F : 0, n_cols = m == 0 ? G : F;
    float mx = 0.f;
    for (int i = threadIdx.x; i < F * n_cols; i += blockDim.x) mx = fmaxf(mx, fabsf(w[(i / n_cols) * ldw + col0 + i % n_cols]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = 14 - e; }      // mx * 2^e in [2^13, 2^14)
        e = e > 60 ? 60 : (e < -60 ? -60 : e);
        out[L.scal + 1 + m] = ldexpf(1.0f, e);
    }
}

__device__ inline void split_img_entry(const float* w, int ldw, int col0, int n_rows, int n_cols, int nt, float scale,
                                       int64_t idx, _Float16* hi, _Float16* lo, bool transposed = false) {
    // idx = ((tp * nt2 + T) * 64 + l) * 8 + j  ->  entry [16 tp + (l & 15)][16 (2T + (j >> 2)) + 4 (l >> 4) + (j & 3)]
    const int nt2 = (nt + 1) / 2;
    const int j = (int)(idx & 7), l = (int)((idx >> 3) & 63);
    const int tt = (int)(idx >> 9);
    const int T = tt % nt2, tp = tt / nt2;
    int row = 16 * tp + (l & 15), col = 16 * (2 * T + (j >> 2)) + 4 * (l >> 4) + (j & 3);
    if (transposed) { const int x = row; row = col; col = x; }
    const float v = (row < n_rows && col < n_cols) ? w[(int64_t)row * ldw + col0 + col] * scale : 0.0f;
    const _Float16 h = (_Float16)v;
    hi[idx] = h;
    lo[idx] = (_Float16)(v - (float)h);
}

__global__ void egnn_pack_split_kernel(tfep_egnn_layer_params p, int nt, float* __restrict__ out) {
    const PackedLayout L = packed_layout(nt);
    const int F = p.F, G = p.G;
    const int64_t n = (int64_t)nt * ((nt + 1) / 2) * 64 * 8;              // halves per image
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 6 * n; i += (int64_t)gridDim.x * blockDim.x) {
        const int im = (int)(i / n);                   // 0..2 forward images, 3..5 their transposes
        const int m = im % 3;
        const bool tr = im >= 3;
        const int64_t idx = i % n;
        _Float16* hi = reinterpret_cast<_Float16*>(out + L.s_w1c + (int64_t)im * split_img_floats(nt));
        const float scale = out[L.scal + 1 + m];
        if (m == 0) split_img_entry(p.msg0_w, 2 * F + G, 2 * F, F, G, nt, scale, idx, hi, hi + n, tr);
        else if (m == 1) split_img_entry(p.msg2_w, F, 0, F, F, nt, scale, idx, hi, hi + n, tr);
        else split_img_entry(p.ux0_w, F, 0, F, F, nt, scale, idx, hi, hi + n, tr);
    }
}

__global__ void egnn_pack_kernel(tfep_egnn_layer_params p, int nt, float* __restrict__ out) {
    const PackedLayout L = packed_layout(nt);
    const int64_t I = img_floats(nt), V = 16 * nt;
    const int F = p.F, G = p.G;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L.s_w1c; i += (int64_t)gridDim.x * blockDim.x) {
        if (i > L.scal && i < L.u1a) continue;                       // the three split scales: egnn_scale_kernel
        float v;
        if (i < L.w2) v = img_entry(p.msg0_w, 2 * F + G, 2 * F, F, G, nt, i - L.w1c);
        else if (i < L.x1) v = img_entry(p.msg2_w, F, 0, F, F, nt, i - L.w2);
        else if (i < L.b2) v = img_entry(p.ux0_w, F, 0, F, F, nt, i - L.x1);
        else if (i < L.d1) v = vec_entry(p.msg2_b, F, i - L.b2);
        else if (i < L.wa) v = vec_entry(p.ux0_b, F, i - L.d1);
        else if (i < L.x2) v = vec_entry(p.att_w, F, i - L.wa);
        else if (i < L.mu) v = vec_entry(p.ux2_w, F, i - L.x2);
        else if (i < L.gamma) v = vec_entry(p.dist_means, G, i - L.mu);
        else if (i < L.scal) v = (i - L.gamma < G) ? expf(p.dist_log_gammas[i - L.gamma]) : 0.0f;   // radial.py:127
        else if (i < L.u1a) v = (i == L.scal) ? p.att_b[0] : 0.0f;
        else if (i < L.u1b) v = img_entry(p.uh0_w, 2 * F, 0, F, F, nt, i - L.u1a);
        else if (i < L.u2) v = img_entry(p.uh0_w, 2 * F, F, F, F, nt, i - L.u1b);
        else if (i < L.c1) v = img_entry(p.uh2_w, F, 0, F, F, nt, i - L.u2);
        else if (i < L.c2) v = vec_entry(p.uh0_b, F, i - L.c1);
        else if (i < L.w1a) v = vec_entry(p.uh2_b, F, i - L.c2);
        else if (i < L.w1b) v = img_entry(p.msg0_w, 2 * F + G, 0, F, F, nt, i - L.w1a);
        else if (i < L.b1) v = img_entry(p.msg0_w, 2 * F + G, F, F, F, nt, i - L.w1b);
        else if (i < L.w1cT) v = vec_entry(p.msg0_b, F, i - L.b1);
        else if (i < L.w2T) v = img_entry(p.msg0_w, 2 * F + G, 2 * F, F, G, nt, i - L.w1cT, true);
        else if (i < L.x1T) v = img_entry(p.msg2_w, F, 0, F, F, nt, i - L.w2T, true);
        else if (i < L.w1aT) v = img_entry(p.ux0_w, F, 0, F, F, nt, i - L.x1T, true);
        else if (i < L.w1bT) v = img_entry(p.msg0_w, 2 * F + G, 0, F, F, nt, i - L.w1aT, true);
        else if (i < L.u2T) v = img_entry(p.msg0_w, 2 * F + G, F, F, F, nt, i - L.w1bT, true);
        else if (i < L.u1aT) v = img_entry(p.uh2_w, F, 0, F, F, nt, i - L.u2T, true);
        else if (i < L.u1bT) v = img_entry(p.uh0_w, 2 * F, 0, F, F, nt, i - L.u1aT, true);
        else v = img_entry(p.uh0_w, 2 * F, F, F, F, nt, i - L.u1bT, true);
        out[i] = v;
        (void)I; (void)V;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// node embedding + the source / destination terms of layer 0 (the same for every sample: h0 depends on t only)
//   h0 = W_emb [one_hot(type), exp(-gamma (t - mu)^2)] + b_emb        (egnn.py:196-219, radial.py:110-130)
//   P0 = W1a h0,  Q0 = W1b h0 + b1                                     (first linear of message_mlp, egnn.py:246)
// ---------------------------------------------------------------------------------------------------------------
__global__ void egnn_embed_kernel(const float* __restrict__ one_hot, int n_nodes, int n_types, float t,
                                  const float* __restrict__ time_means, const float* __restrict__ time_log_gammas,
                                  int time_dim, const float* __restrict__ w_emb, const float* __restrict__ b_emb, int F,
                                  const float* __restrict__ msg0_w, const float* __restrict__ msg0_b, int G, int f_pad,
                                  float* __restrict__ h0, float* __restrict__ P0, float* __restrict__ Q0) {
    extern __shared__ float smem[];           // h0 of this node (F)
    const int node = blockIdx.x;
    const int K = n_types + time_dim;
    for (int f = threadIdx.x; f < f_pad; f += blockDim.x) {
        float acc = 0.0f;
        if (f < F) {
            acc = b_emb[f];
            for (int k = 0; k < n_types; ++k) acc += w_emb[f * K + k] * one_hot[node * n_types + k];
            for (int k = 0; k < time_dim; ++k) {
                const float d = t - time_means[k];
                acc += w_emb[f * K + n_types + k] * expf(-expf(time_log_gammas[k]) * d * d);
            }
        }
        smem[f] = acc;
        h0[(int64_t)node * f_pad + f] = acc;
    }
    __syncthreads();
    const int ldw = 2 * F + G;
    for (int f = threadIdx.x; f < f_pad; f += blockDim.x) {
        float p = 0.0f, q = 0.0f;
        if (f < F) {
            q = msg0_b[f];
            for (int k = 0; k < F; ++k) {
                p += msg0_w[f * ldw + k] * smem[k];
                q += msg0_w[f * ldw + F + k] * smem[k];
            }
        }
        P0[(int64_t)node * f_pad + f] = p;
        Q0[(int64_t)node * f_pad + f] = q;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// edge kernel: one _EGLayer without its node MLP (egnn.py:272-369)
// ---------------------------------------------------------------------------------------------------------------
template <int NT, bool TAN, bool SPLIT>
__global__ __launch_bounds__(EDGE_WAVES * 64, 2) void egnn_edge_kernel(tfep_egnn_edge_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int FP = 16 * NT;
    // f4 (16 bytes) per operand image in LDS: fp32 lane-linear image, or split-f16 hi + lo images
    constexpr int IMG4 = SPLIT ? NT * ((NT + 1) / 2) * 64 * 2 : NT * NT * 64;
    const PackedLayout L = packed_layout(NT);
    const int n = a.n_nodes;
    const int n_blk = (n + 15) / 16;
    // Workgroup ids go round-robin over the 8 XCDs (each with its own L2): remap so that an XCD works through a
    // CONTIGUOUS range of (sample, destination block) pairs -- the 16 blocks of a sample then share one L2 for the
    // sample's positions and source terms (bijective for any grid size; a different placement only costs speed).
    int blk;
    {
        const int total = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q8 = total >> 3, r8 = total & 7;
        blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int b = blk / n_blk, jb = blk % n_blk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c = lane & 15;

    // ---- LDS: [3 images][6 vectors][pos n*3][dpos n*3][reduction]
    f4* const w_img = reinterpret_cast<f4*>(smem);
    float* const vecs = smem + 3 * IMG4 * 4;
    float* const s_pos = vecs + EDGE_CONST_VECS * FP;
    float* const s_dpos = s_pos + ((3 * n + 3) & ~3);
    f4* const s_q = reinterpret_cast<f4*>(s_dpos + (TAN ? ((3 * n + 3) & ~3) : 0));      // Q of the 16 destinations
    f4* const s_dq = s_q + NT * 64;
    float* const s_red = reinterpret_cast<float*>(s_dq + (TAN ? NT * 64 : 0));
    {
        const f4* src = reinterpret_cast<const f4*>(a.packed + (SPLIT ? L.s_w1c : L.w1c));
        for (int i = tid; i < 3 * IMG4; i += EDGE_WAVES * 64) w_img[i] = src[i];
        const f4* vsrc = reinterpret_cast<const f4*>(a.packed + L.b2);
        for (int i = tid; i < EDGE_CONST_VECS * FP / 4; i += EDGE_WAVES * 64) w_img[3 * IMG4 + i] = vsrc[i];
        const float* px = a.pos + (int64_t)b * 3 * n;
        for (int i = tid; i < 3 * n; i += EDGE_WAVES * 64) s_pos[i] = px[i];
        if (TAN) {
            const float* pdx = a.dpos + (int64_t)b * 3 * n;
            for (int i = tid; i < 3 * n; i += EDGE_WAVES * 64) s_dpos[i] = pdx[i];
        }
    }
    // destination terms Q_j (+ tangent) of the block, lane-linear: entry (4t + q) * 16 + c is Q[j0 + c][16t + 4q ..]
    const int64_t pq_b = (int64_t)blk / n_blk * a.pq_bstride;
    for (int e = tid; e < NT * 64 * (TAN ? 2 : 1); e += EDGE_WAVES * 64) {
        const bool tan = e >= NT * 64;
        const int ee = tan ? e - NT * 64 : e;
        const int cc = ee & 15, tq = ee >> 4;
        const int jj = jb * 16 + cc;
        f4 v = f4{0.f, 0.f, 0.f, 0.f};
        if (jj < n) {
            if (!tan) v = reinterpret_cast<const f4*>(a.Q + (pq_b + jj) * FP)[tq];
            else if (a.dQ != nullptr) v = reinterpret_cast<const f4*>(a.dQ + ((int64_t)b * n + jj) * FP)[tq];
        }
        (tan ? s_dq : s_q)[ee] = v;
    }
    const float att_b = a.packed[L.scal];
    // Activations enter the split products PRE-SCALED by XS = 16 (|activation| < 4094 fits fp16; the producers -- radial
    // basis, SiLU -- fold the factor into arithmetic they do anyway); the products come out scaled by (weight scale of the
    // matrix) x XS: exact powers of two, removed by the multiply that adds the bias.  The exact-fp32 chain uses XS = 1.
    constexpr int LOG2S = SPLIT ? 4 : 0;
    constexpr float XS = (float)(1 << LOG2S), IXS = 1.0f / XS;
    const float inv1 = SPLIT ? 1.0f / (a.packed[L.scal + 1] * XS) : 1.0f;
    const float inv2 = SPLIT ? 1.0f / (a.packed[L.scal + 2] * XS) : 1.0f;
    const float inv3 = SPLIT ? 1.0f / (a.packed[L.scal + 3] * XS) : 1.0f;
    __syncthreads();
    const f4* const img_w1c = w_img;
    const f4* const img_w2 = w_img + IMG4;
    const f4* const img_x1 = w_img + 2 * IMG4;
    const f4* const v_b2 = reinterpret_cast<const f4*>(vecs);
    const f4* const v_d1 = v_b2 + FP / 4;
    const f4* const v_wa = v_d1 + FP / 4;
    const f4* const v_x2 = v_wa + FP / 4;
    const f4* const v_mu = v_x2 + FP / 4;
    const f4* const v_ga = v_mu + FP / 4;
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    // out = W x, dout = W dx (raw: the caller un-scales and adds the bias in one fused multiply-add per element)
    auto product = [&](const f4* img, const f4 (&x)[NT], const f4 (&dx)[NT], f4 (&out)[NT], f4 (&dout)[NT]) {
        if constexpr (SPLIT) {
            chain_gemm_split_pre<NT, TAN>(reinterpret_cast<const u4*>(img), x, dx, out, dout, lane);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) { out[t] = zero4; dout[t] = zero4; }
            chain_gemm<NT, TAN>(img, x, dx, out, dout, lane);
        }
    };

    // ---- per-lane destination state
    const int j = jb * 16 + c;
    const bool j_ok = j < n;
    const int jc = j_ok ? j : 0;
    const float xj0 = s_pos[3 * jc], xj1 = s_pos[3 * jc + 1], xj2 = s_pos[3 * jc + 2];
    float dxj0 = 0.f, dxj1 = 0.f, dxj2 = 0.f;
    if (TAN) { dxj0 = s_dpos[3 * jc]; dxj1 = s_dpos[3 * jc + 1]; dxj2 = s_dpos[3 * jc + 2]; }
    f4 nm[NT], dnm[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { nm[t] = f4{0.f, 0.f, 0.f, 0.f}; dnm[t] = f4{0.f, 0.f, 0.f, 0.f}; }
    float disp0 = 0.f, disp1 = 0.f, disp2 = 0.f, dd0 = 0.f, dd1 = 0.f, dd2 = 0.f;
    const float rc = a.r_cutoff, pi_rc = 3.14159265358979323846f / rc, half_rc = 0.5f / rc;
    (void)half_rc;
    const float* const Pb = a.P + pq_b * FP;
    const float* const dPb = (TAN && a.dP != nullptr) ? a.dP + (int64_t)b * n * FP : nullptr;
    // source terms P_i (+ tangent) of the NEXT source are fetched one iteration ahead: with two waves per SIMD nothing
    // else hides an L2 / HBM round trip
    f4 Pn[NT], dPn[NT];
    auto fetch_source = [&](int i) {
        const int ic = i < n ? i : 0;
        const f4* Pp = reinterpret_cast<const f4*>(Pb + (int64_t)ic * FP);
#pragma unroll
        for (int t = 0; t < NT; ++t) Pn[t] = Pp[4 * t + q];
        if (TAN) {
            if (dPb != nullptr) {
                const f4* dPp = reinterpret_cast<const f4*>(dPb + (int64_t)ic * FP);
#pragma unroll
                for (int t = 0; t < NT; ++t) dPn[t] = dPp[4 * t + q];
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) dPn[t] = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    fetch_source(wave);
    for (int i = wave; i < n; i += EDGE_WAVES) {
        // ---- geometry of the 16 edges (i -> j0 + c); the four q-groups compute it redundantly (a few dozen VALU)
        const float v0 = xj0 - s_pos[3 * i], v1 = xj1 - s_pos[3 * i + 1], v2 = xj2 - s_pos[3 * i + 2];
#if TFEP_EGNN_FAST_GEOM
        // square root, reciprocal, the switching function's sine / cosine and the attention's sigmoid on the
        // transcendental unit (1 ulp each; v_sin_f32 / v_cos_f32 take their argument in revolutions): per-edge scalars
        // that every lane of a column evaluates -- the IEEE division / libm forms cost ~100 instructions per 16 edges
        const float d = __builtin_amdgcn_sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
#else
        const float d = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);              // graph.py:257
#endif
        const bool keep = j_ok && (j != i) && (d <= rc);                 // graph.py:297 (and no self edges, :143)
        if (__ballot(keep) == 0ull) {                                    // wave-uniform: nothing survives the cutoff
            fetch_source(i + EDGE_WAVES);
            continue;
        }
#if TFEP_EGNN_FAST_GEOM
        const float inv_d = keep ? fast_rcp(d) : 0.0f;
#else
        const float inv_d = keep ? 1.0f / d : 0.0f;
#endif
        const float u0 = v0 * inv_d, u1 = v1 * inv_d, u2 = v2 * inv_d;   // normalised direction (graph.py:260)
        float ddist = 0.f, du0 = 0.f, du1 = 0.f, du2 = 0.f;
        if (TAN) {
            const float w0 = dxj0 - s_dpos[3 * i], w1 = dxj1 - s_dpos[3 * i + 1], w2 = dxj2 - s_dpos[3 * i + 2];
            ddist = u0 * w0 + u1 * w1 + u2 * w2;
            du0 = (w0 - u0 * ddist) * inv_d; du1 = (w1 - u1 * ddist) * inv_d; du2 = (w2 - u2 * ddist) * inv_d;
        }
        float sn, cs;
#if TFEP_EGNN_FAST_GEOM
        {
            const float rev = fminf(d, rc) * half_rc;                     // (pi d / rc) / (2 pi), in [0, 1/2]
            sn = __builtin_amdgcn_sinf(rev);
            cs = __builtin_amdgcn_cosf(rev);
        }
#else
        sincosf(pi_rc * d, &sn, &cs);
#endif
        const float sw = (0.5f * XS) * cs + (0.5f * XS);                  // XS x the switching function of radial.py:173
        const float dsw = (-0.5f * XS) * pi_rc * sn;                      // XS d sw / d dist
        f4 z[NT], dz[NT], y[NT], dy[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {                                    // y, dy <- XS rbf, XS d rbf
            const f4 mu = v_mu[4 * t + q], ga = v_ga[4 * t + q];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dm = d - mu[r];
                const float gd = ga[r] * dm;
                const float g = __builtin_amdgcn_exp2f(gd * dm * -1.44269504088896340736f);   // radial.py:126-128
                y[t][r] = g * sw;                                         // radial.py:291
                if (TAN) dy[t][r] = (ddist * g) * fmaf(gd * -2.0f, sw, dsw);
            }
        }
        // ---- message MLP: z1 = P_i + Q_j + W1c rbf (egnn.py:246-251 with the first linear split by input block)
        // (what the multiply-add after a product reads from LDS is fetched BEFORE the product: behind its scheduling
        // barrier the round trip would be exposed, in front of it the 48 MFMAs cover it)
        f4 cA[NT], cB[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {                                    // P_i + Q_j in the registers of P_i
            const f4 qv = s_q[(4 * t + q) * 16 + c];
            f4 dqv = zero4;
            if (TAN) dqv = s_dq[(4 * t + q) * 16 + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Pn[t][r] += qv[r];
                if (TAN) dPn[t][r] += dqv[r];
            }
        }
        product(img_w1c, y, dy, z, dz);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                z[t][r] = fmaf(z[t][r], inv1, Pn[t][r]);
                if (TAN) dz[t][r] = fmaf(dz[t][r], inv1, dPn[t][r]);
            }
        }
        fetch_source(i + EDGE_WAVES);                                    // P of the next source: lands under the rest of this one
        silu_tile_scaled<NT, TAN, LOG2S>(z, dz);
#pragma unroll
        for (int t = 0; t < NT; ++t) cA[t] = v_b2[4 * t + q];
        product(img_w2, z, dz, y, dy);
#pragma unroll
        for (int t = 0; t < NT; ++t) cB[t] = v_wa[4 * t + q];                 // (lands under the SiLU that comes first)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                y[t][r] = fmaf(y[t][r], inv2, cA[t][r]);
                if (TAN) dy[t][r] *= inv2;
            }
        }
        silu_tile_scaled<NT, TAN, LOG2S>(y, dy);                          // y = XS m2, dy = XS d m2
        // ---- attention (egnn.py:254-257, 323-325): m = m2 * sigmoid(wa . m2 + ba)
        float s = 0.f, ds = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s = fmaf(cB[t][r], y[t][r], s);
                if (TAN) ds = fmaf(cB[t][r], dy[t][r], ds);
            }
        }
        s = sum_over_q(s) * IXS + att_b;
#if TFEP_EGNN_FAST_GEOM
        const float att = keep ? fast_rcp(1.0f + fast_exp(-s)) : 0.0f;    // pruned edges carry no message
#else
        const float att = keep ? 1.0f / (1.0f + expf(-s)) : 0.0f;         // pruned edges carry no message
#endif
        float datt = 0.f;
        if (TAN) datt = att * (1.0f - att) * (sum_over_q(ds) * IXS);
        {
            const float att_s = att * IXS, datt_s = datt * IXS;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {                             // segment sum over the sources (egnn.py:331)
                    nm[t][r] = fmaf(y[t][r], att_s, nm[t][r]);
                    if (TAN) dnm[t][r] = fmaf(dy[t][r], att_s, fmaf(y[t][r], datt_s, dnm[t][r]));
                }
            }
        }
        // ---- displacement magnitude (egnn.py:260-267, 347-361): tanh(x2 . SiLU(X1 m + d1)).  X1 (att m2) = att (X1 m2):
        // the product runs on m2 as it stands and the attention weight rides on the un-scaling multiply
#pragma unroll
        for (int t = 0; t < NT; ++t) cA[t] = v_d1[4 * t + q];
        product(img_x1, y, dy, z, dz);
#pragma unroll
        for (int t = 0; t < NT; ++t) cB[t] = v_x2[4 * t + q];
        s = 0.f; ds = 0.f;
        {
            const float ia = inv3 * att, ida = inv3 * datt;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = fmaf(z[t][r], ia, cA[t][r]);
                    const float e = __builtin_amdgcn_exp2f(v * -1.44269504088896340736f);
                    const float sig = fast_rcp(1.0f + e);
                    const float sv = v * sig;
                    s = fmaf(cB[t][r], sv, s);
                    if (TAN) {
                        const float dv3 = fmaf(dz[t][r], ia, z[t][r] * ida);
                        ds = fmaf(cB[t][r] * fmaf(sv, 1.0f - sig, sig), dv3, ds);
                    }
                }
            }
        }
        #if TFEP_EGNN_FAST_GEOM
        const float mag = fast_tanh(sum_over_q(s));
#else
        const float mag = tanhf(sum_over_q(s));
#endif
        const float k = keep ? a.speed_factor : 0.0f;
        disp0 += k * u0 * mag; disp1 += k * u1 * mag; disp2 += k * u2 * mag;
        if (TAN) {
            const float dmag = (1.0f - mag * mag) * sum_over_q(ds);
            dd0 += k * (du0 * mag + u0 * dmag); dd1 += k * (du1 * mag + u1 * dmag); dd2 += k * (du2 * mag + u2 * dmag);
        }
    }

    // ---- cross-wave reduction in a fixed order (no atomics), then the node outputs
    constexpr int NM4 = NT * 64;                                          // f4 per wave of the message partials
    f4* const r_nm = reinterpret_cast<f4*>(s_red);
    f4* const r_dnm = r_nm + EDGE_WAVES * NM4;
    float* const r_disp = reinterpret_cast<float*>(r_dnm + (TAN ? EDGE_WAVES * NM4 : 0));
    __syncthreads();                                                      // (the weight images are no longer read)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        r_nm[wave * NM4 + t * 64 + lane] = nm[t];
        if (TAN) r_dnm[wave * NM4 + t * 64 + lane] = dnm[t];
    }
    if (q == 0) {
        float* p = r_disp + (wave * 16 + c) * 6;
        p[0] = disp0; p[1] = disp1; p[2] = disp2; p[3] = dd0; p[4] = dd1; p[5] = dd2;
    }
    __syncthreads();
    if (a.nm != nullptr) {
        for (int it = tid; it < NM4 * (TAN ? 2 : 1); it += EDGE_WAVES * 64) {
            const bool tan = it >= NM4;
            const int e = tan ? it - NM4 : it;
            const f4* src = tan ? r_dnm : r_nm;
            f4 sum = src[e];
#pragma unroll
            for (int w = 1; w < EDGE_WAVES; ++w) sum += src[w * NM4 + e];
            const int t = e >> 6, l = e & 63;
            const int jj = jb * 16 + (l & 15);
            if (jj < n) {
                float* dst = (tan ? a.dnm : a.nm) + ((int64_t)b * n + jj) * FP;
                reinterpret_cast<f4*>(dst)[4 * t + (l >> 4)] = sum;
            }
        }
    }
    if (tid < 16 * (TAN ? 6 : 3)) {
        const int comp = tid % (TAN ? 6 : 3), cc = tid / (TAN ? 6 : 3);
        const int jj = jb * 16 + cc;
        if (jj < n) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < EDGE_WAVES; ++w) sum += r_disp[(w * 16 + cc) * 6 + comp];
            if (comp < 3) a.pos_out[((int64_t)b * n + jj) * 3 + comp] = s_pos[3 * jj + comp] + sum;      // egnn.py:364
            else a.dpos_out[((int64_t)b * n + jj) * 3 + comp - 3] = s_dpos[3 * jj + comp - 3] + sum;
        }
    }
}

template <int NT, bool TAN, bool SPLIT>
size_t edge_lds_bytes(int n) {
    const size_t fp = 16 * NT, img4 = SPLIT ? (size_t)NT * ((NT + 1) / 2) * 64 * 2 : (size_t)NT * NT * 64;
    size_t fl = 3 * img4 * 4 + EDGE_CONST_VECS * fp + (size_t)((3 * n + 3) & ~3) * (TAN ? 2 : 1);
    fl += (size_t)NT * 64 * 4 * (TAN ? 2 : 1);                       // Q / dQ of the destination block
    fl += (size_t)EDGE_WAVES * NT * 64 * 4 * (TAN ? 2 : 1) + EDGE_WAVES * 16 * 6;
    return fl * sizeof(float);
}

// ---------------------------------------------------------------------------------------------------------------
// node kernel between two layers: h' = h + U2 SiLU(U1 [h, nm] + c1) + c2 (egnn.py:327-342) of layer l, then the
// source / destination terms P = W1a h', Q = W1b h' + b1 of layer l + 1 -- the same transposed MFMA chain, 16 nodes
// per column group.
// ---------------------------------------------------------------------------------------------------------------
template <int NT, bool TAN>
__global__ __launch_bounds__(NODE_WAVES * 64) void egnn_node_kernel(tfep_egnn_node_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int FP = 16 * NT;
    constexpr int IMG4 = NT * NT * 64;
    const PackedLayout L = packed_layout(NT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    // LDS: U1a, U1b, U2 (+ c1, c2) of this layer; W1a, W1b (+ b1) of the next
    f4* const post = reinterpret_cast<f4*>(smem);
    f4* const pre = post + 3 * IMG4 + 2 * FP / 4;
    {
        const f4* s0 = reinterpret_cast<const f4*>(a.packed + L.u1a);
        for (int i = tid; i < 3 * IMG4 + 2 * FP / 4; i += NODE_WAVES * 64) post[i] = s0[i];
        const f4* s1 = reinterpret_cast<const f4*>(a.packed_next + L.w1a);
        for (int i = tid; i < 2 * IMG4 + FP / 4; i += NODE_WAVES * 64) pre[i] = s1[i];
    }
    __syncthreads();
    const f4 *img_u1a = post, *img_u1b = post + IMG4, *img_u2 = post + 2 * IMG4;
    const f4 *v_c1 = post + 3 * IMG4, *v_c2 = v_c1 + FP / 4;
    const f4 *img_w1a = pre, *img_w1b = pre + IMG4, *v_b1 = pre + 2 * IMG4;
    const int64_t n_total = (int64_t)a.B * a.n_nodes;
    const int64_t n_groups = (n_total + 15) / 16;
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    for (int64_t g = (int64_t)blockIdx.x * NODE_WAVES + wave; g < n_groups; g += (int64_t)gridDim.x * NODE_WAVES) {
        const int64_t node = g * 16 + c;
        const bool ok = node < n_total;
        const int64_t nd = ok ? node : 0;
        const int64_t hb = (nd / a.n_nodes) * a.h_bstride + (nd % a.n_nodes);
        f4 h[NT], dh[NT], m[NT], dm[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            h[t] = reinterpret_cast<const f4*>(a.h + hb * FP)[4 * t + q];
            m[t] = reinterpret_cast<const f4*>(a.nm + nd * FP)[4 * t + q];
            dh[t] = (TAN && a.dh != nullptr) ? reinterpret_cast<const f4*>(a.dh + nd * FP)[4 * t + q] : zero4;
            dm[t] = TAN ? reinterpret_cast<const f4*>(a.dnm + nd * FP)[4 * t + q] : zero4;
        }
        f4 z[NT], dz[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { z[t] = v_c1[4 * t + q]; dz[t] = zero4; }
        chain_gemm<NT, TAN>(img_u1a, h, dh, z, dz, lane);
        chain_gemm<NT, TAN>(img_u1b, m, dm, z, dz, lane);
        silu_tile<NT, TAN>(z, dz);
        f4 y[NT], dy[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { y[t] = v_c2[4 * t + q] + h[t]; dy[t] = dh[t]; }      // residual (egnn.py:342)
        chain_gemm<NT, TAN>(img_u2, z, dz, y, dy, lane);
        f4 P[NT], dP[NT], Q[NT], dQ[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { P[t] = zero4; dP[t] = zero4; Q[t] = v_b1[4 * t + q]; dQ[t] = zero4; }
        chain_gemm<NT, TAN>(img_w1a, y, dy, P, dP, lane);
        chain_gemm<NT, TAN>(img_w1b, y, dy, Q, dQ, lane);
        if (ok) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                reinterpret_cast<f4*>(a.h_out + node * FP)[4 * t + q] = y[t];
                reinterpret_cast<f4*>(a.P_out + node * FP)[4 * t + q] = P[t];
                reinterpret_cast<f4*>(a.Q_out + node * FP)[4 * t + q] = Q[t];
                if (TAN) {
                    reinterpret_cast<f4*>(a.dh_out + node * FP)[4 * t + q] = dy[t];
                    reinterpret_cast<f4*>(a.dP_out + node * FP)[4 * t + q] = dP[t];
                    reinterpret_cast<f4*>(a.dQ_out + node * FP)[4 * t + q] = dQ[t];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Reverse pass (vector-Jacobian product) of one layer's edge part: what autograd computes for e^T J in the reference's
// Hutchinson estimators (continuous.py:307-361).  Per live edge (source i -> destination j) the forward chain is
// recomputed and walked backwards:
//   g_mag = c (u . gp_j),  g_u = c mag gp_j            (c = speed_factor; gp = cotangent of the layer's output positions)
//   g_z3 = silu'(z3) x2 (1 - mag^2) g_mag,   g_m = X1^T g_z3 + g_nm_j
//   g_a2 = att g_m + wa att (1 - sigma(e)) (g_m . a2),   g_z2 = silu'(z2) g_a2,   g_z1 = silu'(z1) (W2^T g_z2)
//   g_P_i += g_z1,  g_Q_j += g_z1,  g_rbf = W1c^T g_z1,  g_d = sum_k g_rbf_k d rbf_k / d d
//   g_v = g_d u + (g_u - u (u . g_u)) / d,   g_x_j += g_v,   g_x_i -= g_v
// The sums over sources (g_Q_j, + g_v) and over destinations (g_P_i, - g_v) cannot both be register accumulations of one
// pass, and per-edge atomics are out of the question: the kernel runs TWICE, once owning 16 destinations per workgroup
// (lanes = destinations, walking the sources) and once owning 16 sources (lanes = sources, walking the destinations).
// Same transposed MFMA chain as the forward kernel, with the transposed weight images for the three reverse products.
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ inline void silu_keep(f4 (&z)[NT], f4 (&ds)[NT]) {          // z <- silu(z), ds <- silu'(z)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = z[t][r];
            const float sig = fast_rcp(1.0f + fast_exp(-v));
            const float sv = v * sig;
            z[t][r] = sv;
            ds[t][r] = fmaf(sv, 1.0f - sig, sig);
        }
    }
}

template <int NT, bool SRC_OWNED, bool SPLIT>
__global__ __launch_bounds__(EDGE_WAVES * 64, 2) void egnn_edge_bwd_kernel(tfep_egnn_edge_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int FP = 16 * NT;
    constexpr int IMG4 = SPLIT ? NT * ((NT + 1) / 2) * 64 * 2 : NT * NT * 64;
    const PackedLayout L = packed_layout(NT);
    const int n = a.n_nodes;
    const int n_blk = (n + 15) / 16;
    int blk;
    {
        const int total = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q8 = total >> 3, r8 = total & 7;
        blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int b = blk / n_blk, ab = blk % n_blk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c = lane & 15;

    // ---- LDS: [3 forward images][3 transposed images][6 vectors][pos][gpos][lane terms][lane g_nm][reduction]
    f4* const w_img = reinterpret_cast<f4*>(smem);
    float* const vecs = smem + 6 * IMG4 * 4;
    float* const s_pos = vecs + EDGE_CONST_VECS * FP;
    float* const s_gpos = s_pos + ((3 * n + 3) & ~3);
    f4* const s_lane = reinterpret_cast<f4*>(s_gpos + ((3 * n + 3) & ~3));
    f4* const s_gnm = s_lane + NT * 64;
    float* const s_red = reinterpret_cast<float*>(s_gnm + NT * 64);
    const int64_t pq_b = (int64_t)b * a.pq_bstride;
    {
        const f4* src = reinterpret_cast<const f4*>(a.packed + (SPLIT ? L.s_w1c : L.w1c));
        for (int i = tid; i < 3 * IMG4; i += EDGE_WAVES * 64) w_img[i] = src[i];
        const f4* srcT = reinterpret_cast<const f4*>(a.packed + (SPLIT ? L.s_w1cT : L.w1cT));
        for (int i = tid; i < 3 * IMG4; i += EDGE_WAVES * 64) w_img[3 * IMG4 + i] = srcT[i];
        const f4* vsrc = reinterpret_cast<const f4*>(a.packed + L.b2);
        for (int i = tid; i < EDGE_CONST_VECS * FP / 4; i += EDGE_WAVES * 64) w_img[6 * IMG4 + i] = vsrc[i];
        const float* px = a.pos + (int64_t)b * 3 * n;
        const float* pg = a.g_pos_out + (int64_t)b * 3 * n;
        for (int i = tid; i < 3 * n; i += EDGE_WAVES * 64) { s_pos[i] = px[i]; s_gpos[i] = pg[i]; }
        // lane-owned node terms: Q of the destinations (dest-owned) or P of the sources (source-owned); and, dest-owned,
        // the cotangent of the aggregated messages of the lane's destination
        const float* lane_src = SRC_OWNED ? a.P : a.Q;
        for (int e = tid; e < NT * 64 * 2; e += EDGE_WAVES * 64) {
            const bool second = e >= NT * 64;
            const int ee = second ? e - NT * 64 : e;
            const int cc = ee & 15, tq = ee >> 4;
            const int node = ab * 16 + cc;
            f4 v = f4{0.f, 0.f, 0.f, 0.f};
            if (node < n) {
                if (!second) v = reinterpret_cast<const f4*>(lane_src + (pq_b + node) * FP)[tq];
                else if (!SRC_OWNED && a.g_nm != nullptr) v = reinterpret_cast<const f4*>(a.g_nm + ((int64_t)b * n + node) * FP)[tq];
            }
            (second ? s_gnm : s_lane)[ee] = v;
        }
    }
    const float att_b = a.packed[L.scal];
    const float inv1 = SPLIT ? 1.0f / (a.packed[L.scal + 1] * SPLIT_X_SCALE) : 1.0f;
    const float inv2 = SPLIT ? 1.0f / (a.packed[L.scal + 2] * SPLIT_X_SCALE) : 1.0f;
    const float inv3 = SPLIT ? 1.0f / (a.packed[L.scal + 3] * SPLIT_X_SCALE) : 1.0f;
    __syncthreads();
    const f4* const img_w1c = w_img;
    const f4* const img_w2 = w_img + IMG4;
    const f4* const img_x1 = w_img + 2 * IMG4;
    const f4* const img_w1cT = w_img + 3 * IMG4;
    const f4* const img_w2T = w_img + 4 * IMG4;
    const f4* const img_x1T = w_img + 5 * IMG4;
    const f4* const v_b2 = reinterpret_cast<const f4*>(vecs);
    const f4* const v_d1 = v_b2 + FP / 4;
    const f4* const v_wa = v_d1 + FP / 4;
    const f4* const v_x2 = v_wa + FP / 4;
    const f4* const v_mu = v_x2 + FP / 4;
    const f4* const v_ga = v_mu + FP / 4;
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    // out = init + W x
    auto product = [&](const f4* img, const f4 (&x)[NT], f4 (&out)[NT], float inv, auto init) {
        f4 dummy[NT];
        if constexpr (SPLIT) {
#pragma unroll
            for (int t = 0; t < NT; ++t) out[t] = zero4;
            chain_gemm_split<NT, false>(reinterpret_cast<const h8*>(img), x, x, out, dummy, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t) out[t] = out[t] * inv + init(t);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) out[t] = init(t);
            chain_gemm<NT, false>(img, x, x, out, dummy, lane);
        }
    };
    auto zero_init = [&](int) { return zero4; };

    const int an = ab * 16 + c;                          // the lane's own node
    const bool a_ok = an < n;
    const int ac = a_ok ? an : 0;
    const float xa0 = s_pos[3 * ac], xa1 = s_pos[3 * ac + 1], xa2 = s_pos[3 * ac + 2];
    const float gpa0 = s_gpos[3 * ac], gpa1 = s_gpos[3 * ac + 1], gpa2 = s_gpos[3 * ac + 2];
    f4 gacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) gacc[t] = zero4;
    float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f;
    const float rc = a.r_cutoff, pi_rc = 3.14159265358979323846f / rc;
    // loop-node terms, fetched one iteration ahead: P of the source (dest-owned) or Q of the destination (source-owned),
    // and, source-owned, the cotangent of the destination's aggregated messages
    const float* const loop_base = (SRC_OWNED ? a.Q : a.P) + pq_b * FP;
    const float* const gnm_base = (SRC_OWNED && a.g_nm != nullptr) ? a.g_nm + (int64_t)b * n * FP : nullptr;
    f4 Ln[NT], Gn[NT];
    auto fetch_loop = [&](int i) {
        const int ic = i < n ? i : 0;
        const f4* p = reinterpret_cast<const f4*>(loop_base + (int64_t)ic * FP);
#pragma unroll
        for (int t = 0; t < NT; ++t) Ln[t] = p[4 * t + q];
        if (SRC_OWNED) {
            if (gnm_base != nullptr) {
                const f4* g = reinterpret_cast<const f4*>(gnm_base + (int64_t)ic * FP);
#pragma unroll
                for (int t = 0; t < NT; ++t) Gn[t] = g[4 * t + q];
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) Gn[t] = zero4;
            }
        }
    };
    fetch_loop(wave);
    for (int i = wave; i < n; i += EDGE_WAVES) {
        // v = x_dest - x_src (graph.py:254): the lane's node is the destination (dest-owned) or the source
        const float sg = SRC_OWNED ? -1.0f : 1.0f;
        const float v0 = sg * (xa0 - s_pos[3 * i]), v1 = sg * (xa1 - s_pos[3 * i + 1]), v2 = sg * (xa2 - s_pos[3 * i + 2]);
#if TFEP_EGNN_FAST_GEOM
        const float d = __builtin_amdgcn_sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
#else
        const float d = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
#endif
        const bool keep = a_ok && (an != i) && (d <= rc);
        if (__ballot(keep) == 0ull) {
            fetch_loop(i + EDGE_WAVES);
            continue;
        }
#if TFEP_EGNN_FAST_GEOM
        const float inv_d = keep ? fast_rcp(d) : 0.0f;
#else
        const float inv_d = keep ? 1.0f / d : 0.0f;
#endif
        const float u0 = v0 * inv_d, u1 = v1 * inv_d, u2 = v2 * inv_d;
        // cotangent of the DESTINATION's output position
        const float gp0 = SRC_OWNED ? s_gpos[3 * i] : gpa0, gp1 = SRC_OWNED ? s_gpos[3 * i + 1] : gpa1,
                    gp2 = SRC_OWNED ? s_gpos[3 * i + 2] : gpa2;
        float sn, cs;
#if TFEP_EGNN_FAST_GEOM
        {
            const float rev = fminf(d, rc) * (0.5f / rc);                 // (pi d / rc) / (2 pi)
            sn = __builtin_amdgcn_sinf(rev);
            cs = __builtin_amdgcn_cosf(rev);
        }
#else
        sincosf(pi_rc * d, &sn, &cs);
#endif
        const float sw = 0.5f * cs + 0.5f;
        const float dsw = -0.5f * pi_rc * sn;
        // ---- forward chain, keeping the activation derivatives (registers are the scarce resource of this kernel: the
        // radial derivatives are recomputed at the end rather than kept, the message cotangent is read where it is used)
        f4 z[NT], ds1[NT], ds2[NT], ds3[NT], a2[NT];
        {
            f4 rbf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f4 mu = v_mu[4 * t + q], ga = v_ga[4 * t + q];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dm = d - mu[r];
                    rbf[t][r] = fast_exp(-ga[r] * dm * dm) * sw;
                }
            }
            product(img_w1c, rbf, z, inv1, [&](int t) { return Ln[t] + s_lane[(4 * t + q) * 16 + c]; });
        }
        silu_keep<NT>(z, ds1);
        product(img_w2, z, a2, inv2, [&](int t) { return v_b2[4 * t + q]; });
        silu_keep<NT>(a2, ds2);
        float e = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f4 wa = v_wa[4 * t + q];
#pragma unroll
            for (int r = 0; r < 4; ++r) e += wa[r] * a2[t][r];
        }
        e = sum_over_q(e) + att_b;
#if TFEP_EGNN_FAST_GEOM
        const float sig_e = fast_rcp(1.0f + fast_exp(-e));
#else
        const float sig_e = 1.0f / (1.0f + expf(-e));
#endif
        const float att = keep ? sig_e : 0.0f;
        f4 m[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) m[t] = a2[t] * att;
        product(img_x1, m, z, inv3, [&](int t) { return v_d1[4 * t + q]; });
        silu_keep<NT>(z, ds3);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f4 x2 = v_x2[4 * t + q];
#pragma unroll
            for (int r = 0; r < 4; ++r) s += x2[r] * z[t][r];
        }
        #if TFEP_EGNN_FAST_GEOM
        const float mag = fast_tanh(sum_over_q(s));
#else
        const float mag = tanhf(sum_over_q(s));
#endif
        // ---- reverse chain
        const float cs_k = keep ? a.speed_factor : 0.0f;
        const float g_mag = cs_k * (u0 * gp0 + u1 * gp1 + u2 * gp2);
        const float gu0 = cs_k * mag * gp0, gu1 = cs_k * mag * gp1, gu2 = cs_k * mag * gp2;
        const float g_s = (1.0f - mag * mag) * g_mag;
        f4 gz[NT], gm[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f4 x2 = v_x2[4 * t + q];
#pragma unroll
            for (int r = 0; r < 4; ++r) gz[t][r] = ds3[t][r] * x2[r] * g_s;
        }
        // g_m = X1^T g_z3 + g_nm_dest
        product(img_x1T, gz, gm, inv3, [&](int t) { return SRC_OWNED ? Gn[t] : s_gnm[(4 * t + q) * 16 + c]; });
        fetch_loop(i + EDGE_WAVES);                 // the loop node's terms of the next iteration land under the rest
        float g_att = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) g_att += gm[t][r] * a2[t][r];
        const float ce = sum_over_q(g_att) * att * (1.0f - sig_e);             // d att / d e = att (1 - sigma(e))
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f4 wa = v_wa[4 * t + q];
#pragma unroll
            for (int r = 0; r < 4; ++r) gz[t][r] = ds2[t][r] * (att * gm[t][r] + wa[r] * ce);
        }
        product(img_w2T, gz, gm, inv2, zero_init);                               // g_a1 = W2^T g_z2
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            gz[t] = ds1[t] * gm[t];                                              // g_z1
            gacc[t] += gz[t];                                                    // g_Q_dest / g_P_src
        }
        product(img_w1cT, gz, gm, inv1, zero_init);                              // g_rbf = W1c^T g_z1
        float g_d = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f4 mu = v_mu[4 * t + q], ga = v_ga[4 * t + q];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                      // d rbf_k / d distance
                const float dm = d - mu[r];
                g_d += gm[t][r] * fast_exp(-ga[r] * dm * dm) * (dsw - 2.0f * ga[r] * dm * sw);
            }
        }
        g_d = sum_over_q(g_d);
        const float ug = u0 * gu0 + u1 * gu1 + u2 * gu2;
        const float gv0 = g_d * u0 + (gu0 - u0 * ug) * inv_d, gv1 = g_d * u1 + (gu1 - u1 * ug) * inv_d,
                    gv2 = g_d * u2 + (gu2 - u2 * ug) * inv_d;
        const float k = keep ? sg : 0.0f;                                        // + for the destination, - for the source
        gx0 += k * gv0; gx1 += k * gv1; gx2 += k * gv2;
    }

    // ---- cross-wave reduction in a fixed order, then the outputs of the lane-owned nodes
    constexpr int NM4 = NT * 64;
    f4* const r_acc = reinterpret_cast<f4*>(s_red);
    float* const r_gx = reinterpret_cast<float*>(r_acc + EDGE_WAVES * NM4);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) r_acc[wave * NM4 + t * 64 + lane] = gacc[t];
    if (q == 0) {
        float* p = r_gx + (wave * 16 + c) * 4;
        p[0] = gx0; p[1] = gx1; p[2] = gx2;
    }
    __syncthreads();
    for (int e = tid; e < NM4; e += EDGE_WAVES * 64) {
        f4 sum = r_acc[e];
#pragma unroll
        for (int w = 1; w < EDGE_WAVES; ++w) sum += r_acc[w * NM4 + e];
        const int t = e >> 6, l = e & 63;
        const int node = ab * 16 + (l & 15);
        if (node < n) reinterpret_cast<f4*>(a.g_lane + ((int64_t)b * n + node) * FP)[4 * t + (l >> 4)] = sum;
    }
    if (tid < 16 * 3) {
        const int comp = tid % 3, cc = tid / 3;
        const int node = ab * 16 + cc;
        if (node < n) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < EDGE_WAVES; ++w) sum += r_gx[(w * 16 + cc) * 4 + comp];
            float* dst = a.g_pos + ((int64_t)b * n + node) * 3 + comp;
            // dest-owned: the identity path pos' = pos + ... plus the destination-side terms; source-owned: adds its terms
            *dst = SRC_OWNED ? *dst + sum : s_gpos[3 * node + comp] + sum;
        }
    }
}

template <int NT, bool SPLIT>
size_t edge_bwd_lds_bytes(int n) {
    const size_t fp = 16 * NT, img4 = SPLIT ? (size_t)NT * ((NT + 1) / 2) * 64 * 2 : (size_t)NT * NT * 64;
    size_t fl = 6 * img4 * 4 + EDGE_CONST_VECS * fp + 2 * (size_t)((3 * n + 3) & ~3) + 2 * (size_t)NT * 64 * 4;
    fl += (size_t)EDGE_WAVES * NT * 64 * 4 + EDGE_WAVES * 16 * 4;
    return fl * sizeof(float);
}

// Reverse pass of the node update between layer l and l + 1 (egnn_node_kernel):
//   G = g_h' + W1a^T g_P' + W1b^T g_Q',   q = silu'(U1a h + U1b nm + c1) (U2^T G),   g_h = G + U1a^T q,   g_nm = U1b^T q
template <int NT>
__global__ __launch_bounds__(NODE_WAVES * 64) void egnn_node_bwd_kernel(tfep_egnn_node_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int FP = 16 * NT;
    constexpr int IMG4 = NT * NT * 64;
    const PackedLayout L = packed_layout(NT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, c = lane & 15;
    // LDS: [U1a, U1b][c1][U2^T, U1a^T, U1b^T] of this layer, [W1a^T, W1b^T] of the next
    f4* const fwd = reinterpret_cast<f4*>(smem);
    f4* const vc1 = fwd + 2 * IMG4;
    f4* const rev = vc1 + FP / 4;
    f4* const nxt = rev + 3 * IMG4;
    {
        const f4* s0 = reinterpret_cast<const f4*>(a.packed + L.u1a);
        for (int i = tid; i < 2 * IMG4; i += NODE_WAVES * 64) fwd[i] = s0[i];
        const f4* s1 = reinterpret_cast<const f4*>(a.packed + L.c1);
        for (int i = tid; i < FP / 4; i += NODE_WAVES * 64) vc1[i] = s1[i];
        const f4* s2 = reinterpret_cast<const f4*>(a.packed + L.u2T);
        for (int i = tid; i < 3 * IMG4; i += NODE_WAVES * 64) rev[i] = s2[i];
        const f4* s3 = reinterpret_cast<const f4*>(a.packed_next + L.w1aT);
        for (int i = tid; i < 2 * IMG4; i += NODE_WAVES * 64) nxt[i] = s3[i];
    }
    __syncthreads();
    const f4 *img_u1a = fwd, *img_u1b = fwd + IMG4;
    const f4 *img_u2T = rev, *img_u1aT = rev + IMG4, *img_u1bT = rev + 2 * IMG4;
    const f4 *img_w1aT = nxt, *img_w1bT = nxt + IMG4;
    const int64_t n_total = (int64_t)a.B * a.n_nodes;
    const int64_t n_groups = (n_total + 15) / 16;
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    for (int64_t g = (int64_t)blockIdx.x * NODE_WAVES + wave; g < n_groups; g += (int64_t)gridDim.x * NODE_WAVES) {
        const int64_t node = g * 16 + c;
        const bool ok = node < n_total;
        const int64_t nd = ok ? node : 0;
        const int64_t hb = (nd / a.n_nodes) * a.h_bstride + (nd % a.n_nodes);
        f4 h[NT], m[NT], gP[NT], gQ[NT], G[NT], dummy[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            h[t] = reinterpret_cast<const f4*>(a.h + hb * FP)[4 * t + q];
            m[t] = reinterpret_cast<const f4*>(a.nm + nd * FP)[4 * t + q];
            gP[t] = reinterpret_cast<const f4*>(a.g_P + nd * FP)[4 * t + q];
            gQ[t] = reinterpret_cast<const f4*>(a.g_Q + nd * FP)[4 * t + q];
            G[t] = a.g_h_next != nullptr ? reinterpret_cast<const f4*>(a.g_h_next + nd * FP)[4 * t + q] : zero4;
        }
        chain_gemm<NT, false>(img_w1aT, gP, gP, G, dummy, lane);
        chain_gemm<NT, false>(img_w1bT, gQ, gQ, G, dummy, lane);
        f4 z[NT], ds[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) z[t] = vc1[4 * t + q];
        chain_gemm<NT, false>(img_u1a, h, h, z, dummy, lane);
        chain_gemm<NT, false>(img_u1b, m, m, z, dummy, lane);
        silu_keep<NT>(z, ds);
        f4 tt[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) tt[t] = zero4;
        chain_gemm<NT, false>(img_u2T, G, G, tt, dummy, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t) tt[t] = tt[t] * ds[t];
        f4 gnm[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) gnm[t] = zero4;
        chain_gemm<NT, false>(img_u1aT, tt, tt, G, dummy, lane);             // g_h = G + U1a^T q
        chain_gemm<NT, false>(img_u1bT, tt, tt, gnm, dummy, lane);           // g_nm = U1b^T q
        if (ok) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                reinterpret_cast<f4*>(a.g_h + node * FP)[4 * t + q] = G[t];
                reinterpret_cast<f4*>(a.g_nm + node * FP)[4 * t + q] = gnm[t];
            }
        }
    }
}

// out = in - mean over the nodes (per sample and component): the centring of the velocity (egnn.py:187-191) is a symmetric
// projector, so this is also its reverse pass.
__global__ void egnn_center_kernel(const float* __restrict__ in, int n_nodes, float sign, float* __restrict__ out) {
    __shared__ double red[4][4];
    __shared__ double mean[3];
    const int b = blockIdx.x, D = 3 * n_nodes;
    const float* p = in + (int64_t)b * D;
    double s[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < n_nodes; i += blockDim.x)
        for (int k = 0; k < 3; ++k) s[k] += (double)p[3 * i + k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (int k = 0; k < 3; ++k) {
        const double v = wave_sum(s[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double tot = 0;
        for (int w = 0; w < n_waves; ++w) tot += red[w][threadIdx.x];
        mean[threadIdx.x] = tot / n_nodes;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += blockDim.x) out[(int64_t)b * D + i] = sign * (float)((double)p[i] - mean[i % 3]);
}

// per row: dot += scale a . b, sumsq += scale |a|^2  (the quadratic forms of continuous.py:307-361 from e^T J)
__global__ void egnn_rowdots_kernel(const float* __restrict__ x, const float* __restrict__ y, int D, float scale,
                                    float* __restrict__ dot, float* __restrict__ sumsq) {
    __shared__ double red[4][2];
    const int b = blockIdx.x;
    double s0 = 0, s1 = 0;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const double a = x[(int64_t)b * D + i];
        s0 += a * (double)y[(int64_t)b * D + i];
        s1 += a * a;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0, t1 = 0;
        for (int w = 0; w < n_waves; ++w) { t0 += red[w][0]; t1 += red[w][1]; }
        if (dot != nullptr) dot[b] += scale * (float)t0;
        if (sumsq != nullptr) sumsq[b] += scale * (float)t1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// velocity (egnn.py:178-193): vel = (pos_L - x) - mean over nodes; with a tangent also d vel = (dpos_L - e) - mean and
// the quadratic forms the trace estimators need: e . (J e) and |J e|^2 (continuous.py:307-324, :285-304).
// One workgroup per sample.
// ---------------------------------------------------------------------------------------------------------------
__global__ void egnn_finish_kernel(const float* __restrict__ pos, const float* __restrict__ x,
                                   const float* __restrict__ dpos, const float* __restrict__ eps, int n_nodes,
                                   float* __restrict__ vel, float* __restrict__ jvp, float scale,
                                   float* __restrict__ trace, float* __restrict__ frob, float* __restrict__ vel_sq) {
    __shared__ double red[8][4];
    const int b = blockIdx.x, D = 3 * n_nodes;
    const float* p = pos + (int64_t)b * D;
    const float* xx = x + (int64_t)b * D;
    const bool tan = dpos != nullptr;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n_nodes; i += blockDim.x) {
        for (int k = 0; k < 3; ++k) {
            s[k] += (double)(p[3 * i + k] - xx[3 * i + k]);
            if (tan) s[3 + k] += (double)(dpos[(int64_t)b * D + 3 * i + k] - eps[(int64_t)b * D + 3 * i + k]);
        }
    }
    __shared__ double mean[6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (int k = 0; k < 6; ++k) {
        const double v = wave_sum(s[k]);
        if (lane == 0) red[wave][k & 3] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot = 0;
            for (int w = 0; w < n_waves; ++w) tot += red[w][k & 3];
            mean[k] = tot / n_nodes;
        }
        __syncthreads();
    }
    double tr = 0, fr = 0, vs = 0;
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
        const int k = i % 3;
        const float v = (float)((double)(p[i] - xx[i]) - mean[k]);
        if (vel != nullptr) vel[(int64_t)b * D + i] = v;
        vs += (double)v * v;
        if (tan) {
            const float e = eps[(int64_t)b * D + i];
            const float dv = (float)((double)(dpos[(int64_t)b * D + i] - e) - mean[3 + k]);
            if (jvp != nullptr) jvp[(int64_t)b * D + i] = dv;
            tr += (double)e * dv;
            fr += (double)dv * dv;
        }
    }
    double out[3] = {tr, fr, vs};
    for (int k = 0; k < 3; ++k) {
        const double v = wave_sum(out[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0, t1 = 0, t2 = 0;
        for (int w = 0; w < n_waves; ++w) { t0 += red[w][0]; t1 += red[w][1]; t2 += red[w][2]; }
        if (tan && trace != nullptr) trace[b] += scale * (float)t0;
        if (tan && frob != nullptr) frob[b] += scale * (float)t1;
        if (vel_sq != nullptr) vel_sq[b] = (float)t2;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// stand-alone helpers of the public API
// ---------------------------------------------------------------------------------------------------------------
// Gaussian basis (radial.py:110-130) with the optional Behler-Parrinello cosine switch (radial.py:161-176, 269-291)
__global__ void radial_kernel(const float* __restrict__ r, int64_t n, const float* __restrict__ means,
                              const float* __restrict__ log_gammas, int n_basis, float r_cutoff, int switching,
                              int force_zero, float* __restrict__ out) {
    const int64_t total = n * n_basis;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / n_basis;
        const int k = (int)(i % n_basis);
        const float d = r[e], dm = d - means[k];
        float v = expf(-expf(log_gammas[k]) * dm * dm);
        if (switching) {
            float sw = 0.5f * cosf(3.14159265358979323846f / r_cutoff * d) + 0.5f;
            if (force_zero && d > r_cutoff) sw = 0.0f;
            v *= sw;
        }
        out[i] = v;
    }
}

// scatter-add of rows into segments (graph.py:304-316); float atomics: the order of the additions is not fixed
__global__ void segment_sum_kernel(const float* __restrict__ data, const int64_t* __restrict__ ids, int64_t n_rows,
                                   int n_cols, int64_t n_segments, float* __restrict__ out) {
    const int64_t total = n_rows * n_cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / n_cols, seg = ids[row];
        if (seg >= 0 && seg < n_segments) atomicAdd(out + seg * n_cols + (i % n_cols), data[i]);
    }
}

__global__ void fill_kernel(float* __restrict__ p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// y = x + sum_k a_k v_k  (the stage combinations of the fixed-grid Runge-Kutta steppers)
__global__ void axpy_kernel(const float* __restrict__ x, const float* v0, const float* v1, const float* v2,
                            const float* v3, float a0, float a1, float a2, float a3, int64_t n, float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float acc = x != nullptr ? x[i] : 0.0f;
        if (v0 != nullptr) acc += a0 * v0[i];
        if (v1 != nullptr) acc += a1 * v1[i];
        if (v2 != nullptr) acc += a2 * v2[i];
        if (v3 != nullptr) acc += a3 * v3[i];
        y[i] = acc;
    }
}

inline int grid_for(int64_t n, int block) { return (int)std::min<int64_t>((n + block - 1) / block, 65536); }

template <typename K>
int allow_lds(K kernel, size_t bytes, const char* what) {
    if (bytes > 160 * 1024) return fail(TFEP_ERR_UNSUPPORTED, "%s needs %zu bytes of LDS (> 160 KiB)", what, bytes);
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    }
    return TFEP_OK;
}

template <int NT, bool TAN, bool SPLIT>
int launch_edge_impl(const tfep_egnn_edge_args& a, hipStream_t st) {
    const size_t lds = edge_lds_bytes<NT, TAN, SPLIT>(a.n_nodes);
    int rc = allow_lds(egnn_edge_kernel<NT, TAN, SPLIT>, lds, "tfep_egnn_edge");
    if (rc != TFEP_OK) return rc;
    const int64_t blocks = (int64_t)a.B * ((a.n_nodes + 15) / 16);
    TFEP_REQUIRE(blocks < (1ll << 31), "tfep_egnn_edge: too many workgroups");
    hipLaunchKernelGGL((egnn_edge_kernel<NT, TAN, SPLIT>), dim3((unsigned)blocks), dim3(EDGE_WAVES * 64), lds, st, a);
    return check_launch("tfep_egnn_edge");
}

template <int NT, bool TAN>
int launch_edge(const tfep_egnn_edge_args& a, hipStream_t st) {
    return a.split ? launch_edge_impl<NT, TAN, true>(a, st) : launch_edge_impl<NT, TAN, false>(a, st);
}

template <int NT, bool TAN>
int launch_node(const tfep_egnn_node_args& a, hipStream_t st) {
    const size_t lds = ((size_t)5 * NT * NT * 64 * 4 + 3 * 16 * NT) * sizeof(float);
    int rc = allow_lds(egnn_node_kernel<NT, TAN>, lds, "tfep_egnn_node");
    if (rc != TFEP_OK) return rc;
    const int64_t groups = ((int64_t)a.B * a.n_nodes + 15) / 16;
    const int blocks = (int)std::min<int64_t>((groups + NODE_WAVES - 1) / NODE_WAVES, 256 * 8);
    hipLaunchKernelGGL((egnn_node_kernel<NT, TAN>), dim3(blocks), dim3(NODE_WAVES * 64), lds, st, a);
    return check_launch("tfep_egnn_node");
}

template <int NT, bool SRC_OWNED, bool SPLIT>
int launch_edge_bwd_impl(const tfep_egnn_edge_bwd_args& a, hipStream_t st) {
    const size_t lds = edge_bwd_lds_bytes<NT, SPLIT>(a.n_nodes);
    int rc = allow_lds(egnn_edge_bwd_kernel<NT, SRC_OWNED, SPLIT>, lds, "tfep_egnn_edge_backward");
    if (rc != TFEP_OK) return rc;
    const int64_t blocks = (int64_t)a.B * ((a.n_nodes + 15) / 16);
    TFEP_REQUIRE(blocks < (1ll << 31), "tfep_egnn_edge_backward: too many workgroups");
    hipLaunchKernelGGL((egnn_edge_bwd_kernel<NT, SRC_OWNED, SPLIT>), dim3((unsigned)blocks), dim3(EDGE_WAVES * 64), lds, st, a);
    return check_launch("tfep_egnn_edge_backward");
}

template <int NT>
int launch_edge_bwd(const tfep_egnn_edge_bwd_args& a, hipStream_t st) {
    if (a.src_owned) return a.split ? launch_edge_bwd_impl<NT, true, true>(a, st) : launch_edge_bwd_impl<NT, true, false>(a, st);
    return a.split ? launch_edge_bwd_impl<NT, false, true>(a, st) : launch_edge_bwd_impl<NT, false, false>(a, st);
}

template <int NT>
int launch_node_bwd(const tfep_egnn_node_bwd_args& a, hipStream_t st) {
    const size_t lds = ((size_t)7 * NT * NT * 64 * 4 + 16 * NT) * sizeof(float);
    int rc = allow_lds(egnn_node_bwd_kernel<NT>, lds, "tfep_egnn_node_backward");
    if (rc != TFEP_OK) return rc;
    const int64_t groups = ((int64_t)a.B * a.n_nodes + 15) / 16;
    const int blocks = (int)std::min<int64_t>((groups + NODE_WAVES - 1) / NODE_WAVES, 256 * 8);
    hipLaunchKernelGGL((egnn_node_bwd_kernel<NT>), dim3(blocks), dim3(NODE_WAVES * 64), lds, st, a);
    return check_launch("tfep_egnn_node_backward");
}

}  // namespace
}  // namespace tfep

using namespace tfep;

extern "C" {

int tfep_egnn_tile(int node_feat_dim, int distance_feat_dim) {
    const int m = node_feat_dim > distance_feat_dim ? node_feat_dim : distance_feat_dim;
    if (node_feat_dim < 1 || distance_feat_dim < 1 || m > 64) return 0;
    const int nt = (m + 15) / 16;
    return nt == 3 ? 4 : nt;
}

int64_t tfep_egnn_packed_floats(int nt) {
    if (nt != 1 && nt != 2 && nt != 4) return -1;
    return packed_layout(nt).total;
}

int tfep_egnn_pack_layer(const tfep_egnn_layer_params* p, int nt, float* packed, void* stream) {
    TFEP_REQUIRE(p != nullptr && packed != nullptr, "tfep_egnn_pack_layer: null argument");
    TFEP_REQUIRE(nt == 1 || nt == 2 || nt == 4, "tfep_egnn_pack_layer: nt must be 1, 2 or 4");
    TFEP_REQUIRE(p->F >= 1 && p->G >= 1 && p->F <= 16 * nt && p->G <= 16 * nt,
                 "tfep_egnn_pack_layer: feature sizes (%d, %d) do not fit a tile of %d", p->F, p->G, 16 * nt);
    TFEP_REQUIRE(p->dist_means && p->dist_log_gammas && p->msg0_w && p->msg0_b && p->msg2_w && p->msg2_b && p->att_w &&
                 p->att_b && p->ux0_w && p->ux0_b && p->ux2_w && p->uh0_w && p->uh0_b && p->uh2_w && p->uh2_b,
                 "tfep_egnn_pack_layer: null parameter tensor");
    const PackedLayout L = packed_layout(nt);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(egnn_scale_kernel, dim3(3), dim3(256), 0, st, *p, nt, packed);
    hipLaunchKernelGGL(egnn_pack_kernel, dim3(grid_for(L.s_w1c, 256)), dim3(256), 0, st, *p, nt, packed);
    hipLaunchKernelGGL(egnn_pack_split_kernel, dim3(grid_for(6 * split_img_floats(nt), 256)), dim3(256), 0, st, *p, nt, packed);
    return check_launch("tfep_egnn_pack_layer");
}

int tfep_egnn_embed(const float* one_hot, int n_nodes, int n_types, float t, const float* time_means,
                    const float* time_log_gammas, int time_dim, const float* w_emb, const float* b_emb,
                    const tfep_egnn_layer_params* layer0, int nt, float* h0, float* P0, float* Q0, void* stream) {
    TFEP_REQUIRE(one_hot && time_means && time_log_gammas && w_emb && b_emb && layer0 && h0 && P0 && Q0,
                 "tfep_egnn_embed: null argument");
    TFEP_REQUIRE(n_nodes >= 1 && n_types >= 1 && time_dim >= 1, "tfep_egnn_embed: bad sizes");
    TFEP_REQUIRE(nt == 1 || nt == 2 || nt == 4, "tfep_egnn_embed: nt must be 1, 2 or 4");
    TFEP_REQUIRE(layer0->F <= 16 * nt, "tfep_egnn_embed: node_feat_dim %d does not fit the tile", layer0->F);
    const int fp = 16 * nt;
    hipLaunchKernelGGL(egnn_embed_kernel, dim3(n_nodes), dim3(64), fp * sizeof(float), (hipStream_t)stream, one_hot, n_nodes,
                       n_types, t, time_means, time_log_gammas, time_dim, w_emb, b_emb, layer0->F, layer0->msg0_w,
                       layer0->msg0_b, layer0->G, fp, h0, P0, Q0);
    return check_launch("tfep_egnn_embed");
}

int tfep_egnn_edge(const tfep_egnn_edge_args* a, void* stream) {
    TFEP_REQUIRE(a != nullptr, "tfep_egnn_edge: null argument");
    TFEP_REQUIRE(a->B >= 1 && a->n_nodes >= 1 && a->n_nodes <= 4096, "tfep_egnn_edge: bad sizes (B=%d, n_nodes=%d)", a->B,
                 a->n_nodes);
    TFEP_REQUIRE(a->packed && a->pos && a->P && a->Q && a->pos_out, "tfep_egnn_edge: null tensor");
    TFEP_REQUIRE(a->r_cutoff > 0.0f, "tfep_egnn_edge: r_cutoff must be positive");
    TFEP_REQUIRE(a->pq_bstride == 0 || a->pq_bstride == a->n_nodes, "tfep_egnn_edge: pq_bstride must be 0 or n_nodes");
    const bool tan = a->dpos != nullptr;
    if (tan) {
        TFEP_REQUIRE(a->dpos_out != nullptr, "tfep_egnn_edge: dpos_out missing");
        TFEP_REQUIRE((a->dP == nullptr) == (a->dQ == nullptr), "tfep_egnn_edge: dP and dQ go together");
        TFEP_REQUIRE(a->nm == nullptr || a->dnm != nullptr, "tfep_egnn_edge: dnm missing");
    }
    hipStream_t st = (hipStream_t)stream;
    switch (a->nt * 2 + (tan ? 1 : 0)) {
        case 2: return launch_edge<1, false>(*a, st);
        case 3: return launch_edge<1, true>(*a, st);
        case 4: return launch_edge<2, false>(*a, st);
        case 5: return launch_edge<2, true>(*a, st);
        case 8: return launch_edge<4, false>(*a, st);
        case 9: return launch_edge<4, true>(*a, st);
    }
    return fail(TFEP_ERR_INVALID_ARGUMENT, "tfep_egnn_edge: nt must be 1, 2 or 4 (got %d)", a->nt);
}

int tfep_egnn_node(const tfep_egnn_node_args* a, void* stream) {
    TFEP_REQUIRE(a != nullptr, "tfep_egnn_node: null argument");
    TFEP_REQUIRE(a->B >= 1 && a->n_nodes >= 1, "tfep_egnn_node: bad sizes");
    TFEP_REQUIRE(a->packed && a->packed_next && a->h && a->nm && a->h_out && a->P_out && a->Q_out,
                 "tfep_egnn_node: null tensor");
    TFEP_REQUIRE(a->h_bstride == 0 || a->h_bstride == a->n_nodes, "tfep_egnn_node: h_bstride must be 0 or n_nodes");
    const bool tan = a->dnm != nullptr;
    if (tan) TFEP_REQUIRE(a->dh_out && a->dP_out && a->dQ_out, "tfep_egnn_node: tangent outputs missing");
    hipStream_t st = (hipStream_t)stream;
    switch (a->nt * 2 + (tan ? 1 : 0)) {
        case 2: return launch_node<1, false>(*a, st);
        case 3: return launch_node<1, true>(*a, st);
        case 4: return launch_node<2, false>(*a, st);
        case 5: return launch_node<2, true>(*a, st);
        case 8: return launch_node<4, false>(*a, st);
        case 9: return launch_node<4, true>(*a, st);
    }
    return fail(TFEP_ERR_INVALID_ARGUMENT, "tfep_egnn_node: nt must be 1, 2 or 4 (got %d)", a->nt);
}

int tfep_egnn_finish(const float* pos, const float* x, const float* dpos, const float* eps, int B, int n_nodes,
                     float* vel, float* jvp, float scale, float* trace, float* frob, float* vel_sq, void* stream) {
    TFEP_REQUIRE(pos && x && B >= 1 && n_nodes >= 1, "tfep_egnn_finish: bad arguments");
    TFEP_REQUIRE((dpos == nullptr) == (eps == nullptr), "tfep_egnn_finish: dpos and eps go together");
    const int block = n_nodes >= 256 ? 256 : 64;
    hipLaunchKernelGGL(egnn_finish_kernel, dim3(B), dim3(block), 0, (hipStream_t)stream, pos, x, dpos, eps, n_nodes, vel,
                       jvp, scale, trace, frob, vel_sq);
    return check_launch("tfep_egnn_finish");
}

int tfep_egnn_edge_backward(const tfep_egnn_edge_bwd_args* a, void* stream) {
    TFEP_REQUIRE(a != nullptr, "tfep_egnn_edge_backward: null argument");
    TFEP_REQUIRE(a->B >= 1 && a->n_nodes >= 1 && a->n_nodes <= 4096, "tfep_egnn_edge_backward: bad sizes (B=%d, n_nodes=%d)",
                 a->B, a->n_nodes);
    TFEP_REQUIRE(a->packed && a->pos && a->P && a->Q && a->g_pos_out && a->g_lane && a->g_pos,
                 "tfep_egnn_edge_backward: null tensor");
    TFEP_REQUIRE(a->r_cutoff > 0.0f, "tfep_egnn_edge_backward: r_cutoff must be positive");
    TFEP_REQUIRE(a->pq_bstride == 0 || a->pq_bstride == a->n_nodes, "tfep_egnn_edge_backward: pq_bstride must be 0 or n_nodes");
    hipStream_t st = (hipStream_t)stream;
    switch (a->nt) {
        case 1: return launch_edge_bwd<1>(*a, st);
        case 2: return launch_edge_bwd<2>(*a, st);
        case 4: return launch_edge_bwd<4>(*a, st);
    }
    return fail(TFEP_ERR_INVALID_ARGUMENT, "tfep_egnn_edge_backward: nt must be 1, 2 or 4 (got %d)", a->nt);
}

int tfep_egnn_node_backward(const tfep_egnn_node_bwd_args* a, void* stream) {
    TFEP_REQUIRE(a != nullptr, "tfep_egnn_node_backward: null argument");
    TFEP_REQUIRE(a->B >= 1 && a->n_nodes >= 1, "tfep_egnn_node_backward: bad sizes");
    TFEP_REQUIRE(a->packed && a->packed_next && a->h && a->nm && a->g_P && a->g_Q && a->g_h && a->g_nm,
                 "tfep_egnn_node_backward: null tensor");
    TFEP_REQUIRE(a->h_bstride == 0 || a->h_bstride == a->n_nodes, "tfep_egnn_node_backward: h_bstride must be 0 or n_nodes");
    hipStream_t st = (hipStream_t)stream;
    switch (a->nt) {
        case 1: return launch_node_bwd<1>(*a, st);
        case 2: return launch_node_bwd<2>(*a, st);
        case 4: return launch_node_bwd<4>(*a, st);
    }
    return fail(TFEP_ERR_INVALID_ARGUMENT, "tfep_egnn_node_backward: nt must be 1, 2 or 4 (got %d)", a->nt);
}

int tfep_egnn_center(const float* in, int B, int n_nodes, float sign, float* out, void* stream) {
    TFEP_REQUIRE(in && out && B >= 1 && n_nodes >= 1, "tfep_egnn_center: bad arguments");
    hipLaunchKernelGGL(egnn_center_kernel, dim3(B), dim3(n_nodes >= 256 ? 256 : 64), 0, (hipStream_t)stream, in, n_nodes,
                       sign, out);
    return check_launch("tfep_egnn_center");
}

int tfep_row_dots(const float* x, const float* y, int B, int D, float scale, float* dot, float* sumsq, void* stream) {
    TFEP_REQUIRE(x && y && B >= 1 && D >= 1, "tfep_row_dots: bad arguments");
    hipLaunchKernelGGL(egnn_rowdots_kernel, dim3(B), dim3(D >= 256 ? 256 : 64), 0, (hipStream_t)stream, x, y, D, scale, dot,
                       sumsq);
    return check_launch("tfep_row_dots");
}

int tfep_radial_expansion(const float* r, int64_t n, const float* means, const float* log_gammas, int n_basis,
                          float r_cutoff, int switching, int force_zero_after_cutoff, float* out, void* stream) {
    TFEP_REQUIRE(n >= 0 && n_basis >= 1, "tfep_radial_expansion: bad sizes");
    if (n == 0) return TFEP_OK;
    TFEP_REQUIRE(r && means && log_gammas && out, "tfep_radial_expansion: null tensor");
    TFEP_REQUIRE(!switching || r_cutoff > 0.0f, "tfep_radial_expansion: r_cutoff must be positive");
    hipLaunchKernelGGL(radial_kernel, dim3(grid_for(n * n_basis, 256)), dim3(256), 0, (hipStream_t)stream, r, n, means,
                       log_gammas, n_basis, r_cutoff, switching, force_zero_after_cutoff, out);
    return check_launch("tfep_radial_expansion");
}

int tfep_segment_sum(const float* data, const int64_t* segment_ids, int64_t n_rows, int n_cols, int64_t n_segments,
                     float* out, void* stream) {
    TFEP_REQUIRE(n_rows >= 0 && n_cols >= 1 && n_segments >= 0, "tfep_segment_sum: bad sizes");
    if (n_segments == 0) return TFEP_OK;
    TFEP_REQUIRE(out != nullptr, "tfep_segment_sum: null output");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n_segments * n_cols, 256)), dim3(256), 0, (hipStream_t)stream, out,
                       n_segments * n_cols, 0.0f);
    if (n_rows == 0) return check_launch("tfep_segment_sum");
    TFEP_REQUIRE(data && segment_ids, "tfep_segment_sum: null tensor");
    hipLaunchKernelGGL(segment_sum_kernel, dim3(grid_for(n_rows * n_cols, 256)), dim3(256), 0, (hipStream_t)stream, data,
                       segment_ids, n_rows, n_cols, n_segments, out);
    return check_launch("tfep_segment_sum");
}

int tfep_ode_axpy(const float* x, const float* const* v, const float* a, int n_terms, int64_t n, float* y, void* stream) {
    TFEP_REQUIRE(n >= 0 && n_terms >= 0 && n_terms <= 4 && y != nullptr, "tfep_ode_axpy: bad arguments");
    if (n == 0) return TFEP_OK;
    const float* vv[4] = {nullptr, nullptr, nullptr, nullptr};
    float aa[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < n_terms; ++k) {
        TFEP_REQUIRE(v != nullptr && a != nullptr && v[k] != nullptr, "tfep_ode_axpy: null term");
        vv[k] = v[k];
        aa[k] = a[k];
    }
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, vv[0], vv[1], vv[2], vv[3],
                       aa[0], aa[1], aa[2], aa[3], n, y);
    return check_launch("tfep_ode_axpy");
}

}  // extern "C"
