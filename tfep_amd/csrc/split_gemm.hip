// Split-precision masked-linear GEMM for gfx950: fp32 operands carried as two fp16 halves, three
// v_mfma_f32_16x16x32_f16 per product (hi*hi + lo*hi + hi*lo, fp32 accumulate).
//
// Why: the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) peaks at 157 TFLOP/s on MI355X; the fp16 MFMA at 2.5 PFLOP/s.
// Three fp16 MFMAs per fp32 product leave a 5.3x higher matrix-pipe ceiling, and the result is fp32-equivalent:
//   v * s = hi + lo + r,  hi = fp16(v * s),  lo = fp16(v * s - hi),  |r| <= 2^-22 |v * s|
// with s a power of two chosen per activation row / per weight matrix so that max |v * s| lies in [2^14, 2^15)
// (fp16 never overflows and the low half of every element within 2^-19 of the maximum stays normal).  hi*hi,
// lo*hi and hi*lo are exact in the fp32 accumulator; only lo*lo (2^-22 relative) is dropped.  Products are
// un-scaled in the epilogue by the exact powers of two.
//
// Operand format ("split rows"): same pitch as the fp32 matrix (4 bytes per element); per row and per group of
// 8 consecutive k:  [8 x fp16 hi][8 x fp16 lo]  (32 bytes).  Written by tfep_split_rows.
//
// Tiling (one workgroup = 4 wavefronts = one per SIMD, 512 registers each, one workgroup per CU):
//   workgroup tile 256 x (16 * NREP), BK = 32; wave w owns rows [64w, 64w + 64) and all columns as 4 x NREP
//   tiles of 16x16x32 (accumulators: 16 * NREP registers).  Per k-tile a wave reads its A fragments once
//   (8 ds_read_b128) and streams the B fragments (2 ds_read_b128 per column group for 12 MFMAs).
//   LDS: weights double-buffered (2 x 16*NREP x 128 B), activations single-buffered and PRIVATE to the wave
//   that owns the rows (8 KB each) -- a wave refills its own A region right after reading its fragments, so the
//   only workgroup barrier per k-tile is the one that publishes the weight tile.
//   LDS image: row r of a k-tile is 128 B = 8 parts of 16 B (part 2g + h = half h of k-group g); part p sits at
//   position p ^ swz(r), swz(r) = (e & 1) | (e & 4), e = (r >> 1) & 7, which makes both ds_read_b128 of a fragment
//   conflict-free for the hardware's 16-lane groups.  The LDS-DMA writes the image directly: lane i of a DMA
//   instruction fills position i & 7 of row i >> 3 of an 8-row chunk and fetches the part that belongs there.
#include "gemm_common.h"

#include <type_traits>

namespace tfep {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int SBK = 32;
constexpr int SWAVES = 4, STHREADS = SWAVES * 64, SMREP = 4;
constexpr int ROW_BYTES = 128;                    // one row of one k-tile: 32 k x (hi, lo)
constexpr int A_WAVE_BYTES = 16 * SMREP * ROW_BYTES;

template <int NREP>
struct STile {
    static constexpr int BM = SWAVES * 16 * SMREP;
    static constexpr int BN = 16 * NREP;
    static constexpr int A_BYTES = SWAVES * A_WAVE_BYTES;
    static constexpr int B_BYTES = BN * ROW_BYTES;
    static constexpr int LDS_BYTES = A_BYTES + 2 * B_BYTES;
    static constexpr int A_DMA = A_WAVE_BYTES / 1024;               // DMA instructions per wave per k-tile
    static constexpr int B_CHUNKS = BN / 8;
    static constexpr int B_DMA = (B_CHUNKS + SWAVES - 1) / SWAVES;
    static constexpr int N_DMA = A_DMA + B_DMA;
};

// ------------------------------------------------------------------------------------------
// fp32 rows -> split rows
// ------------------------------------------------------------------------------------------
__device__ inline float pow2_scale_for(float amax) {
    // amax * s in [2^14, 2^15); 1 for an all-zero / non-finite row
    if (!(amax > 0.f) || !(amax < INFINITY)) return 1.0f;
    int ex;
    frexpf(amax, &ex);                       // amax = f * 2^ex, f in [0.5, 1)
    int k = 15 - ex;
    k = k > 100 ? 100 : k;
    return ldexpf(1.0f, k);
}

__global__ void zero_u32_kernel(uint32_t* p, int n = 1) {
    for (int i = 0; i < n; ++i) p[i] = 0u;
}

__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int64_t cols,
                                                     uint32_t* __restrict__ out_bits) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* sr = src + row * ld;
    float m = 0.f;
    if (((uintptr_t)sr & 15u) == 0) {                  // 16-byte loads (a maximum: the order of the comparisons does not matter)
        const float4* s4 = reinterpret_cast<const float4*>(sr);
        const int64_t n4 = cols >> 2;
        for (int64_t i = lane; i < n4; i += 64) {
            const float4 q = s4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
        }
        for (int64_t i = 4 * n4 + lane; i < cols; i += 64) m = fmaxf(m, fabsf(sr[i]));
    } else {
        for (int64_t i = lane; i < cols; i += 64) m = fmaxf(m, fabsf(sr[i]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    // non-negative floats order like their bits; look before the atomic (same-address atomics serialise in L2)
    if (lane == 0 && __float_as_uint(m) > __atomic_load_n(out_bits, __ATOMIC_RELAXED)) atomicMax(out_bits, __float_as_uint(m));
}

// One wave per row.  per_tensor: the scale comes from *tensor_max_bits (absmax_kernel) and inv_scale[0] is written
// once; otherwise each row gets its own scale and inv_scale[row].
__global__ void __launch_bounds__(256) split_rows_kernel(const float* __restrict__ src, int64_t ld_src, int64_t rows,
                                                         int64_t cols, uint4* __restrict__ dst, int64_t ld_dst,
                                                         int64_t cols_padded, float* __restrict__ inv_scale,
                                                         const uint32_t* __restrict__ tensor_max_bits) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* sr = src + row * ld_src;
    const bool vec = ((uintptr_t)sr & 15u) == 0;       // wave-uniform: 16-byte loads when the row starts on a 16-byte boundary
    float amax;
    if (tensor_max_bits) {
        amax = __uint_as_float(*tensor_max_bits);
    } else {
        amax = 0.f;
        if (vec) {                                // (a maximum: the order of the comparisons does not matter)
            const float4* s4 = reinterpret_cast<const float4*>(sr);
            const int64_t n4 = cols >> 2;
            for (int64_t i = lane; i < n4; i += 64) {
                const float4 q = s4[i];
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
            }
            for (int64_t i = 4 * n4 + lane; i < cols; i += 64) amax = fmaxf(amax, fabsf(sr[i]));
        } else {
            for (int64_t i = lane; i < cols; i += 64) amax = fmaxf(amax, fabsf(sr[i]));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    }
    const float s = pow2_scale_for(amax);
    if (lane == 0 && (!tensor_max_bits || row == 0)) inv_scale[tensor_max_bits ? 0 : row] = 1.0f / s;
    uint4* dr = dst + row * (ld_dst / 4);        // 16 bytes = 4 elements of pitch
    for (int64_t g8 = lane; g8 * 8 < cols_padded; g8 += 64) {
        f16x8 hi, lo;
        float e[8];
        if (vec && g8 * 8 + 8 <= cols) {          // two 16-byte loads per group (element loads: 8 instructions at a 32-byte lane stride)
            const float4 a = reinterpret_cast<const float4*>(sr)[g8 * 2], b = reinterpret_cast<const float4*>(sr)[g8 * 2 + 1];
            e[0] = a.x; e[1] = a.y; e[2] = a.z; e[3] = a.w; e[4] = b.x; e[5] = b.y; e[6] = b.z; e[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int64_t c = g8 * 8 + j;
                e[j] = c < cols ? sr[c] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = e[j] * s;
            const _Float16 h = (_Float16)v;
            hi[j] = h;
            lo[j] = (_Float16)(v - (float)h);
        }
        dr[g8 * 2] = *reinterpret_cast<uint4*>(&hi);
        dr[g8 * 2 + 1] = *reinterpret_cast<uint4*>(&lo);
    }
}

// out (C rows of split groups: per 8 consecutive r  [8 x fp16 hi][8 x fp16 lo], groups up to R_pad) = split(in^T) for
// in (R x C fp32), rows r >= R read as zero.  One pass instead of transpose + absmax + split_rows.  The scale:
//   mode 0: *scale_src = 1 / s is known (a matrix and its transpose share the per-tensor maximum: the weights);
//   mode 1: per tensor from *max_bits (absmax_kernel ran on `in`): inv_scale_out[0] = 1 / s;
//   mode 2: per output row from scale_src[c] = max_r |in[r, c]| (colsum_absmax_kernel): inv_scale_out[c] = 1 / s_c.
// 64 x 64 tiles; R_pad a multiple of 8.
__global__ void __launch_bounds__(256) transpose_split_kernel(const float* __restrict__ in, int64_t ld_in, int R, int C,
                                                              uint4* __restrict__ out, int64_t ld_out, int R_pad, int mode,
                                                              const float* __restrict__ scale_src,
                                                              const uint32_t* __restrict__ max_bits,
                                                              float* __restrict__ inv_scale_out) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
        const int r = r0 + ty + i, c = c0 + tx;
        tile[ty + i][tx] = (r < R && c < C) ? in[(int64_t)r * ld_in + c] : 0.f;
    }
    __syncthreads();
    float s_all = 1.0f;
    if (mode == 0) s_all = 1.0f / scale_src[0];                  // a power of two: exact
    if (mode == 1) {
        s_all = pow2_scale_for(__uint_as_float(*max_bits));
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) inv_scale_out[0] = 1.0f / s_all;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int item = threadIdx.x + 256 * p;                  // 64 output rows x 8 groups of 8
        const int cc = item & 63, g = item >> 6;
        const int c = c0 + cc, rg = r0 + 8 * g;
        if (c < C && rg < R_pad) {
            float s = s_all;
            if (mode == 2) {
                s = pow2_scale_for(scale_src[c]);
                if (blockIdx.y == 0 && g == 0) inv_scale_out[c] = 1.0f / s;
            }
            f16x8 hi, lo;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = tile[8 * g + k][cc] * s;
                const _Float16 h = (_Float16)v;
                hi[k] = h;
                lo[k] = (_Float16)(v - (float)h);
            }
            uint4* dr = out + (int64_t)c * (ld_out / 4) + (rg >> 3) * 2;
            dr[0] = *reinterpret_cast<uint4*>(&hi);
            dr[1] = *reinterpret_cast<uint4*>(&lo);
        }
    }
}

// |.| reductions for row-scale bounds, one wave per row.  mode 0: out[row] = max_k |src[row, k]|;  mode 1: out[0] =
// max_row sum_k |src[row, k]| (the infinity norm; out[0] cleared by zero_u32_kernel first; non-negative floats order
// like their bit patterns, so the maximum is an integer atomicMax -- exact and order-independent).
__global__ void __launch_bounds__(256) abs_reduce_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int64_t cols,
                                                         int mode, float* __restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* sr = src + row * ld;
    float acc = 0.f;
    if (mode == 0) {
        for (int64_t i = lane; i < cols; i += 64) acc = fmaxf(acc, fabsf(sr[i]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc = fmaxf(acc, __shfl_xor(acc, off, 64));
        if (lane == 0) out[row] = acc;
    } else {
        for (int64_t i = lane; i < cols; i += 64) acc += fabsf(sr[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) {                              // running maximum: look before the same-address atomic
            unsigned int* mx = reinterpret_cast<unsigned int*>(out);
            if (__float_as_uint(acc) > __atomic_load_n(mx, __ATOMIC_RELAXED)) atomicMax(mx, __float_as_uint(acc));
        }
    }
}

// Columns [g0 * 8, (g0 + n_groups) * 8) of fp32 rows -> the same columns of split rows, with a scale the CALLER fixed
// beforehand (inv_scale[row], a power of two): the hidden-activation panels of the blocked inverse grow by a few columns
// per block, so their row scale has to be known before the values are (a bound; see flows/autoregressive.py).
__global__ void __launch_bounds__(256) split_columns_scaled_kernel(const float* __restrict__ src, int64_t ld_src, int64_t rows,
                                                                   int64_t g0, int64_t n_groups, uint4* __restrict__ dst,
                                                                   int64_t ld_dst, const float* __restrict__ inv_scale) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * n_groups) return;
    const int64_t row = idx / n_groups, g8 = g0 + idx % n_groups;
    const float s = 1.0f / inv_scale[row];
    const float4* sr = reinterpret_cast<const float4*>(src + row * ld_src + g8 * 8);
    const float4 lo4 = sr[0], hi4 = sr[1];
    const float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float t = v[j] * s;
        const _Float16 h = (_Float16)t;
        hi[j] = h;
        lo[j] = (_Float16)(t - (float)h);
    }
    uint4* dr = dst + row * (ld_dst / 4) + g8 * 2;
    dr[0] = *reinterpret_cast<uint4*>(&hi);
    dr[1] = *reinterpret_cast<uint4*>(&lo);
}

// Masked weight preparation straight into split rows (masked_linear.hip's weight_prepare_kernel + split_rows in one
// pass over the weights).  One wave per output row: the row of v is read once coalesced for the weight-norm, then
// gathered in PACKED column order (in_of_col: packed column -> input column; the row sits in L2 by then) so that
// the split row is written with coalesced 32-byte pieces.  The matrix scale comes from *max_bits, an upper bound of
// max |w|: max |g| with weight norm (|w_ij| = |g_i| |v_ij| / ||v_i|| <= |g_i|), max |v| without.
__global__ void __launch_bounds__(256) weight_prepare_split_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                   const float* __restrict__ mask, int N, int K,
                                                                   const int32_t* __restrict__ row_of_out,
                                                                   const int32_t* __restrict__ in_of_col,
                                                                   const int32_t* __restrict__ col_cut,
                                                                   uint4* __restrict__ w_out, int64_t ldw, int k_padded,
                                                                   const uint32_t* __restrict__ max_bits,
                                                                   float* __restrict__ inv_scale) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= N) return;
    const int lane = threadIdx.x & 63;
    const float* vr = v + (int64_t)o * K;
    // col_cut: the mask row is the prefix [0, cut) of the packed columns -- nothing to read
    const int cut = col_cut ? col_cut[o] : 0;
    const float* mr = (mask && !col_cut) ? mask + (int64_t)o * K : nullptr;
    float wn = 1.0f;
    if (g) {
        float ss = 0.f;
        for (int i = lane; i < K; i += 64) ss += vr[i] * vr[i];
        ss = wave_sum(ss);
        wn = g[o] / sqrtf(ss);             // may be inf/NaN for a fully-masked row: never used below
    }
    const float s = pow2_scale_for(__uint_as_float(*max_bits));
    if (o == 0 && lane == 0) inv_scale[0] = 1.0f / s;
    const int64_t orow = row_of_out ? row_of_out[o] : o;
    uint4* dr = w_out + orow * (ldw / 4);
    float l1 = 0.f;                              // sum_k |w_ok|: bounds this unit's pre-activation (see EPI_ELU_SPLIT)
    for (int g8 = lane; g8 * 8 < k_padded; g8 += 64) {
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = g8 * 8 + j;
            float val = 0.f;
            if (c < K) {
                const int i = in_of_col ? in_of_col[c] : c;
                const bool on = col_cut ? c < cut : !(mr && mr[i] == 0.0f);
                if (on) val = g ? vr[i] * wn : (mr ? vr[i] * mr[i] : vr[i]);
            }
            l1 += fabsf(val);
            val *= s;
            const _Float16 h = (_Float16)val;
            hi[j] = h;
            lo[j] = (_Float16)(val - (float)h);
        }
        dr[g8 * 2] = *reinterpret_cast<uint4*>(&hi);
        dr[g8 * 2 + 1] = *reinterpret_cast<uint4*>(&lo);
    }
    l1 = wave_sum(l1);
    // (a running maximum: look before the atomic -- 75 000 same-address atomics serialise in L2 and were most of this kernel's
    // time; a stale read only costs an unnecessary atomic)
    if (lane == 0 && l1 < INFINITY) {
        uint32_t* mx = reinterpret_cast<uint32_t*>(inv_scale + 2);
        if (__float_as_uint(l1) > __atomic_load_n(mx, __ATOMIC_RELAXED)) atomicMax(mx, __float_as_uint(l1));
    }
}

// The same for prefix masks (col_cut), one WORKGROUP per output row with the row of v staged in LDS: v is read exactly
// once from HBM, fully coalesced (sum of squares on the way in), the permuted gather v[in_of_col[c]] then comes from LDS
// (the one-wave-per-row kernel above gathers 4-byte elements through L2: 1.7 TB/s on the cfg2 output layer), and only the
// live prefix [0, cut) of the packed row is written, in whole 32-byte groups -- the masked suffix was zeroed when the
// buffer was allocated and nothing ever writes it (the k-ranges of the GEMMs do not even read it).
constexpr int PFX_THREADS = 512;
__global__ void __launch_bounds__(PFX_THREADS) weight_prepare_split_prefix_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                          int N, int K, const int32_t* __restrict__ row_of_out,
                                                                          const int32_t* __restrict__ in_of_col,
                                                                          const int32_t* __restrict__ col_cut,
                                                                          uint4* __restrict__ w_out, int64_t ldw, int k_padded,
                                                                          const uint32_t* __restrict__ max_bits,
                                                                          float* __restrict__ inv_scale) {
    extern __shared__ float srow[];
    const int o = blockIdx.x, tid = threadIdx.x;
    const float* vr = v + (int64_t)o * K;
    {
        // 16-byte loads with several in flight per lane: at two workgroups per CU (the row takes 60 KB of LDS at
        // K = 14 998) 4-byte loads leave too few bytes in flight to cover the HBM latency.  Rows start 8-byte aligned at
        // best: a scalar head up to the first 16-byte boundary, float4 body, scalar tail.
        const int head = min(K, (int)(((16u - (uint32_t)((uintptr_t)vr & 15u)) & 15u) >> 2));
        const int n4 = (K - head) >> 2;
        const float4* v4 = reinterpret_cast<const float4*>(vr + head);
        if (tid < head) srow[tid] = vr[tid];
        // eight loads issued before the first is consumed (the compiler does not hoist them over the LDS stores itself:
        // one HBM round trip per 16 bytes and lane made this kernel latency-bound at 1 TB/s)
        for (int i0 = tid; i0 < n4; i0 += 8 * PFX_THREADS) {
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * PFX_THREADS;
                q[u] = i < n4 ? v4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * PFX_THREADS;
                if (i < n4) {
                    float* d = srow + head + 4 * i;
                    d[0] = q[u].x; d[1] = q[u].y; d[2] = q[u].z; d[3] = q[u].w;
                }
            }
        }
        for (int i = head + 4 * n4 + tid; i < K; i += PFX_THREADS) srow[i] = vr[i];
    }
    __syncthreads();
    float wn = 1.0f;
    if (g) {
        // the row norm in the summation order of weight_prepare_split_kernel (lane l: l, l + 64, ...; then the butterfly),
        // every wave for itself from LDS: the packed bits are identical whichever kernel packs the layer
        float ss = 0.f;
        for (int i = tid & 63; i < K; i += 64) ss += srow[i] * srow[i];
        ss = wave_sum(ss);
        wn = g[o] / sqrtf(ss);             // may be inf/NaN for a fully-masked row: never used below
    }
    const float s = pow2_scale_for(__uint_as_float(*max_bits));
    if (o == 0 && tid == 0) inv_scale[0] = 1.0f / s;
    const int cut = min(col_cut[o], K);
    const int64_t orow = row_of_out ? row_of_out[o] : o;
    uint4* dr = w_out + orow * (ldw / 4);
    float l1 = 0.f;
    const int n_groups = (cut + 7) >> 3;
    for (int g8 = tid; g8 < n_groups; g8 += PFX_THREADS) {
        f16x8 hi, lo;
        int idx[8];
        if (in_of_col && g8 * 8 + 8 <= K && ((uintptr_t)in_of_col & 15u) == 0) {
            const int4 a = reinterpret_cast<const int4*>(in_of_col)[2 * g8], b = reinterpret_cast<const int4*>(in_of_col)[2 * g8 + 1];
            idx[0] = a.x; idx[1] = a.y; idx[2] = a.z; idx[3] = a.w; idx[4] = b.x; idx[5] = b.y; idx[6] = b.z; idx[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = g8 * 8 + j;
                idx[j] = (in_of_col && c < K) ? in_of_col[c] : min(c, K - 1);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = g8 * 8 + j;
            float val = 0.f;
            if (c < cut) val = srow[idx[j]] * wn;
            l1 += fabsf(val);
            val *= s;
            const _Float16 h = (_Float16)val;
            hi[j] = h;
            lo[j] = (_Float16)(val - (float)h);
        }
        dr[g8 * 2] = *reinterpret_cast<uint4*>(&hi);
        dr[g8 * 2 + 1] = *reinterpret_cast<uint4*>(&lo);
    }
    l1 = wave_sum(l1);
    if ((tid & 63) == 0 && l1 < INFINITY) {                   // look before the atomic (see weight_prepare_split_kernel)
        uint32_t* mx = reinterpret_cast<uint32_t*>(inv_scale + 2);
        if (__float_as_uint(l1) > __atomic_load_n(mx, __ATOMIC_RELAXED)) atomicMax(mx, __float_as_uint(l1));
    }
}

// ------------------------------------------------------------------------------------------
// GEMM
// ------------------------------------------------------------------------------------------
// One v_mfma_f32_16x16x32_f16 with the accumulator pinned to a register file.  A wave of this kernel owns 16 * NREP
// accumulator registers -- 400 for the fused spline tile -- and has 256 AGPRs + 256 VGPRs: the column groups
// below N_ACC_AGPR accumulate in AGPRs, the rest in VGPRs.  (Left to the register allocator, the builtin form
// shuffles tiles between the two files inside the k-loop and spills.)
constexpr int N_ACC_AGPR = 16;
constexpr int B_AHEAD = 2;         // column groups of B fragments in flight ahead of the MFMAs

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <bool IN_AGPR>
__device__ __forceinline__ void mfma16(f32x4& acc, const f16x8& a, const f16x8& b) {
    if constexpr (IN_AGPR)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

__device__ __forceinline__ void keep_alive(const f32x4& v) { asm volatile("" ::"v"(v)); }

// TFEP_DIAG & 16: per-phase cycle totals of the fused kernel (wave 0 of every workgroup): [k-loop, epilogue, workgroups]
__device__ unsigned long long g_split_cycles[4];   // [3]: workgroup lifetimes in 100 MHz real-time ticks

struct SplitCtx {
    __amdgpu_buffer_rsrc_t ra, rw;
    uint32_t va_even, va_odd, vw;   // per-lane byte offsets inside an 8-row chunk (A: even / odd chunks)
    uint32_t piece_a, piece_w;      // bytes between consecutive 8-row chunks
};

template <int NREP>
__device__ inline void split_dma(const SplitCtx& sc, char* a_wave, char* b_stage, int k0, int wave, int d) {
    using T = STile<NREP>;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    if (d < T::A_DMA) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(sc.ra, (lds_ptr)(a_wave + d * 1024), 16,
                                                 ((d & 1) ? sc.va_odd : sc.va_even) + (uint32_t)d * sc.piece_a, k0 * 4, 0, 0);
    } else {
        const int c = wave + SWAVES * (d - T::A_DMA);
        if (c < T::B_CHUNKS)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sc.rw, (lds_ptr)(b_stage + c * 1024), 16,
                                                     sc.vw + (uint32_t)c * sc.piece_w, k0 * 4, 0, 0);
    }
}

// Fused RQ-spline epilogue of the split kernel (P = 25 parameters per feature, one feature per lane column,
// 16 samples per lane).  With one wave per SIMD there is no partner wave to hide latency behind, and 400
// accumulator registers cannot be indexed by a run-time (m, i): so the wave stages its tile through its own LDS
// region, one 16-row block at a time -- element (row, feature) becomes a 28-float record [25 parameters (bias and
// un-scaling applied), x, pad] -- and a 4-iteration run-time loop evaluates one record per lane per iteration.
// Same arithmetic as gemm_epilogue<EPI_SPLINE> (spline.h), same outputs.
// the records are written float by float and read back 16 bytes at a time: the vector type must be allowed to alias
typedef float f32x4_alias __attribute__((ext_vector_type(4), may_alias));
// floats per record: 16-byte aligned, > P, and a stride that spreads the 16 features of a block over 16 banks (28 for the
// 25 parameters of 8 bins; 20 for the 16 / 13 of 5 / 4 bins -- 16 would put them on 4)
template <int KSPL> constexpr int spl_rec() { return 3 * KSPL + 2 <= 20 ? 20 : 28; }
template <int KSPL> constexpr int spl_wave_bytes() { return 16 * 16 * spl_rec<KSPL>() * 4; }     // one 16-row block of one wave

template <int KSPL>
__device__ __forceinline__ void split_spline_epilogue(const GemmArgs& g, f32x4 (&acc)[3 * KSPL + 1][SMREP],
                                                      const f32x4 (&rs)[SMREP], int nt, int n0, int wrow0, int lane,
                                                      float* rec_base) {
    constexpr int P = 3 * KSPL + 1;
    constexpr int SPL_REC = spl_rec<KSPL>();
    static_assert(P + 1 <= SPL_REC, "record too small");
    const FusedArgs& fu = g.fu;
    const int cj = lane & 15, gq = lane >> 4;
    const int slot = nt * 16 + cj;
    const int fcol = fu.feat_index[slot];
    const bool live = fcol >= 0;
    float bias_p[P];
    static_for<0, P>([&](auto pc) __attribute__((always_inline)) {
        bias_p[pc.value] = g.bias ? g.bias[n0 + pc.value * 16 + cj] : 0.f;
    });
    float x0 = 0.f, xf = 1.f, y0 = 0.f, yf = 1.f;
    if (live) {
        const int ftr = fu.feat_tr[slot];
        x0 = fu.x0[ftr];
        xf = fu.xf[ftr];
        y0 = fu.y0[ftr];
        yf = fu.yf[ftr];
    }
    // all 16 inputs of this lane in flight at once
    f32x4 xin[SMREP];
    static_for<0, SMREP * 4>([&](auto ic) __attribute__((always_inline)) {
        constexpr int m = ic.value / 4, i = ic.value % 4;
        const int row = wrow0 + m * 16 + gq * 4 + i;
        xin[m][i] = (live && row < g.B) ? fu.x[(int64_t)row * fu.ldx + fcol] : 0.f;
    });
    float* rec0 = rec_base + ((gq * 4) * 16 + cj) * SPL_REC;      // record of (row gq*4, feature cj)
    static_for<0, SMREP>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = mc.value;
        static_for<0, P * 4>([&](auto ic) __attribute__((always_inline)) {
            constexpr int n = ic.value / 4, i = ic.value % 4;
            rec0[i * 16 * SPL_REC + n] = acc[n][m][i] * rs[m][i] + bias_p[n];
        });
        static_for<0, 4>([&](auto ic) __attribute__((always_inline)) { rec0[ic.value * 16 * SPL_REC + P] = xin[m][ic.value]; });
        // two records per iteration, in one basic block: their fp64 chains are independent and interleave
#pragma nounroll
        for (int i0 = 0; i0 < 4; i0 += 2) {
            float prm[2][SPL_REC];
            static_for<0, 2 * (SPL_REC / 4)>([&](auto qc) __attribute__((always_inline)) {
                constexpr int e = qc.value / (SPL_REC / 4), q4 = qc.value % (SPL_REC / 4);
                const f32x4_alias q = ((const f32x4_alias*)(rec0 + (i0 + e) * 16 * SPL_REC))[q4];
                prm[e][4 * q4] = q[0];
                prm[e][4 * q4 + 1] = q[1];
                prm[e][4 * q4 + 2] = q[2];
                prm[e][4 * q4 + 3] = q[3];
            });
            double ld[2];
            float outv[2];
            static_for<0, 2>([&](auto ec) __attribute__((always_inline)) {
                constexpr int e = ec.value;
                float w[KSPL], h[KSPL], sraw[KSPL + 1];
                static_for<0, KSPL>([&](auto kc) __attribute__((always_inline)) {
                    w[kc.value] = prm[e][kc.value];
                    h[kc.value] = prm[e][KSPL + kc.value];
                    sraw[kc.value] = prm[e][2 * KSPL + kc.value];
                });
                const float lastp = prm[e][3 * KSPL];
                sraw[KSPL] = fu.sf.circular ? sraw[0] : lastp;      // plain: K+1 slopes; circular: slope_K := slope_0
                outv[e] = (float)rq_spline_forward_full<KSPL>(w, h, sraw, lastp, fu.sf, x0, xf, y0, yf, prm[e][P], &ld[e]);
            });
            static_for<0, 2>([&](auto ec) __attribute__((always_inline)) {
                constexpr int e = ec.value;
                const int row = wrow0 + m * 16 + gq * 4 + i0 + e;
                const bool ok = live && row < g.B;
                if (ok) fu.y[(int64_t)row * fu.ldy + fcol] = outv[e];
                double l = ok ? ld[e] : 0.0;
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
                if (cj == 0 && row < g.B) fu.ldj_partial[(int64_t)nt * g.B + row] = l;
            });
        }
    });
}

template <int NREP, int EPI, int P, int KSPL>
__global__ void __launch_bounds__(STHREADS, 1) split_gemm_kernel(GemmArgs g, int n_rows_w) {
    using T = STile<NREP>;
    extern __shared__ __attribute__((aligned(16))) char slds[];

    // wave index in an SGPR: LDS addresses of the DMA (M0) are then scalar arithmetic, no v_readfirstlane per instruction
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int mt, ntp;
    if (!map_block(g, mt, ntp)) return;
    int nt = g.tile_order ? g.tile_order[ntp] : ntp;
    // split-K (the short-and-wide block GEMMs of the inverse): ksplit x as many column positions, position -> (column
    // tile, k slice); slice s writes its partial sums to y + s * slab_stride (gemm_common.h)
    int k_slice = 0;
    if constexpr (EPI == EPI_LINEAR) {              // (compile-time: the fused instantiations stay exactly as they were)
        if (g.ksplit > 1) {
            const int n_real = g.n_tiles / g.ksplit;
            k_slice = nt / n_real;
            nt -= k_slice * n_real;
        }
    }
    if (g.tile_live && !g.tile_live[(int64_t)mt * g.n_tiles + nt]) return;
    const int m0 = mt * T::BM, n0 = nt * T::BN;
    const unsigned long long t_start = (g.diag & 16) ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long r_start = (g.diag & 16) ? __builtin_amdgcn_s_memrealtime() : 0ull;

    int kb = 0, ke = g.k_padded;
    if (g.k_ranges) {
        kb = g.k_ranges[2 * nt];
        ke = g.k_ranges[2 * nt + 1];
    }
    if constexpr (EPI == EPI_LINEAR) {
        if (g.ksplit > 1) {
            const int per = (((ke - kb) / SBK + g.ksplit - 1) / g.ksplit) * SBK;   // k per slice, whole tiles
            kb = min(ke, kb + k_slice * per);
            ke = min(ke, kb + per);
        }
    }

    f32x4 acc[NREP][SMREP];
    static_for<0, NREP * SMREP>([&](auto ic) __attribute__((always_inline)) {
        acc[ic.value / SMREP][ic.value % SMREP] = (f32x4){0.f, 0.f, 0.f, 0.f};
    });

    SplitCtx sc;
    {
        constexpr int FLAGS = 0x00020000;     // gfx9 raw buffer, 32-bit data; rows past the end read as 0
        const int rows_a = min(g.B - m0, T::BM), rows_w = min(n_rows_w - n0, T::BN);
        sc.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a + (int64_t)m0 * g.lda), 0,
                                                  (int)clamp_u32((int64_t)rows_a * g.lda * 4), FLAGS);
        sc.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w + (int64_t)n0 * g.ldw), 0,
                                                  (int)clamp_u32((int64_t)rows_w * g.ldw * 4), FLAGS);
        const int drow = lane >> 3, p0 = (lane & 7) ^ ((lane >> 4) & 1);
        const uint32_t arow = (uint32_t)(((int64_t)(wave * 16 * SMREP + drow) * g.lda) * 4);
        sc.va_even = arow + p0 * 16;
        sc.va_odd = arow + (p0 ^ 4) * 16;
        sc.vw = (uint32_t)(((int64_t)drow * g.ldw) * 4) + (((wave & 1) ? (p0 ^ 4) : p0) * 16);
        sc.piece_a = (uint32_t)(8 * g.lda * 4);
        sc.piece_w = (uint32_t)(8 * g.ldw * 4);
    }
    char* a_wave = slds + wave * A_WAVE_BYTES;
    char* b_base = slds + T::A_BYTES;

    const int nk = (g.diag & 2) ? 0 : (ke - kb) / SBK;      // diag 2: epilogue only (timing)
    if (nk > 0) {
#pragma unroll
        for (int d = 0; d < T::N_DMA; ++d) split_dma<NREP>(sc, a_wave, b_base, kb, wave, d);
    }
    // fragment of row (l & 15), k-group (l >> 4): hi at part 2g, lo at part 2g + 1, swizzled
    const int fr = lane & 15, fg = lane >> 4;
    const int fe = (fr >> 1) & 7, fsw = (fe & 1) | (fe & 4);
    const int off_hi = fr * ROW_BYTES + (((2 * fg) ^ fsw) << 4);
    const int off_lo = fr * ROW_BYTES + (((2 * fg + 1) ^ fsw) << 4);

    // One k-tile.  DMA = true: also issue the LDS-DMA of tile t + 1 (every tile but the last; a separate
    // instantiation, so the k-loop has no branch around the DMA instructions and they sit BETWEEN the MFMAs).
    auto tile = [&](auto dma_c, int t) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma_c)::value;
        // Own A DMA + everybody's B DMA of tile t landed; other B stage free.  The wait is explicit: the compiler's
        // own LDS-DMA tracking was seen to emit vmcnt(1) here, leaving the last-issued chunk in flight.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(g.diag & 8)) __syncthreads();
        const char* Bs = b_base + (t & 1) * T::B_BYTES;
        char* Bn = b_base + ((t + 1) & 1) * T::B_BYTES;
        f16x8 ah[SMREP], al[SMREP];
#pragma unroll
        for (int m = 0; m < SMREP; ++m) {
            ah[m] = *(const f16x8*)(a_wave + m * 16 * ROW_BYTES + off_hi);
            al[m] = *(const f16x8*)(a_wave + m * 16 * ROW_BYTES + off_lo);
        }
        // B fragments run B_AHEAD column groups ahead of the MFMAs that use them (ring of B_AHEAD + 1 register sets):
        // a lone wave per SIMD has to cover the LDS latency itself.
        f16x8 bh[B_AHEAD + 1], bl[B_AHEAD + 1];
        static_for<0, B_AHEAD>([&](auto pc) __attribute__((always_inline)) {
            if constexpr (pc.value < NREP) {
                bh[pc.value] = *(const f16x8*)(Bs + pc.value * 16 * ROW_BYTES + off_hi);
                bl[pc.value] = *(const f16x8*)(Bs + pc.value * 16 * ROW_BYTES + off_lo);
            }
        });
#ifdef TFEP_PROBE_KWINDOW
        // TIMING PROBE (wrong results): every k-tile is fetched from the first TFEP_PROBE_KWINDOW columns of the operand
        // panels, so that the streamed operands stay resident in the XCD's L2 -- what removing the super-tile over-fetch
        // from beyond L2 could buy at most (python -m tfep_amd.build --probe ... -DTFEP_PROBE_KWINDOW=128).
        const int k_next = kb + (((t + 1) * SBK) % TFEP_PROBE_KWINDOW);
#else
        const int k_next = kb + (t + 1) * SBK;
#endif
        // The next tile's DMA goes out EARLY in this tile (a DMA issued late meets the barrier before it has landed),
        // one instruction after each group of four MFMAs -- a lone wave that issues a clump of DMA instructions
        // starves its matrix pipe meanwhile.  Weights first: their stage is free since the barrier; then the wave's
        // own A rows, whose region is free once the fragments above are in registers (group 0 has waited for them).
        auto dma_slot = [&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;                                   // issue order
            if constexpr (DMA && q < T::N_DMA) {
                constexpr int dd = q < T::B_DMA ? T::A_DMA + q : q - T::B_DMA;       // split_dma index: B first
                static_assert(q < T::B_DMA || q / 3 >= 1, "A DMA before the A fragments are read");
                __builtin_amdgcn_sched_barrier(0);
                split_dma<NREP>(sc, a_wave, Bn, k_next, wave, dd);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // Column groups as a compile-time loop: the accumulator indices must be constants in the frontend,
        // or the 400-register array is not promoted out of scratch memory.
        static_for<0, NREP>([&](auto nc) __attribute__((always_inline)) {
            constexpr int n = decltype(nc)::value;
            if constexpr (n + B_AHEAD < NREP) {
                bh[(n + B_AHEAD) % (B_AHEAD + 1)] = *(const f16x8*)(Bs + (n + B_AHEAD) * 16 * ROW_BYTES + off_hi);
                bl[(n + B_AHEAD) % (B_AHEAD + 1)] = *(const f16x8*)(Bs + (n + B_AHEAD) * 16 * ROW_BYTES + off_lo);
            }
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 h = bh[n % (B_AHEAD + 1)], l = bl[n % (B_AHEAD + 1)];
            static_for<0, SMREP>([&](auto mc) __attribute__((always_inline)) { mfma16<(n < N_ACC_AGPR)>(acc[n][mc.value], ah[mc.value], h); });
            dma_slot(std::integral_constant<int, 3 * n>{});
            static_for<0, SMREP>([&](auto mc) __attribute__((always_inline)) { mfma16<(n < N_ACC_AGPR)>(acc[n][mc.value], al[mc.value], h); });
            dma_slot(std::integral_constant<int, 3 * n + 1>{});
            static_for<0, SMREP>([&](auto mc) __attribute__((always_inline)) { mfma16<(n < N_ACC_AGPR)>(acc[n][mc.value], ah[mc.value], l); });
            dma_slot(std::integral_constant<int, 3 * n + 2>{});
        });
    };
    static_assert(T::N_DMA <= 3 * NREP, "not enough DMA slots in a k-tile");
    for (int t = 0; t + 1 < nk; ++t) tile(std::true_type{}, t);
    if (nk > 0) tile(std::false_type{}, nk - 1);
    // The MFMAs are inline asm: leave the matrix pipe's result latency behind before anything reads acc.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    if (g.diag & 1) {   // timing-only build of the main loop: keep the accumulators alive, store nothing
        static_for<0, NREP * SMREP>([&](auto ic) __attribute__((always_inline)) {
            keep_alive(acc[ic.value / SMREP][ic.value % SMREP]);
        });
        if (EPI == EPI_SPLINE && (g.diag & 16) && threadIdx.x == 0) {
            atomicAdd(&g_split_cycles[0], __builtin_readcyclecounter() - t_start);
            atomicAdd(&g_split_cycles[2], 1ull);
            atomicAdd(&g_split_cycles[3], __builtin_amdgcn_s_memrealtime() - r_start);
        }
        return;
    }
    const unsigned long long t_loop = (g.diag & 16) ? __builtin_readcyclecounter() : 0ull;
    // Un-scale by the exact powers of two (rs: per-row 1/scale of the activations times the weights' 1/scale).
    // `acc` is only ever indexed by constants, so it lives in registers through the k-loop.
    const int wrow0 = m0 + wave * 16 * SMREP;
    f32x4 rs[SMREP];
    {
        const float ws = g.w_inv_scale[0];
        const int rq = (lane >> 4) * 4;
        static_for<0, SMREP * 4>([&](auto ic) __attribute__((always_inline)) {
            constexpr int m = ic.value / 4, i = ic.value % 4;
            const int row = wrow0 + m * 16 + rq + i;
            rs[m][i] = row < g.B ? g.a_inv_scale[row] * ws : 0.f;
        });
    }
    if constexpr (EPI == EPI_SPLINE) {
        static_assert(SWAVES * spl_wave_bytes<KSPL>() <= T::LDS_BYTES, "epilogue records do not fit in LDS");
        __syncthreads();                        // every wave is done with the operand stages: LDS is reused below
        split_spline_epilogue<KSPL>(g, acc, rs, nt, n0, wrow0, lane, (float*)(slds + wave * spl_wave_bytes<KSPL>()));
    } else if constexpr (EPI == EPI_ELU_SPLIT) {
        // y = ELU(x W^T + b) written straight as split rows for the next GEMM.  The row scale cannot wait for the row
        // maximum (other workgroups hold the other columns), so it comes from a bound every workgroup can compute:
        //   |y| <= max(1, max|x_row| * max_j sum_k |w_jk| + max|b|),   max|x_row| < 2^15 * x_inv_scale[row].
        // A bound even 100x above the true maximum costs nothing: the split format keeps an absolute error of
        // 2^-40 of the scaled maximum, far below fp32 rounding of the sums that consume it.
        const float wl1 = g.w_l1max[0], bmax = g.bias_absmax[0];
        const int cj = lane & 15, rq = (lane >> 4) * 4;
        f32x4 so[SMREP];
        static_for<0, SMREP * 4>([&](auto ic) __attribute__((always_inline)) {
            constexpr int m = ic.value / 4, i = ic.value % 4;
            const int row = wrow0 + m * 16 + rq + i;
            float s_out = 1.f;
            if (row < g.B) {
                s_out = pow2_scale_for(fmaxf(1.f, 32768.f * g.a_inv_scale[row] * wl1 + bmax));
                if (nt == 0 && cj == 0) g.y_inv_scale[row] = 1.0f / s_out;
            }
            so[m][i] = s_out;
        });
        // The accumulator layout (a lane: 4 rows x 1 column per tile) would store 2 bytes at a time; the tile goes through
        // LDS instead, 64 columns at a time, and leaves as whole 32-byte groups (8 hi halves, 8 lo halves) -- 8 lanes write
        // 256 contiguous bytes of a row.  (4-byte pair-packed stores ran this epilogue at 1.5 TB/s: 17 % of the K = 3008
        // hidden layer.)
        constexpr int EP_PITCH = 68;                    // floats per staged row: 64 columns + 4 so that the 4 row groups
                                                        // of a tile write to different banks
        static_assert(NREP % 4 == 0, "the split-row epilogue stages 4 column tiles at a time");
        static_assert(SWAVES * 64 * EP_PITCH * 4 <= T::LDS_BYTES, "epilogue stage does not fit in LDS");
        __syncthreads();                                // every wave is done with the operand stages: LDS is reused below
        float* stage = reinterpret_cast<float*>(slds + wave * (64 * EP_PITCH * 4));
        static_for<0, NREP / 4>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = qc.value;
            static_for<0, 4>([&](auto n4c) __attribute__((always_inline)) {
                constexpr int n = 4 * q + n4c.value;
                const int col = n0 + n * 16 + cj;
                const bool in_range = col < g.N;
                const float bv = (in_range && g.bias) ? g.bias[col] : 0.f;
                static_for<0, SMREP * 4>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int m = ic.value / 4, i = ic.value % 4;
                    const float v = in_range ? elu_f(acc[n][m][i] * rs[m][i] + bv) * so[m][i] : 0.f;
                    stage[(m * 16 + rq + i) * EP_PITCH + n4c.value * 16 + cj] = v;
                });
            });
            __builtin_amdgcn_wave_barrier();            // (one wave per stage: LDS is in order, this pins the compiler)
            const int grp = lane & 7;
            const int colg = n0 + q * 64 + grp * 8;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row_l = it * 8 + (lane >> 3);
                const int row = wrow0 + row_l;
                const f32x4_alias lo4 = *reinterpret_cast<const f32x4_alias*>(stage + row_l * EP_PITCH + grp * 8);
                const f32x4_alias hi4 = *reinterpret_cast<const f32x4_alias*>(stage + row_l * EP_PITCH + grp * 8 + 4);
                const float v8[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                f16x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const _Float16 h = (_Float16)v8[j];
                    hi[j] = h;
                    lo[j] = (_Float16)(v8[j] - (float)h);
                }
                if (row < g.B && colg < g.N) {
                    uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(g.y) + (int64_t)row * g.ldy * 4 + (int64_t)(colg >> 3) * 32);
                    dst[0] = *reinterpret_cast<uint4*>(&hi);
                    dst[1] = *reinterpret_cast<uint4*>(&lo);
                }
            }
            __builtin_amdgcn_wave_barrier();
        });
    } else {
        // the other epilogues get a copy they may index from unrolled loops
        f32x4 out[NREP][SMREP];
        static_for<0, NREP * SMREP>([&](auto ic) __attribute__((always_inline)) {
            constexpr int n = ic.value / SMREP, m = ic.value % SMREP;
            out[n][m] = acc[n][m] * rs[m];
        });
        gemm_epilogue<SMREP, NREP, EPI, P, KSPL>(g, out, nt, n0, wrow0, lane, k_slice);
    }
    if (EPI == EPI_SPLINE && (g.diag & 16) && threadIdx.x == 0) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        atomicAdd(&g_split_cycles[0], t_loop - t_start);
        atomicAdd(&g_split_cycles[1], t_end - t_loop);
        atomicAdd(&g_split_cycles[2], 1ull);
        atomicAdd(&g_split_cycles[3], __builtin_amdgcn_s_memrealtime() - r_start);
    }
}

// The matrix-pipe ceiling of THIS device for the split GEMM's instruction mix (no memory).
template <int NREP>
__global__ void __launch_bounds__(STHREADS, 1) split_peak_kernel(float* out, int iters) {
    f32x4 acc[NREP][SMREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
        for (int m = 0; m < SMREP; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 a[SMREP], b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        b[j] = (_Float16)(1e-3f * (threadIdx.x + j));
#pragma unroll
        for (int m = 0; m < SMREP; ++m) a[m][j] = (_Float16)(0.5f + 0.01f * (threadIdx.x & 31) + m + j);
    }
    for (int it = 0; it < iters; ++it) {
        static_for<0, NREP>([&](auto nc) __attribute__((always_inline)) {
            constexpr int n = decltype(nc)::value;
            static_for<0, 3 * SMREP>([&](auto mc) __attribute__((always_inline)) {
                mfma16<(n < N_ACC_AGPR)>(acc[n][mc.value % SMREP], a[mc.value % SMREP], b);
            });
        });
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
        for (int m = 0; m < SMREP; ++m) sum += acc[n][m][0] + acc[n][m][1] + acc[n][m][2] + acc[n][m][3];
    out[blockIdx.x * STHREADS + threadIdx.x] = sum;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <int NREP, int EPI, int P, int KSPL>
static int launch_split(const GemmArgs& g, int n_rows_w, int n_col_tiles, hipStream_t s) {
    using T = STile<NREP>;
    auto kern = split_gemm_kernel<NREP, EPI, P, KSPL>;
    // per device: a process may drive several GPUs
    static bool attr_set_on[TFEP_MAX_DEVICES] = {};
    bool& attr_set = attr_set_on[current_device_slot()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
        if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "hipFuncSetAttribute(LDS=%d): %s", T::LDS_BYTES, hipGetErrorString(e));
        attr_set = true;
    }
    GemmArgs ga = g;
    ga.m_tiles = (g.B + T::BM - 1) / T::BM;
    if (g.ksplit > 1 && EPI != EPI_LINEAR) return fail(TFEP_ERR_INVALID_ARGUMENT, "split gemm: k_split needs the linear epilogue");
    ga.n_tiles = n_col_tiles * (g.ksplit > 1 ? g.ksplit : 1);
    ga.map_mode = block_map_mode();
    ga.diag = env_int("TFEP_DIAG", 0);
    const long long blocks = gemm_grid_blocks(ga.map_mode, ga.m_tiles, ga.n_tiles);
    if (blocks > 0x7fffffffLL) return fail(TFEP_ERR_INVALID_ARGUMENT, "split gemm: grid too large");
    kern<<<dim3((unsigned)blocks), STHREADS, T::LDS_BYTES, s>>>(ga, n_rows_w);
    return check_launch("split_gemm_kernel");
}

int check_split_operands(const GemmArgs& g) {
    TFEP_REQUIRE(g.a_inv_scale && g.w_inv_scale, "split gemm: NULL scale pointer");
    TFEP_REQUIRE(g.k_padded > 0 && g.k_padded % SBK == 0, "split gemm: k_padded=%d must be a positive multiple of %d", g.k_padded, SBK);
    TFEP_REQUIRE(g.lda % SBK == 0 && g.ldw % SBK == 0, "split gemm: row strides must be multiples of %d elements", SBK);
    TFEP_REQUIRE((int64_t)STile<25>::BM * g.lda * 4 < 0x7fffffffLL && (int64_t)STile<25>::BN * g.ldw * 4 < 0x7fffffffLL,
                 "split gemm: row stride too large");
    return TFEP_OK;
}

int launch_split_linear(const GemmArgs& g, int n_rows_w, int act, hipStream_t s) {
    int rc = check_split_operands(g);
    if (rc) return rc;
    constexpr int NREP = 16;
    const int n_tiles = (g.N + STile<NREP>::BN - 1) / STile<NREP>::BN;
    if (act == 1 && g.y_inv_scale) {
        TFEP_REQUIRE(g.w_l1max && g.bias_absmax, "split gemm: split output needs w_l1max and bias_absmax");
        TFEP_REQUIRE(!g.col_map && !g.aux && !g.pre_add && !g.accumulate && g.ldy % 8 == 0 && g.N <= g.ldy && g.N % 2 == 0,
                     "split gemm: split output supports the plain ELU layer only");
        return launch_split<NREP, EPI_ELU_SPLIT, 1, 1>(g, n_rows_w, n_tiles, s);
    }
    if (act == 1) return launch_split<NREP, EPI_ELU, 1, 1>(g, n_rows_w, n_tiles, s);
    // the hidden-layer block GEMMs of the inverse are ~100 outputs wide: a 128-column tile instead of a 256-column one
    // that would be 60 % padding
    constexpr int NARROW = 8;
    if (g.N <= STile<NARROW>::BN && env_int("TFEP_SPLIT_NARROW", 1))
        return launch_split<NARROW, EPI_LINEAR, 1, 1>(g, n_rows_w, 1, s);
    return launch_split<NREP, EPI_LINEAR, 1, 1>(g, n_rows_w, n_tiles, s);
}

int launch_split_fused(const GemmArgs& g, int n_rows_w, int kind, int n_col_tiles, hipStream_t s) {
    int rc = check_split_operands(g);
    if (rc) return rc;
    if (kind == TFEP_FUSED_AFFINE) return launch_split<16, EPI_AFFINE, 2, 1>(g, n_rows_w, n_col_tiles, s);
    // one feature group (16 features x P = 3 K + 1 parameters) per column tile: 8, 5 or 4 bins
    if (g.fu.sf.K == 5) return launch_split<16, EPI_SPLINE, 16, 5>(g, n_rows_w, n_col_tiles, s);
    if (g.fu.sf.K == 4) return launch_split<13, EPI_SPLINE, 13, 4>(g, n_rows_w, n_col_tiles, s);
    return launch_split<25, EPI_SPLINE, 25, 8>(g, n_rows_w, n_col_tiles, s);
}

}  // namespace tfep

using namespace tfep;

extern "C" {

int tfep_split_tile_k(void) { return SBK; }

int tfep_transpose_split(const float* in, int64_t ld_in, int R, int C, void* out_split, int64_t ld_out, int R_pad, int mode,
                         const float* scale_src, float* inv_scale_out, void* stream) {
    TFEP_REQUIRE(R >= 0 && C >= 0 && ld_in >= C && R_pad >= R, "transpose_split: bad sizes");
    if (R_pad == 0 || C == 0) return TFEP_OK;
    TFEP_REQUIRE(in && out_split, "transpose_split: NULL pointer");
    TFEP_REQUIRE(mode >= 0 && mode <= 2, "transpose_split: mode must be 0 (known scale), 1 (per tensor) or 2 (per output row)");
    TFEP_REQUIRE((mode == 1 || scale_src) && (mode == 0 || inv_scale_out), "transpose_split: NULL scale pointer");
    TFEP_REQUIRE(R_pad % 8 == 0 && ld_out >= R_pad && ld_out % 4 == 0 && ((uintptr_t)out_split & 15) == 0,
                 "transpose_split: R_pad must be a multiple of 8, output rows 16-byte aligned and at least R_pad wide");
    hipStream_t s = (hipStream_t)stream;
    uint32_t* max_bits = nullptr;
    if (mode == 1) {                                   // inv_scale_out[1] is scratch for the tensor maximum (as bits)
        max_bits = reinterpret_cast<uint32_t*>(inv_scale_out + 1);
        zero_u32_kernel<<<1, 1, 0, s>>>(max_bits);
        if (R > 0) absmax_kernel<<<(unsigned)((R + 3) / 4), 256, 0, s>>>(in, ld_in, R, C, max_bits);
    }
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R_pad + 63) / 64));
    transpose_split_kernel<<<grid, 256, 0, s>>>(in, ld_in, R, C, (uint4*)out_split, ld_out, R_pad, mode, scale_src, max_bits,
                                                inv_scale_out);
    return check_launch("transpose_split_kernel");
}



int tfep_split_rows(const float* src, int64_t ld_src, int64_t rows, int64_t cols, void* dst, int64_t ld_dst,
                    int64_t cols_padded, float* inv_scale, int per_tensor, void* stream) {
    TFEP_REQUIRE(rows >= 0 && cols >= 0, "split_rows: negative size");
    if (rows == 0) return TFEP_OK;
    TFEP_REQUIRE(src && dst && inv_scale, "split_rows: NULL pointer");
    TFEP_REQUIRE(cols_padded >= cols && cols_padded % SBK == 0, "split_rows: cols_padded=%lld must be a multiple of %d >= cols",
                 (long long)cols_padded, SBK);
    TFEP_REQUIRE(ld_src >= cols && ld_dst >= cols_padded && ld_dst % 4 == 0, "split_rows: bad row strides");
    TFEP_REQUIRE((uintptr_t)dst % 16 == 0, "split_rows: dst must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    uint32_t* max_bits = nullptr;
    if (per_tensor) {
        // inv_scale[1] is scratch for the tensor maximum (as bits)
        max_bits = reinterpret_cast<uint32_t*>(inv_scale + 1);
        zero_u32_kernel<<<1, 1, 0, s>>>(max_bits);
        absmax_kernel<<<blocks, 256, 0, s>>>(src, ld_src, rows, cols, max_bits);
    }
    split_rows_kernel<<<blocks, 256, 0, s>>>(src, ld_src, rows, cols, (uint4*)dst, ld_dst, cols_padded, inv_scale, max_bits);
    return check_launch("split_rows_kernel");
}

int tfep_abs_reduce(const float* src, int64_t ld_src, int64_t rows, int64_t cols, int mode, float* out, void* stream) {
    TFEP_REQUIRE(rows >= 0 && cols >= 0, "abs_reduce: negative size");
    TFEP_REQUIRE(mode == 0 || mode == 1, "abs_reduce: mode must be 0 (row maxima) or 1 (maximum row sum)");
    TFEP_REQUIRE(out && (src || rows == 0 || cols == 0), "abs_reduce: NULL pointer");
    TFEP_REQUIRE(ld_src >= cols, "abs_reduce: ld_src < cols");
    hipStream_t s = (hipStream_t)stream;
    if (mode == 1) zero_u32_kernel<<<1, 1, 0, s>>>(reinterpret_cast<uint32_t*>(out), 1);
    if (rows > 0) {
        TFEP_REQUIRE((rows + 3) / 4 <= 0x7fffffffLL, "abs_reduce: grid too large");
        abs_reduce_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, s>>>(src, ld_src, rows, cols, mode, out);
    }
    return check_launch("abs_reduce_kernel");
}

int tfep_split_columns_scaled(const float* src, int64_t ld_src, int64_t rows, int64_t col0, int64_t cols, void* dst,
                              int64_t ld_dst, const float* inv_scale, void* stream) {
    TFEP_REQUIRE(rows >= 0 && cols >= 0 && col0 >= 0, "split_columns_scaled: negative size");
    if (rows == 0 || cols == 0) return TFEP_OK;
    TFEP_REQUIRE(src && dst && inv_scale, "split_columns_scaled: NULL pointer");
    TFEP_REQUIRE(col0 % 8 == 0, "split_columns_scaled: col0=%lld must be a multiple of 8", (long long)col0);
    const int64_t n_groups = (cols + 7) / 8;
    TFEP_REQUIRE(ld_src % 4 == 0 && ld_dst % 4 == 0 && col0 + n_groups * 8 <= ld_src && col0 + n_groups * 8 <= ld_dst,
                 "split_columns_scaled: the 8-column groups covering [col0, col0 + cols) must lie inside both rows");
    TFEP_REQUIRE((uintptr_t)dst % 16 == 0 && (uintptr_t)src % 16 == 0, "split_columns_scaled: src and dst must be 16-byte aligned");
    const int64_t n = rows * n_groups;
    TFEP_REQUIRE((n + 255) / 256 <= 0x7fffffffLL, "split_columns_scaled: grid too large");
    split_columns_scaled_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, ld_src, rows, col0 / 8, n_groups,
                                                                                          (uint4*)dst, ld_dst, inv_scale);
    return check_launch("split_columns_scaled_kernel");
}

// Read and reset the TFEP_DIAG=16 cycle counters: out[0] k-loop, out[1] epilogue (shader cycles summed over workgroups), out[2] workgroups, out[3] lifetimes (100 MHz ticks).
int tfep_diag_split_cycles(unsigned long long* out) {
    TFEP_REQUIRE(out, "diag_split_cycles: NULL");
    unsigned long long zero[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_split_cycles), sizeof(zero));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_split_cycles), zero, sizeof(zero));
    if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "diag_split_cycles: %s", hipGetErrorString(e));
    return TFEP_OK;
}

int tfep_masked_weight_prepare_split(const float* weight_v, const float* weight_g, const float* mask, int out_features,
                                     int in_features, const int32_t* row_of_out, const int32_t* in_of_col, const int32_t* col_cut,
                                     void* w_split_out, int64_t ldw, int k_padded, float* inv_scale, void* stream) {
    TFEP_REQUIRE(weight_v && w_split_out && inv_scale, "masked_weight_prepare_split: NULL pointer");
    TFEP_REQUIRE(out_features >= 0 && in_features >= 0, "masked_weight_prepare_split: negative size");
    TFEP_REQUIRE(k_padded >= in_features && k_padded % SBK == 0 && ldw >= k_padded && ldw % 4 == 0,
                 "masked_weight_prepare_split: k_padded=%d must be a multiple of %d >= in_features, ldw >= k_padded", k_padded, SBK);
    TFEP_REQUIRE((uintptr_t)w_split_out % 16 == 0, "masked_weight_prepare_split: output must be 16-byte aligned");
    if (out_features == 0) return TFEP_OK;
    hipStream_t s = (hipStream_t)stream;
    uint32_t* max_bits = reinterpret_cast<uint32_t*>(inv_scale + 1);
    zero_u32_kernel<<<1, 1, 0, s>>>(max_bits, 2);          // the maximum and the row-L1 maximum (inv_scale[1], [2])
    if (weight_g)
        absmax_kernel<<<1, 256, 0, s>>>(weight_g, 0, 1, out_features, max_bits);          // one wave over the N gains
    else if (in_features > 0)
        absmax_kernel<<<(unsigned)((out_features + 3) / 4), 256, 0, s>>>(weight_v, in_features, out_features, in_features, max_bits);
    static const bool lds_rows = env_int("TFEP_PACK_LDS", 1) != 0;            // A/B switch
    // (short rows stay with the one-wave-per-row kernel: a workgroup per 3000-element row is mostly launch overhead)
    if (col_cut && lds_rows && in_features >= 8192 && (size_t)in_features * 4 <= 64 * 1024) {
        // prefix masks: the row goes through LDS once; the masked suffix of each packed row is NOT written (see the kernel)
        weight_prepare_split_prefix_kernel<<<(unsigned)out_features, PFX_THREADS, (size_t)in_features * 4, s>>>(
            weight_v, weight_g, out_features, in_features, row_of_out, in_of_col, col_cut, (uint4*)w_split_out, ldw, k_padded,
            max_bits, inv_scale);
        return check_launch("weight_prepare_split_prefix_kernel");
    }
    weight_prepare_split_kernel<<<(unsigned)((out_features + 3) / 4), 256, 0, s>>>(
        weight_v, weight_g, mask, out_features, in_features, row_of_out, in_of_col, col_cut, (uint4*)w_split_out, ldw,
        k_padded, max_bits, inv_scale);
    return check_launch("weight_prepare_split_kernel");
}

int tfep_diag_split_mfma_peak(float* scratch, int blocks, int iters, void* stream) {
    TFEP_REQUIRE(scratch && blocks > 0 && iters > 0, "diag_split_mfma_peak: bad arguments");
    split_peak_kernel<25><<<blocks, STHREADS, 0, (hipStream_t)stream>>>(scratch, iters);
    return check_launch("split_peak_kernel");
}

}  // extern "C"
