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
//
// The GEMM kernel itself (tiling, LDS image, epilogues) is in split_gemm_kernel.h.
#include "split_gemm_kernel.h"

namespace tfep {

// ------------------------------------------------------------------------------------------
// fp32 rows -> split rows
// ------------------------------------------------------------------------------------------

__global__ void zero_u32_kernel(uint32_t* p, int n = 1) {
    for (int i = 0; i < n; ++i) p[i] = 0u;
}
__global__ void zero_u32_n_kernel(uint32_t* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
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

// out_bits[c] = max_r |src[r, c]| as bits (cleared by zero_u32_kernel first; the integer maximum of the magnitudes' bits:
// non-negative floats order like their bits and a NaN, above them all, survives).  A workgroup covers 1024 columns
// (16-byte loads when the rows allow) and every gridDim.y-th row: 4 KB contiguous per row, running maxima in registers, one
// atomic per column and workgroup at the end.
__global__ void __launch_bounds__(256) column_absmax_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int64_t cols,
                                                            uint32_t* __restrict__ out_bits, int vec4) {
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (c0 >= cols) return;
    uint32_t m[4] = {0u, 0u, 0u, 0u};
    const int nc = (int)(cols - c0 < 4 ? cols - c0 : 4);
    if (vec4 && nc == 4) {
        // 16 rows in flight per lane (one load per iteration leaves the kernel latency bound: 2.8 TB/s)
        constexpr int U = 16;
        // a contiguous slab of rows per workgroup
        const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
        const int64_t step = 1, r_end = min(rows, (int64_t)(blockIdx.y + 1) * per);
        int64_t r = (int64_t)blockIdx.y * per;
        for (; r + (U - 1) * step < r_end; r += U * step) {
            float4 q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) q[u] = *reinterpret_cast<const float4*>(src + (r + u * step) * ld + c0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                m[0] = max(m[0], __float_as_uint(fabsf(q[u].x)));
                m[1] = max(m[1], __float_as_uint(fabsf(q[u].y)));
                m[2] = max(m[2], __float_as_uint(fabsf(q[u].z)));
                m[3] = max(m[3], __float_as_uint(fabsf(q[u].w)));
            }
        }
        for (; r < r_end; r += step) {
            const float4 q = *reinterpret_cast<const float4*>(src + r * ld + c0);
            m[0] = max(m[0], __float_as_uint(fabsf(q.x)));
            m[1] = max(m[1], __float_as_uint(fabsf(q.y)));
            m[2] = max(m[2], __float_as_uint(fabsf(q.z)));
            m[3] = max(m[3], __float_as_uint(fabsf(q.w)));
        }
    } else {
        for (int64_t r = blockIdx.y; r < rows; r += gridDim.y)
            for (int j = 0; j < nc; ++j) m[j] = max(m[j], __float_as_uint(fabsf(src[r * ld + c0 + j])));
    }
    for (int j = 0; j < nc; ++j)
        if (m[j] > __atomic_load_n(out_bits + c0 + j, __ATOMIC_RELAXED)) atomicMax(out_bits + c0 + j, m[j]);
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
        const int item = threadIdx.x + 256 * p;                  // 64 output rows x 8 groups of 8: the 8 groups of a row
        const int g = item & 7, cc = item >> 3;                  // on neighbouring lanes (256 contiguous bytes per row)
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

// Transpose of split rows that share ONE scale (a packed weight matrix): out[c][r] = in[r][c], halves moved as they are.
// 64 x 64 tiles through two fp16 planes in LDS, 16-byte accesses on both sides; R, C multiples of 8.
__global__ void __launch_bounds__(256) transpose_split_rows_kernel(const uint4* __restrict__ in, int64_t ld_in, int R, int C,
                                                                   uint4* __restrict__ out, int64_t ld_out) {
    constexpr int PITCH = 72;                                     // halves per plane row: 144 B, 16-byte aligned
    __shared__ __attribute__((aligned(16))) _Float16 plane[2][64 * PITCH];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int item = threadIdx.x + 256 * p;                  // 64 input rows x 8 groups of 8 columns
        const int rr = item & 63, g = item >> 6;
        const int r = r0 + rr, cg = c0 + 8 * g;
        uint4 hq = make_uint4(0u, 0u, 0u, 0u), lq = hq;
        if (r < R && cg < C) {
            const uint4* src = in + (int64_t)r * (ld_in / 4) + (cg >> 3) * 2;
            hq = src[0];
            lq = src[1];
        }
        const f16x8 hi = *reinterpret_cast<const f16x8*>(&hq), lo = *reinterpret_cast<const f16x8*>(&lq);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            plane[0][(8 * g + k) * PITCH + rr] = hi[k];
            plane[1][(8 * g + k) * PITCH + rr] = lo[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int item = threadIdx.x + 256 * p;                  // 64 output rows x 8 groups of 8, a row's groups on
        const int g = item & 7, cc = item >> 3;                  // neighbouring lanes
        const int c = c0 + cc, rg = r0 + 8 * g;
        if (c < C && rg < R) {
            uint4* dr = out + (int64_t)c * (ld_out / 4) + (rg >> 3) * 2;
            dr[0] = *reinterpret_cast<const uint4*>(&plane[0][cc * PITCH + 8 * g]);
            dr[1] = *reinterpret_cast<const uint4*>(&plane[1][cc * PITCH + 8 * g]);
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

// Rows whose own dynamic range is wider than the split format keeps at fp32 accuracy: with one power-of-two scale per row
// an element below 2^-bits of the row maximum (bits = 19: its low half leaves the normal fp16 range) is carried with
// fewer than 22 significant bits, and an output that sees ONLY such elements (possible under MADE's prefix masks) inherits
// that error component-wise.  count[0] += rows with a non-zero element below the threshold (non-finite rows count too: they
// belong to the exact kernel's NaN semantics).  One wave per row, atomics only from flagged rows.
__global__ void __launch_bounds__(256) range_flag_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int64_t cols,
                                                         float ratio, int* __restrict__ count) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* sr = src + row * ld;
    float amax = 0.f, amin = 3.0e38f;
    bool bad = false;
    for (int64_t i = lane; i < cols; i += 64) {
        const float a = fabsf(sr[i]);
        bad |= !(a <= 3.0e38f);
        amax = fmaxf(amax, a);
        if (a > 0.f) amin = fminf(amin, a);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, off, 64));
        amin = fminf(amin, __shfl_xor(amin, off, 64));
    }
    const bool flagged = __any(bad) || (amax > 0.f && amin < amax * ratio);
    if (lane == 0 && flagged) atomicAdd(count, 1);
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
// BOTH: the same effective weights also as fp32 rows (w32, the packing of tfep_masked_weight_prepare_prefix: same bits) --
// the blocked inverse needs the fp32 matrix for its block kernel and the split one for its block GEMMs: one read of v.
template <bool BOTH>
__global__ void __launch_bounds__(PFX_THREADS) weight_prepare_split_prefix_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                          int N, int K, const int32_t* __restrict__ row_of_out,
                                                                          const int32_t* __restrict__ in_of_col,
                                                                          const int32_t* __restrict__ col_cut,
                                                                          uint4* __restrict__ w_out, int64_t ldw, int k_padded,
                                                                          const uint32_t* __restrict__ max_bits,
                                                                          float* __restrict__ inv_scale,
                                                                          float* __restrict__ w32, int64_t ldw32) {
    extern __shared__ float srow[];
    const int o = blockIdx.x, tid = threadIdx.x;
    const float* vr = v + (int64_t)o * K;
    // The permutation entries of this thread's column groups (at most 4: K <= 16 384 columns, 512 threads, 8 columns per
    // group) are fetched FIRST, so that their L2 round trips run beside the row load and the norm instead of in front of
    // every gather pass (2.6 -> 2.4 ms for the cfg2 output layer; a persistent, register-pipelined version of this kernel
    // measured the same: the kernel is not latency bound any more, see profiles/NOTES.md).
    const int cut = min(col_cut[o], K);
    const int n_groups = (cut + 7) >> 3;
    const bool vec_idx = in_of_col && ((uintptr_t)in_of_col & 15u) == 0;
    int4 ia[4], ib[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int g8 = tid + u * PFX_THREADS;
        ia[u] = ib[u] = make_int4(0, 0, 0, 0);
        if (vec_idx && g8 < n_groups && g8 * 8 + 8 <= K) {
            ia[u] = reinterpret_cast<const int4*>(in_of_col)[2 * g8];
            ib[u] = reinterpret_cast<const int4*>(in_of_col)[2 * g8 + 1];
        }
    }
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
    const int64_t orow = row_of_out ? row_of_out[o] : o;
    uint4* dr = w_out + orow * (ldw / 4);
    float l1 = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int g8 = tid + u * PFX_THREADS;
        if (g8 >= n_groups) break;
        f16x8 hi, lo;
        int idx[8];
        if (vec_idx && g8 * 8 + 8 <= K) {
            const int4 a = ia[u], b = ib[u];
            idx[0] = a.x; idx[1] = a.y; idx[2] = a.z; idx[3] = a.w; idx[4] = b.x; idx[5] = b.y; idx[6] = b.z; idx[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = g8 * 8 + j;
                idx[j] = (in_of_col && c < K) ? in_of_col[c] : min(c, K - 1);
            }
        }
        float e32[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = g8 * 8 + j;
            float val = 0.f;
            if (c < cut) val = srow[idx[j]] * wn;
            e32[j] = val;
            l1 += fabsf(val);
            val *= s;
            const _Float16 h = (_Float16)val;
            hi[j] = h;
            lo[j] = (_Float16)(val - (float)h);
        }
        dr[g8 * 2] = *reinterpret_cast<uint4*>(&hi);
        dr[g8 * 2 + 1] = *reinterpret_cast<uint4*>(&lo);
        if constexpr (BOTH) {
            float4* d32 = reinterpret_cast<float4*>(w32 + orow * ldw32 + (int64_t)g8 * 8);
            d32[0] = make_float4(e32[0], e32[1], e32[2], e32[3]);
            d32[1] = make_float4(e32[4], e32[5], e32[6], e32[7]);
        }
    }
    l1 = wave_sum(l1);
    if ((tid & 63) == 0 && l1 < INFINITY) {                   // look before the atomic (see weight_prepare_split_kernel)
        uint32_t* mx = reinterpret_cast<uint32_t*>(inv_scale + 2);
        if (__float_as_uint(l1) > __atomic_load_n(mx, __ATOMIC_RELAXED)) atomicMax(mx, __float_as_uint(l1));
    }
}


int check_split_operands(const GemmArgs& g) {
    TFEP_REQUIRE(g.a_inv_scale && g.w_inv_scale, "split gemm: NULL scale pointer");
    TFEP_REQUIRE(g.k_padded > 0 && g.k_padded % SBK == 0, "split gemm: k_padded=%d must be a positive multiple of %d", g.k_padded, SBK);
    TFEP_REQUIRE(g.lda % SBK == 0 && g.ldw % SBK == 0, "split gemm: row strides must be multiples of %d elements", SBK);
    TFEP_REQUIRE((int64_t)STile<27>::BM * g.lda * 4 < 0x7fffffffLL && (int64_t)STile<27>::BN * g.ldw * 4 < 0x7fffffffLL,
                 "split gemm: row stride too large");
    return TFEP_OK;
}

int split_wide_tile_n() { return STile<25>::BN; }
int split_half_wide_tile_n() { return STile<13>::BN; }

int launch_split_linear(const GemmArgs& g, int n_rows_w, int act, hipStream_t s, bool wide_tile, bool half_wide_tile) {
    int rc = check_split_operands(g);
    if (rc) return rc;
    if (half_wide_tile) {
        // 208 columns: two such tiles cover the 400 output rows of a block of 16 degrees of an 8-bin spline layer (the
        // blocked inverse's output-layer GEMM) with 4 % padding where two 256-column tiles carry 22 %
        TFEP_REQUIRE(act == 0 && !g.y_inv_scale, "split gemm: the 208-column tile takes the plain linear product only");
        return launch_split<13, EPI_LINEAR, 1, 1>(g, n_rows_w, (g.N + STile<13>::BN - 1) / STile<13>::BN, s);
    }
    if (wide_tile) {
        // the 400-column tile of the fused spline kernel for a plain product: a wave's A fragments serve 25 column groups
        // instead of 16 (+5 % on a dense 16 384 x 76 800 x 9 024 product: 468 -> 494 TFLOP/s)
        TFEP_REQUIRE(act == 0 && !g.y_inv_scale, "split gemm: the 400-column tile takes the plain linear product only");
        return launch_split<25, EPI_LINEAR, 1, 1>(g, n_rows_w, (g.N + STile<25>::BN - 1) / STile<25>::BN, s);
    }
    constexpr int NREP = 16;
    const int n_tiles = (g.N + STile<NREP>::BN - 1) / STile<NREP>::BN;
    // short ELU layers (K <= TFEP_SPLIT_SHORT_K, default 2048; the hidden layers of BASELINE cfg4-ii: 1024 columns): 128-column
    // tiles, two workgroups per CU (split_gemm_kernel.h, OCC); the k-ranges / tile order stay those of the 256-column tiles.
    // Measured (cfg4-ii, B = 131 072): 557 -> 519 us per hidden layer; the plain linear product did not gain (545 -> 556 us) and
    // keeps the 256-column tile: a k-tile's LDS-DMA round trip (~6 000 cycles from HBM under load) is 2 - 4 x its matrix products
    // either way, and one k-tile ahead is all the 160 KB of LDS hold
    static const int short_k = env_int("TFEP_SPLIT_SHORT_K", 2048);
    const bool occ2 = g.k_padded <= short_k && g.ksplit <= 1 && !g.tile_list && !g.tile_live && !g.col_map && !g.aux && !g.pre_add &&
                      !g.accumulate && g.N > STile<8>::BN;
    if (act == 1 && g.y_inv_scale) {
        TFEP_REQUIRE(g.w_l1max && g.bias_absmax, "split gemm: split output needs w_l1max and bias_absmax");
        TFEP_REQUIRE(!g.col_map && !g.aux && !g.pre_add && !g.accumulate && g.ldy % 8 == 0 && g.N <= g.ldy && g.N % 2 == 0,
                     "split gemm: split output supports the plain ELU layer only");
        if (occ2) {
            GemmArgs g2 = g;
            g2.kr_shift = 1;
            return launch_split<8, EPI_ELU_SPLIT, 1, 1, false, 2>(g2, n_rows_w, 2 * n_tiles, s);
        }
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

bool split_fused_saving_supported(const SplineFlags& f) {
    const int P = spline_n_params(f.K, f.circular, f.identity, f.learn_lower, f.learn_upper);
    return (f.K == 8 || f.K == 5 || f.K == 4) && P == 3 * f.K + 1 && !f.identity;
}

int launch_split_fused(const GemmArgs& g, int n_rows_w, int kind, int n_col_tiles, hipStream_t s) {
    int rc = check_split_operands(g);
    if (rc) return rc;
    if (kind == TFEP_FUSED_AFFINE) return launch_split<16, EPI_AFFINE, 2, 1>(g, n_rows_w, n_col_tiles, s);
    // one feature group (16 features x P parameters) per column tile: 8, 5 or 4 bins; P = 3 K + 1 for the plain and the
    // circular layout (here); 3 K - 1 .. 3 K + 3 with identity boundary slopes and / or learnable bounds
    // (spline_n_params; split_gemm_layouts.hip, also for identity slopes + both bounds, whose count is 3 K + 1 again)
    const SplineFlags& f = g.fu.sf;
    const int K = f.K, P = spline_n_params(K, f.circular, f.identity, f.learn_lower, f.learn_upper);
    if (g.fu.feature_major) {
        // the training forward (tfep_fused_output_transformer_forward_split_saving): plain and circular layouts
        if (!split_fused_saving_supported(g.fu.sf)) return fail(TFEP_ERR_UNSUPPORTED, "fused split saving: unsupported spline layout");
        if (K == 8) return launch_split<25, EPI_SPLINE, 25, 8, true>(g, n_rows_w, n_col_tiles, s);
        if (K == 5) return launch_split<16, EPI_SPLINE, 16, 5, true>(g, n_rows_w, n_col_tiles, s);
        return launch_split<13, EPI_SPLINE, 13, 4, true>(g, n_rows_w, n_col_tiles, s);
    }
    if (P != 3 * K + 1 || f.identity) return launch_split_fused_layouts(g, n_rows_w, K, P, n_col_tiles, s);
    if (K == 8) return launch_split<25, EPI_SPLINE, 25, 8>(g, n_rows_w, n_col_tiles, s);
    if (K == 5) return launch_split<16, EPI_SPLINE, 16, 5>(g, n_rows_w, n_col_tiles, s);
    if (K == 4) return launch_split<13, EPI_SPLINE, 13, 4>(g, n_rows_w, n_col_tiles, s);
    return fail(TFEP_ERR_UNSUPPORTED, "fused split: no kernel for %d bins", K);
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



int tfep_transpose_split_rows(const void* in_split, int64_t ld_in, int R, int C, void* out_split, int64_t ld_out, void* stream) {
    TFEP_REQUIRE(R >= 0 && C >= 0, "transpose_split_rows: negative size");
    if (R == 0 || C == 0) return TFEP_OK;
    TFEP_REQUIRE(in_split && out_split, "transpose_split_rows: NULL pointer");
    TFEP_REQUIRE(R % 8 == 0 && C % 8 == 0 && ld_in >= C && ld_out >= R && ld_in % 4 == 0 && ld_out % 4 == 0,
                 "transpose_split_rows: R, C must be multiples of 8 and the rows at least that wide, in whole 16-byte units");
    TFEP_REQUIRE((uintptr_t)in_split % 16 == 0 && (uintptr_t)out_split % 16 == 0, "transpose_split_rows: operands must be 16-byte aligned");
    TFEP_REQUIRE((R + 63) / 64 <= 65535, "transpose_split_rows: too many rows for one launch");
    dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64));
    transpose_split_rows_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const uint4*)in_split, ld_in, R, C, (uint4*)out_split, ld_out);
    return check_launch("transpose_split_rows_kernel");
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

int tfep_column_absmax(const float* src, int64_t ld_src, int64_t rows, int64_t cols, float* out, void* stream) {
    TFEP_REQUIRE(rows >= 0 && cols >= 0, "column_absmax: negative size");
    if (cols == 0) return TFEP_OK;
    TFEP_REQUIRE(out && (rows == 0 || src) && ld_src >= cols, "column_absmax: bad arguments");
    TFEP_REQUIRE(cols <= 0x7fffffffLL, "column_absmax: too many columns");
    hipStream_t s = (hipStream_t)stream;
    uint32_t* bits = reinterpret_cast<uint32_t*>(out);
    zero_u32_n_kernel<<<(unsigned)((cols + 255) / 256), 256, 0, s>>>(bits, (int)cols);
    if (rows > 0) {
        const unsigned gx = (unsigned)((cols + 1023) / 1024);
        // ~512 workgroups: more of them only queue up on the final atomics (measured, 512 MB: 256 WGs 121 us, 512 131,
        // 2048 165, 8192 217; torch's reduction 238: tools/probe/colmax_bench.py)
        const int64_t want = (env_int("TFEP_COLMAX_WGS", 512) + gx - 1) / gx;
        const unsigned gy = (unsigned)(rows < want ? rows : want);
        const int vec4 = ((uintptr_t)src % 16 == 0 && ld_src % 4 == 0) ? 1 : 0;
        column_absmax_kernel<<<dim3(gx, gy), 256, 0, s>>>(src, ld_src, rows, cols, bits, vec4);
    }
    return check_launch("column_absmax_kernel");
}

int tfep_range_flag(const float* src, int64_t ld_src, int64_t rows, int64_t cols, int bits, int32_t* count, void* stream) {
    TFEP_REQUIRE(rows >= 0 && cols >= 0 && bits > 0 && bits < 60, "range_flag: bad size / bits");
    TFEP_REQUIRE(count && (src || rows == 0 || cols == 0), "range_flag: NULL pointer");
    TFEP_REQUIRE(ld_src >= cols, "range_flag: ld_src < cols");
    if (rows == 0 || cols == 0) return TFEP_OK;
    TFEP_REQUIRE((rows + 3) / 4 <= 0x7fffffffLL, "range_flag: grid too large");
    range_flag_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(src, ld_src, rows, cols, ldexpf(1.0f, -bits),
                                                                              count);
    return check_launch("range_flag_kernel");
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
        weight_prepare_split_prefix_kernel<false><<<(unsigned)out_features, PFX_THREADS, (size_t)in_features * 4, s>>>(
            weight_v, weight_g, out_features, in_features, row_of_out, in_of_col, col_cut, (uint4*)w_split_out, ldw, k_padded,
            max_bits, inv_scale, nullptr, 0);
        return check_launch("weight_prepare_split_prefix_kernel");
    }
    weight_prepare_split_kernel<<<(unsigned)((out_features + 3) / 4), 256, 0, s>>>(
        weight_v, weight_g, mask, out_features, in_features, row_of_out, in_of_col, col_cut, (uint4*)w_split_out, ldw,
        k_padded, max_bits, inv_scale);
    return check_launch("weight_prepare_split_kernel");
}

int tfep_masked_weight_prepare_split_both(const float* weight_v, const float* weight_g, int out_features, int in_features,
                                          const int32_t* row_of_out, const int32_t* in_of_col, const int32_t* col_cut,
                                          void* w_split_out, int64_t ldw, float* w32_out, int64_t ldw32, int k_padded,
                                          float* inv_scale, void* stream) {
    TFEP_REQUIRE(weight_v && w_split_out && w32_out && inv_scale && col_cut, "masked_weight_prepare_split_both: NULL pointer");
    TFEP_REQUIRE(out_features >= 0 && in_features >= 0, "masked_weight_prepare_split_both: negative size");
    TFEP_REQUIRE(k_padded >= in_features && k_padded % SBK == 0 && ldw >= k_padded && ldw % 4 == 0 && ldw32 >= k_padded && ldw32 % 4 == 0,
                 "masked_weight_prepare_split_both: k_padded=%d must be a multiple of %d >= in_features, rows at least that wide", k_padded, SBK);
    TFEP_REQUIRE((uintptr_t)w_split_out % 16 == 0 && (uintptr_t)w32_out % 16 == 0, "masked_weight_prepare_split_both: outputs must be 16-byte aligned");
    if (!(in_features >= 8192 && (size_t)in_features * 4 <= 64 * 1024))
        return fail(TFEP_ERR_UNSUPPORTED, "masked_weight_prepare_split_both: rows of 8192 .. 16384 weights only (the LDS-staged kernel)");
    if (out_features == 0) return TFEP_OK;
    hipStream_t s = (hipStream_t)stream;
    uint32_t* max_bits = reinterpret_cast<uint32_t*>(inv_scale + 1);
    zero_u32_kernel<<<1, 1, 0, s>>>(max_bits, 2);
    if (weight_g)
        absmax_kernel<<<1, 256, 0, s>>>(weight_g, 0, 1, out_features, max_bits);
    else
        absmax_kernel<<<(unsigned)((out_features + 3) / 4), 256, 0, s>>>(weight_v, in_features, out_features, in_features, max_bits);
    weight_prepare_split_prefix_kernel<true><<<(unsigned)out_features, PFX_THREADS, (size_t)in_features * 4, s>>>(
        weight_v, weight_g, out_features, in_features, row_of_out, in_of_col, col_cut, (uint4*)w_split_out, ldw, k_padded,
        max_bits, inv_scale, w32_out, ldw32);
    return check_launch("weight_prepare_split_prefix_kernel<both>");
}

int tfep_diag_split_mfma_peak(float* scratch, int blocks, int iters, void* stream) {
    TFEP_REQUIRE(scratch && blocks > 0 && iters > 0, "diag_split_mfma_peak: bad arguments");
    split_peak_kernel<25><<<blocks, STHREADS, 0, (hipStream_t)stream>>>(scratch, iters);
    return check_launch("split_peak_kernel");
}

}  // extern "C"
