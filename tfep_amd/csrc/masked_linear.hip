// MADE conditioner kernels for gfx950: masked weight preparation, mask k-ranges, and the
// fp32-MFMA masked-linear GEMM with three epilogues (bias, bias+ELU, fused transformer).
//
// GEMM shape:  Y[b, n] = sum_k X[b, k] * W[n, k]      (both operands K-contiguous, "NT")
//
// Tiling (one workgroup = 8 wavefronts = 512 threads, one workgroup per CU):
//   workgroup tile  BM x BN = (8 * 16 * MREP) x (16 * NREP),  BK = 16
//   wave w owns rows [w*16*MREP, (w+1)*16*MREP) and ALL BN columns, as MREP x NREP tiles of
//   v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate; 4 accumulator VGPRs each).
//   Operands are staged global -> LDS by the LDS-DMA (global_load_lds_dwordx4, no staging
//   VGPRs) into a double buffer: the DMA of k-tile t+1 flies under the MFMAs of k-tile t.
//   Each lane reads its fragments with ONE ds_read_b128 per 16-deep k-tile: lane l takes
//   floats [4q, 4q+4) (q = l >> 4) of row (l & 15) and uses element s in MFMA step s; A and B
//   use the same k permutation so the dot product is unchanged.
//
// Mask sparsity: the weights are stored already masked (zeros materialised) and, per column
// tile, a [k_begin, k_end) range bounds the non-zeros.  After sorting hidden units by degree the
// MADE masks are block triangular (reference conditioners/made.py:286-329), so ~half of the
// k-tiles are skipped without touching them.
#include "gemm_common.h"

namespace tfep {

constexpr int BK = 16;
constexpr int WAVES = 8;
constexpr int THREADS = WAVES * 64;

// ------------------------------------------------------------------------------------------
// Weight preparation (reference masked.py:369-371, :433-439, :270)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) weight_prepare_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                             const float* __restrict__ mask, int N, int K,
                                                             const int32_t* __restrict__ row_of_out,
                                                             const int32_t* __restrict__ col_of_in,
                                                             const int32_t* __restrict__ col_cut,
                                                             float* __restrict__ w_out, int64_t ldw) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= N) return;
    const int lane = threadIdx.x & 63;
    const float* vr = v + (int64_t)o * K;
    const int cut = col_cut ? col_cut[o] : 0;         // prefix mask: packed columns [0, cut) are on, nothing to read
    const float* mr = (mask && !col_cut) ? mask + (int64_t)o * K : nullptr;
    float scale = 1.0f;
    if (g) {
        float ss = 0.f;
        for (int i = lane; i < K; i += 64) ss += vr[i] * vr[i];
        ss = wave_sum(ss);
        scale = g[o] / sqrtf(ss);          // may be inf/NaN for a fully-masked row: never used below
    }
    const int64_t orow = row_of_out ? row_of_out[o] : o;
    float* wr = w_out + orow * ldw;
    for (int i = lane; i < K; i += 64) {
        const int c = col_of_in ? col_of_in[i] : i;
        float val;
        if (col_cut ? c >= cut : (mr && mr[i] == 0.0f))
            val = 0.0f;                    // _ApplyMask: exact zero, also where v*scale is NaN
        else
            val = g ? vr[i] * scale : (mr ? vr[i] * mr[i] : vr[i]);
        wr[c] = val;
    }
}

// The fp32 pack of a layer whose mask rows are prefixes of its packed columns (tfep_masked_weight_prepare_split's col_cut):
// one workgroup per output row, the row of v staged in LDS with coalesced 16-byte loads, the permuted gather out of LDS,
// only the live prefix [0, cut) written, in whole 32-byte groups -- the masked suffix was zeroed when the buffer was
// allocated and nothing writes it.  Same values, bit for bit, as weight_prepare_kernel (same norm summation order), which
// gathers 4-byte elements through L2 and writes the zeros of the suffix too.
constexpr int PFX32_THREADS = 512;
__global__ void __launch_bounds__(PFX32_THREADS) weight_prepare_prefix_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                              int N, int K, const int32_t* __restrict__ row_of_out,
                                                                              const int32_t* __restrict__ in_of_col,
                                                                              const int32_t* __restrict__ col_cut,
                                                                              float* __restrict__ w_out, int64_t ldw) {
    extern __shared__ float srow[];
    const int o = blockIdx.x, tid = threadIdx.x;
    const float* vr = v + (int64_t)o * K;
    {
        // rows start 8-byte aligned at best: scalar head up to the first 16-byte boundary, float4 body with eight loads in
        // flight per lane, scalar tail
        const int head = min(K, (int)(((16u - (uint32_t)((uintptr_t)vr & 15u)) & 15u) >> 2));
        const int n4 = (K - head) >> 2;
        const float4* v4 = reinterpret_cast<const float4*>(vr + head);
        if (tid < head) srow[tid] = vr[tid];
        for (int i0 = tid; i0 < n4; i0 += 8 * PFX32_THREADS) {
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * PFX32_THREADS;
                q[u] = i < n4 ? v4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * PFX32_THREADS;
                if (i < n4) {
                    float* d = srow + head + 4 * i;
                    d[0] = q[u].x; d[1] = q[u].y; d[2] = q[u].z; d[3] = q[u].w;
                }
            }
        }
        for (int i = head + 4 * n4 + tid; i < K; i += PFX32_THREADS) srow[i] = vr[i];
    }
    __syncthreads();
    float wn = 1.0f;
    if (g) {
        // the row norm in weight_prepare_kernel's summation order (lane l: l, l + 64, ...; then the butterfly), every wave
        // for itself from LDS
        float ss = 0.f;
        for (int i = tid & 63; i < K; i += 64) ss += srow[i] * srow[i];
        ss = wave_sum(ss);
        wn = g[o] / sqrtf(ss);             // may be inf/NaN for a fully-masked row: never used below
    }
    const int cut = min(col_cut[o], K);
    const int64_t orow = row_of_out ? row_of_out[o] : o;
    float4* dr = reinterpret_cast<float4*>(w_out + orow * ldw);
    const int n_groups = (cut + 7) >> 3;
    for (int g8 = tid; g8 < n_groups; g8 += PFX32_THREADS) {
        float val[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = g8 * 8 + j;
            const int idx = (in_of_col && c < K) ? in_of_col[c] : min(c, K - 1);
            val[j] = c < cut ? (g ? srow[idx] * wn : srow[idx]) : 0.f;
        }
        dr[g8 * 2] = make_float4(val[0], val[1], val[2], val[3]);
        dr[g8 * 2 + 1] = make_float4(val[4], val[5], val[6], val[7]);
    }
}

// Bounding k-range of the mask non-zeros per tile of `tile_n` packed rows.
__global__ void __launch_bounds__(256) mask_k_ranges_kernel(const float* __restrict__ mask, int N, int K,
                                                            const int32_t* __restrict__ row_of_out,
                                                            const int32_t* __restrict__ col_of_in, int tile_n,
                                                            int32_t* __restrict__ lo_hi) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= N) return;
    const int lane = threadIdx.x & 63;
    const float* mr = mask + (int64_t)o * K;
    int lo = 0x7fffffff, hi = -1;
    for (int i = lane; i < K; i += 64) {
        if (mr[i] != 0.0f) {
            const int c = col_of_in ? col_of_in[i] : i;
            lo = min(lo, c);
            hi = max(hi, c);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off, 64));
        hi = max(hi, __shfl_xor(hi, off, 64));
    }
    if (lane == 0 && hi >= 0) {
        const int t = (row_of_out ? row_of_out[o] : o) / tile_n;
        atomicMin(&lo_hi[2 * t], lo);
        atomicMax(&lo_hi[2 * t + 1], hi);
    }
}

__global__ void init_k_ranges_kernel(int32_t* __restrict__ lo_hi, int n_tiles) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_tiles) {
        lo_hi[2 * t] = 0x7fffffff;
        lo_hi[2 * t + 1] = -1;
    }
}

__global__ void finish_k_ranges_kernel(int32_t* __restrict__ lo_hi, int n_tiles, int tile_k, int k_padded) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    int lo = lo_hi[2 * t], hi = lo_hi[2 * t + 1];
    if (hi < 0) {
        lo = 0;
        hi = 0;                            // empty tile: no k-tiles at all
    } else {
        lo = (lo / tile_k) * tile_k;
        hi = min(((hi + tile_k) / tile_k) * tile_k, k_padded);
    }
    lo_hi[2 * t] = lo;
    lo_hi[2 * t + 1] = hi;
}

// ------------------------------------------------------------------------------------------
// GEMM (launch arguments, block mapping and epilogues: gemm_common.h)
// ------------------------------------------------------------------------------------------
template <int MREP, int NREP>
struct Tile {
    static constexpr int BM = WAVES * 16 * MREP;
    static constexpr int BN = 16 * NREP;
    static constexpr int A_FLOATS = BM * BK;
    static constexpr int B_FLOATS = BN * BK;
    static constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int STAGES = 2;
    static constexpr int LDS_BYTES = STAGES * STAGE_FLOATS * 4;
    // 1 KiB (16 rows x 64 B) per LDS-DMA wave instruction
    static constexpr int A_CHUNKS = BM / 16;
    static constexpr int B_CHUNKS = BN / 16;
};

// LDS-DMA staging through buffer descriptors (buffer_load_dwordx4 ... offen lds).
//
// A workgroup owns two descriptors: the activation rows from m0 on and the weight rows from n0 on.
// num_records ends each buffer at the last valid row, so rows past the end of a matrix read as 0 from
// the hardware range check -- no clamping, and no access outside the allocation.  A lane needs ONE
// VGPR per operand for all pieces of all k-tiles: its byte offset inside a 16-row piece (row l>>2,
// floats [4*(l&3), +4)); the piece's row offset is a uniform add, the k offset rides in soffset.
// The LDS image is [row][16 floats] linear, exactly base + 16*lane as the DMA requires.
struct StageCtx {
    __amdgpu_buffer_rsrc_t ra, rw;
    uint32_t voff_a, voff_w;      // per-lane byte offset inside a piece
    uint32_t piece_a, piece_w;    // bytes between consecutive 16-row pieces
};

__device__ inline StageCtx make_stage_ctx(const GemmArgs& g, int m0, int n0, int lane, int n_rows_w) {
    StageCtx c;
    constexpr int FLAGS = 0x00020000;     // gfx9 raw buffer, 32-bit data
    c.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a + (int64_t)m0 * g.lda), 0,
                                             (int)clamp_u32((int64_t)(g.B - m0) * g.lda * 4), FLAGS);
    c.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w + (int64_t)n0 * g.ldw), 0,
                                             (int)clamp_u32((int64_t)(n_rows_w - n0) * g.ldw * 4), FLAGS);
    const int r = lane >> 2, q = lane & 3;
    c.voff_a = (uint32_t)((r * g.lda + 4 * q) * 4);
    c.voff_w = (uint32_t)((r * g.ldw + 4 * q) * 4);
    c.piece_a = (uint32_t)(16 * g.lda * 4);
    c.piece_w = (uint32_t)(16 * g.ldw * 4);
    return c;
}

// Issue the LDS-DMA of one k-tile: chunk c (16 rows) is issued by wave (c % WAVES).
template <int MREP, int NREP>
__device__ inline void stage_tile(const StageCtx& sc, float* lds_stage, int k0, int wave) {
    using T = Tile<MREP, NREP>;
#pragma unroll
    for (int c = wave; c < T::A_CHUNKS + T::B_CHUNKS; c += WAVES) {
        if (c < T::A_CHUNKS) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sc.ra, (__attribute__((address_space(3))) void*)(lds_stage + c * 16 * BK),
                                                     16, sc.voff_a + (uint32_t)c * sc.piece_a, k0 * 4, 0, 0);
        } else {
            const int cb = c - T::A_CHUNKS;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sc.rw,
                                                     (__attribute__((address_space(3))) void*)(lds_stage + T::A_FLOATS + cb * 16 * BK),
                                                     16, sc.voff_w + (uint32_t)cb * sc.piece_w, k0 * 4, 0, 0);
        }
    }
}

// P parameters x FT feature groups per column tile (NREP = P * FT) for the fused epilogues.
template <int MREP, int NREP, int EPI, int P, int KSPL>
__global__ void __launch_bounds__(THREADS, (MREP == 1 ? 4 : 2)) gemm_kernel(GemmArgs g, int n_rows_w) {
    using T = Tile<MREP, NREP>;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Workgroup -> (row tile, column tile).  Workgroups are dealt round-robin over the 8 XCDs
    // (id % 8 labels the XCD) and each XCD has its own L2, so in mode 1 the 32 workgroups that are
    // co-resident on one XCD form an 8 (row tiles) x 4 (column tiles) super-tile: 8 activation
    // panels + 4 weight panels are fetched once into that L2 instead of 32 + 1.  Pure speed: any
    // placement gives the same result.
    int mt, ntp;
    if (!map_block(g, mt, ntp)) return;
    int nt = g.tile_order ? g.tile_order[ntp] : ntp;
    // split-K (short-and-wide products, e.g. the block GEMMs of the inverse: too few output tiles to fill 256 CUs):
    // the launch has ksplit x as many column positions; position -> (column tile, k slice)
    int k_slice = 0;
    if (g.ksplit > 1) {
        const int n_real = g.n_tiles / g.ksplit;
        k_slice = nt / n_real;
        nt -= k_slice * n_real;
    }
    if (g.tile_live && !g.tile_live[(int64_t)mt * g.n_tiles + nt]) return;
    const int m0 = mt * T::BM, n0 = nt * T::BN;

    int kb = 0, ke = g.k_padded;
    if (g.k_ranges) {
        kb = g.k_ranges[2 * nt];
        ke = g.k_ranges[2 * nt + 1];
    }
    if (g.ksplit > 1) {
        const int per = (((ke - kb) / BK + g.ksplit - 1) / g.ksplit) * BK;     // k per slice, whole tiles
        kb = min(ke, kb + k_slice * per);
        ke = min(ke, kb + per);
    }

    f32x4 acc[NREP][MREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
        for (int m = 0; m < MREP; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Two-stage LDS double buffer: the DMA of k-tile t+1 flies under the MFMAs of k-tile t; one
    // barrier per k-tile.
    const int nk = (ke - kb) / BK;
    const StageCtx sc = make_stage_ctx(g, m0, n0, lane, n_rows_w);
    if (nk > 0) stage_tile<MREP, NREP>(sc, lds, kb, wave);
    const int frag_off = (lane & 15) * BK + (lane >> 4) * 4;   // row (l&15), floats [4q, 4q+4)
    // The matrix pipe is the bottleneck, so it is started first after the barrier and the ~60
    // scalar/vector instructions that issue the next tile's LDS-DMA run in the shadow of MFMA groups.
    // Waves w and w+4 share a SIMD and leave the barrier together: they issue their DMA at
    // DIFFERENT column groups (N_DMA_A / N_DMA_B) so one of them always feeds the matrix pipe.
    constexpr int N_DMA_A = NREP >= 4 ? 2 : 0, N_DMA_B = NREP >= 4 ? NREP / 2 + 1 : NREP - 1;
    static_assert(NREP >= 2 && NREP > N_DMA_B && N_DMA_A != N_DMA_B, "column groups per wave");
    const bool late_half = wave >= WAVES / 2;
    for (int t = 0; t < nk; ++t) {
        if (!(g.diag & 8)) __syncthreads();   // DMA of tile t landed (vmcnt(0) + barrier); other buffer free
        float* cur = lds + (t & 1) * T::STAGE_FLOATS;
        const float* As = cur + (wave * 16 * MREP) * BK + frag_off;
        const float* Bs = cur + T::A_FLOATS + frag_off;
        f32x4 af[MREP];
#pragma unroll
        for (int m = 0; m < MREP; ++m) af[m] = *(const f32x4*)(As + m * 16 * BK);
        // B fragments are software-pipelined one column group ahead (bq[n & 1]) so a wave does not
        // depend on its SIMD partner to cover the LDS latency of every group.
        f32x4 bq[2];
        bq[0] = *(const f32x4*)(Bs);
        const bool dma = t + 1 < nk && !(g.diag & 4);
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
            if (n == N_DMA_A || n == N_DMA_B) {
                __builtin_amdgcn_sched_barrier(0);
                if (dma && late_half == (n == N_DMA_B))
                    stage_tile<MREP, NREP>(sc, lds + ((t + 1) & 1) * T::STAGE_FLOATS, kb + (t + 1) * BK, wave);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (n + 1 < NREP) bq[(n + 1) & 1] = *(const f32x4*)(Bs + (n + 1) * 16 * BK);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch AHEAD of this group's MFMAs
            const f32x4 bf = bq[n & 1];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MREP; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][s], bf[s], acc[n][m], 0, 0, 0);
        }
    }

    if (g.diag & 1) {   // timing-only build of the main loop: keep the accumulators alive, store nothing
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int m = 0; m < MREP; ++m) asm volatile("" ::"v"(acc[n][m]));
        return;
    }
    gemm_epilogue<MREP, NREP, EPI, P, KSPL>(g, acc, nt, n0, m0 + wave * 16 * MREP, lane, k_slice);
}

// Diagnostic: the matrix-pipe ceiling of THIS device for the GEMM's own instruction mix -- the same
// 2 x NREP accumulator tiles of v_mfma_f32_16x16x4_f32 per wave, operands in registers, no memory.
template <int MREP, int NREP>
__global__ void __launch_bounds__(THREADS, 2) mfma_peak_kernel(float* out, int iters) {
    f32x4 acc[NREP][MREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
        for (int m = 0; m < MREP; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 af[MREP], bf;
#pragma unroll
    for (int m = 0; m < MREP; ++m) af[m] = (f32x4){1.0f + threadIdx.x, 0.5f, 0.25f, 2.0f};
    bf = (f32x4){1e-3f * threadIdx.x, 1e-3f, 2e-3f, 3e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NREP; ++n)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < MREP; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][s], bf[s], acc[n][m], 0, 0, 0);
    }
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
        for (int m = 0; m < MREP; ++m) sum += acc[n][m][0] + acc[n][m][1] + acc[n][m][2] + acc[n][m][3];
    out[blockIdx.x * THREADS + threadIdx.x] = sum;
}

// ldj[b] (+)= sum_t partial[t][b]
__global__ void __launch_bounds__(256) ldj_reduce_kernel(const double* __restrict__ partial, int n_tiles, int B,
                                                         float* __restrict__ ldj, int accumulate) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int t = 0; t < n_tiles; ++t) s += partial[(int64_t)t * B + b];
    ldj[b] = accumulate ? (float)((double)ldj[b] + s) : (float)s;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
constexpr int LIN_MREP = 2, LIN_NREP = 16;          // 256 x 256 tile for the hidden layers
constexpr int NARROW_NREP = 2;                      // 256 x 32 tile: row slices of the blocked inverse


// The 256 x 32 narrow tile at half height (128 x 32, MREP = 1) while the launch would have fewer workgroups than twice the
// CUs: these products are bound by the latency of one workgroup's k-loop (a barrier and an LDS-DMA round trip per 16
// columns), and two resident workgroups per CU overlap theirs.  (A/B: TFEP_NARROW_HALF=0.)
static bool narrow_half_height(int B, int n_col_positions) {
    static const int on = env_int("TFEP_NARROW_HALF", 1);
    const long long wgs = (long long)((B + 255) / 256) * n_col_positions;
    return on && wgs < 512;
}

template <int MREP, int NREP, int EPI, int P, int KSPL>
static int launch_gemm(const GemmArgs& g, int n_rows_w, int n_col_tiles, hipStream_t s) {
    using T = Tile<MREP, NREP>;
    auto kern = gemm_kernel<MREP, NREP, EPI, P, KSPL>;
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
    ga.n_tiles = n_col_tiles * (g.ksplit > 1 ? g.ksplit : 1);
    ga.map_mode = block_map_mode();
    ga.diag = env_int("TFEP_DIAG", 0);
    const long long blocks = gemm_grid_blocks(ga.map_mode, ga.m_tiles, ga.n_tiles, ga.tile_list ? ga.n_tile_list : 0);
    if (blocks > 0x7fffffffLL) return fail(TFEP_ERR_INVALID_ARGUMENT, "gemm: grid too large");
    kern<<<dim3((unsigned)blocks), THREADS, T::LDS_BYTES, s>>>(ga, n_rows_w);
    return check_launch("gemm_kernel");
}

static int check_gemm_operands(const float* a, int64_t lda, const float* w, int64_t ldw, int k_padded) {
    TFEP_REQUIRE(a && w, "masked_linear: NULL operand");
    TFEP_REQUIRE(k_padded > 0 && k_padded % BK == 0, "masked_linear: k_padded=%d must be a positive multiple of %d", k_padded, BK);
    TFEP_REQUIRE(lda >= k_padded && ldw >= k_padded, "masked_linear: lda=%lld / ldw=%lld smaller than k_padded=%d",
                 (long long)lda, (long long)ldw, k_padded);
    TFEP_REQUIRE(lda % 4 == 0 && ldw % 4 == 0, "masked_linear: row strides must be multiples of 4 floats");
    TFEP_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)w % 16 == 0), "masked_linear: operands must be 16-byte aligned");
    return TFEP_OK;
}

}  // namespace tfep

namespace tfep {
// p[0 .. n) = 0 (grid-stride; see tfep_masked_weight_prepare for why not hipMemsetAsync)
__global__ void __launch_bounds__(256) fill_zero_kernel(float* __restrict__ p, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n && ((uintptr_t)(p + i) & 15) == 0) {
            *reinterpret_cast<float4*>(p + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (size_t j = i; j < n && j < i + 4; ++j) p[j] = 0.f;
        }
    }
}
}  // namespace tfep

using namespace tfep;

extern "C" {

int tfep_masked_linear_tile_m(void) { return Tile<LIN_MREP, LIN_NREP>::BM; }
int tfep_masked_linear_tile_n(void) { return Tile<LIN_MREP, LIN_NREP>::BN; }
// Alignment of k_padded and of the k-ranges: the larger of the two kernels' k-tiles (fp32: 16, split-f16: 32).
int tfep_masked_linear_tile_k(void) { return 2 * BK; }
int tfep_masked_linear_narrow_tile_n(void) { return Tile<LIN_MREP, NARROW_NREP>::BN; }
int tfep_split_wide_tile_n(void) { return split_wide_tile_n(); }
int tfep_split_half_wide_tile_n(void) { return split_half_wide_tile_n(); }
int tfep_fused_tile_features(void) { return FUSED_TILE_FEATURES; }

int tfep_masked_weight_prepare(const float* weight_v, const float* weight_g, const float* mask, int out_features,
                               int in_features, const int32_t* row_of_out, const int32_t* col_of_in, const int32_t* col_cut,
                               int clear, float* w_out, int n_rows_padded, int64_t ldw, void* stream) {
    TFEP_REQUIRE(weight_v && w_out, "masked_weight_prepare: NULL pointer");
    TFEP_REQUIRE(out_features >= 0 && in_features >= 0, "masked_weight_prepare: negative size");
    TFEP_REQUIRE(n_rows_padded >= out_features && ldw >= in_features, "masked_weight_prepare: output too small");
    hipStream_t s = (hipStream_t)stream;
    // Cleared by a kernel, not hipMemsetAsync: captured in a HIP graph a memset becomes a memset node, and those were seen
    // to leave garbage behind on replay (the padding rows of this buffer read back as ~1e36 by a later kernel of the same
    // graph, while eager runs were clean; an earlier 4-byte case was traced to the same node type).
    const size_t n_clear = (size_t)n_rows_padded * (size_t)ldw;
    if (clear && n_clear > 0) {
        fill_zero_kernel<<<(unsigned)((n_clear + 1023) / 1024 < 65535 * 16 ? (n_clear + 1023) / 1024 : 65535 * 16), 256, 0, s>>>(w_out, n_clear);
        int rc = check_launch("fill_zero_kernel");
        if (rc) return rc;
    }
    if (out_features == 0 || in_features == 0) return TFEP_OK;
    weight_prepare_kernel<<<(unsigned)((out_features + 3) / 4), 256, 0, s>>>(weight_v, weight_g, mask, out_features,
                                                                              in_features, row_of_out, col_of_in, col_cut, w_out, ldw);
    return check_launch("weight_prepare_kernel");
}

int tfep_masked_weight_prepare_prefix(const float* weight_v, const float* weight_g, int out_features, int in_features,
                                      const int32_t* row_of_out, const int32_t* in_of_col, const int32_t* col_cut,
                                      float* w_out, int n_rows_padded, int64_t ldw, void* stream) {
    TFEP_REQUIRE(weight_v && w_out && col_cut, "masked_weight_prepare_prefix: NULL pointer");
    TFEP_REQUIRE(out_features >= 0 && in_features >= 0, "masked_weight_prepare_prefix: negative size");
    TFEP_REQUIRE(n_rows_padded >= out_features && ldw >= ((in_features + 7) & ~7) && ldw % 4 == 0 && ((uintptr_t)w_out & 15) == 0,
                 "masked_weight_prepare_prefix: output too small or not aligned to 16 bytes");
    TFEP_REQUIRE((size_t)in_features * 4 <= 64 * 1024, "masked_weight_prepare_prefix: a row of %d weights does not fit the LDS stage",
                 in_features);
    if (out_features == 0 || in_features == 0) return TFEP_OK;
    weight_prepare_prefix_kernel<<<(unsigned)out_features, PFX32_THREADS, (size_t)in_features * 4, (hipStream_t)stream>>>(
        weight_v, weight_g, out_features, in_features, row_of_out, in_of_col, col_cut, w_out, ldw);
    return check_launch("weight_prepare_prefix_kernel");
}

int tfep_mask_k_ranges(const float* mask, int out_features, int in_features, const int32_t* row_of_out,
                       const int32_t* col_of_in, int tile_n, int tile_k, int n_tiles, int k_padded,
                       int32_t* k_ranges, void* stream) {
    TFEP_REQUIRE(k_ranges, "mask_k_ranges: NULL output");
    TFEP_REQUIRE(tile_n > 0 && tile_k > 0 && n_tiles >= 0, "mask_k_ranges: bad tile sizes");
    hipStream_t s = (hipStream_t)stream;
    if (n_tiles == 0) return TFEP_OK;
    TFEP_REQUIRE(mask, "mask_k_ranges: mask is NULL (pass k_ranges = NULL to the GEMM for dense weights)");
    init_k_ranges_kernel<<<(unsigned)((n_tiles + 255) / 256), 256, 0, s>>>(k_ranges, n_tiles);
    if (out_features > 0 && in_features > 0)
        mask_k_ranges_kernel<<<(unsigned)((out_features + 3) / 4), 256, 0, s>>>(mask, out_features, in_features,
                                                                                 row_of_out, col_of_in, tile_n, k_ranges);
    finish_k_ranges_kernel<<<(unsigned)((n_tiles + 255) / 256), 256, 0, s>>>(k_ranges, n_tiles, tile_k, k_padded);
    return check_launch("mask_k_ranges");
}

int tfep_masked_linear_forward(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                               const int32_t* k_ranges, const int32_t* tile_order, const int32_t* col_map, float* y,
                               int64_t ldy, int B, int N, int n_rows_w, int k_padded, int act, int tile_n,
                               void* stream) {
    TFEP_REQUIRE(B >= 0 && N >= 0 && n_rows_w >= N, "masked_linear: bad sizes B=%d N=%d rows=%d", B, N, n_rows_w);
    TFEP_REQUIRE(act == 0 || act == 1, "masked_linear: act must be 0 (identity) or 1 (ELU)");
    if (B == 0 || N == 0) return TFEP_OK;                  // (an empty batch has no storage: its pointers are NULL)
    int rc = check_gemm_operands(x, ldx, w, ldw, k_padded);
    if (rc) return rc;
    TFEP_REQUIRE(y, "masked_linear: y is NULL");
    GemmArgs g = {};
    g.a = x; g.lda = ldx; g.w = w; g.ldw = ldw; g.bias = bias; g.k_ranges = k_ranges; g.col_map = col_map;
    g.y = y; g.ldy = ldy; g.B = B; g.N = N; g.k_padded = k_padded; g.tile_order = tile_order;
    constexpr int WIDE_BN = Tile<LIN_MREP, LIN_NREP>::BN, NARROW_BN = Tile<LIN_MREP, NARROW_NREP>::BN;
    TFEP_REQUIRE(tile_n == 0 || tile_n == WIDE_BN || tile_n == NARROW_BN,
                 "masked_linear: tile_n=%d unsupported (0, %d or %d)", tile_n, WIDE_BN, NARROW_BN);
    if (tile_n == NARROW_BN) {
        // narrow column tile for the row slices of the blocked autoregressive inverse
        const int n_tiles = (N + tile_n - 1) / tile_n;
        if (narrow_half_height(B, n_tiles)) {
            if (act == 1) return launch_gemm<1, NARROW_NREP, EPI_ELU, 1, 1>(g, n_rows_w, n_tiles, (hipStream_t)stream);
            return launch_gemm<1, NARROW_NREP, EPI_LINEAR, 1, 1>(g, n_rows_w, n_tiles, (hipStream_t)stream);
        }
        if (act == 1) return launch_gemm<LIN_MREP, NARROW_NREP, EPI_ELU, 1, 1>(g, n_rows_w, n_tiles, (hipStream_t)stream);
        return launch_gemm<LIN_MREP, NARROW_NREP, EPI_LINEAR, 1, 1>(g, n_rows_w, n_tiles, (hipStream_t)stream);
    }
    const int n_tiles = (N + Tile<LIN_MREP, LIN_NREP>::BN - 1) / Tile<LIN_MREP, LIN_NREP>::BN;
    if (act == 1) return launch_gemm<LIN_MREP, LIN_NREP, EPI_ELU, 1, 1>(g, n_rows_w, n_tiles, (hipStream_t)stream);
    return launch_gemm<LIN_MREP, LIN_NREP, EPI_LINEAR, 1, 1>(g, n_rows_w, n_tiles, (hipStream_t)stream);
}

int tfep_masked_linear_gemm(const tfep_gemm_desc* d, void* stream) {
    TFEP_REQUIRE(d != nullptr, "masked_linear_gemm: NULL descriptor");
    TFEP_REQUIRE(d->B >= 0 && d->N >= 0 && d->n_rows_w >= 1, "masked_linear_gemm: bad sizes");
    TFEP_REQUIRE(d->act == 0 || d->act == 1, "masked_linear_gemm: act must be 0 or 1");
    if (d->B == 0 || d->N == 0) return TFEP_OK;            // (an empty batch has no storage: its pointers are NULL)
    int rc = check_gemm_operands(d->x, d->ldx, d->w, d->ldw, d->k_padded);
    if (rc) return rc;
    TFEP_REQUIRE(d->y, "masked_linear_gemm: y is NULL");
    GemmArgs g = {};
    g.a = d->x; g.lda = d->ldx; g.w = d->w; g.ldw = d->ldw; g.bias = d->bias; g.k_ranges = d->k_ranges;
    g.col_map = d->col_map; g.y = d->y; g.ldy = d->ldy; g.B = d->B; g.N = d->N; g.k_padded = d->k_padded;
    g.tile_order = d->tile_order; g.aux = d->elu_grad_of; g.ldaux = d->ld_elu_grad_of; g.accumulate = d->accumulate;
    g.tile_live = d->tile_live; g.pre_add = d->pre_add; g.ld_pre_add = d->ld_pre_add;
    if (d->tile_list) {
        TFEP_REQUIRE(d->n_tile_list > 0 && !d->tile_order && d->k_split <= 1 &&
                         (d->tile_n == 0 || d->tile_n == Tile<LIN_MREP, LIN_NREP>::BN ||
                          (d->split && (d->tile_n == split_wide_tile_n() || d->tile_n == split_half_wide_tile_n()))),
                     "masked_linear_gemm: tile_list needs n_tile_list > 0, a wide tile, no tile_order and no k_split");
        g.tile_list = d->tile_list; g.n_tile_list = d->n_tile_list;
    }
    if (d->k_split > 1) {
        TFEP_REQUIRE(!d->split_out && !d->accumulate && !d->tile_order && !d->tile_live && !d->elu_grad_of && d->act == 0,
                     "masked_linear_gemm: k_split needs a plain linear product (no activation / accumulate / tile_order / tile_live)");
        TFEP_REQUIRE(d->k_split <= 64 && d->slab_stride > 0, "masked_linear_gemm: bad k_split / slab_stride");
        g.ksplit = d->k_split; g.slab_stride = d->slab_stride;
    }
    if (d->split) {
        constexpr int SPLIT_BN = Tile<LIN_MREP, LIN_NREP>::BN;
        const bool xwide = d->tile_n == split_wide_tile_n(), hwide = d->tile_n == split_half_wide_tile_n();
        TFEP_REQUIRE(d->tile_n == 0 || d->tile_n == SPLIT_BN || xwide || hwide,
                     "masked_linear_gemm: split operands need a wide tile (0, %d, %d or %d)", SPLIT_BN, split_wide_tile_n(),
                     split_half_wide_tile_n());
        TFEP_REQUIRE(!(xwide || hwide) || (d->act == 0 && !d->split_out),
                     "masked_linear_gemm: the %d- and %d-column tiles take the plain linear product only", split_wide_tile_n(),
                     split_half_wide_tile_n());
        g.a_inv_scale = d->x_inv_scale; g.w_inv_scale = d->w_inv_scale;
        if (d->split_out) {
            TFEP_REQUIRE(d->act == 1 && d->y_inv_scale, "masked_linear_gemm: split_out needs act = 1 (ELU) and y_inv_scale");
            g.y_inv_scale = d->y_inv_scale; g.w_l1max = d->w_l1max; g.bias_absmax = d->bias_absmax;
        }
        return launch_split_linear(g, d->n_rows_w, d->act, (hipStream_t)stream, xwide, hwide);
    }
    constexpr int WIDE_BN = Tile<LIN_MREP, LIN_NREP>::BN, NARROW_BN = Tile<LIN_MREP, NARROW_NREP>::BN;
    TFEP_REQUIRE(d->tile_n == 0 || d->tile_n == WIDE_BN || d->tile_n == NARROW_BN,
                 "masked_linear_gemm: tile_n=%d unsupported (0, %d or %d)", d->tile_n, WIDE_BN, NARROW_BN);
    if (d->tile_n == NARROW_BN) {
        const int n_tiles = (d->N + NARROW_BN - 1) / NARROW_BN;
        if (!d->tile_live && narrow_half_height(d->B, n_tiles * (d->k_split > 1 ? d->k_split : 1))) {    // (tile_live: per 256-row tile)
            if (d->act == 1) return launch_gemm<1, NARROW_NREP, EPI_ELU, 1, 1>(g, d->n_rows_w, n_tiles, (hipStream_t)stream);
            return launch_gemm<1, NARROW_NREP, EPI_LINEAR, 1, 1>(g, d->n_rows_w, n_tiles, (hipStream_t)stream);
        }
        if (d->act == 1) return launch_gemm<LIN_MREP, NARROW_NREP, EPI_ELU, 1, 1>(g, d->n_rows_w, n_tiles, (hipStream_t)stream);
        return launch_gemm<LIN_MREP, NARROW_NREP, EPI_LINEAR, 1, 1>(g, d->n_rows_w, n_tiles, (hipStream_t)stream);
    }
    const int n_tiles = (d->N + WIDE_BN - 1) / WIDE_BN;
    if (d->act == 1) return launch_gemm<LIN_MREP, LIN_NREP, EPI_ELU, 1, 1>(g, d->n_rows_w, n_tiles, (hipStream_t)stream);
    return launch_gemm<LIN_MREP, LIN_NREP, EPI_LINEAR, 1, 1>(g, d->n_rows_w, n_tiles, (hipStream_t)stream);
}

int tfep_diag_mfma_peak(float* scratch, int blocks, int iters, void* stream) {
    TFEP_REQUIRE(scratch && blocks > 0 && iters > 0, "diag_mfma_peak: bad arguments");
    mfma_peak_kernel<2, 25><<<blocks, THREADS, 0, (hipStream_t)stream>>>(scratch, iters);
    return check_launch("mfma_peak_kernel");
}

// parameters per feature of a spline descriptor (spline_n_params)
static int desc_n_params(const tfep_spline_desc* d) {
    return spline_n_params(d->n_bins, d->circular != 0, d->identity_boundary_slopes != 0, d->learn_lower_bound != 0,
                           d->learn_upper_bound != 0);
}

int tfep_fused_supported(int kind, const tfep_spline_desc* d) {
    if (kind == TFEP_FUSED_AFFINE) return 1;
    if (kind == TFEP_FUSED_SPLINE && d) {
        if (d->n_bins != 8 && d->n_bins != 5 && d->n_bins != 4) return 0;
        if (d->circular && (d->learn_lower_bound || d->learn_upper_bound)) return 0;       // (not a valid spline)
        // (8 bins with learnable bounds: 26 / 27 accumulator tiles per wave -- 432 of the split kernel's 512 registers)
        return desc_n_params(d) <= 27;
    }
    return 0;
}

int tfep_fused_tile_columns(int kind, const tfep_spline_desc* d) {
    if (kind == TFEP_FUSED_AFFINE) return 16 * 16;      // P = 2, FT = 8
    if (kind == TFEP_FUSED_SPLINE && tfep_fused_supported(kind, d)) return 16 * desc_n_params(d);   // FT = 1
    return fail(TFEP_ERR_UNSUPPORTED, "fused: unsupported transformer configuration");
}

static int fused_forward(const float* h, int64_t ldh, const float* w, int64_t ldw, const float* bias_packed,
                         const int32_t* k_ranges, const int32_t* tile_order, int kind, const tfep_spline_desc* desc,
                         const float* x, int64_t ldx, float* y, int64_t ldy, const int32_t* feat_index,
                         const int32_t* feat_tr, int n_feature_slots, double* ldj_partial, float* log_det_J,
                         int accumulate, int B, int n_rows_w, int k_padded, bool split, const float* h_inv_scale,
                         const float* w_inv_scale, void* stream, int feature_major = 0, float* theta_out = nullptr,
                         int64_t ld_theta = 0) {
    TFEP_REQUIRE(n_feature_slots > 0 && n_feature_slots % FUSED_TILE_FEATURES == 0,
                 "fused: n_feature_slots=%d must be a positive multiple of %d", n_feature_slots, FUSED_TILE_FEATURES);
    if (!tfep_fused_supported(kind, desc)) return fail(TFEP_ERR_UNSUPPORTED, "fused: unsupported transformer configuration");
    TFEP_REQUIRE(B >= 0, "fused: negative batch");
    if (B == 0) return TFEP_OK;                            // (an empty batch has no storage: its pointers are NULL)
    int rc = check_gemm_operands(h, ldh, w, ldw, k_padded);
    if (rc) return rc;
    TFEP_REQUIRE(x && y && feat_index && ldj_partial && log_det_J, "fused: NULL pointer");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g = {};
    g.a = h; g.lda = ldh; g.w = w; g.ldw = ldw; g.bias = bias_packed; g.k_ranges = k_ranges;
    g.B = B; g.k_padded = k_padded; g.tile_order = tile_order;
    g.fu.x = x; g.fu.ldx = ldx; g.fu.y = y; g.fu.ldy = ldy; g.fu.feat_index = feat_index; g.fu.feat_tr = feat_tr;
    g.fu.ldj_partial = ldj_partial;
    g.fu.feature_major = feature_major; g.fu.theta_out = theta_out; g.fu.ld_theta = ld_theta;
    g.a_inv_scale = h_inv_scale; g.w_inv_scale = w_inv_scale;
    const int n_groups = n_feature_slots / FUSED_TILE_FEATURES;
    if (kind == TFEP_FUSED_AFFINE) {
        constexpr int P = 2, FT = 8;
        TFEP_REQUIRE(n_groups % FT == 0, "fused affine: feature slots must be a multiple of %d", FT * 16);
        TFEP_REQUIRE(n_rows_w >= n_feature_slots * P, "fused: weight has too few rows");
        g.N = n_feature_slots * P;
        rc = split ? launch_split_fused(g, n_rows_w, kind, n_groups / FT, s)
                   : launch_gemm<2, P * FT, EPI_AFFINE, P, 1>(g, n_rows_w, n_groups / FT, s);
    } else {
        const int KS = desc->n_bins, P = desc_n_params(desc);
        TFEP_REQUIRE(feat_tr && desc->x0 && desc->xf && desc->y0 && desc->yf, "fused spline: NULL descriptor arrays");
        // (feature-major rows: the dead slots of the last tile lie past the packing; the kernel reads them as zeros)
        TFEP_REQUIRE(n_rows_w >= (feature_major ? n_feature_slots - FUSED_TILE_FEATURES + 1 : n_feature_slots) * P,
                     "fused: weight has too few rows");
        g.N = n_feature_slots * P;
        g.fu.x0 = desc->x0; g.fu.xf = desc->xf; g.fu.y0 = desc->y0; g.fu.yf = desc->yf;
        g.fu.sf.K = KS; g.fu.sf.circular = desc->circular != 0; g.fu.sf.identity = desc->identity_boundary_slopes != 0;
        g.fu.sf.learn_lower = desc->learn_lower_bound != 0; g.fu.sf.learn_upper = desc->learn_upper_bound != 0;
        g.fu.sf.min_bin = desc->min_bin_size; g.fu.sf.min_slope = desc->min_slope;
        g.fu.sf.slope_offset = (float)log(exp(1.0 - (double)desc->min_slope) - 1.0);
        if (split) {
            rc = launch_split_fused(g, n_rows_w, kind, n_groups, s);
        } else {
            rc = fail(TFEP_ERR_UNSUPPORTED, "fused: no kernel for this spline layout");
#define TFEP_FUSED_SPLINE(KK, PP) \
    if (KS == KK && P == PP) rc = launch_gemm<2, PP, EPI_SPLINE, PP, KK>(g, n_rows_w, n_groups, s);
            // (identity slopes + both bounds learnable count 3 K + 1 parameters like the plain layout: EPI_SPLINE_IDB)
            const bool idb = g.fu.sf.identity && P == 3 * KS + 1;
#define TFEP_FUSED_SPLINE_IDB(KK) \
    if (KS == KK && idb) rc = launch_gemm<2, 3 * KK + 1, EPI_SPLINE_IDB, 3 * KK + 1, KK>(g, n_rows_w, n_groups, s);
            TFEP_FUSED_SPLINE_IDB(8) TFEP_FUSED_SPLINE_IDB(5) TFEP_FUSED_SPLINE_IDB(4)
#undef TFEP_FUSED_SPLINE_IDB
            if (!idb) {
            TFEP_FUSED_SPLINE(8, 25) TFEP_FUSED_SPLINE(8, 23) TFEP_FUSED_SPLINE(8, 24) TFEP_FUSED_SPLINE(8, 26) TFEP_FUSED_SPLINE(8, 27)
            TFEP_FUSED_SPLINE(5, 16) TFEP_FUSED_SPLINE(5, 14) TFEP_FUSED_SPLINE(5, 15) TFEP_FUSED_SPLINE(5, 17) TFEP_FUSED_SPLINE(5, 18)
            TFEP_FUSED_SPLINE(4, 13) TFEP_FUSED_SPLINE(4, 11) TFEP_FUSED_SPLINE(4, 12) TFEP_FUSED_SPLINE(4, 14) TFEP_FUSED_SPLINE(4, 15)
            }
#undef TFEP_FUSED_SPLINE
        }
    }
    if (rc) return rc;
    ldj_reduce_kernel<<<(unsigned)((B + 255) / 256), 256, 0, s>>>(ldj_partial, n_groups, B, log_det_J, accumulate);
    return check_launch("ldj_reduce_kernel");
}

int tfep_fused_output_transformer_forward(const float* h, int64_t ldh, const float* w, int64_t ldw,
                                          const float* bias_packed, const int32_t* k_ranges,
                                          const int32_t* tile_order, int kind,
                                          const tfep_spline_desc* desc, const float* x, int64_t ldx, float* y,
                                          int64_t ldy, const int32_t* feat_index, const int32_t* feat_tr,
                                          int n_feature_slots, double* ldj_partial, float* log_det_J, int accumulate,
                                          int B, int n_rows_w, int k_padded, void* stream) {
    return fused_forward(h, ldh, w, ldw, bias_packed, k_ranges, tile_order, kind, desc, x, ldx, y, ldy, feat_index, feat_tr,
                         n_feature_slots, ldj_partial, log_det_J, accumulate, B, n_rows_w, k_padded, false, nullptr,
                         nullptr, stream);
}

int tfep_fused_output_transformer_forward_split(const void* h_split, int64_t ldh, const float* h_inv_scale,
                                                const void* w_split, int64_t ldw, const float* w_inv_scale,
                                                const float* bias_packed, const int32_t* k_ranges,
                                                const int32_t* tile_order, int kind, const tfep_spline_desc* desc,
                                                const float* x, int64_t ldx, float* y, int64_t ldy,
                                                const int32_t* feat_index, const int32_t* feat_tr, int n_feature_slots,
                                                double* ldj_partial, float* log_det_J, int accumulate, int B,
                                                int n_rows_w, int k_padded, void* stream) {
    TFEP_REQUIRE(h_inv_scale && w_inv_scale, "fused split: NULL scale pointer");
    return fused_forward((const float*)h_split, ldh, (const float*)w_split, ldw, bias_packed, k_ranges, tile_order, kind,
                         desc, x, ldx, y, ldy, feat_index, feat_tr, n_feature_slots, ldj_partial, log_det_J, accumulate, B,
                         n_rows_w, k_padded, true, h_inv_scale, w_inv_scale, stream);
}

int tfep_fused_saving_supported(const tfep_spline_desc* d) {
    if (!d || !tfep_fused_supported(TFEP_FUSED_SPLINE, d)) return 0;
    SplineFlags f = {};
    f.K = d->n_bins; f.circular = d->circular != 0; f.identity = d->identity_boundary_slopes != 0;
    f.learn_lower = d->learn_lower_bound != 0; f.learn_upper = d->learn_upper_bound != 0;
    return split_fused_saving_supported(f) ? 1 : 0;
}

int tfep_fused_output_transformer_forward_split_saving(const void* h_split, int64_t ldh, const float* h_inv_scale,
                                                       const void* w_split, int64_t ldw, const float* w_inv_scale,
                                                       const float* bias_packed, const int32_t* k_ranges,
                                                       const int32_t* tile_order, const tfep_spline_desc* desc,
                                                       const float* x, int64_t ldx, float* y, int64_t ldy,
                                                       const int32_t* feat_index, const int32_t* feat_tr, int n_feature_slots,
                                                       double* ldj_partial, float* log_det_J, int accumulate, int B,
                                                       int n_rows_w, int k_padded, int w_feature_major, float* theta_out,
                                                       int64_t ld_theta, void* stream) {
    TFEP_REQUIRE(h_inv_scale && w_inv_scale, "fused split: NULL scale pointer");
    TFEP_REQUIRE(desc, "fused saving: NULL spline descriptor");
    TFEP_REQUIRE(w_feature_major, "fused saving: the kernel takes feature-major weight rows (w_feature_major = 1) only");
    if (theta_out)
        TFEP_REQUIRE(ld_theta >= (int64_t)(n_feature_slots - FUSED_TILE_FEATURES + 1) * desc_n_params(desc),
                     "fused saving: theta rows too short");
    return fused_forward((const float*)h_split, ldh, (const float*)w_split, ldw, bias_packed, k_ranges, tile_order,
                         TFEP_FUSED_SPLINE, desc, x, ldx, y, ldy, feat_index, feat_tr, n_feature_slots, ldj_partial, log_det_J,
                         accumulate, B, n_rows_w, k_padded, true, h_inv_scale, w_inv_scale, stream, w_feature_major ? 1 : 0,
                         theta_out, ld_theta);
}

}  // extern "C"
