// One MAF layer in ONE launch: the MADE conditioner (every masked linear, ELU between them) and the transformer with its
// log|det J| -- for conditioners small enough that the weights stay cache-resident (BASELINE config 4-ii: Moebius maps of
// 512 torsions as unit 2-vectors, MADE 1024 -> 1024 -> 1024 -> 1024, 3.1 M weights).
//
// Reference: conditioners/made.py:286-329, 355 (three MaskedLinear + ELU), masked.py:265-277, transformers/moebius.py:374-478,
// flows/autoregressive.py:144-177.
//
// Launched as four kernels per layer (split rows, two hidden GEMMs, output GEMM, Moebius map) every (B, 1024) intermediate
// made a round trip through HBM and every GEMM -- 16 to 32 k-tiles deep -- paid its prologue and epilogue with nothing to
// overlap them: 10.6 ms for the 4-layer flow at B = 131 072, 0.05 of the HBM bound, 0.19 of the matrix-pipe bound.  Here a
// workgroup (4 waves, one per SIMD) OWNS 256 sample rows for the whole layer:
//   for every masked linear, for every column tile: the split-f16 k-loop of split_gemm_kernel.h (weights double-buffered
//     in LDS by LDS-DMA, a wave's own 64 activation rows in its private LDS region), over the tile's k-range only
//     (block-triangular masks: hidden units sorted by degree);
//   hidden layers: ELU, the bound-based row scale, and the activations written straight as split rows -- to a scratch panel
//     that only THIS wave reads back (rows are private to a wave from the first layer to the last: no cross-wave or
//     cross-workgroup dependency, the only barriers are the k-loop's own), through L2 / the Infinity Cache;
//   output layer: the parameters of 32 columns at a time go through the wave's LDS region and a run-time loop evaluates the
//     Moebius map in fp64 (moebius.h, the stand-alone kernel's arithmetic) -- 16 lanes per row, one 2-vector per lane --
//     stores y and keeps the rows' log|det J| in LDS until the last column tile.
// MEASURED (round 4, profiles/r04_cfg4ii_layer_kernel.txt; cfg4-ii, 4 layers, B = 131 072; per-phase cycle counters of
// TFEP_DIAG_MAF_LAYER=1): the four launches 10.6 ms; this kernel with the 256-column tile of split_gemm_kernel.h (256
// accumulator registers, one workgroup per CU) 12.5 ms, with 128-column tiles and two workgroups per CU (what is built) 11.5
// ms -- SLOWER, so the host keeps the launch-by-launch path unless `layer_kernel = True`.  Why:
//   (1) a workgroup that is alone with its 256 rows re-reads its activation panel once per column tile (8 x 1 MB per layer
//       at 128 columns), and with every CU holding panels of its own (512 MB in flight) those reads come from HBM: 7.4 GB
//       per layer, a k-tile every ~5 000 cycles where its matrix products take 1 536.  The launch-by-launch GEMMs share a
//       panel among the 4 column tiles that are co-resident on an XCD through that XCD's L2 (gemm_common.h: map_block);
//   (2) the epilogues are vector code on a wave that issues one instruction per ~8 cycles when it is alone on its SIMD
//       (tools/probe/valu_rate_probe.hip): the fp64 Moebius map of a row tile costs 1.2 - 1.4 M cycles here, as much as
//       its k-loops, where the stand-alone kernel (8 waves per SIMD) needs 0.3 ms per layer for the whole batch.
// The floor of this blocking is ~4 ms for the flow either way (HBM re-reads at 256 columns, or L2 weight streams at 64
// rows per workgroup): the route to the 5 ms this kernel was built for is workgroups that share a row panel through L2
// AND epilogue work moved onto co-resident waves -- not built.
// HBM traffic per layer: x (split rows + fp32) in, y out, the scratch panels' write-back and the re-reads of (1).
#include "split_gemm_kernel.h"
#include "moebius.h"

namespace tfep {

constexpr int ML_MAX_LINEARS = 4;
#ifndef TFEP_ML_NREP
#define TFEP_ML_NREP 8
#endif
constexpr int ML_NREP = TFEP_ML_NREP;            // 8: 128-column tiles, 128 accumulator registers, two workgroups per CU
constexpr int ML_OCC = ML_NREP <= 8 ? 2 : 1;     //    (probe builds: -DTFEP_ML_NREP=16, one workgroup per CU)
constexpr int ML_PIECE = 32;                     // columns per pass of the epilogues' LDS stage
constexpr int ML_PITCH = ML_PIECE + 4;           // floats per staged row

struct MafLayerArgs {
    int B, n_lin;
    const float* a0; int64_t lda0; const float* a0_inv_scale;          // split rows of the conditioner input
    const float* w[ML_MAX_LINEARS]; int64_t ldw[ML_MAX_LINEARS];
    int n_rows_w[ML_MAX_LINEARS], n_out[ML_MAX_LINEARS], n_tiles[ML_MAX_LINEARS];
    const float* w_scales[ML_MAX_LINEARS];                              // [1/scale, -, max_j sum_k |w_jk|, -]
    const float* bias[ML_MAX_LINEARS]; const float* bias_absmax[ML_MAX_LINEARS];
    const int32_t* k_ranges[ML_MAX_LINEARS];
    float* h[2]; int64_t ldh;                                           // scratch split rows (ping-pong between layers)
    const float* x; int64_t ldx; float* y; int64_t ldy; float* ldj;
    int n_features, mb_unit_sphere; float mb_max_radius;
};

// The k-loop of split_gemm_kernel (same schedule: B fragments two column groups ahead, the next tile's LDS-DMA one
// instruction after each group of four MFMAs), as a function of its own so that one kernel can run it for every tile of
// every layer.  acc is indexed by constants only.
template <int NREP>
__device__ __forceinline__ void ml_mainloop(f32x4 (&acc)[NREP][SMREP], const SplitCtx& sc, char* slds, int kb, int nk, int wave,
                                            int lane) {
    using T = STile<NREP>;
    char* a_wave = slds + wave * A_WAVE_BYTES;
    char* b_base = slds + T::A_BYTES;
    if (nk > 0) {
#pragma unroll
        for (int d = 0; d < T::N_DMA; ++d) split_dma<NREP>(sc, a_wave, b_base, kb, wave, d);
    }
    const int fr = lane & 15, fg = lane >> 4;
    const int fe = (fr >> 1) & 7, fsw = (fe & 1) | (fe & 4);
    const int off_hi = fr * ROW_BYTES + (((2 * fg) ^ fsw) << 4);
    const int off_lo = fr * ROW_BYTES + (((2 * fg + 1) ^ fsw) << 4);
    auto tile = [&](auto dma_c, int t) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma_c)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* Bs = b_base + (t & 1) * T::B_BYTES;
        char* Bn = b_base + ((t + 1) & 1) * T::B_BYTES;
        f16x8 ah[SMREP], al[SMREP];
#pragma unroll
        for (int m = 0; m < SMREP; ++m) {
            ah[m] = *(const f16x8*)(a_wave + m * 16 * ROW_BYTES + off_hi);
            al[m] = *(const f16x8*)(a_wave + m * 16 * ROW_BYTES + off_lo);
        }
        f16x8 bh[B_AHEAD + 1], bl[B_AHEAD + 1];
        static_for<0, B_AHEAD>([&](auto pc) __attribute__((always_inline)) {
            if constexpr (pc.value < NREP) {
                bh[pc.value] = *(const f16x8*)(Bs + pc.value * 16 * ROW_BYTES + off_hi);
                bl[pc.value] = *(const f16x8*)(Bs + pc.value * 16 * ROW_BYTES + off_lo);
            }
        });
        const int k_next = kb + (t + 1) * SBK;
        auto dma_slot = [&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            if constexpr (DMA && q < T::N_DMA) {
                constexpr int dd = q < T::B_DMA ? T::A_DMA + q : q - T::B_DMA;
                static_assert(q < T::B_DMA || q / 3 >= 1, "A DMA before the A fragments are read");
                __builtin_amdgcn_sched_barrier(0);
                split_dma<NREP>(sc, a_wave, Bn, k_next, wave, dd);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
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
    if (nk > 0) {
        tile(std::false_type{}, nk - 1);
        // the MFMAs are inline asm: leave their result latency behind here, then re-define the accumulators so that every
        // later use (epilogue arithmetic, spill stores) depends on a value that exists only after the wait (see
        // split_gemm_kernel.h)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        static_for<0, NREP * SMREP>([&](auto ic) __attribute__((always_inline)) {
            constexpr int n = ic.value / SMREP, m = ic.value % SMREP;
            redefine<(n < N_ACC_AGPR)>(acc[n][m]);
        });
    }
}

// TFEP_DIAG_MAF_LAYER: cycles of wave 0 of every workgroup in [k-loops, hidden epilogues, Moebius epilogues, whole kernel], [4] = workgroups
static __device__ unsigned long long g_ml_cycles[8];

// ELU for the split rows of the next layer: v > 0 ? v : exp(v) - 1 with the hardware exponential (v_exp_f32) where the
// result is not small and the series to v^6 where it is -- ~5e-7 relative everywhere, a third of expm1f's instructions, on
// values the split format then carries to 2^-22 of their row's bound.
__device__ __forceinline__ float elu_fast(float v) {
    const float e = __expf(v) - 1.0f;
    float s = fmaf(v, 1.0f / 720.0f, 1.0f / 120.0f);
    s = fmaf(v, s, 1.0f / 24.0f);
    s = fmaf(v, s, 1.0f / 6.0f);
    s = fmaf(v, s, 0.5f);
    s = fmaf(v, s, 1.0f);
    s *= v;
    const float neg = v > -0.25f ? s : e;
    return v > 0.f ? v : neg;
}

template <bool UNIT, bool DIAG>
__global__ void __launch_bounds__(STHREADS, ML_OCC) maf_layer_moebius2_kernel(MafLayerArgs a) {
    constexpr int NREP = ML_NREP;
    using T = STile<NREP>;
    extern __shared__ __attribute__((aligned(16))) char slds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned long long t_kernel = DIAG ? __builtin_readcyclecounter() : 0ull;
    const int m0 = blockIdx.x * T::BM;
    const int wrow0 = m0 + wave * 16 * SMREP;
    const int cj = lane & 15, rq = (lane >> 4) * 4;

    // Per-row scales of the wave's 64 rows, in LDS behind the operand stages (nothing but the accumulators and the loop state
    // is live across a k-loop: with two workgroups per CU a wave has 128 vector registers beside its 128 accumulators, and
    // scales kept in registers were spilled and re-loaded one by one in the epilogues): [0] 1/scale of the activations the
    // current layer reads, [1] that times the weights' 1/scale (un-scaling of the accumulators), [2] the scale of the split
    // rows the current layer writes.
    float* const row_scales = reinterpret_cast<float*>(slds + T::LDS_BYTES + SWAVES * 64 * 8) + wave * (3 * 64);
    row_scales[lane] = wrow0 + lane < a.B ? a.a0_inv_scale[wrow0 + lane] : 0.f;
    __builtin_amdgcn_wave_barrier();
    // log|det J| of the wave's 64 rows, summed over every column tile: LDS, behind the operand stages
    double* ldj_lds = reinterpret_cast<double*>(slds + T::LDS_BYTES) + wave * 64;
    ldj_lds[lane] = 0.0;
    __builtin_amdgcn_wave_barrier();

    for (int l = 0; l < a.n_lin; ++l) {
        const bool last = l + 1 == a.n_lin;
        const float* A = l == 0 ? a.a0 : a.h[(l - 1) & 1];
        const int64_t lda = l == 0 ? a.lda0 : a.ldh;
        const float* W = a.w[l];
        const int64_t ldw = a.ldw[l];
        const float* bias = a.bias[l];
        const int N = a.n_out[l];
        const float ws = a.w_scales[l][0];
        // un-scaling of the accumulators, and (hidden layers) the scale of the split rows written for the next layer:
        //   |ELU(x W^T + b)| <= max(1, max|x_row| max_j sum_k |w_jk| + max|b|),  max|x_row| < 2^15 / scale(row)
        {
            const float wl1 = last ? 0.f : a.w_scales[l][2], bmax = last ? 0.f : a.bias_absmax[l][0];
            const float ai = row_scales[lane];
            row_scales[64 + lane] = ai * ws;
            row_scales[128 + lane] = pow2_scale_for(fmaxf(1.f, 32768.f * ai * wl1 + bmax));
            __builtin_amdgcn_wave_barrier();
        }
        SplitCtx sc;
        constexpr int FLAGS = 0x00020000;
        {
            const int rows_a = min(a.B - m0, T::BM);
            sc.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (int64_t)m0 * lda), 0,
                                                      (int)clamp_u32((int64_t)rows_a * lda * 4), FLAGS);
            const int drow = lane >> 3, p0 = (lane & 7) ^ ((lane >> 4) & 1);
            const uint32_t arow = (uint32_t)(((int64_t)(wave * 16 * SMREP + drow) * lda) * 4);
            sc.va_even = arow + p0 * 16;
            sc.va_odd = arow + (p0 ^ 4) * 16;
            sc.vw = (uint32_t)(((int64_t)drow * ldw) * 4) + (((wave & 1) ? (p0 ^ 4) : p0) * 16);
            sc.piece_a = (uint32_t)(8 * lda * 4);
            sc.piece_w = (uint32_t)(8 * ldw * 4);
        }
        for (int nt = 0; nt < a.n_tiles[l]; ++nt) {
            const int n0 = nt * T::BN;
            const int kb = a.k_ranges[l][2 * nt], ke = a.k_ranges[l][2 * nt + 1];
            const int nk = (ke - kb) / SBK;
            const int rows_w = min(a.n_rows_w[l] - n0, T::BN);
            sc.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (int64_t)n0 * ldw), 0,
                                                      (int)clamp_u32((int64_t)rows_w * ldw * 4), FLAGS);
            f32x4 acc[NREP][SMREP];
            static_for<0, NREP * SMREP>([&](auto ic) __attribute__((always_inline)) {
                acc[ic.value / SMREP][ic.value % SMREP] = (f32x4){0.f, 0.f, 0.f, 0.f};
            });
            __syncthreads();                            // the previous tile's k-loop / epilogue stage is done with LDS in every wave
            const unsigned long long t_tile = DIAG ? __builtin_readcyclecounter() : 0ull;
            ml_mainloop<NREP>(acc, sc, slds, kb, nk, wave, lane);

            const unsigned long long t_loop = DIAG ? __builtin_readcyclecounter() : 0ull;
            __syncthreads();                            // every wave is done with the operand stages: LDS is reused below
            float* stage = reinterpret_cast<float*>(slds + wave * (64 * ML_PITCH * 4));
            f32x4 rs[SMREP], so[SMREP];
            static_for<0, SMREP>([&](auto mc) __attribute__((always_inline)) {
                rs[mc.value] = *reinterpret_cast<const f32x4_alias*>(row_scales + 64 + mc.value * 16 + rq);
                so[mc.value] = *reinterpret_cast<const f32x4_alias*>(row_scales + 128 + mc.value * 16 + rq);
            });
            float bias_n[NREP];                          // the tile's bias values: all in flight at once
            static_for<0, NREP>([&](auto nc) __attribute__((always_inline)) {
                const int col = n0 + nc.value * 16 + cj;
                bias_n[nc.value] = col < N ? bias[col] : 0.f;
            });
            static_assert(SWAVES * 64 * ML_PITCH * 4 <= T::LDS_BYTES, "epilogue stage does not fit in LDS");
            static_assert(ML_PIECE == 32 && NREP % 2 == 0, "a pass of the epilogue stages two column groups");
            if (!last) {
                // ---- ELU, split rows for the next layer (the arithmetic of EPI_ELU_SPLIT, split_gemm_kernel.h; ELU by elu_fast)
                float* out = a.h[l & 1];
                static_for<0, NREP / 2>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = qc.value;
                    static_for<0, 2>([&](auto n2c) __attribute__((always_inline)) {
                        constexpr int n = 2 * q + n2c.value;
                        const int col = n0 + n * 16 + cj;
                        const bool in_range = col < N;
                        const float bv = bias_n[n];
                        static_for<0, SMREP * 4>([&](auto ic) __attribute__((always_inline)) {
                            constexpr int m = ic.value / 4, i = ic.value % 4;
                            const float v = in_range ? elu_fast(acc[n][m][i] * rs[m][i] + bv) * so[m][i] : 0.f;
                            stage[(m * 16 + rq + i) * ML_PITCH + n2c.value * 16 + cj] = v;
                        });
                    });
                    __builtin_amdgcn_wave_barrier();
                    // a row's 32 columns leave as 4 groups of 8 (16 bytes of hi halves, 16 of lo halves): 4 lanes write 128
                    // contiguous bytes of a row, 16 rows per instruction
                    const int grp = lane & 3;
                    const int colg = n0 + q * ML_PIECE + grp * 8;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row_l = it * 16 + (lane >> 2);
                        const int row = wrow0 + row_l;
                        const f32x4_alias lo4 = *reinterpret_cast<const f32x4_alias*>(stage + row_l * ML_PITCH + grp * 8);
                        const f32x4_alias hi4 = *reinterpret_cast<const f32x4_alias*>(stage + row_l * ML_PITCH + grp * 8 + 4);
                        const float v8[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                        f16x8 hi, lo;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const _Float16 hh = (_Float16)v8[j];
                            hi[j] = hh;
                            lo[j] = (_Float16)(v8[j] - (float)hh);
                        }
                        if (row < a.B && colg < N) {
                            uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + (int64_t)row * a.ldh * 4 + (int64_t)(colg >> 3) * 32);
                            dst[0] = *reinterpret_cast<uint4*>(&hi);
                            dst[1] = *reinterpret_cast<uint4*>(&lo);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                });
                if (DIAG && threadIdx.x == 0) atomicAdd(&g_ml_cycles[1], __builtin_readcyclecounter() - t_loop);
            } else {
                // ---- Moebius map of the 2-vectors (columns 2v, 2v + 1), moebius.py:374-478.  The accumulator registers cannot
                // be indexed by a run-time (row, vector), and unrolled fp64 maps for every register would be 50 KB of code: 32
                // columns of parameters at a time go through the wave's LDS region and a run-time loop maps them -- 16 lanes
                // per row (a 128-byte line of x / y per row and instruction), one vector per lane; the log-det partials of a
                // row are summed over its 16 lanes and kept in LDS.
                const int D = a.n_features;
                const int rr = lane >> 4, vv = lane & 15;
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                typedef f32x2 __attribute__((may_alias)) f32x2_alias;
                static_for<0, NREP / 2>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = qc.value;
                    static_for<0, 2>([&](auto n2c) __attribute__((always_inline)) {
                        constexpr int n = 2 * q + n2c.value;
                        const float bv = bias_n[n];
                        static_for<0, SMREP * 4>([&](auto ic) __attribute__((always_inline)) {
                            constexpr int m = ic.value / 4, i = ic.value % 4;
                            stage[(m * 16 + rq + i) * ML_PITCH + n2c.value * 16 + cj] = acc[n][m][i] * rs[m][i] + bv;
                        });
                    });
                    __builtin_amdgcn_wave_barrier();
                    const int col = n0 + q * ML_PIECE + 2 * vv;
                    const bool col_ok = col < D;
                    auto load_x = [&](int it) __attribute__((always_inline)) {
                        const int row = wrow0 + it * 4 + rr;
                        f32x2 v = (f32x2){1.f, 0.f};                             // (a harmless point for dead lanes: no NaN)
                        if (col_ok && row < a.B) v = *(const f32x2_alias*)(a.x + (int64_t)row * a.ldx + col);
                        return v;
                    };
                    f32x2 xnext = load_x(0), xnext2 = load_x(1);
#pragma nounroll
                    for (int it = 0; it < 16; ++it) {
                        const f32x2 xcur = xnext;
                        xnext = xnext2;
                        xnext2 = load_x(it < 14 ? it + 2 : 15);                  // the inputs fly two passes ahead of the maps
                        const int row_l = it * 4 + rr, row = wrow0 + row_l;
                        const bool ok = col_ok && row < a.B;
                        const f32x2 p2 = *(const f32x2_alias*)(stage + row_l * ML_PITCH + 2 * vv);
                        double xd[MOEBIUS_MAX_DIM], wd[MOEBIUS_MAX_DIM], yd[MOEBIUS_MAX_DIM];
                        xd[0] = (double)xcur[0];
                        xd[1] = (double)xcur[1];
                        wd[0] = ok ? (double)p2[0] : 0.0;
                        wd[1] = ok ? (double)p2[1] : 0.0;
                        double ld = moebius_vector(xd, wd, 2, a.mb_max_radius, UNIT ? 1 : 0, yd);
                        if (ok) *(f32x2_alias*)(a.y + (int64_t)row * a.ldy + col) = (f32x2){(float)yd[0], (float)yd[1]};
                        ld = ok ? ld : 0.0;
#pragma unroll
                        for (int off = 8; off > 0; off >>= 1) ld += __shfl_xor(ld, off, 64);
                        if (vv == 0) ldj_lds[row_l] += ld;
                    }
                    __builtin_amdgcn_wave_barrier();
                });
                if (DIAG && threadIdx.x == 0) atomicAdd(&g_ml_cycles[2], __builtin_readcyclecounter() - t_loop);
            }
            if (DIAG && threadIdx.x == 0) atomicAdd(&g_ml_cycles[0], t_loop - t_tile);
        }
        if (!last) {
            // this wave reads its own rows of the panel back in the next layer: its stores must have reached L2
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            row_scales[lane] = 1.0f / row_scales[128 + lane];
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (wrow0 + lane < a.B) a.ldj[wrow0 + lane] = (float)ldj_lds[lane];
    if (DIAG && threadIdx.x == 0) {
        atomicAdd(&g_ml_cycles[3], __builtin_readcyclecounter() - t_kernel);
        atomicAdd(&g_ml_cycles[4], 1ull);
    }
}

}  // namespace tfep

using namespace tfep;

extern "C" {

int tfep_maf_layer_tile_n(void) { return 16 * ML_NREP; }

/* TFEP_DIAG_MAF_LAYER=1 builds of the launch: out[0..4] = cycles of wave 0 summed over the workgroups in the k-loops, the hidden
 * epilogues, the Moebius epilogues, the whole kernel; the number of workgroups.  Reads and clears the counters. */
int tfep_diag_maf_layer_cycles(unsigned long long* out) {
    TFEP_REQUIRE(out != nullptr, "diag_maf_layer_cycles: NULL");
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ml_cycles), 5 * sizeof(unsigned long long));
    if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "hipMemcpyFromSymbol: %s", hipGetErrorString(e));
    unsigned long long zero[8] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_ml_cycles), zero, sizeof(zero));
    if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "hipMemcpyToSymbol: %s", hipGetErrorString(e));
    return TFEP_OK;
}

int tfep_maf_layer_forward_split(const tfep_maf_layer_desc* d, void* stream) {
    TFEP_REQUIRE(d != nullptr, "maf_layer: NULL descriptor");
    TFEP_REQUIRE(d->B >= 0, "maf_layer: negative batch");
    if (d->B == 0) return TFEP_OK;
    TFEP_REQUIRE(d->n_linears >= 2 && d->n_linears <= ML_MAX_LINEARS, "maf_layer: 2..%d masked linears (hidden layers + the output layer)",
                 ML_MAX_LINEARS);
    TFEP_REQUIRE(d->kind == 2 && d->moebius_dim == 2, "maf_layer: only the Moebius transformer of dimension 2 (kind 2) is built");
    TFEP_REQUIRE(d->a0 && d->a0_inv_scale && d->x && d->y && d->log_det_J && d->scratch[0] && d->scratch[1], "maf_layer: NULL pointer");
    TFEP_REQUIRE(d->n_features >= 2 && d->n_features % 2 == 0, "maf_layer: n_features=%d is not a multiple of the dimension 2", d->n_features);
    MafLayerArgs a = {};
    a.B = d->B; a.n_lin = d->n_linears;
    a.a0 = (const float*)d->a0; a.lda0 = d->lda0; a.a0_inv_scale = d->a0_inv_scale;
    TFEP_REQUIRE(((uintptr_t)d->a0 & 15) == 0 && d->lda0 % 32 == 0, "maf_layer: split input rows not aligned to a k-tile");
    const int tn = 16 * ML_NREP;
    for (int l = 0; l < d->n_linears; ++l) {
        TFEP_REQUIRE(d->w[l] && d->w_scales[l] && d->bias[l] && d->k_ranges[l], "maf_layer: NULL pointer (layer %d)", l);
        TFEP_REQUIRE(((uintptr_t)d->w[l] & 15) == 0 && d->ldw[l] % 32 == 0 && d->n_rows_w[l] > 0 && d->n_out[l] > 0 && d->n_out[l] <= d->n_rows_w[l],
                     "maf_layer: bad weight panel of layer %d", l);
        // the k-ranges of layer l index its input: the conditioner input for l = 0, the scratch panel after that
        TFEP_REQUIRE(d->ldw[l] <= (l == 0 ? d->lda0 : d->ld_scratch), "maf_layer: layer %d reads %lld input columns, its input panel has %lld",
                     l, (long long)d->ldw[l], (long long)(l == 0 ? d->lda0 : d->ld_scratch));
        a.w[l] = (const float*)d->w[l]; a.ldw[l] = d->ldw[l]; a.n_rows_w[l] = d->n_rows_w[l]; a.n_out[l] = d->n_out[l];
        a.n_tiles[l] = (d->n_out[l] + tn - 1) / tn;
        a.w_scales[l] = d->w_scales[l]; a.bias[l] = d->bias[l]; a.k_ranges[l] = d->k_ranges[l];
        if (l + 1 < d->n_linears) {
            TFEP_REQUIRE(d->bias_absmax[l], "maf_layer: hidden layer %d needs bias_absmax", l);
            TFEP_REQUIRE(d->n_out[l] % 8 == 0 && d->n_out[l] <= d->ld_scratch, "maf_layer: hidden width %d does not fit the scratch panel", d->n_out[l]);
            a.bias_absmax[l] = d->bias_absmax[l];
        }
    }
    TFEP_REQUIRE(d->n_out[d->n_linears - 1] == d->n_features, "maf_layer: the output layer must hold one parameter per feature");
    TFEP_REQUIRE(((uintptr_t)d->scratch[0] & 15) == 0 && ((uintptr_t)d->scratch[1] & 15) == 0 && d->ld_scratch % 32 == 0,
                 "maf_layer: scratch panels not aligned to a k-tile");
    a.h[0] = (float*)d->scratch[0]; a.h[1] = (float*)d->scratch[1]; a.ldh = d->ld_scratch;
    a.x = d->x; a.ldx = d->ldx; a.y = d->y; a.ldy = d->ldy; a.ldj = d->log_det_J;
    a.n_features = d->n_features; a.mb_unit_sphere = d->moebius_unit_sphere; a.mb_max_radius = d->moebius_max_radius;
    using T = STile<ML_NREP>;
    constexpr int LDS = T::LDS_BYTES + SWAVES * 64 * 8 + SWAVES * 3 * 64 * 4;      // operand stages + the log-det accumulators + the row scales
    static const bool diag = env_int("TFEP_DIAG_MAF_LAYER", 0) != 0;
    void (*kernel)(MafLayerArgs) = d->moebius_unit_sphere ? (diag ? maf_layer_moebius2_kernel<true, true> : maf_layer_moebius2_kernel<true, false>)
                                                          : (diag ? maf_layer_moebius2_kernel<false, true> : maf_layer_moebius2_kernel<false, false>);
    static bool attr_set_on[4][TFEP_MAX_DEVICES] = {};
    bool& attr_set = attr_set_on[(d->moebius_unit_sphere ? 1 : 0) + (diag ? 2 : 0)][current_device_slot()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "hipFuncSetAttribute(LDS=%d): %s", LDS, hipGetErrorString(e));
        attr_set = true;
    }
    const int m_tiles = (d->B + T::BM - 1) / T::BM;
    kernel<<<dim3((unsigned)m_tiles), STHREADS, LDS, (hipStream_t)stream>>>(a);
    return check_launch("maf_layer_moebius2_kernel");
}

}  // extern "C"
