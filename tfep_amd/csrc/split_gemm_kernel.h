// The split-f16 GEMM kernel template and its launcher (see split_gemm.hip for the arithmetic and the operand format).
// A header, so that the instantiations can be compiled in more than one translation unit: split_gemm.hip holds the
// linear / ELU / affine kernels and the plain spline layouts, split_gemm_layouts.hip the other spline layouts.
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
#pragma once
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

// ELU of the split-row epilogue: v > 0 ? v : exp(v) - 1 by the hardware exponential (v_exp_f32) where the result is not small
// and by the series to v^6 where it is -- ~5e-7 relative, on values that leave as split halves carrying 2^-22 of their row's
// bound.  expm1f() is ~40 instructions per value on a wave that issues one per ~5 cycles (one wave per SIMD beside the
// accumulators): at K <= 1024 (BASELINE cfg4-ii: 16 - 32 k-tiles per output tile) the epilogue of a tile then costs what its
// k-loop costs (0.64 -> 0.5 ms per hidden layer of cfg4-ii); at cfg2's K = 15 008 it is 0.5 % of the step.
__device__ __forceinline__ float elu_split_rows(float v) {
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

__device__ __forceinline__ void keep_alive(const f32x4& v) { asm volatile("" ::"v"(v)); }

// An empty asm that re-defines an accumulator where it lives: what follows depends on a value made HERE (see the kernel).
template <bool IN_AGPR>
__device__ __forceinline__ void redefine(f32x4& v) {
    if constexpr (IN_AGPR)
        asm volatile("" : "+a"(v));
    else
        asm volatile("" : "+v"(v));
}

// TFEP_DIAG & 16: per-phase cycle totals of the fused kernel (wave 0 of every workgroup): [k-loop, epilogue, workgroups]
static __device__ unsigned long long g_split_cycles[4];   // (per translation unit) [3]: workgroup lifetimes in 100 MHz real-time ticks

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
// floats per record: 16-byte aligned, > P, and a stride that spreads the 16 features of a block over 16 banks (28 for
// the 20 .. 27 parameters of 8-bin (and the widest 5-bin) layouts, 20 below -- 16 or 24 would put them on 4 or 8)
template <int P> constexpr int spl_rec() { return P + 1 <= 20 ? 20 : 28; }
template <int P> constexpr int spl_wave_bytes() { return 16 * 16 * spl_rec<P>() * 4; }     // one 16-row block of one wave

// dynamic LDS of one kernel instantiation: the operand stages, or the epilogue records of a narrow spline tile
template <int NREP, int EPI, int P>
constexpr int split_lds_bytes() {
    constexpr int stages = STile<NREP>::LDS_BYTES;
    if constexpr (epi_is_spline(EPI)) return stages > SWAVES * spl_wave_bytes<P>() ? stages : SWAVES * spl_wave_bytes<P>();
    return stages;
}

template <int KSPL, int P, bool IDB, bool SAVE>
__device__ __forceinline__ void split_spline_epilogue(const GemmArgs& g, f32x4 (&acc)[P][SMREP],
                                                      const f32x4 (&rs)[SMREP], int nt, int n0, int wrow0, int lane,
                                                      float* rec_base) {
    constexpr int SPL_REC = spl_rec<P>();
    static_assert(P + 1 <= SPL_REC, "record too small");
    const FusedArgs& fu = g.fu;
    const SplineFlags sf = spline_flags_of_layout<KSPL, P, IDB>(fu.sf);
    const int cj = lane & 15, gq = lane >> 4;
    const int slot = nt * 16 + cj;
    const int fcol = fu.feat_index[slot];
    const bool live = fcol >= 0;
    float bias_p[P];
    // SAVE (the training forward): a feature-major tile -- its dead slots lie past the packed bias
    constexpr bool fmaj = SAVE;
    static_for<0, P>([&](auto pc) __attribute__((always_inline)) {
        bias_p[pc.value] = !g.bias ? 0.f : fmaj ? (live ? g.bias[n0 + cj * P + pc.value] : 0.f) : g.bias[n0 + pc.value * 16 + cj];
    });
    float* theta_col = (SAVE && fu.theta_out) ? fu.theta_out + (int64_t)slot * P : nullptr;
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
            if constexpr (SAVE) {
                // the parameters the backward needs: this lane's two records, straight from the registers just loaded
                static_for<0, 2>([&](auto ec) __attribute__((always_inline)) {
                    constexpr int e = ec.value;
                    const int row = wrow0 + m * 16 + gq * 4 + i0 + e;
                    if (theta_col && live && row < g.B) {
                        float* th = theta_col + (int64_t)row * fu.ld_theta;
                        static_for<0, P>([&](auto pc) __attribute__((always_inline)) { th[pc.value] = prm[e][pc.value]; });
                    }
                });
            }
            double ld[2];
            float outv[2];
            static_for<0, 2>([&](auto ec) __attribute__((always_inline)) {
                constexpr int e = ec.value;
                float w[KSPL], h[KSPL], sraw[KSPL + 1], lastp, last2;
                spline_expand<KSPL, P>(sf, [&](auto pc) __attribute__((always_inline)) { return prm[e][pc.value]; },
                                       w, h, sraw, lastp, last2);
                outv[e] = (float)rq_spline_forward_full<KSPL>(w, h, sraw, lastp, last2, sf, x0, xf, y0, yf, prm[e][P],
                                                              &ld[e]);
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

// SAVE: the spline kernels of a training forward (feature-major weight rows, parameter store) -- instantiations of their
// own, so that the inference kernels keep their register allocation
// OCC = 2 (NREP <= 8: 128 accumulator registers): two workgroups per CU.  At K <= ~2 000 an output tile is 16 - 64 k-tiles deep and
// its prologue, the exposed part of every LDS-DMA round trip and its epilogue (256 ELUs + conversions per lane on a wave that is
// alone on its SIMD) cost as much as its matrix products: 35 % matrix-pipe utilisation for BASELINE cfg4-ii's 1024 x 1024 layers
// with the 256-column tile.  With two workgroups per CU one tile's epilogue and waits run beside the other's products.
template <int NREP, int EPI, int P, int KSPL, bool SAVE = false, int OCC = 1>
__global__ void __launch_bounds__(STHREADS, OCC) split_gemm_kernel(GemmArgs g, int n_rows_w) {
    using T = STile<NREP>;
    extern __shared__ __attribute__((aligned(16))) char slds[];

    // wave index in an SGPR: LDS addresses of the DMA (M0) are then scalar arithmetic, no v_readfirstlane per instruction
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int mt, ntp;
    if (!map_block(g, mt, ntp)) return;
    int nt = g.tile_order ? ((g.tile_order[ntp >> g.kr_shift] << g.kr_shift) | (ntp & ((1 << g.kr_shift) - 1))) : ntp;
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
        kb = g.k_ranges[2 * (nt >> g.kr_shift)];
        ke = g.k_ranges[2 * (nt >> g.kr_shift) + 1];
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
        if constexpr (SAVE && epi_is_spline(EPI)) {
            {
                // Tile row q = p * 16 + j (what the accumulator layout wants) sits at packed row j * P + p.  Chunk
                // c = wave + 4 e holds q = 8 c + drow: p = c >> 1 = (wave >> 1) + 2 e, j = 8 (wave & 1) + drow, so its rows
                // are ((8 (wave & 1) + drow) P + (wave >> 1) + 2 e) -- with c * piece_w = (wave + 4 e) * 2 ldw bytes the
                // e-dependent part is 2 rows per step, the rest goes into the per-lane offset (mod 2^32 arithmetic).
                sc.piece_w = (uint32_t)(2 * g.ldw);
                sc.vw = (uint32_t)(((int64_t)((8 * (wave & 1) + drow) * P + (wave >> 1)) * g.ldw) * 4) - (uint32_t)wave * sc.piece_w +
                        (((wave & 1) ? (p0 ^ 4) : p0) * 16);
            }
        }
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
        const char* Bs = b_base + (t & 1) * T::B_BYTES;
        char* Bn = b_base + ((t + 1) & 1) * T::B_BYTES;
        f16x8 ah[SMREP], al[SMREP];
#ifdef TFEP_PROBE_OLD_TILE_HEAD
        if (!(g.diag & 8)) __syncthreads();
#endif
        // The wave's A rows are its own (its own DMA, waited for above): their fragment reads go out BEFORE the barrier and travel
        // while the wave waits for the others -- a lone wave per SIMD has nothing else to cover them with.  The barrier is a
        // bare s_barrier: everybody's B DMA of tile t has landed (each wave's vmcnt(0) above), and the stage the next DMA
        // overwrites was read by MFMAs that have been issued; __syncthreads() would add a fence that waits for the reads.
#pragma unroll
        for (int m = 0; m < SMREP; ++m) {
            ah[m] = *(const f16x8*)(a_wave + m * 16 * ROW_BYTES + off_hi);
            al[m] = *(const f16x8*)(a_wave + m * 16 * ROW_BYTES + off_lo);
        }
#ifndef TFEP_PROBE_OLD_TILE_HEAD
        if (!(g.diag & 8)) asm volatile("s_barrier" ::: "memory");
#endif
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
    if (nk > 0) {
        tile(std::false_type{}, nk - 1);
        // The MFMAs are inline asm: the compiler neither knows their result latency nor pads for it.  Leave it behind IN THE
        // BLOCK OF THE LAST MFMAS, then re-define every accumulator through an (empty) asm: each later use -- the epilogue's
        // arithmetic, the copies that merge this path with the nk == 0 one, and above all the register allocator's SPILL
        // stores, which it otherwise places right after the last MFMAs, ahead of the wait (seen: scratch_store and v_mov
        // of accumulator registers before the s_nop; the last products of a few rows then miss from a result, ~1e-7
        // relative and different from run to run) -- depends on a value that exists only after the wait.
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        static_for<0, NREP * SMREP>([&](auto ic) __attribute__((always_inline)) {
            constexpr int n = ic.value / SMREP, m = ic.value % SMREP;
            redefine<(n < N_ACC_AGPR)>(acc[n][m]);
        });
    }

    if (g.diag & 1) {   // timing-only build of the main loop: keep the accumulators alive, store nothing
        static_for<0, NREP * SMREP>([&](auto ic) __attribute__((always_inline)) {
            keep_alive(acc[ic.value / SMREP][ic.value % SMREP]);
        });
        if (epi_is_spline(EPI) && (g.diag & 16) && threadIdx.x == 0) {
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
    if constexpr (epi_is_spline(EPI)) {
        static_assert(SWAVES * spl_wave_bytes<P>() <= split_lds_bytes<NREP, EPI, P>(), "epilogue records do not fit in LDS");
        __syncthreads();                        // every wave is done with the operand stages: LDS is reused below
        split_spline_epilogue<KSPL, P, EPI == EPI_SPLINE_IDB, SAVE>(g, acc, rs, nt, n0, wrow0, lane, (float*)(slds + wave * spl_wave_bytes<P>()));
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
        if constexpr (OCC > 1) {
            // the 64 KB of a two-per-CU workgroup hold 32 columns of the tile at a time (36-float rows): a row's 4 groups of 8
            // leave as 128 contiguous bytes from 4 lanes, 16 rows per instruction
            constexpr int EPN = 36;
            static_assert(NREP % 2 == 0 && SWAVES * 64 * EPN * 4 <= T::LDS_BYTES, "epilogue stage does not fit in LDS");
            __syncthreads();
            float* stage = reinterpret_cast<float*>(slds + wave * (64 * EPN * 4));
            static_for<0, NREP / 2>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = qc.value;
                static_for<0, 2>([&](auto n2c) __attribute__((always_inline)) {
                    constexpr int n = 2 * q + n2c.value;
                    const int col = n0 + n * 16 + cj;
                    const bool in_range = col < g.N;
                    const float bv = (in_range && g.bias) ? g.bias[col] : 0.f;
                    static_for<0, SMREP * 4>([&](auto ic) __attribute__((always_inline)) {
                        constexpr int m = ic.value / 4, i = ic.value % 4;
                        const float v = in_range ? elu_split_rows(acc[n][m][i] * rs[m][i] + bv) * so[m][i] : 0.f;
                        stage[(m * 16 + rq + i) * EPN + n2c.value * 16 + cj] = v;
                    });
                });
                __builtin_amdgcn_wave_barrier();
                const int grp = lane & 3;
                const int colg = n0 + q * 32 + grp * 8;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row_l = it * 16 + (lane >> 2);
                    const int row = wrow0 + row_l;
                    const f32x4_alias lo4 = *reinterpret_cast<const f32x4_alias*>(stage + row_l * EPN + grp * 8);
                    const f32x4_alias hi4 = *reinterpret_cast<const f32x4_alias*>(stage + row_l * EPN + grp * 8 + 4);
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
                    const float v = in_range ? elu_split_rows(acc[n][m][i] * rs[m][i] + bv) * so[m][i] : 0.f;
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
        }
    } else {
        // The plain product (no column map, no addend, nothing accumulated into): the tile leaves through LDS, QW column groups at
        // a time, as 16-byte stores -- 16 lanes write 256 contiguous bytes of a row.  (The element-wise path below stores 4 bytes
        // per lane with its own 64-bit address: ~190 000 cycles for a 256 x 400 tile, 18 % of a K = 4480 tile and ~90 us of every
        // super-block GEMM of the inverse, whose workgroups run one tile each.)
        bool staged = false;
        if constexpr (EPI == EPI_LINEAR) {
            constexpr int QW = (T::LDS_BYTES >= SWAVES * 64 * 68 * 4) ? 4 : 2;          // column groups per pass
            constexpr int PITCH = QW * 16 + 4, LPR = QW * 4, RPI = 64 / LPR;           // floats per staged row, lanes per row, rows per instruction
            static_assert(SWAVES * 64 * PITCH * 4 <= T::LDS_BYTES, "epilogue stage does not fit in LDS");
            staged = !g.col_map && !g.pre_add && !g.aux && !g.accumulate && (g.ldy & 3) == 0 && (g.slab_stride & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(g.y) & 15) == 0 && !(g.diag & 32);
            if (staged) {
                __syncthreads();                        // every wave is done with the operand stages: LDS is reused below
                float* stage = reinterpret_cast<float*>(slds + wave * (64 * PITCH * 4));
                float* const ybase = g.y + (int64_t)k_slice * g.slab_stride;
                const bool with_bias = g.bias && k_slice == 0;
                const int cj = lane & 15, rq = (lane >> 4) * 4;
                static_for<0, (NREP + QW - 1) / QW>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q = qc.value;
                    constexpr int groups = (NREP - QW * q) < QW ? (NREP - QW * q) : QW;
                    static_for<0, groups>([&](auto ngc) __attribute__((always_inline)) {
                        constexpr int n = QW * q + ngc.value;
                        const int col = n0 + n * 16 + cj;
                        const float bv = (with_bias && col < g.N) ? g.bias[col] : 0.f;
                        static_for<0, SMREP * 4>([&](auto ic) __attribute__((always_inline)) {
                            constexpr int m = ic.value / 4, i = ic.value % 4;
                            stage[(m * 16 + rq + i) * PITCH + ngc.value * 16 + cj] = acc[n][m][i] * rs[m][i] + bv;
                        });
                    });
                    __builtin_amdgcn_wave_barrier();    // (one wave per stage: LDS is in order, this pins the compiler)
                    const int c4 = (lane % LPR) * 4;
                    const int colg = n0 + q * (QW * 16) + c4;
                    if (c4 < groups * 16 && colg < g.N) {
#pragma unroll
                        for (int it = 0; it < 64 / RPI; ++it) {
                            const int row_l = it * RPI + lane / LPR;
                            const int row = wrow0 + row_l;
                            if (row < g.B) {
                                const f32x4_alias v = *reinterpret_cast<const f32x4_alias*>(stage + row_l * PITCH + c4);
                                float* dst = ybase + (int64_t)row * g.ldy + colg;
                                if (colg + 4 <= g.N) {
                                    *reinterpret_cast<f32x4_alias*>(dst) = v;
                                } else {
#pragma unroll
                                    for (int j = 0; j < 3; ++j)
                                        if (colg + j < g.N) dst[j] = v[j];
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                });
            }
        }
        if (staged) {
        } else if constexpr ((EPI == EPI_LINEAR || EPI == EPI_ELU) && (NREP > 16 || OCC > 1)) {
            // a wide tile has no registers for a copy of its accumulators: one column group at a time
            static_for<0, NREP>([&](auto nc) __attribute__((always_inline)) {
                constexpr int n = nc.value;
                f32x4 out1[1][SMREP];
                static_for<0, SMREP>([&](auto mc) __attribute__((always_inline)) { out1[0][mc.value] = acc[n][mc.value] * rs[mc.value]; });
                gemm_epilogue<SMREP, 1, EPI, P, KSPL>(g, out1, nt, n0 + n * 16, wrow0, lane, k_slice);
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
    }
    if ((g.diag & 16) && threadIdx.x == 0) {
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
template <int NREP, int EPI, int P, int KSPL, bool SAVE = false, int OCC = 1>
static int launch_split(const GemmArgs& g, int n_rows_w, int n_col_tiles, hipStream_t s) {
    using T = STile<NREP>;
    auto kern = split_gemm_kernel<NREP, EPI, P, KSPL, SAVE, OCC>;
    constexpr int LDS = split_lds_bytes<NREP, EPI, P>();
    // per device: a process may drive several GPUs
    static bool attr_set_on[TFEP_MAX_DEVICES] = {};
    bool& attr_set = attr_set_on[current_device_slot()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "hipFuncSetAttribute(LDS=%d): %s", LDS, hipGetErrorString(e));
        attr_set = true;
    }
    GemmArgs ga = g;
    ga.m_tiles = (g.B + T::BM - 1) / T::BM;
    if (g.ksplit > 1 && EPI != EPI_LINEAR) return fail(TFEP_ERR_INVALID_ARGUMENT, "split gemm: k_split needs the linear epilogue");
    ga.n_tiles = n_col_tiles * (g.ksplit > 1 ? g.ksplit : 1);
    ga.map_mode = block_map_mode();
    ga.diag = env_int("TFEP_DIAG", 0);
    const long long blocks = gemm_grid_blocks(ga.map_mode, ga.m_tiles, ga.n_tiles, ga.tile_list ? ga.n_tile_list : 0);
    if (blocks > 0x7fffffffLL) return fail(TFEP_ERR_INVALID_ARGUMENT, "split gemm: grid too large");
    kern<<<dim3((unsigned)blocks), STHREADS, LDS, s>>>(ga, n_rows_w);
    return check_launch("split_gemm_kernel");
}

}  // namespace tfep
