// Pieces shared by the two masked-linear GEMM kernels (fp32 MFMA in masked_linear.hip, split-f16 MFMA in
// split_gemm.hip): launch arguments, workgroup -> tile mapping and the epilogues.
#pragma once

#include "common.h"
#include "spline.h"

#include <stdlib.h>

namespace tfep {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// EPI_SPLINE_IDB: the RQ-spline epilogue for identity boundary slopes with BOTH bounds learnable -- the one layout whose
// parameter count (3 K + 1) does not tell it apart from the plain one
enum Epilogue { EPI_LINEAR = 0, EPI_ELU = 1, EPI_AFFINE = 2, EPI_SPLINE = 3, EPI_ELU_SPLIT = 4, EPI_SPLINE_IDB = 5 };
constexpr bool epi_is_spline(int epi) { return epi == EPI_SPLINE || epi == EPI_SPLINE_IDB; }

struct FusedArgs {
    const float* x;            // transformer input  (B, ldx)
    int64_t ldx;
    float* y;                  // transformer output (B, ldy)
    int64_t ldy;
    const int32_t* feat_index; // packed feature slot -> column of x / y, -1 = padding slot
    const int32_t* feat_tr;    // packed feature slot -> index among the transformed features (x0/xf/...)
    double* ldj_partial;       // (n_col_tiles, B)
    const float *x0, *xf, *y0, *yf;
    SplineFlags sf;
    // split spline kernels only (the activation-saving forward of a training step):
    int feature_major;         // the packed weight / bias rows of a column tile are feature-major (row = slot * P + p, the
                               //   backward's packing) instead of parameter-major (row = p * 16 + slot): the weight DMA
                               //   gathers the tile's rows in the kernel's order, one packing serves forward and backward
    float* theta_out;          // optional (B, ld_theta): the transformer parameters (bias included), column slot * P + p of
    int64_t ld_theta;          //   the live feature slots -- what the backward needs, written from the epilogue's records
};

struct GemmArgs {
    const float* a;            // activations (B, lda), zero padded up to k_padded columns
    int64_t lda;
    const float* w;            // packed masked weights (n_padded, ldw)
    int64_t ldw;
    const float* bias;         // packed bias (n_padded) or NULL
    const int32_t* k_ranges;   // per column tile [begin, end) or NULL
    const int32_t* col_map;    // linear epilogues: packed column -> output column, -1 = drop; NULL = identity
    float* y;
    int64_t ldy;
    int B, N, k_padded;
    const int32_t* tile_order; // optional: launch position -> column tile (heaviest k-range first), or NULL
    int map_mode;              // 0: row-tile fastest; 1 / 2: XCD-aware 8x4 super-tiles, row / column super-tiles fastest
    int m_tiles, n_tiles;
    const float* aux;          // linear epilogue: if set, y = acc * elu'(aux) with elu'(h) = h > 0 ? 1 : h + 1
    int64_t ldaux;             //   (ELU backward from the saved activation h; same indexing as y)
    int accumulate;            // linear epilogue: y += value instead of y = value
    const float* pre_add;      // linear epilogue: added BEFORE the activation (partial pre-activations of the
    int64_t ld_pre_add;        //   two-level blocked inverse); same indexing as y
    const uint8_t* tile_live;  // optional (m_tiles x n_tiles): 0 = the whole output tile is masked, skip it
    const int32_t* tile_list;  // optional: launch position p -> (row tile, column tile) = tile_list[2p], tile_list[2p + 1]
    int n_tile_list;           //   (negative = no tile); replaces map_mode / tile_order / tile_live: the host lists the
                               //   live tiles of a block-triangular product so that every XCD gets the same number
    int diag;                  // diagnostics only (TFEP_DIAG): 1 = skip the epilogue, 4 = skip the LDS-DMA,
                               // 8 = skip the barriers (garbage results; timing only), 16 = per-phase cycle counters, 32 = element-wise linear epilogue
    int ksplit;                // linear epilogues: > 1 = split-K: launch position (column tile, k slice); slice s writes
    int64_t slab_stride;       //   its partial sums to y + s * slab_stride (bias / pre_add in slice 0 only); the
                               //   consumer adds the slabs (deterministic, no atomics)
    float* y_inv_scale;        // EPI_ELU_SPLIT (split kernel): y receives split rows; per-row 1/scale written here (B)
    const float* w_l1max;      //   max_j sum_k |w_jk| of the weights and max |bias|: bound the outputs, hence the scale
    const float* bias_absmax;
    const float* a_inv_scale;  // split-f16 operands only: per-row 1/scale of the activations (B)
    const float* w_inv_scale;  //   and the single 1/scale of the weights
    int kr_shift;              // split kernels with a narrower tile than the k-ranges / tile order were built for: column tile nt
                               //   takes k_ranges[nt >> kr_shift] and launch position p the tile (tile_order[p >> kr_shift] << kr_shift) + low bits
    FusedArgs fu;
};

__device__ inline uint32_t clamp_u32(int64_t v) { return v > 0xffffffffLL ? 0xffffffffu : (v < 0 ? 0u : (uint32_t)v); }

__device__ inline float elu_f(float v) { return v > 0.f ? v : expm1f(v); }

// Workgroup -> (row tile, position in the column-tile order); false = nothing to do.
__device__ inline bool map_block(const GemmArgs& g, int& mt, int& ntp) {
    if (g.tile_list) {
        // Workgroup ids are dealt round-robin over the 8 XCDs, each XCD works through its own share: a triangular live
        // region under the super-tile walk below leaves some XCDs with twice the tiles of others (8 row super-tiles on
        // 8 XCDs: one row of super-tiles each).  The host's list holds the live tiles only, 32 neighbours per XCD turn.
        if ((int)blockIdx.x >= g.n_tile_list) return false;
        mt = g.tile_list[2 * blockIdx.x];
        ntp = g.tile_list[2 * blockIdx.x + 1];
        return mt >= 0 && mt < g.m_tiles && ntp >= 0 && ntp < g.n_tiles;
    }
    // Workgroup -> (row tile, column tile).  Workgroups are dealt round-robin over the 8 XCDs
    // (id % 8 labels the XCD) and each XCD has its own L2, so in mode 1 the 32 workgroups that are
    // co-resident on one XCD form an 8 (row tiles) x 4 (column tiles) super-tile: 8 activation
    // panels + 4 weight panels are fetched once into that L2 instead of 32 + 1.  Pure speed: any
    // placement gives the same result.
    if (g.map_mode == 1 || g.map_mode == 2) {
        const int id = blockIdx.x;
        const int xcd = id & 7, seq = id >> 3;
        const int S = (seq >> 5) * 8 + xcd, w = seq & 31;
        const int SM = (g.m_tiles + 7) >> 3, SN = (g.n_tiles + 3) >> 2;
        // mode 1: the 8 XCDs walk 8 ROW super-tiles of one column super-tile at a time (its 4 weight panels stay in the
        // Infinity Cache, the activation panels stream from HBM once per column super-tile); mode 2: 8 COLUMN super-tiles
        // of one row super-tile (the 123 MB activation panel of a cfg2 layer stays resident, the weights stream)
        const int sm = g.map_mode == 1 ? S % SM : S / SN, sn = g.map_mode == 1 ? S / SM : S % SN;
        mt = sm * 8 + (w & 7);
        ntp = sn * 4 + (w >> 3);
        if (mt >= g.m_tiles || ntp >= g.n_tiles) return false;
    } else {
        mt = blockIdx.x % g.m_tiles;
        ntp = blockIdx.x / g.m_tiles;
    }
    return true;
}

// Epilogues on the accumulators of one wave: MREP x NREP tiles of 16 x 16, C layout of the 16x16 MFMAs
// (column = lane & 15, row = (lane >> 4) * 4 + reg).  The wave owns rows [wrow0, wrow0 + 16 * MREP).
template <int MREP, int NREP, int EPI, int P, int KSPL>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[NREP][MREP], int nt, int n0, int wrow0, int lane,
                                              int k_slice = 0) {
    const int cj = lane & 15, rq = (lane >> 4) * 4;

    if constexpr (EPI == EPI_LINEAR || EPI == EPI_ELU) {
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
            const int col = n0 + n * 16 + cj;
            if (col >= g.N) continue;
            const float bv = (g.bias && k_slice == 0) ? g.bias[col] : 0.f;
            const int ocol = g.col_map ? g.col_map[col] : col;
            if (ocol < 0) continue;
#pragma unroll
            for (int m = 0; m < MREP; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = wrow0 + m * 16 + rq + i;
                    if (row < g.B) {
                        float v = acc[n][m][i] + bv;
                        if (g.pre_add && k_slice == 0) v += g.pre_add[(int64_t)row * g.ld_pre_add + ocol];
                        if (EPI == EPI_ELU) v = elu_f(v);
                        if (g.aux) {
                            const float h = g.aux[(int64_t)row * g.ldaux + ocol];
                            v *= h > 0.f ? 1.f : h + 1.f;
                        }
                        float* dst = g.y + (int64_t)k_slice * g.slab_stride + (int64_t)row * g.ldy + ocol;
                        *dst = g.accumulate ? *dst + v : v;
                    }
                }
        }
    } else {
        // Fused transformer: packed column (ft*P + p)*16 + j of this tile is parameter p of
        // feature slot (nt*FT + ft)*16 + j; the lane owns that feature for 4*MREP samples.
        constexpr int FT = NREP / P;
        const FusedArgs& fu = g.fu;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
            const int slot = (nt * FT + ft) * 16 + cj;
            const int fcol = fu.feat_index[slot];
            const bool live = fcol >= 0;
            float bias_p[P];
#pragma unroll
            for (int p = 0; p < P; ++p) bias_p[p] = g.bias ? g.bias[n0 + (ft * P + p) * 16 + cj] : 0.f;
            float x0 = 0.f, xf = 1.f, y0 = 0.f, yf = 1.f;
            if (epi_is_spline(EPI) && live) {
                const int ftr = fu.feat_tr[slot];
                x0 = fu.x0[ftr];
                xf = fu.xf[ftr];
                y0 = fu.y0[ftr];
                yf = fu.yf[ftr];
            }
#pragma unroll
            for (int m = 0; m < MREP; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = wrow0 + m * 16 + rq + i;
                    const bool ok = live && row < g.B;
                    double ld = 0.0;
                    if (ok) {
                        const float xv = fu.x[(int64_t)row * fu.ldx + fcol];
                        float out;
                        if constexpr (EPI == EPI_AFFINE) {
                            const float shift = acc[ft * P + 0][m][i] + bias_p[0];
                            const float ls = acc[ft * P + 1][m][i] + bias_p[1];
                            out = xv * expf(ls) + shift;           // affine.py:321-323
                            ld = (double)ls;
                        } else {
                            float w[KSPL], h[KSPL], sraw[KSPL + 1], lastp, last2;
                            const SplineFlags sf = spline_flags_of_layout<KSPL, P, EPI == EPI_SPLINE_IDB>(fu.sf);
                            spline_expand<KSPL, P>(sf, [&](auto pc) __attribute__((always_inline)) {
                                return acc[ft * P + pc.value][m][i] + bias_p[pc.value];
                            }, w, h, sraw, lastp, last2);
                            out = (float)rq_spline_element<KSPL, false>(w, h, sraw, lastp, last2, sf, x0, xf, y0, yf, xv, &ld);
                        }
                        fu.y[(int64_t)row * fu.ldy + fcol] = out;
                    }
                    // sum over the 16 features held by lanes with the same (lane >> 4)
#pragma unroll
                    for (int off = 8; off > 0; off >>= 1) ld += __shfl_xor(ld, off, 64);
                    if (cj == 0 && row < g.B) {
                        double* dst = fu.ldj_partial + (int64_t)(nt * FT + ft) * g.B + row;
                        *dst = ld;
                    }
                }
        }
    }
}

inline int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
// Tuning switches (A/B experiments; defaults are the measured best).
inline int block_map_mode() { static int m = env_int("TFEP_BLOCK_MAP", 1); return m; }

// Grid size for map_block().
inline long long gemm_grid_blocks(int map_mode, int m_tiles, int n_tiles, int n_tile_list = 0) {
    if (n_tile_list > 0) return n_tile_list;
    if (map_mode == 1 || map_mode == 2) {
        const long long SM = (m_tiles + 7) / 8, SN = (n_tiles + 3) / 4;
        return ((SM * SN + 7) / 8) * 8 * 32;
    }
    return (long long)m_tiles * n_tiles;
}

constexpr int FUSED_TILE_FEATURES = 16;

// split_gemm.hip: the same GEMMs on split-f16 operands (g.a / g.w point to split rows, g.a_inv_scale / g.w_inv_scale set)
int launch_split_linear(const GemmArgs& g, int n_rows_w, int act, hipStream_t s, bool wide_tile = false, bool half_wide_tile = false);
int split_wide_tile_n();
int split_half_wide_tile_n();
int launch_split_fused(const GemmArgs& g, int n_rows_w, int kind, int n_col_tiles, hipStream_t s);
bool split_fused_saving_supported(const SplineFlags& f);
// (split_gemm_layouts.hip) the spline layouts with identity boundary slopes / learnable bounds: P != 3 K + 1
int launch_split_fused_layouts(const GemmArgs& g, int n_rows_w, int K, int P, int n_col_tiles, hipStream_t s);

}  // namespace tfep
