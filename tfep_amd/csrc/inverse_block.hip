// Fused inner loop of the blocked autoregressive inverse (reference flows/autoregressive.py:179-229).
//
// The two-level blocked inverse (tfep_amd/nn/flows/autoregressive.py) handles `inverse_block` degrees at a time:
// the contribution of every EARLIER degree to the block's rows is one wide GEMM per layer (the "z" pre-activations);
// what is left is a chain that is sequential in the degree but independent across samples -- per degree a few
// hidden units per layer, the P parameters of that degree's features, the transformer inverse and the new x.
// Launched kernel by kernel that chain is ~7 tiny dependent launches per degree (24 000 per cfg2 layer), each
// dominated by launch and pipeline latency.  Here ONE kernel runs the whole chain of a block:
//   one thread per sample row, one wave per workgroup;
//   the row's in-block hidden activations (and the block's new x values) live in LDS as [unit][lane] (conflict free),
//     the old units of the bottom partial k-tile are preloaded from HBM;
//   the masked packed weights of a step are the same for all 64 rows: the wave fetches them COOPERATIVELY (one
//     global_load_dwordx4 per lane = 8 weight rows x 32 columns per instruction, up to 16 instructions in flight) into
//     an LDS stage and reads them back as same-address (broadcast) ds_read_b128.  Wave-uniform loads straight from
//     HBM/L2 cost one ~600-cycle round trip per 4 columns on this latency-bound chain; the stage pays one per dot;
//   the pre-activations (wide-GEMM slabs) are fetched cooperatively too (8 rows x 8 values per instruction) and
//     transposed through LDS;
//   hidden units: packed fp32 FMA; the 25 spline parameters of a feature: two 32 x 32 fp32 MFMA tiles (out_dot_mfma);
//   the transformer inverse is spline.h's rq_spline_element / moebius.h's moebius_vector (same code as the stand-alone
//     kernels) or the affine map.
#include <stdlib.h>

#include "common.h"
#include "spline.h"
#include "moebius.h"

namespace tfep {

constexpr int IB_MAX_LAYERS = 4;
constexpr int IB_STEP_INTS = 4 * IB_MAX_LAYERS + 6;
constexpr int IB_MAX_P = 32;
constexpr int IB_MAX_GROUPS = 8;       // kind 3: members of a mixed transformer (all RQ splines)

struct InverseBlockArgs {
    int B, L, n_steps, kind, P;
    float* x; int64_t ldx;
    float* xpad; int64_t ldxpad;
    const float* y; int64_t ldy;
    float* h[IB_MAX_LAYERS]; int64_t ldh[IB_MAX_LAYERS];
    const float* z[IB_MAX_LAYERS]; int64_t ldz[IB_MAX_LAYERS];
    const float* zout; int64_t ldzout;
    int z_slabs[IB_MAX_LAYERS]; int64_t z_slab_stride[IB_MAX_LAYERS];   // split-K partial sums of the wide GEMMs
    int zout_slabs; int64_t zout_slab_stride;
    float* ldj;
    const float* w[IB_MAX_LAYERS]; int64_t ldw[IB_MAX_LAYERS];
    const float* wout; int64_t ldwout;
    const int32_t* steps;
    const int32_t* feat_cols;
    const int32_t* feat_sel;
    const int32_t* feat_in;       // per feature: first conditioner-input entry of the block it feeds (index into in_cols / LDS)
    const int32_t* feat_per;      // per feature: 1 = periodic (two entries: cos, sin), 0 = plain (one entry)
    const int32_t* in_cols;       // per entry: column of the (embedded) conditioner input
    float emb_lower, emb_scale;   // periodic embedding: t = (x - lower) * scale
    int c0[IB_MAX_LAYERS];        // first packed column of layer l held in the LDS cache
    int n_old[IB_MAX_LAYERS];     // units [c0, c0 + n_old) were computed by earlier blocks: preloaded from h
    int cache_len;                // LDS entries per layer
    int max_feats;                // LDS entries for the block's new x values
    int stage_gstride;            // floats per 8-row group of the LDS weight stage
    int lds_floats;               // whole dynamic LDS allocation (a multiple of 4)
    int mb_dim, mb_unit_sphere;   // kind 2: Moebius
    float mb_max_radius;
    SplineArgs sp;
    SplineArgs spg[IB_MAX_GROUPS];   // kind 3: one per member; the step record names the member of its features
};

__device__ inline float elu_ib(float v) { return v > 0.f ? v : expm1f(v); }

constexpr int IB_STAGE_ROWS = 32;      // weight rows staged at once (>= the 25 spline parameters of a feature)

__host__ __device__ inline int ib_round4(int n) { return (n + 3) & ~3; }
// The weight stage holds 4 groups of 8 rows, each group TRANSPOSED: [column j][8 rows].  One same-address (broadcast)
// ds_read_b128 then yields 4 rows of one column in adjacent registers, which is what v_pk_fma_f32 wants (row-major
// staging cost ~2 v_mov per packed FMA to pair the operands).  Columns per group: the longest dot of the launch.
__host__ __device__ inline int ib_round8(int n) { return (n + 7) & ~7; }
__host__ __device__ inline int ib_stage_cols(int cache_len, int max_feats) {
    // + 1: the four 8-row groups then start 8 banks apart (a multiple of 8 columns would put them all on bank 0; the
    // matrix-core path of the spline parameters reads one element of all 32 rows per instruction)
    return ib_round8(cache_len > max_feats ? cache_len : max_feats) + 1;
}
constexpr int IB_LDS_SLACK = 12 * 64;    // floats after the weight stage: the one-slice-ahead reads of dot_staged stay in bounds
constexpr int IB_Z_PITCH = 72;           // floats per value of the pre-activation stage [IB_STAGE_ROWS values][64 sample rows + 8]
__host__ __device__ inline size_t ib_lds_floats(int L, int cache_len, int max_feats) {
    return ((size_t)L * ib_round4(cache_len) + ib_round4(max_feats)) * 64 + (size_t)IB_STAGE_ROWS * ib_stage_cols(cache_len, max_feats) +
           IB_LDS_SLACK + (size_t)IB_STAGE_ROWS * IB_Z_PITCH;
}

typedef float ib_f4 __attribute__((ext_vector_type(4)));
typedef ib_f4 __attribute__((may_alias)) ib_f4_alias;

// Stage element (row r, column j):  st[(r >> 3) * gstride + j * 8 + (r & 7)],  gstride = 8 * ib_stage_cols.
//
// stage_rows: w[(row0 + r * row_stride) * ldw + kb + j]  for r < nrows (<= 32), j < ke - kb; columns up to the next
// multiple of 8 are zero-filled.  Lane t fetches row (t >> 3) (+8, +16, +24), columns 4 (t & 7) .. +3 (+32, +64, +96)
// with one 16-byte load: all loads of a 128-column batch are issued before the first LDS write.  kb, ldw and w must be
// multiples of 4 floats.
__device__ __forceinline__ void stage_rows(float* __restrict__ st, int gstride, const float* __restrict__ w, int64_t ldw,
                                           int row0, int row_stride, int nrows, int kb, int ke, int lane) {
    const int g = lane >> 3, c = (lane & 7) * 4;
    const int len = ke - kb, len8 = ib_round8(len);
    for (int s0 = 0; s0 < len; s0 += 128) {
        ib_f4 v[4][4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rg * 8 + g, j = s0 + u * 32 + c;
                v[rg][u] = ib_f4{0.f, 0.f, 0.f, 0.f};
                if (r < nrows && j < len)
                    v[rg][u] = *(const ib_f4_alias*)(w + (int64_t)(row0 + r * row_stride) * ldw + kb + j);
            }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rg * 8 + g, j = s0 + u * 32 + c;
                if (r < nrows && j < len8) {
                    float* d = st + rg * gstride + j * 8 + g;
                    d[0] = j < len ? v[rg][u].x : 0.f;
                    d[8] = j + 1 < len ? v[rg][u].y : 0.f;
                    d[16] = j + 2 < len ? v[rg][u].z : 0.f;
                    d[24] = j + 3 < len ? v[rg][u].w : 0.f;
                }
            }
    }
    __builtin_amdgcn_wave_barrier();      // one wave: LDS is in order, this only pins the compiler's schedule
}

// Layer 0: element (r, e) = w[(row0 + r) * ldw + in_cols[e]]  for r < nrows, e < ke (the block's input entries so far)
__device__ __forceinline__ void stage_gather(float* __restrict__ st, int gstride, const float* __restrict__ w, int64_t ldw,
                                             int row0, int nrows, const int32_t* __restrict__ in_cols, int ke, int lane) {
    const int g = lane & 7, c = lane >> 3;            // 8 rows x 8 entries per instruction; LDS writes are contiguous
    const int ke8 = ib_round8(ke);
    for (int s0 = 0; s0 < ke8; s0 += 32) {
        int col[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = s0 + u * 8 + c;
            col[u] = e < ke ? in_cols[e] : -1;
        }
        float v[4][4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rg * 8 + g;
                v[rg][u] = 0.f;
                if (r < nrows && col[u] >= 0) v[rg][u] = w[(int64_t)(row0 + r) * ldw + col[u]];
            }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = rg * 8 + g, e = s0 + u * 8 + c;
                if (r < nrows && e < ke8) st[rg * gstride + e * 8 + g] = v[rg][u];
            }
    }
    __builtin_amdgcn_wave_barrier();
}

// Pre-activations of a step (the wide GEMMs' contribution of all earlier degrees): value v < nv (<= 32) of sample row i
// of this wave is  sum_s z[s * slab_stride + (wave_row0 + i) * ldz + base + v * vstride].  One lane per sample row would
// touch 64 cache lines per load instruction, on the critical path; here the wave loads 8 rows x 8 values per instruction
// (8 lines when vstride = 1), everything in flight at once, and transposes through LDS: zs[v * IB_Z_PITCH + i].
__device__ __forceinline__ void stage_z(float* __restrict__ zs, const float* __restrict__ z, int64_t ldz, int wave_row0, int n_rows_total,
                                        int base, int vstride, int nv, int slabs, int64_t slab_stride, int lane) {
    const int vq = lane & 7, rq = lane >> 3;
    float val[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) val[q][i] = 0.f;
    // slabs outermost: the (up to) 32 loads of one slab are independent and all in flight; the sum runs in slab order.
    // No per-lane predicate on the loads (a predicated `val += load` makes the compiler wait for each load inside its
    // own exec region): lanes past nv re-read value nv - 1 and are dropped at the LDS write; the q blocks are skipped
    // wave-uniformly.
    // two slabs per pass, both fetches in flight before the first add (a slab per pass is a dependent round trip of its own:
    // the trip count is not known to the compiler); a pass past the last slab re-reads it and adds zero
    for (int sl0 = 0; sl0 < slabs; sl0 += 2) {
        float t[2][4][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int sl = min(sl0 + u, slabs - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q * 8 < nv) {
                    const int v = min(q * 8 + vq, nv - 1);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int row = min(wave_row0 + rq + 8 * i, n_rows_total - 1);      // dead rows shadow the last one
                        t[u][q][i] = z[sl * slab_stride + (int64_t)row * ldz + base + v * vstride];
                    }
                }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool on = sl0 + u < slabs;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q * 8 < nv) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) val[q][i] += on ? t[u][q][i] : 0.f;
                }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int v = q * 8 + vq;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (v < nv) zs[v * IB_Z_PITCH + rq + 8 * i] = val[q][i];              // bank = 8 vq + rq (+8i): conflict free
    }
    __builtin_amdgcn_wave_barrier();
}

// One 4-column slice of a staged dot: the activations of this lane's sample row and 8 weight rows x 4 columns.
struct IbSlice {
    float h[4];
    ib_f4 wa[4], wb[4];
    __device__ __forceinline__ void load(const float* __restrict__ st, const float* __restrict__ act, int j, int lane) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = act[(j + i) * 64 + lane];
            wa[i] = *(const ib_f4_alias*)(st + (j + i) * 8);
            wb[i] = *(const ib_f4_alias*)(st + (j + i) * 8 + 4);
        }
    }
    __device__ __forceinline__ void fma_into(float (&acc)[8]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[0] = fmaf(wa[i].x, h[i], acc[0]);
            acc[1] = fmaf(wa[i].y, h[i], acc[1]);
            acc[2] = fmaf(wa[i].z, h[i], acc[2]);
            acc[3] = fmaf(wa[i].w, h[i], acc[3]);
            acc[4] = fmaf(wb[i].x, h[i], acc[4]);
            acc[5] = fmaf(wb[i].y, h[i], acc[5]);
            acc[6] = fmaf(wb[i].z, h[i], acc[6]);
            acc[7] = fmaf(wb[i].w, h[i], acc[7]);
        }
    }
};

// acc[g] += sum_{j < len} stage(g, j) * act[j * 64 + lane],  g < 8, in the order of j.  `st` points at the row group; len is
// a multiple of 8 (the stage holds zeros in the round-up, and everything in LDS is finite: zero-initialised).  The LDS
// reads run one 4-column slice ahead of the multiplies, unconditionally (a lone wave has nobody else to hide that latency;
// conditional loads or multiplies make the compiler sink or copy the slice registers): they touch up to 4 columns past
// the dot, inside the LDS allocation (IB_LDS_SLACK).
__device__ __forceinline__ void dot_staged(float (&acc)[8], const float* __restrict__ st, const float* __restrict__ act,
                                           int len, int lane) {
    IbSlice A, B;
    A.load(st, act, 0, lane);
    for (int j = 0; j < len; j += 8) {
        B.load(st, act, j + 4, lane);
        A.fma_into(acc);
        A.load(st, act, j + 8, lane);
        B.fma_into(acc);
    }
}

typedef float ib_f16v __attribute__((ext_vector_type(16)));

// The P <= 32 parameters of one feature for the wave's 64 sample rows on the matrix cores:
//   D^T[parameter m][sample n] = z[m][n] + sum_k W[m][k] act[k][n]   as two 32 x 32 tiles (samples 0..31 / 32..63) of
// v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate): A = the staged weights (lane l: row l % 32, column
// k + l / 32), B = the activations from the [unit][64 samples] cache, C = the staged pre-activations.  A lane ends up
// with D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32] in register r of each tile: half the parameters of sample l % 32 and of
// sample 32 + l % 32.  v_permlane32_swap exchanges the halves, after which lane l holds all 32 parameters of ITS sample.
// 2 MFMAs per 2 columns instead of 32 packed FMAs + 20 LDS reads per 8: the out dot was 25 % of the kernel.
__device__ __forceinline__ void out_dot_mfma(float (&prm)[IB_MAX_P], const float* __restrict__ stg, int gstride,
                                             const float* __restrict__ zso, const float* __restrict__ act, int len, int P,
                                             int lane) {
    const int half = lane >> 5, col = lane & 31;
    ib_f16v d0, d1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = (r >> 2) * 8 + half * 4 + (r & 3);
        const bool on = m < P;                                 // rows past P: never staged (stale LDS), keep them zero
        d0[r] = on ? zso[m * IB_Z_PITCH + col] : 0.f;
        d1[r] = on ? zso[m * IB_Z_PITCH + 32 + col] : 0.f;
    }
    const float* wp = stg + (col >> 3) * gstride + half * 8 + (col & 7);       // element (row col, column k + half)
    const float* ap = act + half * 64 + col;
    const bool row_on = col < P;
    // len is a multiple of 8: four column pairs per iteration, their 12 LDS reads issued before the first MFMA; the reads
    // of the next iteration are issued before this iteration's MFMAs so that they land behind them
    float w[4], b0[4], b1[4];
    auto load = [&](int k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w[q] = row_on ? wp[(k + 2 * q) * 8] : 0.f;
            b0[q] = ap[(k + 2 * q) * 64];
            b1[q] = ap[(k + 2 * q) * 64 + 32];
        }
    };
    load(0);
    for (int k = 0; k < len; k += 8) {
        float wc[4], b0c[4], b1c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { wc[q] = w[q]; b0c[q] = b0[q]; b1c[q] = b1[q]; }
        load(k + 8);                                    // (one slice past the end on the last pass: inside the LDS slack)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[q], b0c[q], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[q], b1c[q], d1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0[r]), __float_as_uint(d1[r]), false, false);
        prm[(r >> 2) * 8 + (r & 3)] = __uint_as_float(sw[0]);
        prm[(r >> 2) * 8 + 4 + (r & 3)] = __uint_as_float(sw[1]);
    }
}

// Step record: per layer l  [row0, n, kb, ke]: units [row0, row0 + n) of layer l are computed from the inputs
//   l == 0: the first `ke` conditioner-input entries of the block (in_cols order);  l >= 1: packed columns [kb, ke) of layer l - 1
// then [out_row0, n_d, out_kb, out_ke, feat_off, member].
// KIND: 0 affine, 1 spline, 2 Moebius, 3 splines of several layouts (a mixed transformer: `member` of the step record
// selects the flags, the parameter count and the domain arrays) -- one instantiation each: the spline code needs most
// of the register file.
template <int KIND>
__global__ void __launch_bounds__(64) inverse_block_kernel(InverseBlockArgs a) {
    extern __shared__ float cache[];              // [L][cache_len][64] hidden activations, then [max_feats][64] x values
    const int lane = threadIdx.x;
    const int row = blockIdx.x * 64 + lane;
    const bool live = row < a.B;
    const int64_t r = live ? row : 0;             // dead lanes shadow row 0 and store nothing
    float* xc = cache + (size_t)a.L * a.cache_len * 64;            // a.cache_len, a.max_feats: multiples of 4
    float* stg = xc + (size_t)a.max_feats * 64;                    // weight stage [4 row groups][column][8 rows]
    const int gstride = a.stage_gstride;
    float* zs = stg + (size_t)IB_STAGE_ROWS * (gstride >> 3) + IB_LDS_SLACK;   // pre-activation stage
    const int wave_row0 = blockIdx.x * 64;

    // dots run over column counts rounded up to 8 (against staged zeros) and read one slice ahead: everything in LDS
    // must be a finite number
    for (int j = lane * 4; j < a.lds_floats; j += 256) *(ib_f4_alias*)(cache + j) = ib_f4{0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < a.L; ++l) {
        const float* hr = a.h[l] + r * a.ldh[l] + a.c0[l];
        float* cl = cache + (size_t)l * a.cache_len * 64;
        for (int j = 0; j < a.n_old[l]; ++j) cl[j * 64 + lane] = hr[j];
    }

    double ldj_acc = 0.0;
    for (int s = 0; s < a.n_steps; ++s) {
        const int32_t* st = a.steps + s * IB_STEP_INTS;
        // ---- hidden units of this degree, layer by layer
        for (int l = 0; l < a.L; ++l) {
            const int row0 = st[4 * l], n = st[4 * l + 1], kb = st[4 * l + 2], ke = st[4 * l + 3];
            float* cl = cache + (size_t)l * a.cache_len * 64;
            float* hr = a.h[l] + r * a.ldh[l];
            const float* act = l == 0 ? xc : cache + ((size_t)(l - 1) * a.cache_len + (kb - a.c0[l - 1])) * 64;
            const int len = ib_round8(l == 0 ? ke : ke - kb);
            for (int ub = row0; ub < row0 + n; ub += IB_STAGE_ROWS) {
                const int nb = min(IB_STAGE_ROWS, row0 + n - ub);
                if (l == 0) stage_gather(stg, gstride, a.w[0], a.ldw[0], ub, nb, a.in_cols, ke, lane);
                else stage_rows(stg, gstride, a.w[l], a.ldw[l], ub, 1, nb, kb, ke, lane);
                stage_z(zs, a.z[l], a.ldz[l], wave_row0, a.B, ub, 1, nb, a.z_slabs[l], a.z_slab_stride[l], lane);
                for (int u0 = ub; u0 < ub + nb; u0 += 8) {
                    const int nu = min(8, ub + nb - u0);
                    float acc[8];
#pragma unroll
                    for (int g = 0; g < 8; ++g)                                 // earlier blocks + bias
                        acc[g] = g < nu ? zs[(u0 - ub + g) * IB_Z_PITCH + lane] : 0.f;
                    dot_staged(acc, stg + ((u0 - ub) >> 3) * gstride, act, len, lane);
#pragma unroll
                    for (int g = 0; g < 8; ++g)
                        if (g < nu) {
                            const float hv = elu_ib(acc[g]);
                            cl[(u0 + g - a.c0[l]) * 64 + lane] = hv;
                            if (live) hr[u0 + g] = hv;
                        }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- parameters and transformer inverse of this degree's features
        const int out_row0 = st[4 * IB_MAX_LAYERS], n_d = st[4 * IB_MAX_LAYERS + 1];
        const int okb = st[4 * IB_MAX_LAYERS + 2], oke = st[4 * IB_MAX_LAYERS + 3], foff = st[4 * IB_MAX_LAYERS + 4];
        const float* cp = cache + ((size_t)(a.L - 1) * a.cache_len + (okb - a.c0[a.L - 1])) * 64;
        const int olen = ib_round8(oke - okb);
        // the new feature becomes conditioner input: itself, or (cos, sin) under a periodic embedding
        // (same arithmetic as periodic_embedding_kernel, mafembed.py:112-145)
        auto emit = [&](int fi, float xv) {
            const int col = a.feat_cols[fi], e0 = a.feat_in[fi], icol = a.in_cols[e0];
            float in0 = xv, in1 = 0.f;
            const bool per = a.feat_per[fi] != 0;
            if (per) sincosf((xv - a.emb_lower) * a.emb_scale, &in1, &in0);
            xc[e0 * 64 + lane] = in0;
            if (per) xc[(e0 + 1) * 64 + lane] = in1;
            if (live) {
                a.x[r * a.ldx + col] = xv;
                a.xpad[r * a.ldxpad + icol] = in0;
                if (per) a.xpad[r * a.ldxpad + icol + 1] = in1;
            }
        };
        if constexpr (KIND == 2) {
            // Moebius (moebius.py:142-147, :374-478): one parameter per feature, `dim` consecutive features of the
            // degree form a vector; the dim parameter rows are adjacent, so one staged dot computes the vector's w.
            const int dim = a.mb_dim;
            for (int f = 0; f < n_d; f += dim) {
                stage_rows(stg, gstride, a.wout, a.ldwout, out_row0 + f, 1, dim, okb, oke, lane);
                stage_z(zs, a.zout, a.ldzout, wave_row0, a.B, out_row0 + f, 1, dim, a.zout_slabs, a.zout_slab_stride, lane);
                float acc[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) acc[g] = g < dim ? zs[g * IB_Z_PITCH + lane] : 0.f;
                dot_staged(acc, stg, cp, olen, lane);
                double yv[MOEBIUS_MAX_DIM], wv[MOEBIUS_MAX_DIM], xv[MOEBIUS_MAX_DIM];
#pragma unroll
                for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
                    if (i < dim) {
                        yv[i] = (double)a.y[r * a.ldy + a.feat_sel[foff + f + i]];
                        wv[i] = (double)(-acc[i]);
                    }
                ldj_acc += moebius_vector(yv, wv, dim, a.mb_max_radius, a.mb_unit_sphere, xv);
#pragma unroll
                for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
                    if (i < dim) emit(foff + f + i, (float)xv[i]);
            }
        } else {
        constexpr int MAXP = KIND == 0 ? 8 : IB_MAX_P;
        const SplineArgs& spa = KIND == 3 ? a.spg[st[4 * IB_MAX_LAYERS + 5]] : a.sp;      // (wave uniform)
        const int nP = KIND == 3 ? spa.P : a.P;
        for (int f = 0; f < n_d; ++f) {
            float prm[MAXP];
            stage_rows(stg, gstride, a.wout, a.ldwout, out_row0 + f, n_d, nP, okb, oke, lane);   // the feature's P rows at once
            stage_z(zs, a.zout, a.ldzout, wave_row0, a.B, out_row0 + f, n_d, nP, a.zout_slabs, a.zout_slab_stride, lane);
            if constexpr (KIND == 1 || KIND == 3) {
                out_dot_mfma(prm, stg, gstride, zs, cp, olen, nP, lane);
            } else {
#pragma unroll
                for (int p0 = 0; p0 < MAXP; p0 += 8) {
                    float acc[8];
#pragma unroll
                    for (int g = 0; g < 8; ++g) acc[g] = 0.f;
                    if (p0 < nP) {                                          // wave-uniform
                        const int np = min(8, nP - p0);
#pragma unroll
                        for (int g = 0; g < 8; ++g)
                            if (g < np) acc[g] = zs[(p0 + g) * IB_Z_PITCH + lane];
                        dot_staged(acc, stg + (p0 >> 3) * gstride, cp, olen, lane);
                    }
#pragma unroll
                    for (int g = 0; g < 8; ++g) prm[p0 + g] = acc[g];
                }
            }
            const int sel = a.feat_sel[foff + f];
            const float yv = a.y[r * a.ldy + sel];
            float xv;
            if constexpr (KIND == 0) {                                  // affine.py:361-363
                xv = (yv - prm[0]) * expf(-prm[1]);
                ldj_acc -= (double)prm[1];
            } else {
                const SplineFlags& fl = spa.f;
                const int K = fl.K;
                if (KIND == 3 && K == 0) {                              // a plain shift member (affine.py:366-456): log-det 0
                    xv = yv - prm[0];
                } else {
                float w[8], hh[8], sraw[9];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    w[k] = k < K ? prm[k] : 0.f;
                    hh[k] = k < K ? prm[K + k] : 0.f;
                }
#pragma unroll
                for (int j = 0; j <= 8; ++j) {
                    sraw[j] = 0.f;
                    if (j <= K) {
                        const int pi = spline_slope_param(j, K, fl.circular, fl.identity);
                        if (pi >= 0) sraw[j] = prm[pi];
                    }
                }
                float last = 0.f, last2 = 0.f;
                if (fl.circular || fl.learn_lower || fl.learn_upper) last = prm[nP - 1];
                if (fl.learn_lower && fl.learn_upper) last2 = prm[nP - 2];
                double ld;
                xv = (float)rq_spline_element<8, true>(w, hh, sraw, last, last2, fl, spa.x0[sel], spa.xf[sel],
                                                       spa.y0[sel], spa.yf[sel], yv, &ld);
                ldj_acc -= ld;
                }
            }
            // (emit, written out: the lambda call here costs the spline kernel ~50% -- different SGPR spill placement)
            const int col = a.feat_cols[foff + f];
            const int e0 = a.feat_in[foff + f], icol = a.in_cols[e0];
            float in0 = xv, in1 = 0.f;
            const bool per = a.feat_per[foff + f] != 0;
            if (per) sincosf((xv - a.emb_lower) * a.emb_scale, &in1, &in0);
            xc[e0 * 64 + lane] = in0;
            if (per) xc[(e0 + 1) * 64 + lane] = in1;
            if (live) {
                a.x[r * a.ldx + col] = xv;
                a.xpad[r * a.ldxpad + icol] = in0;
                if (per) a.xpad[r * a.ldxpad + icol + 1] = in1;
            }
        }
        }
    }
    if (live) a.ldj[row] = (float)((double)a.ldj[row] + ldj_acc);
}


// =====================================================================================================================
// Sixteen sample rows per wave ("q4": four lanes per row in the per-row phases).
//
// The one-row-per-lane kernel above is bound by the instruction issue rate of its single wave per workgroup, and at the
// batch sizes the inverse usually runs at (B / 64 workgroups <= 256 CUs) three SIMDs of every CU it touches -- and at
// B = 8192 half the CUs -- sit idle.  Here the same chain runs on 4x as many waves of 16 rows each:
//   every dot product runs on the matrix cores as 16 x 16 tiles of v_mfma_f32_16x16x4_f32 (exact fp32), D^T[unit or
//     parameter][row] with A = the staged weights, B = the [unit][16 rows] activation cache, C = the staged
//     pre-activations: hidden units are finished (ELU, stores) straight from the accumulator layout (hidden_mfma16),
//     the <= 32 parameters of a feature / Moebius vector are handed to the rows' lanes through LDS (out_dot_mfma16);
//   the pre-activation stage loads 16 rows instead of 64 per wave; the weight stage is the same cooperative fetch;
//   the transformer inverse of a row is evaluated by its four lanes (lane = 4 row + part) redundantly -- identical
//     instructions and results; lane part 0 stores.
// Per wave the dot products and the pre-activation loads shrink, weight staging and the transformer inverse do not: the
// total instruction count over all waves is higher than the one-row-per-lane kernel's, so this layout is for batches
// that leave SIMDs idle (host: rows_per_wave = 16 up to 16 384 rows).  Sums associate differently from the kernel above
// (MFMA accumulation order), i.e. the last bits of an inverse depend on which layout the batch size selects.
// =====================================================================================================================
constexpr int Q4_ROWS = 16;
// floats after a weight stage of the 16-row layouts.  Their dots prefetch one slice ahead of the last one they use: at most 3
// columns (x 8 rows) past a row group -- values that are never used, they only have to be addresses inside the allocation.
// (The 64-row kernel's IB_LDS_SLACK of 768 floats was carried over at first: 10 KB per pair, and the difference between one
// and two resident pairs per CU for cfg4-i.)
constexpr int Q4_LDS_SLACK = 128;
constexpr int Q4_Z_PITCH = 24;       // floats per value of the pre-activation stage: 8 values x 8 rows per store instruction, conflict free
constexpr int IB_FTAB_WORDS = 28 + IB_STEP_INTS;   // super-block kernel: per feature 9 table words (padded to 12) + y of the pair's 16 rows +
                                                   // a step record (a block has at most as many steps as features)
constexpr int IB_REC_WORDS = 40;                   // ... and the block's own record (IB_BLK_INTS, padded)
constexpr int Q4_P_PITCH = 20;       // floats per parameter of the MFMA hand-over [IB_MAX_P parameters][16 rows + 4]

// columns per 8-row group of the weight stage: + 4 puts consecutive groups 32 banks apart -- a 16 x 4 operand fragment (two
// groups x four columns x eight rows) then touches every bank once
__host__ __device__ inline int ib_stage_cols_q4(int cache_len, int max_feats) {
    return ib_round8(cache_len > max_feats ? cache_len : max_feats) + 4;
}
__host__ __device__ inline size_t ib_lds_floats_q4(int L, int cache_len, int max_feats) {
    return ((size_t)L * ib_round4(cache_len) + ib_round4(max_feats)) * Q4_ROWS + (size_t)IB_STAGE_ROWS * ib_stage_cols_q4(cache_len, max_feats) +
           Q4_LDS_SLACK + (size_t)IB_STAGE_ROWS * Q4_Z_PITCH + (size_t)IB_MAX_P * Q4_P_PITCH;
}

// weight stages of the paired layouts, in floats: the fp32 image [row group][column][8 rows], or (super-block kernel, split
// dots) the split image [k-group][row][32 bytes] + 16 bytes per k-group -- whichever is larger
__host__ __device__ inline size_t ib_q4_hstage_floats(int cols, int cache_len) {
    const size_t f = 8 * (size_t)cols, sd = ((size_t)(ib_round8(cache_len) >> 3) * (8 * 32 + 16) + 3) / 4;
    return f > sd ? f : sd;
}
__host__ __device__ inline size_t ib_q4_ostage_floats(int cols, int cache_len) {
    const size_t f = (size_t)IB_STAGE_ROWS * cols, sd = ((size_t)(ib_round8(cache_len) >> 3) * (IB_STAGE_ROWS * 32 + 16) + 3) / 4;
    return f > sd ? f : sd;
}
constexpr int IB_SCALE_FLOATS = 2 * (IB_MAX_LAYERS + 1) * Q4_ROWS;     // super-block kernel: per-row scales of the split dots

// the paired (loader + consumer) launch: a second weight stage and a second pre-activation stage
__host__ __device__ inline size_t ib_lds_floats_q4_paired(int L, int cache_len, int max_feats) {
    const size_t cols = ib_stage_cols_q4(cache_len, max_feats);
    const size_t h = ib_q4_hstage_floats((int)cols, cache_len) + Q4_LDS_SLACK + 8 * (size_t)Q4_Z_PITCH;                  // hidden-layer stage (8 rows)
    const size_t o = ib_q4_ostage_floats((int)cols, cache_len) + Q4_LDS_SLACK + (size_t)IB_STAGE_ROWS * Q4_Z_PITCH;      // output-rows stage
    // (+ the per-feature table of the super-block kernel: IB_FTAB_WORDS words per feature of a block, and its scale table)
    return ((size_t)L * ib_round8(cache_len) + ib_round4(max_feats)) * Q4_ROWS + (size_t)IB_MAX_P * Q4_P_PITCH + 2 * h + 2 * o +
           (size_t)IB_FTAB_WORDS * ib_round4(max_feats) + IB_SCALE_FLOATS + IB_REC_WORDS;
}

// zs[v * Q4_Z_PITCH + i] = sum_s z[s * slab_stride + (wave_row0 + i) * ldz + base + v * vstride],  v < nv <= 32, i < 16
__device__ __forceinline__ void stage_z16(float* __restrict__ zs, const float* __restrict__ z, int64_t ldz, int wave_row0, int n_rows_total,
                                          int base, int vstride, int nv, int slabs, int64_t slab_stride, int lane) {
    const int vq = lane & 7, rq = lane >> 3;
    float val[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) { val[q][0] = 0.f; val[q][1] = 0.f; }
    // four slabs per pass with every load of the pass in flight before the first add (a slab per pass made each slab a
    // dependent round trip on the chain: the trip count is not known to the compiler); no predicates on the loads -- a
    // pass past the last slab re-reads it and adds zero.  The sum still runs in slab order.
    for (int sl0 = 0; sl0 < slabs; sl0 += 4) {
        float t[4][4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sl = min(sl0 + u, slabs - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q * 8 < nv) {
                    const int v = min(q * 8 + vq, nv - 1);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = min(wave_row0 + rq + 8 * i, n_rows_total - 1);      // dead rows shadow the last one
                        t[u][q][i] = z[sl * slab_stride + (int64_t)row * ldz + base + v * vstride];
                    }
                }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool on = sl0 + u < slabs;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q * 8 < nv) {
                    val[q][0] += on ? t[u][q][0] : 0.f;
                    val[q][1] += on ? t[u][q][1] : 0.f;
                }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int v = q * 8 + vq;
        if (v < nv) {
            zs[v * Q4_Z_PITCH + rq] = val[q][0];                     // bank = 24 vq + rq (+8): conflict free
            zs[v * Q4_Z_PITCH + rq + 8] = val[q][1];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// nu <= 16 hidden units (stage rows ug .. ug + nu) for the wave's 16 sample rows on the matrix cores:
//   D[unit][row] = z[unit][row] + sum_k W[unit][k] act[k][row]   as one 16 x 16 tile of v_mfma_f32_16x16x4_f32 per 4 columns
// (A: lane l = stage row ug + l % 16, column k + l / 16;  B: activation k + l / 16 of sample row l % 16;  D: register r of
// lane l = unit 4 (l / 16) + r, sample row l % 16), then ELU and the stores straight from the accumulator layout.  Against
// packed FMAs on quarter dots: 2 LDS dwords per lane and 4 columns instead of 36 bytes, a third of the instructions --
// with four of these waves per CU the LDS pipe, not the issue rate, was the limit of the FMA form.
__device__ __forceinline__ void hidden_mfma16(const float* __restrict__ stg, int gstride, int ug, const float* __restrict__ zs,
                                              const float* __restrict__ act, int len, int nu, float* __restrict__ cl_u0,
                                              float* __restrict__ h_u0, bool live16, int lane) {
    const int col = lane & 15, kq = lane >> 4;
    ib_f4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = 4 * kq + r < nu ? zs[(ug + 4 * kq + r) * Q4_Z_PITCH + col] : 0.f;
    const int srow = ug + col;
    const float* wp = stg + (srow >> 3) * gstride + kq * 8 + (srow & 7);         // element (stage row ug + col, column k + kq)
    const float* ap = act + kq * Q4_ROWS + col;
    const bool on = col < nu;                                                    // rows past nu: not staged (stale LDS)
    float w[2], b[2];
    auto load = [&](int k) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            w[q] = on ? wp[(k + 4 * q) * 8] : 0.f;
            b[q] = ap[(k + 4 * q) * Q4_ROWS];
        }
    };
    load(0);
    for (int k = 0; k < len; k += 8) {                                   // len is a multiple of 8
        float wc[2], bc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { wc[q] = w[q]; bc[q] = b[q]; }
        load(k + 8);                                    // (one slice past the end on the last pass: inside the LDS slack)
#pragma unroll
        for (int q = 0; q < 2; ++q) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[q], bc[q], d, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int u = 4 * kq + r;
        if (u < nu) {
            const float hv = elu_ib(d[r]);
            cl_u0[u * Q4_ROWS + col] = hv;
            if (live16) h_u0[u] = hv;
        }
    }
}

// prm[m] (m < P <= 32) of this lane's row:  z[m][row] + sum_k W[m][k] act[k][row]  as two 16 x 16 tiles of
// v_mfma_f32_16x16x4_f32 (A: lane l = row l % 16 of the tile, column k + l / 16;  B: activation k + l / 16 of sample row
// l % 16;  D: register r of lane l = parameter 4 (l / 16) + r, sample row l % 16), then through LDS to the rows' lanes.
__device__ __forceinline__ void out_dot_mfma16(float (&prm)[IB_MAX_P], const float* __restrict__ stg, int gstride,
                                               const float* __restrict__ zso, const float* __restrict__ act, float* __restrict__ pb,
                                               int len, int P, int lane) {
    const int col = lane & 15, kq = lane >> 4;
    ib_f4 d0, d1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 4 * kq + r;
        d0[r] = m < P ? zso[m * Q4_Z_PITCH + col] : 0.f;                  // rows past P: never staged (stale LDS), keep them zero
        d1[r] = 16 + m < P ? zso[(16 + m) * Q4_Z_PITCH + col] : 0.f;
    }
    const float* w0 = stg + (col >> 3) * gstride + kq * 8 + (col & 7);           // element (row col, column k + kq)
    const float* w1 = stg + ((16 + col) >> 3) * gstride + kq * 8 + (col & 7);    // element (row 16 + col, column k + kq)
    const float* ap = act + kq * Q4_ROWS + col;
    const bool on0 = col < P, on1 = 16 + col < P;
    const bool two = P > 16;                                            // wave-uniform
    float a0[2], a1[2], b[2];
    auto load = [&](int k) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a0[q] = on0 ? w0[(k + 4 * q) * 8] : 0.f;
            a1[q] = on1 ? w1[(k + 4 * q) * 8] : 0.f;
            b[q] = ap[(k + 4 * q) * Q4_ROWS];
        }
    };
    load(0);
    for (int k = 0; k < len; k += 8) {                                   // len is a multiple of 8: two k-groups of 4 per pass
        float a0c[2], a1c[2], bc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { a0c[q] = a0[q]; a1c[q] = a1[q]; bc[q] = b[q]; }
        load(k + 8);                                    // (one slice past the end on the last pass: inside the LDS slack)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0c[q], bc[q], d0, 0, 0, 0);
            if (two) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1c[q], bc[q], d1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pb[(4 * kq + r) * Q4_P_PITCH + col] = d0[r];                     // bank = 16 kq + 20 r + col: conflict free
        pb[(16 + 4 * kq + r) * Q4_P_PITCH + col] = d1[r];
    }
    __builtin_amdgcn_wave_barrier();
    const int s = lane >> 2;
#pragma unroll
    for (int m = 0; m < IB_MAX_P; ++m) prm[m] = m < P ? pb[m * Q4_P_PITCH + s] : 0.f;
    __builtin_amdgcn_wave_barrier();                                     // (the next feature overwrites pb)
}

// PAIR: the workgroup is TWO waves on the same 16 rows -- a CONSUMER (the chain: dots, transformer inverse, stores) and a
// LOADER that stages the weights and pre-activations of the NEXT stage into the other half of a double-buffered stage while
// the consumer works on the current one.  Nothing the loader fetches depends on the chain (packed weights, the block GEMMs'
// slabs), and a lone wave issues a vector instruction only every ~8 cycles (tools/probe/valu_rate_probe.hip): the staging
// was 48 % of the chain's instructions (probe build without it: cfg2 layer, B = 8192, 106.8 -> 81.3 ms).  One workgroup
// barrier per stage: the loader fills buffer t & 1 then meets the barrier, the consumer meets it then reads buffer t & 1;
// the loader's next fill (t + 1, the other buffer) overlaps the consumer's stage t, and its fill t + 2 comes after barrier
// t + 1, which the consumer reaches only when it is done with buffer t & 1.  The stages of a degree are unequal -- small
// hidden-layer stages, then per feature a big one (its P output rows; the dot + transformer inverse) -- so with ONE kind of
// buffer the big fill met a small dot and the big dot a small fill (cfg2 layer, B = 8192: 107.3 -> 95.7 ms only).  The output
// rows therefore have their own double buffer, filled a whole FEATURE ahead: right after the barrier that hands feature j to
// the consumer the loader fetches feature j + 1's rows (the next step's if need be), beside the consumer's dot and spline;
// the hidden-layer stages keep a small double buffer (8 rows per chunk).  LDS per pair 75 KB at cfg2 (two pairs per CU).
// Measured: B = 8192 107.3 -> 89.4 ms, 4096 92.2 -> 72.0, 2048 88.5 -> 68.1; bit-identical to the single-wave kernel.
template <int KIND, bool PAIR>
__global__ void __launch_bounds__(512) inverse_block_q4_kernel(InverseBlockArgs a) {
    extern __shared__ float cache_all[];          // per wave (pair): [L][cache_len][16] hidden activations, then [max_feats][16] x values
    // not PAIR: blockDim.x / 64 INDEPENDENT waves per workgroup (no workgroup barrier): wave w of the launch owns rows
    // [16 w, 16 w + 16) and its own LDS region.  More than one wave per workgroup only packs the launch onto fewer CUs.
    const int lane = threadIdx.x & 63;
    const int wave_in_wg = threadIdx.x >> 6;
    const bool consumer = !PAIR || wave_in_wg == 0, loader = !PAIR || wave_in_wg == 1;
    float* const cache = cache_all + (PAIR ? (size_t)0 : (size_t)wave_in_wg * a.lds_floats);
    const int s = lane >> 2, part = lane & 3;
    const int wave_row0 = (PAIR ? (int)blockIdx.x : (int)(blockIdx.x * (blockDim.x >> 6) + wave_in_wg)) * Q4_ROWS;
    if (wave_row0 >= a.B) return;                  // (both waves of a pair alike)
    const int row = wave_row0 + s;
    const bool live = row < a.B;
    const bool writer = live && part == 0;
    const int64_t r = live ? row : 0;             // dead lanes shadow row 0 and store nothing
    const bool live16 = wave_row0 + (lane & 15) < a.B;
    const int64_t r16 = live16 ? wave_row0 + (lane & 15) : 0;
    float* xc = cache + (size_t)a.L * a.cache_len * Q4_ROWS;
    const int gstride = a.stage_gstride;
    // not PAIR: one weight stage (IB_STAGE_ROWS rows) + one pre-activation stage.  PAIR: two SMALL stages for the hidden
    // layers (8 rows: a degree has a handful of units per layer; more go through in chunks of 8), alternating, and two
    // FULL stages for the output rows of a feature, alternating per feature -- the loader fills the output stage of the
    // NEXT feature while the consumer works on the current one (the big fill beside the big dot + transformer inverse).
    constexpr int HROWS = PAIR ? 8 : IB_STAGE_ROWS;
    const size_t h_floats = (size_t)HROWS * (gstride >> 3) + Q4_LDS_SLACK + (size_t)HROWS * Q4_Z_PITCH;
    const size_t o_floats = (size_t)IB_STAGE_ROWS * (gstride >> 3) + Q4_LDS_SLACK + (size_t)IB_STAGE_ROWS * Q4_Z_PITCH;
    float* const stg0 = xc + (size_t)a.max_feats * Q4_ROWS;
    float* const zs0 = stg0 + (size_t)HROWS * (gstride >> 3) + Q4_LDS_SLACK;
    float* pb = zs0 + (size_t)HROWS * Q4_Z_PITCH;
    float* const stg1 = pb + (size_t)IB_MAX_P * Q4_P_PITCH;                      // PAIR only from here on
    float* const zs1 = stg1 + (size_t)HROWS * (gstride >> 3) + Q4_LDS_SLACK;
    float* const ostg0 = stg0 + (PAIR ? 2 * h_floats + (size_t)IB_MAX_P * Q4_P_PITCH : 0);
    float* const ozs0 = ostg0 + (size_t)IB_STAGE_ROWS * (gstride >> 3) + Q4_LDS_SLACK;
    float* const ostg1 = ostg0 + o_floats;
    float* const ozs1 = ozs0 + o_floats;
    const int lds_total = a.lds_floats;                 // (per wave, or per pair: ib_lds_floats_q4 / _paired)
    float* stg = stg0;
    float* zs = zs0;
    int parity = 0;
    auto stage_done = [&]() __attribute__((always_inline)) {       // PAIR: the hand-over of one stage (see the kernel's header)
        // (__syncthreads(), vmcnt drain included: a barrier that waits for LDS only -- s_waitcnt lgkmcnt(0); s_barrier --
        // measured SLOWER, 95.7 -> 98.2 ms per cfg2 layer at B = 8192)
        if (PAIR) __syncthreads();
    };
    auto next_stage = [&]() __attribute__((always_inline)) {
        if (PAIR) {
            parity ^= 1;
            stg = parity ? stg1 : stg0;
            zs = parity ? zs1 : zs0;
        }
    };

    if (consumer) {
        for (int j = lane * 4; j < lds_total; j += 256) *(ib_f4_alias*)(cache + j) = ib_f4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_wave_barrier();
        for (int l = 0; l < a.L; ++l) {
            const float* hr = a.h[l] + r * a.ldh[l] + a.c0[l];
            float* cl = cache + (size_t)l * a.cache_len * Q4_ROWS;
            for (int j = part; j < a.n_old[l]; j += 4) cl[j * Q4_ROWS + s] = hr[j];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (PAIR) __syncthreads();                      // (the loader's first fill must not meet the zero fill)

    // PAIR: the output stages form one sequence over the features of all steps; stage j lives in output buffer j & 1.
    int oj = 0;
    auto fill_out = [&](int step, int f, int buf) __attribute__((always_inline)) {
        const int32_t* sr = a.steps + step * IB_STEP_INTS;
        const int o_row0 = sr[4 * IB_MAX_LAYERS], o_nd = sr[4 * IB_MAX_LAYERS + 1], o_kb = sr[4 * IB_MAX_LAYERS + 2],
                  o_ke = sr[4 * IB_MAX_LAYERS + 3];
        float* os = buf ? ostg1 : ostg0;
        float* oz = buf ? ozs1 : ozs0;
        if constexpr (KIND == 2) {
            stage_rows(os, gstride, a.wout, a.ldwout, o_row0 + f, 1, a.mb_dim, o_kb, o_ke, lane);
            stage_z16(oz, a.zout, a.ldzout, wave_row0, a.B, o_row0 + f, 1, a.mb_dim, a.zout_slabs, a.zout_slab_stride, lane);
        } else {
            const int o_P = KIND == 3 ? a.spg[sr[4 * IB_MAX_LAYERS + 5]].P : a.P;
            stage_rows(os, gstride, a.wout, a.ldwout, o_row0 + f, o_nd, o_P, o_kb, o_ke, lane);
            stage_z16(oz, a.zout, a.ldzout, wave_row0, a.B, o_row0 + f, o_nd, o_P, a.zout_slabs, a.zout_slab_stride, lane);
        }
    };
    // the output stage after feature f of step `step`: the next feature of the step, or the first one of a later step
    auto fill_next_out = [&](int step, int f) __attribute__((always_inline)) {
        const int df = KIND == 2 ? a.mb_dim : 1;
        int ns = step, nf = f + df;
        while (ns < a.n_steps && nf >= a.steps[ns * IB_STEP_INTS + 4 * IB_MAX_LAYERS + 1]) { ++ns; nf = 0; }
        if (ns < a.n_steps) fill_out(ns, nf, (oj + 1) & 1);
    };
    if (PAIR && loader) {                               // the first output stage: before anything is handed over
        int ns = 0;
        while (ns < a.n_steps && a.steps[ns * IB_STEP_INTS + 4 * IB_MAX_LAYERS + 1] <= 0) ++ns;
        if (ns < a.n_steps) fill_out(ns, 0, 0);
    }

    double ldj_acc = 0.0;
    for (int st_i = 0; st_i < a.n_steps; ++st_i) {
        const int32_t* st = a.steps + st_i * IB_STEP_INTS;
        // ---- hidden units of this degree, layer by layer
        for (int l = 0; l < a.L; ++l) {
            const int row0 = st[4 * l], n = st[4 * l + 1], kb = st[4 * l + 2], ke = st[4 * l + 3];
            float* cl = cache + (size_t)l * a.cache_len * Q4_ROWS;
            float* h16 = a.h[l] + r16 * a.ldh[l];                    // the matrix-core phases: lane = sample row lane % 16
            const float* act = l == 0 ? xc : cache + ((size_t)(l - 1) * a.cache_len + (kb - a.c0[l - 1])) * Q4_ROWS;
            const int len = ib_round8(l == 0 ? ke : ke - kb);
            for (int ub = row0; ub < row0 + n; ub += HROWS) {
                const int nb = min(HROWS, row0 + n - ub);
#ifndef TFEP_PROBE_NO_STAGE          // (timing probe, wrong results: what a loader wave could take off the chain)
                if (loader) {
                    if (l == 0) stage_gather(stg, gstride, a.w[0], a.ldw[0], ub, nb, a.in_cols, ke, lane);
                    else stage_rows(stg, gstride, a.w[l], a.ldw[l], ub, 1, nb, kb, ke, lane);
                    stage_z16(zs, a.z[l], a.ldz[l], wave_row0, a.B, ub, 1, nb, a.z_slabs[l], a.z_slab_stride[l], lane);
                }
#endif
                stage_done();
                if (consumer) {
                    for (int u0 = ub; u0 < ub + nb; u0 += 16)
                        hidden_mfma16(stg, gstride, u0 - ub, zs, act, len, min(16, ub + nb - u0), cl + (size_t)(u0 - a.c0[l]) * Q4_ROWS,
                                      h16 + u0, live16, lane);
                    __builtin_amdgcn_wave_barrier();
                }
                next_stage();
            }
        }
        // ---- parameters and transformer inverse of this degree's features
        const int out_row0 = st[4 * IB_MAX_LAYERS], n_d = st[4 * IB_MAX_LAYERS + 1];
        const int okb = st[4 * IB_MAX_LAYERS + 2], oke = st[4 * IB_MAX_LAYERS + 3], foff = st[4 * IB_MAX_LAYERS + 4];
        const float* cp = cache + ((size_t)(a.L - 1) * a.cache_len + (okb - a.c0[a.L - 1])) * Q4_ROWS;
        const int olen = ib_round8(oke - okb);
        auto emit = [&](int fi, float xv) {
            const int col = a.feat_cols[fi], e0 = a.feat_in[fi], icol = a.in_cols[e0];
            float in0 = xv, in1 = 0.f;
            const bool per = a.feat_per[fi] != 0;
            if (per) sincosf((xv - a.emb_lower) * a.emb_scale, &in1, &in0);
            xc[e0 * Q4_ROWS + s] = in0;                                          // (the same value from the four lanes of the row)
            if (per) xc[(e0 + 1) * Q4_ROWS + s] = in1;
            if (writer) {
                a.x[r * a.ldx + col] = xv;
                a.xpad[r * a.ldxpad + icol] = in0;
                if (per) a.xpad[r * a.ldxpad + icol + 1] = in1;
            }
        };
        if constexpr (KIND == 2) {
            const int dim = a.mb_dim;
            for (int f = 0; f < n_d; f += dim) {
                if (!PAIR) {
                    stage_rows(stg, gstride, a.wout, a.ldwout, out_row0 + f, 1, dim, okb, oke, lane);
                    stage_z16(zs, a.zout, a.ldzout, wave_row0, a.B, out_row0 + f, 1, dim, a.zout_slabs, a.zout_slab_stride, lane);
                }
                stage_done();                                           // PAIR: output stage oj was filled a feature ago
                if (PAIR && loader) fill_next_out(st_i, f);
                if (consumer) {
                    float acc[IB_MAX_P];
                    out_dot_mfma16(acc, PAIR ? ((oj & 1) ? ostg1 : ostg0) : stg, gstride, PAIR ? ((oj & 1) ? ozs1 : ozs0) : zs, cp, pb,
                                   olen, dim, lane);
                    double yv[MOEBIUS_MAX_DIM], wv[MOEBIUS_MAX_DIM], xv[MOEBIUS_MAX_DIM];
#pragma unroll
                    for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
                        if (i < dim) {
                            yv[i] = (double)a.y[r * a.ldy + a.feat_sel[foff + f + i]];
                            wv[i] = (double)(-acc[i]);
                        }
                    ldj_acc += moebius_vector(yv, wv, dim, a.mb_max_radius, a.mb_unit_sphere, xv);
#pragma unroll
                    for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
                        if (i < dim) emit(foff + f + i, (float)xv[i]);
                    __builtin_amdgcn_wave_barrier();
                }
                ++oj;
            }
        } else {
        const SplineArgs& spa = KIND == 3 ? a.spg[st[4 * IB_MAX_LAYERS + 5]] : a.sp;      // (wave uniform)
        const int nP = KIND == 3 ? spa.P : a.P;
        for (int f = 0; f < n_d; ++f) {
            float prm[IB_MAX_P];
#ifndef TFEP_PROBE_NO_STAGE
            if (!PAIR) {
                stage_rows(stg, gstride, a.wout, a.ldwout, out_row0 + f, n_d, nP, okb, oke, lane);   // the feature's P rows at once
                stage_z16(zs, a.zout, a.ldzout, wave_row0, a.B, out_row0 + f, n_d, nP, a.zout_slabs, a.zout_slab_stride, lane);
            }
#endif
            stage_done();                                               // PAIR: output stage oj was filled a feature ago
            if (!consumer) {                     // (PAIR: the loader fetches the NEXT feature's rows beside this one's dot)
                fill_next_out(st_i, f);
                ++oj;
                continue;
            }
            out_dot_mfma16(prm, PAIR ? ((oj & 1) ? ostg1 : ostg0) : stg, gstride, PAIR ? ((oj & 1) ? ozs1 : ozs0) : zs, cp, pb, olen,
                           nP, lane);
            const int sel = a.feat_sel[foff + f];
            const float yv = a.y[r * a.ldy + sel];
            float xv;
            if constexpr (KIND == 0) {                                  // affine.py:361-363
                xv = (yv - prm[0]) * expf(-prm[1]);
                ldj_acc -= (double)prm[1];
            } else {
                const SplineFlags& fl = spa.f;
                const int K = fl.K;
                if (KIND == 3 && K == 0) {                              // a plain shift member (affine.py:366-456): log-det 0
                    xv = yv - prm[0];
                } else {
                float w[8], hh[8], sraw[9];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    w[k] = k < K ? prm[k] : 0.f;
                    hh[k] = k < K ? prm[K + k] : 0.f;
                }
#pragma unroll
                for (int j = 0; j <= 8; ++j) {
                    sraw[j] = 0.f;
                    if (j <= K) {
                        const int pi = spline_slope_param(j, K, fl.circular, fl.identity);
                        if (pi >= 0) sraw[j] = prm[pi];
                    }
                }
                float last = 0.f, last2 = 0.f;
                if (fl.circular || fl.learn_lower || fl.learn_upper) last = prm[nP - 1];
                if (fl.learn_lower && fl.learn_upper) last2 = prm[nP - 2];
                double ld;
                xv = (float)rq_spline_element<8, true>(w, hh, sraw, last, last2, fl, spa.x0[sel], spa.xf[sel],
                                                       spa.y0[sel], spa.yf[sel], yv, &ld);
                ldj_acc -= ld;
                }
            }
            const int col = a.feat_cols[foff + f];
            const int e0 = a.feat_in[foff + f], icol = a.in_cols[e0];
            float in0 = xv, in1 = 0.f;
            const bool per = a.feat_per[foff + f] != 0;
            if (per) sincosf((xv - a.emb_lower) * a.emb_scale, &in1, &in0);
            xc[e0 * Q4_ROWS + s] = in0;
            if (per) xc[(e0 + 1) * Q4_ROWS + s] = in1;
            if (writer) {
                a.x[r * a.ldx + col] = xv;
                a.xpad[r * a.ldxpad + icol] = in0;
                if (per) a.xpad[r * a.ldxpad + icol + 1] = in1;
            }
            __builtin_amdgcn_wave_barrier();
            ++oj;
        }
        }
    }
    if (consumer && writer) a.ldj[row] = (float)((double)a.ldj[row] + ldj_acc);
}


// =====================================================================================================================
// SUPER-BLOCK launch (round 4): ONE launch of the paired 16-row kernel runs every block of a super-block.
//
// Launched block by block (the kernel above), a block of 16 degrees cost its chain (~173 us at cfg2, B = 8192) PLUS, in series
// on the stream, three short GEMMs over what the super-block had produced so far (their launch floor: 25 - 40 us each on 64
// workgroups) and two re-scaling launches for the split copies of the new units: ~108 us per block, 21 of the 81 ms of a cfg2
// layer's inverse.  Nothing in those products couples sample rows: a pair of waves can form them for ITS OWN 16 rows.  Here
// the pair loops over the blocks, and at the head of every block both waves compute, on the matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 products of the fp32 packs, the h / x values the pair itself stored a moment ago),
//   z_extra[l][rows of the block] = sum over the columns [kb, ke) that EARLIER blocks of this super-block produced
// for every layer -- layer 0 from the conditioner-input entries of those blocks (gathered columns of W0 and xpad) -- and
// store it to the slab the short GEMMs used to write, which the chain then adds to the super-block GEMM's slabs exactly as
// before.  Five 16-unit tiles share one fetch of the activations; operands go global -> registers as 16-byte loads in the
// MFMA layouts (lane = (unit or row) l % 16, four consecutive k at 4 (l / 16)), two 32-column steps in flight.
// Same-workgroup visibility of the stores (h, xpad, z_extra) across the block boundary: __syncthreads().
// =====================================================================================================================
constexpr int IB_BLK_INTS = 36;      // per-block record: [n_steps, steps_off, feat_off, in_off, c0[4], n_old[4], (row0, n, kb, ke) x 5 layers, n_feats, pad]
constexpr int IB_PROD_DEPTH = 3;     // 32-column steps of the split products in flight
constexpr int IB_PROD_TILES = 5;     // 16-unit tiles that share a fetch of the activations

struct InverseSuperArgs {
    InverseBlockArgs a;
    int n_blocks;
    const int32_t* blocks;
    float* z_extra[IB_MAX_LAYERS];      // slab written by the in-kernel products (same indexing as a.z[l]: by packed row)
    float* zout_extra;
    // optional (all or none, layers 1 .. L; index L = the output layer): the split-f16 pack of the layer's weights (its rows in
    // the order of w[l] / wout), its 1/scale, and the per-row 1/scale fixed for the panel the layer READS (h[l - 1]) -- the
    // products then run as 3 x v_mfma_f32_16x16x32_f16 per 32 columns instead of 8 x v_mfma_f32_16x16x4_f32
    const float* ws[IB_MAX_LAYERS + 1]; int64_t ldws[IB_MAX_LAYERS + 1];
    const float* ws_inv[IB_MAX_LAYERS + 1];
    const float* hs_inv[IB_MAX_LAYERS + 1];
    int diag_who;                       // DIAG kernel: 0 = the phase counters of the chain wave of pair 0, 1 = of its loader wave
};

// One group of up to IB_PROD_TILES 16-row tiles of W (rows row0 + 16 t .. ) against NSETS sets of 16 sample rows (the pairs of
// the workgroup: rows row_base + 16 s + j) -- one fetch of the weights serves every set:
//   dst[(row_base + 16 s + j) * ldd + row0 + u] = sum_{k in [kb, ke)} W[row0 + u][k] src[row_base + 16 s + j][k]      (GATHER: k -> cols[k])
// kb, ke multiples of 16 unless GATHER (entries past ke read as zero).
template <bool GATHER, int NSETS>
__device__ __forceinline__ void ib_product_group(const float* __restrict__ W, int64_t ldw, int row0, int n_rows, int ntiles,
                                                 const float* __restrict__ src, int64_t lds_, int kb, int ke,
                                                 const int32_t* __restrict__ cols, float* __restrict__ dst, int64_t ldd,
                                                 int row_base, int B, int lane) {
    const int u = lane & 15, kq = lane >> 4;
    bool live[NSETS];
    const float* srow[NSETS];
#pragma unroll
    for (int s = 0; s < NSETS; ++s) {
        live[s] = row_base + 16 * s + u < B;
        srow[s] = src + (int64_t)(live[s] ? row_base + 16 * s + u : 0) * lds_;
    }
    ib_f4 acc[NSETS][IB_PROD_TILES];
#pragma unroll
    for (int s = 0; s < NSETS; ++s)
#pragma unroll
        for (int t = 0; t < IB_PROD_TILES; ++t) acc[s][t] = ib_f4{0.f, 0.f, 0.f, 0.f};
    const float* wrow[IB_PROD_TILES];
    bool won[IB_PROD_TILES];
#pragma unroll
    for (int t = 0; t < IB_PROD_TILES; ++t) {
        const int r = t * 16 + u;
        won[t] = t < ntiles && r < n_rows;
        wrow[t] = W + (int64_t)(row0 + (won[t] ? r : 0)) * ldw;
    }
    auto fetch = [&](ib_f4 (&wa)[IB_PROD_TILES], ib_f4 (&xb)[NSETS], int k) __attribute__((always_inline)) {
        if constexpr (GATHER) {
            int c[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = k + 4 * kq + j < ke ? cols[k + 4 * kq + j] : -1;
#pragma unroll
            for (int s = 0; s < NSETS; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) xb[s][j] = c[j] >= 0 ? srow[s][c[j]] : 0.f;
#pragma unroll
            for (int t = 0; t < IB_PROD_TILES; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) wa[t][j] = (won[t] && c[j] >= 0) ? wrow[t][c[j]] : 0.f;
        } else {
#pragma unroll
            for (int s = 0; s < NSETS; ++s) xb[s] = *(const ib_f4_alias*)(srow[s] + k + 4 * kq);
#pragma unroll
            for (int t = 0; t < IB_PROD_TILES; ++t)
                wa[t] = won[t] ? *(const ib_f4_alias*)(wrow[t] + k + 4 * kq) : ib_f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto multiply = [&](const ib_f4 (&wa)[IB_PROD_TILES], const ib_f4 (&xb)[NSETS]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < IB_PROD_TILES; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < NSETS; ++s) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[t][j], xb[s][j], acc[s][t], 0, 0, 0);
    };
    // two 16-column steps per pass, the next pass' loads issued before this pass' products
    ib_f4 wa0[IB_PROD_TILES], wa1[IB_PROD_TILES], xb0[NSETS], xb1[NSETS];
    if (kb < ke) {
        fetch(wa0, xb0, kb);
        fetch(wa1, xb1, kb + 16);                                 // (past ke: GATHER reads zeros; else ke - kb is a multiple of 32)
    }
    for (int k = kb; k < ke; k += 32) {
        ib_f4 wc0[IB_PROD_TILES], wc1[IB_PROD_TILES], xc0[NSETS], xc1[NSETS];
#pragma unroll
        for (int t = 0; t < IB_PROD_TILES; ++t) { wc0[t] = wa0[t]; wc1[t] = wa1[t]; }
#pragma unroll
        for (int s = 0; s < NSETS; ++s) { xc0[s] = xb0[s]; xc1[s] = xb1[s]; }
        if (k + 32 < ke) {
            fetch(wa0, xb0, k + 32);
            fetch(wa1, xb1, k + 48);
        }
        multiply(wc0, xc0);
        multiply(wc1, xc1);
    }
    // D: register r of lane l = unit 4 (l / 16) + r of the tile, sample row l % 16
#pragma unroll
    for (int s = 0; s < NSETS; ++s)
        if (live[s]) {
            float* drow = dst + (int64_t)(row_base + 16 * s + u) * ldd + row0;
#pragma unroll
            for (int t = 0; t < IB_PROD_TILES; ++t)
                if (t < ntiles) {
                    const int r0 = t * 16 + 4 * kq;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r0 + r < n_rows) drow[r0 + r] = acc[s][t][r];
                }
        }
}

typedef _Float16 ib_h8 __attribute__((ext_vector_type(8)));

// The same products on split-f16 operands (csrc/split_gemm.hip: v s = hi + lo in fp16, hi hi + lo hi + hi lo on
// v_mfma_f32_16x16x32_f16, fp32 accumulate): the weights from the layer's split pack (32 contiguous bytes per lane and 8 columns:
// the bytes of the fp32 form), the activations converted in registers with the row scale the split copy of the panel uses
// (split_columns_scaled_kernel's arithmetic).  A = weights (unit l % 16, columns 8 (l / 16) .. + 7), B = activations, D as in
// the fp32 form.  kb, ke multiples of 32.
template <int NSETS>
__device__ __forceinline__ void ib_product_group_split(const float* __restrict__ Ws, int64_t ldws, float w_inv, int row0, int n_rows,
                                                       int ntiles, const float* __restrict__ src, int64_t lds_,
                                                       const float* __restrict__ src_inv, int kb, int ke, float* __restrict__ dst,
                                                       int64_t ldd, int row_base, int B, int lane) {
    const int u = lane & 15, kg = lane >> 4;
    bool live[NSETS];
    const float* srow[NSETS];
    float sc[NSETS], unsc[NSETS];
#pragma unroll
    for (int s = 0; s < NSETS; ++s) {
        live[s] = row_base + 16 * s + u < B;
        const int64_t rr = live[s] ? row_base + 16 * s + u : 0;
        srow[s] = src + rr * lds_;
        const float inv = src_inv[rr];
        sc[s] = 1.0f / inv;
        unsc[s] = inv * w_inv;
    }
    ib_f4 acc[NSETS][IB_PROD_TILES];
#pragma unroll
    for (int s = 0; s < NSETS; ++s)
#pragma unroll
        for (int t = 0; t < IB_PROD_TILES; ++t) acc[s][t] = ib_f4{0.f, 0.f, 0.f, 0.f};
    const float* wrow[IB_PROD_TILES];
    bool won[IB_PROD_TILES];
#pragma unroll
    for (int t = 0; t < IB_PROD_TILES; ++t) {
        const int r = t * 16 + u;
        won[t] = t < ntiles && r < n_rows;
        wrow[t] = Ws + (int64_t)(row0 + (won[t] ? r : 0)) * ldws;
    }
    struct Step { ib_f4 wh[IB_PROD_TILES], wl[IB_PROD_TILES], x0[NSETS], x1[NSETS]; };
    auto fetch = [&](Step& st, int k) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            st.x0[s] = *(const ib_f4_alias*)(srow[s] + k + 8 * kg);
            st.x1[s] = *(const ib_f4_alias*)(srow[s] + k + 8 * kg + 4);
        }
#pragma unroll
        for (int t = 0; t < IB_PROD_TILES; ++t) {
            st.wh[t] = won[t] ? *(const ib_f4_alias*)(wrow[t] + k + 8 * kg) : ib_f4{0.f, 0.f, 0.f, 0.f};           // 8 hi halves
            st.wl[t] = won[t] ? *(const ib_f4_alias*)(wrow[t] + k + 8 * kg + 4) : ib_f4{0.f, 0.f, 0.f, 0.f};       // 8 lo halves
        }
    };
    auto multiply = [&](const Step& st) __attribute__((always_inline)) {
        ib_h8 xh[NSETS], xl[NSETS];
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            const float v[8] = {st.x0[s][0], st.x0[s][1], st.x0[s][2], st.x0[s][3], st.x1[s][0], st.x1[s][1], st.x1[s][2], st.x1[s][3]};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = v[j] * sc[s];
                const _Float16 h = (_Float16)t;
                xh[s][j] = h;
                xl[s][j] = (_Float16)(t - (float)h);
            }
        }
#pragma unroll
        for (int t = 0; t < IB_PROD_TILES; ++t) {
            const ib_h8 wh = __builtin_bit_cast(ib_h8, st.wh[t]), wl = __builtin_bit_cast(ib_h8, st.wl[t]);
#pragma unroll
            for (int s = 0; s < NSETS; ++s) {
                acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[s], acc[s][t], 0, 0, 0);
                acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[s], acc[s][t], 0, 0, 0);
                acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[s], acc[s][t], 0, 0, 0);
            }
        }
    };
    // one 32-column step per pass, IB_PROD_DEPTH passes in flight: a pass is ~500 cycles of matrix products, a fetch from L2
    // ~2 000 -- with two in flight the phase waited for its loads (the fp32 and the split form took the same time)
    Step ring[IB_PROD_DEPTH];
#pragma unroll
    for (int d = 0; d < IB_PROD_DEPTH; ++d)
        if (kb + 32 * d < ke) fetch(ring[d], kb + 32 * d);
    for (int k = kb; k < ke; k += 32 * IB_PROD_DEPTH) {
#pragma unroll
        for (int d = 0; d < IB_PROD_DEPTH; ++d)
            if (k + 32 * d < ke) {
                const Step cur = ring[d];
                if (k + 32 * (d + IB_PROD_DEPTH) < ke) fetch(ring[d], k + 32 * (d + IB_PROD_DEPTH));
                multiply(cur);
            }
    }
#pragma unroll
    for (int s = 0; s < NSETS; ++s)
        if (live[s]) {
            float* drow = dst + (int64_t)(row_base + 16 * s + u) * ldd + row0;
#pragma unroll
            for (int t = 0; t < IB_PROD_TILES; ++t)
                if (t < ntiles) {
                    const int r0 = t * 16 + 4 * kg;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (r0 + r < n_rows) drow[r0 + r] = acc[s][t][r] * unsc[s];
                }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// SPLIT DOTS (round 4): the dots of the chain itself on split-f16 operands.  A degree's chain is three short dots and the
// transformer inverse, strictly one after the other; as exact-fp32 products (v_mfma_f32_16x16x4_f32: 32 cycles per 4 columns,
// one 4-byte LDS read per lane, operand and step) the two dots over hidden units were 40 % of it.  On split operands a dot
// takes 3 x v_mfma_f32_16x16x32_f16 (48 cycles) and four 16-byte LDS reads per 32 columns.  The layer-0 dot (the block's own
// inputs, gathered columns of the fp32 pack) stays as it was.
//   activations of layers 0 .. L - 1 in LDS as split halves, [k-group of 8][sample row 16][32 bytes: 8 hi, 8 lo], with the per-row
//     scale of the panel's split copy (the bound-based one of the block GEMMs);
//   weight stages [k-group][row][32 bytes] from the layer's split pack; a k-group block is padded by 16 bytes (the loader writes
//     consecutive k-groups of a row from consecutive lanes);
//   in both, rows 4 q .. 4 q + 3 with odd q hold (lo, hi) instead of (hi, lo): the 16 rows of a k-group then spread a 16-byte
//     read over all banks twice, the minimum.
// Arithmetic: fp32-equivalent like the block GEMMs and the forward pass (2^-22 of the row's bound), not bit-identical to the
// exact-fp32 dots of the one-block kernels.
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int sd_group_bytes(int rows) { return rows * 32 + 16; }
__device__ __forceinline__ int sd_hi_off(int row) { return row * 32 + (((row >> 2) & 1) ? 16 : 0); }
__device__ __forceinline__ int sd_lo_off(int row) { return row * 32 + (((row >> 2) & 1) ? 0 : 16); }

// stage rows r < nrows (packed rows row0 + r * row_stride of the split pack ws), k-groups of columns [kb, ke) (kb a multiple of 8)
// CH chunks (one row's k-group: 32 bytes) per lane and pass, all of a pass's loads in flight before its first LDS write: a pass is a
// round trip of the loader's slot.  Two for the hidden stages (<= 8 rows: one pass), four for the output rows of a degree (25 rows x
// 14 k-groups at cfg2: two passes instead of three).
template <int CH>
__device__ __forceinline__ void stage_rows_split_impl(char* __restrict__ st, int R, const float* __restrict__ ws, int64_t ldws, int row0,
                                                      int row_stride, int nrows, int kb, int ke, int lane) {
    const int ng = (ke - kb + 7) >> 3;
    const int gb = sd_group_bytes(R);
    const int total = nrows * ng;
    // i / ng without the ~25 instructions of an integer division (the loader wave's slot is what the chain waits for):
    // floor((i + 0.5) / ng) in fp32 is exact for i < 2^12, ng <= 2^8 -- the quotient's fraction stays 0.5 / ng away from an integer
    const float inv_ng = __builtin_amdgcn_rcpf((float)ng);        // (v_rcp_f32, 1 ulp: still exact, checked for 1 ulp either way)
    for (int i0 = 0; i0 < total; i0 += 64 * CH) {
        ib_f4 hi[CH], lo[CH];
        int rr[CH], gg[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            const int i = i0 + 64 * t + lane;
            rr[t] = i < total ? (int)(((float)i + 0.5f) * inv_ng) : -1;
            gg[t] = i - (rr[t] < 0 ? 0 : rr[t]) * ng;
            if (rr[t] >= 0) {
                const float* src = ws + (int64_t)(row0 + rr[t] * row_stride) * ldws + kb + 8 * gg[t];
                hi[t] = *(const ib_f4_alias*)src;
                lo[t] = *(const ib_f4_alias*)(src + 4);
            }
        }
#pragma unroll
        for (int t = 0; t < CH; ++t)
            if (rr[t] >= 0) {
                *(ib_f4_alias*)(st + gg[t] * gb + sd_hi_off(rr[t])) = hi[t];
                *(ib_f4_alias*)(st + gg[t] * gb + sd_lo_off(rr[t])) = lo[t];
            }
    }
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void stage_rows_split(char* __restrict__ st, int R, const float* __restrict__ ws, int64_t ldws, int row0,
                                                 int row_stride, int nrows, int kb, int ke, int lane) {
    if (nrows * ((ke - kb + 7) >> 3) > 128) stage_rows_split_impl<4>(st, R, ws, ldws, row0, row_stride, nrows, kb, ke, lane);
    else stage_rows_split_impl<2>(st, R, ws, ldws, row0, row_stride, nrows, kb, ke, lane);
}

// The same stage by LDS-DMA (global_load_lds_dwordx4), one instruction per k-group: lane 2 r + s fetches half s ^ swizzle(r) of row r's
// 32 bytes straight into the stage -- no registers, no LDS-write instructions, and nothing waits: the loader's later loads complete
// after it (in order), and it drains its vector-memory counter before the hand-over that follows.  For the output rows of a degree
// (nrows <= 32); the hidden stages (<= 8 rows) keep the register path, 4 loads per lane.
__device__ __forceinline__ void stage_rows_split_dma(char* __restrict__ st, int R, const float* __restrict__ ws, int64_t ldws, int row0,
                                                     int nrows, int kb, int ke, int lane) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int ng = (ke - kb + 7) >> 3;
    const int gb = sd_group_bytes(R);
    const int row = lane >> 1, slot = lane & 1;
    const int half = slot ^ ((row >> 2) & 1);                                  // 0: the 8 hi halves, 1: the 8 lo halves
    const float* src = ws + (int64_t)(row0 + min(row, nrows - 1)) * ldws + kb + 4 * half;
    if (row < nrows)
        for (int g = 0; g < ng; ++g)
            __builtin_amdgcn_global_load_lds((glb_ptr)(src + 8 * g), (lds_ptr)(st + g * gb), 16, 0, 0);
}

// value `hv` of unit k (position in the layer's cache) for sample row `row`, scaled, into the split activation cache
__device__ __forceinline__ void sd_store_act(char* __restrict__ act, int k, int row, float hv, float scale) {
    const float t = hv * scale;
    const _Float16 h = (_Float16)t;
    const _Float16 l = (_Float16)(t - (float)h);
    char* g = act + (size_t)(k >> 3) * (16 * 32);
    *reinterpret_cast<_Float16*>(g + sd_hi_off(row) + 2 * (k & 7)) = h;
    *reinterpret_cast<_Float16*>(g + sd_lo_off(row) + 2 * (k & 7)) = l;
}

// ELU of the nu <= 16 units of a tile held in the MFMA accumulator layout (register r of lane l = unit 4 (l / 16) + r, sample row
// l % 16), stored to the fp32 panel in global memory and to the split cache of the layer
__device__ __forceinline__ void sd_finish_hidden(const ib_f4& d, int nu, int k0, char* __restrict__ act_out, float scale_out,
                                                 float* __restrict__ h_u0, bool live16, int lane) {
    const int col = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int u = 4 * kq + r;
        if (u < nu) {
            // ELU with the hardware exponential where the result is not small and the series to v^6 where it is (~5e-7
            // relative; expm1f is ~40 instructions on a wave that issues one per ~8 cycles: 1.0 of 71.2 ms per cfg2 inverse)
            const float v_ = d[r];
            const float e_ = __expf(v_) - 1.0f;
            float s_ = fmaf(v_, 1.0f / 720.0f, 1.0f / 120.0f);
            s_ = fmaf(v_, s_, 1.0f / 24.0f); s_ = fmaf(v_, s_, 1.0f / 6.0f); s_ = fmaf(v_, s_, 0.5f); s_ = fmaf(v_, s_, 1.0f); s_ *= v_;
            const float hv = v_ > 0.f ? v_ : (v_ > -0.25f ? s_ : e_);
            sd_store_act(act_out, k0 + u, col, hv, scale_out);
            if (live16) h_u0[u] = hv;
        }
    }
}

// layer 0: the fp32 dot of hidden_mfma16 (gathered inputs), its result into the split cache
__device__ __forceinline__ void hidden_mfma16_l0_sd(const float* __restrict__ stg, int gstride, int ug, const float* __restrict__ zs,
                                                    const float* __restrict__ act, int len, int nu, int k0, char* __restrict__ act_out,
                                                    float scale_out, float* __restrict__ h_u0, bool live16, int lane) {
    const int col = lane & 15, kq = lane >> 4;
    ib_f4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = 4 * kq + r < nu ? zs[(ug + 4 * kq + r) * Q4_Z_PITCH + col] : 0.f;
    const int srow = ug + col;
    const float* wp = stg + (srow >> 3) * gstride + kq * 8 + (srow & 7);
    const float* ap = act + kq * Q4_ROWS + col;
    const bool on = col < nu;
    float w[2], b[2];
    auto load = [&](int k) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            w[q] = on ? wp[(k + 4 * q) * 8] : 0.f;
            b[q] = ap[(k + 4 * q) * Q4_ROWS];
        }
    };
    load(0);
    for (int k = 0; k < len; k += 8) {
        float wc[2], bc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { wc[q] = w[q]; bc[q] = b[q]; }
        load(k + 8);
#pragma unroll
        for (int q = 0; q < 2; ++q) d = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[q], bc[q], d, 0, 0, 0);
    }
    sd_finish_hidden(d, nu, k0, act_out, scale_out, h_u0, live16, lane);
}

// D[unit][sample row] = sum over the k-groups [0, ng) of stage rows `srow` x the split activations; 3 MFMAs per 32 columns
__device__ __forceinline__ ib_f4 sd_dot(const char* __restrict__ stg, int gb, int srow, bool row_on, const char* __restrict__ act, int ng,
                                        int lane) {
    const int col = lane & 15, kg = lane >> 4;
    const int a_hi = sd_hi_off(srow), a_lo = sd_lo_off(srow), b_hi = sd_hi_off(col), b_lo = sd_lo_off(col);
    const ib_h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    ib_f4 d = ib_f4{0.f, 0.f, 0.f, 0.f};
    ib_h8 ah, al, bh, bl;
    auto load = [&](int g0) {
        const int G = g0 + kg;
        const bool v = G < ng;
        ah = (v && row_on) ? *(const ib_h8*)(stg + G * gb + a_hi) : zero;
        al = (v && row_on) ? *(const ib_h8*)(stg + G * gb + a_lo) : zero;
        bh = v ? *(const ib_h8*)(act + G * (16 * 32) + b_hi) : zero;
        bl = v ? *(const ib_h8*)(act + G * (16 * 32) + b_lo) : zero;
    };
    load(0);
    for (int g0 = 0; g0 < ng; g0 += 4) {
        const ib_h8 cah = ah, cal = al, cbh = bh, cbl = bl;
        load(g0 + 4);                                   // (past ng: zeros, no LDS access)
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(cah, cbh, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(cal, cbh, d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(cah, cbl, d, 0, 0, 0);
    }
    return d;
}

// hidden units of a layer l >= 1: stage rows ug .. ug + nu of the split stage against the split cache of layer l - 1
__device__ __forceinline__ void hidden_mfma16_sd(const char* __restrict__ stg, int gb, int ug, const float* __restrict__ zs,
                                                 const char* __restrict__ act, int ng, int nu, float unscale, int k0,
                                                 char* __restrict__ act_out, float scale_out, float* __restrict__ h_u0, bool live16,
                                                 int lane) {
    const int col = lane & 15, kq = lane >> 4;
    ib_f4 d = sd_dot(stg, gb, ug + col, col < nu, act, ng, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = 4 * kq + r < nu ? fmaf(d[r], unscale, zs[(ug + 4 * kq + r) * Q4_Z_PITCH + col]) : 0.f;
    sd_finish_hidden(d, nu, k0, act_out, scale_out, h_u0, live16, lane);
}

// the P <= 32 parameters of a feature (out_dot_mfma16 on split operands): two tiles, then through LDS to the rows' lanes
__device__ __forceinline__ void out_dot_mfma16_sd(float (&prm)[IB_MAX_P], const char* __restrict__ stg, int gb, const float* __restrict__ zso,
                                                  const char* __restrict__ act, float* __restrict__ pb, int ng, int P, float unscale,
                                                  int lane) {
    const int col = lane & 15, kq = lane >> 4;
    ib_f4 d0 = ib_f4{0.f, 0.f, 0.f, 0.f}, d1 = ib_f4{0.f, 0.f, 0.f, 0.f};
    {
        const int kg = kq;
        const int a0_hi = sd_hi_off(col), a0_lo = sd_lo_off(col), a1_hi = sd_hi_off(16 + col), a1_lo = sd_lo_off(16 + col);
        const int b_hi = sd_hi_off(col), b_lo = sd_lo_off(col);
        const bool on0 = col < P, on1 = 16 + col < P, two = P > 16;
        const ib_h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        ib_h8 a0h, a0l, a1h, a1l, bh, bl;
        auto load = [&](int g0) {
            const int G = g0 + kg;
            const bool v = G < ng;
            a0h = (v && on0) ? *(const ib_h8*)(stg + G * gb + a0_hi) : zero;
            a0l = (v && on0) ? *(const ib_h8*)(stg + G * gb + a0_lo) : zero;
            a1h = (v && on1) ? *(const ib_h8*)(stg + G * gb + a1_hi) : zero;
            a1l = (v && on1) ? *(const ib_h8*)(stg + G * gb + a1_lo) : zero;
            bh = v ? *(const ib_h8*)(act + G * (16 * 32) + b_hi) : zero;
            bl = v ? *(const ib_h8*)(act + G * (16 * 32) + b_lo) : zero;
        };
        load(0);
        for (int g0 = 0; g0 < ng; g0 += 4) {
            const ib_h8 c0h = a0h, c0l = a0l, c1h = a1h, c1l = a1l, cbh = bh, cbl = bl;
            load(g0 + 4);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c0h, cbh, d0, 0, 0, 0);
            if (two) d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1h, cbh, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c0l, cbh, d0, 0, 0, 0);
            if (two) d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1l, cbh, d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c0h, cbl, d0, 0, 0, 0);
            if (two) d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(c1h, cbl, d1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = 4 * kq + r;
        pb[m * Q4_P_PITCH + col] = m < P ? fmaf(d0[r], unscale, zso[m * Q4_Z_PITCH + col]) : 0.f;
        pb[(16 + m) * Q4_P_PITCH + col] = 16 + m < P ? fmaf(d1[r], unscale, zso[(16 + m) * Q4_Z_PITCH + col]) : 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    const int s = lane >> 2;
#pragma unroll
    for (int m = 0; m < IB_MAX_P; ++m) prm[m] = m < P ? pb[m * Q4_P_PITCH + s] : 0.f;
    __builtin_amdgcn_wave_barrier();
}

// TFEP_DIAG_INVERSE=1 (kind 1 only): cycles of the CONSUMER wave of every pair, summed over the pairs, in
// [0] products at the head of a block, [1] cache / table initialisation, [2] waiting at a hidden-layer hand-over, [3] hidden dots,
// [4] waiting at an output hand-over, [5] parameter dot, [6] transformer inverse + stores, [7] whole kernel; [8] = pairs counted
static __device__ unsigned long long g_ib_cycles[12];

// PAIRS = 2: two pairs per workgroup (4 waves, each pair its own LDS region and its own 16 rows).  The chain of a pair is what it was;
// the hand-overs are workgroup barriers, so the two pairs move in lockstep -- they run the same records on different rows -- and
// the PRODUCTS at the head of a block are dealt over the four waves, every wave multiplying its tiles with BOTH row sets: the
// weight rows, the 60 GB per inverse that bound that phase (every pair read them all), are fetched once per workgroup.
// Wave-uniform ints kept in LDS (the records of the block being worked on): a read is a broadcast ds_read, brought back to an SGPR.
// From global memory each record field was a VECTOR load (the addresses are uniform, but the stores in between keep the compiler
// from scalar loads) with its s_waitcnt vmcnt(0) -- which also waits for every global store the wave has issued: the chain wave
// paid a store round trip at the head of every stage for three integers.
struct LdsInts {
    const int* p;
    __device__ __forceinline__ int operator[](int i) const { return __builtin_amdgcn_readfirstlane(p[i]); }
    __device__ __forceinline__ LdsInts operator+(int o) const { return LdsInts{p + o}; }
};

// SPEC (kind 1): 0 = any spline layout (bins, flags and parameter count at run time: the unpacking of a feature's parameters then
// indexes a 32-register array dynamically, every flag is a select); 1 = 8 bins, plain layout (25 parameters); 2 = 8 bins, circular
// (25 parameters): the layouts of BASELINE cfg2 / cfg4-i as compile-time constants.
template <int KIND, bool DIAG = false, int PAIRS = 1, bool SD = false, int SPEC = 0>
#ifdef TFEP_PROBE_SK_OCC2
// (timing probe: two workgroups = four pairs per CU, two waves per SIMD at 256 registers each.  Measured at B = 16 384 with 5 degrees
// per block so that four pairs' LDS fits (TFEP_INV_BLOCK=5 TFEP_INV_SUPER=25): 146.6 ms against 135.6 ms with one workgroup per CU
// in two rounds -- the chain spills ~280 registers at that budget and both chain waves of a CU land on the same two SIMDs.)
__global__ void __launch_bounds__(128 * PAIRS, 2) inverse_superblock_kernel
#else
__global__ void __launch_bounds__(128 * PAIRS) inverse_superblock_kernel
#endif
(InverseSuperArgs sa) {
    unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_mark = DIAG ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long t_kernel0 = t_mark;
    auto lap = [&](int slot) __attribute__((always_inline)) {
        if (DIAG) {
            const unsigned long long now = __builtin_readcyclecounter();
            dg[slot] += now - t_mark;
            t_mark = now;
        }
    };
    extern __shared__ float cache_all[];
    const InverseBlockArgs& a = sa.a;
#ifdef TFEP_PROBE_FIXED_L
    constexpr int LL = TFEP_PROBE_FIXED_L;          // (timing probe: the number of hidden layers as a compile-time constant)
#else
    const int LL = sa.a.L;
#endif
    const int lane = threadIdx.x & 63;
    const int wave_in_wg = threadIdx.x >> 6;
    const int pair = wave_in_wg >> 1;
    const bool consumer = (wave_in_wg & 1) == 0, loader = !consumer;
    float* const cache = cache_all + (size_t)pair * a.lds_floats;
    const int s = lane >> 2, part = lane & 3;
    const int wg_row0 = (int)blockIdx.x * PAIRS * Q4_ROWS;
    if (wg_row0 >= a.B) return;                    // (the whole workgroup alike)
    // a pair past the end of the batch (an odd number of pairs) keeps walking -- every barrier needs it -- on row 0, storing nothing
    const int wave_row0 = wg_row0 + pair * Q4_ROWS;
    const int row = wave_row0 + s;
    const bool live = row < a.B;
    const bool writer = live && part == 0;
    const int64_t r = live ? row : 0;
    const bool live16 = wave_row0 + (lane & 15) < a.B;
    const int64_t r16 = live16 ? wave_row0 + (lane & 15) : 0;
    float* xc = cache + (size_t)LL * a.cache_len * Q4_ROWS;            // (a.cache_len: a multiple of 8 here)
    const int gstride = a.stage_gstride;
    constexpr int HROWS = 8;
    const size_t hst = ib_q4_hstage_floats(gstride >> 3, a.cache_len), ost = ib_q4_ostage_floats(gstride >> 3, a.cache_len);
    const size_t h_floats = hst + Q4_LDS_SLACK + (size_t)HROWS * Q4_Z_PITCH;
    const size_t o_floats = ost + Q4_LDS_SLACK + (size_t)IB_STAGE_ROWS * Q4_Z_PITCH;
    float* const stg0 = xc + (size_t)a.max_feats * Q4_ROWS;
    float* const zs0 = stg0 + hst + Q4_LDS_SLACK;
    float* pb = zs0 + (size_t)HROWS * Q4_Z_PITCH;
    float* const stg1 = pb + (size_t)IB_MAX_P * Q4_P_PITCH;
    float* const zs1 = stg1 + hst + Q4_LDS_SLACK;
    float* const ostg0 = stg0 + 2 * h_floats + (size_t)IB_MAX_P * Q4_P_PITCH;
    float* const ozs0 = ostg0 + ost + Q4_LDS_SLACK;
    float* const ostg1 = ostg0 + o_floats;
    float* const ozs1 = ozs0 + o_floats;
    const size_t hstride = (size_t)(stg1 - stg0);       // (buffers addressed by arithmetic on the parity, not by selects: the
    (void)ostg1; (void)ozs1; (void)zs1;                 //  compiler cloned the whole chain per combination of parities)
    const int lds_total = a.lds_floats;
    // SD (split dots): the activation caches as split halves (same bytes: 64 per unit), byte views of the stages
    char* const cache_b = reinterpret_cast<char*>(cache);
    const int layer_bytes = a.cache_len * Q4_ROWS * 4;
    constexpr int gb_h = HROWS * 32 + 16, gb_o = IB_STAGE_ROWS * 32 + 16;
    // Per-feature table of the current block, filled by the consumer at the head of the block: what the chain used to fetch
    // from global memory feature by feature, each fetch a dependent round trip on the critical path (index -> y, domain)
    // -- 3.9 of 76.6 ms per cfg2 layer (probe build without them).  [feature][12]: sel, x column, first input entry, its
    // xpad column, periodic, x0, xf, y0, yf;  then [feature][16 rows]: y of the pair's rows.
    float* const ftab = ozs1 + (size_t)IB_STAGE_ROWS * Q4_Z_PITCH;
    float* const ytab = ftab + 12 * (size_t)a.max_feats;
    // SD: [l][16] the un-scaling of layer l's dot (1/scale of the rows of its input panel x 1/scale of its weights), then
    // [L + 1 + l][16] the scale of the split halves of layer l's OUTPUT (the panel layer l + 1 reads), per sample row of the pair
    float* const scl = ytab + (size_t)a.max_feats * Q4_ROWS;
    int* const lrec = reinterpret_cast<int*>(scl + IB_SCALE_FLOATS);    // the block's record and its step records
    int* const lsteps = lrec + IB_REC_WORDS;
    const SplineArgs& sp0 = KIND == 3 ? a.spg[0] : a.sp;            // (kind 3: every member indexes the same domain arrays)

    // The hand-over of a stage inside the chain.  (__syncthreads() drains the wave's vector-memory counter too, so the consumer
    // waits at every hand-over for the global stores it has just issued; a consumer-side barrier that waits for LDS only measured
    // the same -- 71.1 / 71.0 ms, same box, alternating -- as it did on the one-block kernel in round 3.)
    // Round 4, later: a bare s_barrier once the wave's LDS traffic has landed.  Nothing the chain stores to global memory (h, x,
    // xpad) is read inside the block -- the barrier at the head of the next block publishes it -- and the loader's loads are in
    // its registers before it writes LDS: neither wave needs the vmcnt(0) of __syncthreads() here.  (With the records read from
    // global memory the same change measured nothing: their loads brought the wait back.)
    auto handover = [&]() __attribute__((always_inline)) {
        if (loader) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (its LDS-DMA of the output rows has landed too)
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    double ldj_acc = 0.0;
    for (int blk = 0; blk < sa.n_blocks; ++blk) {
        const int32_t* grec = sa.blocks + (size_t)blk * IB_BLK_INTS;
        const int n_steps = min(grec[0], (int)ib_round4(a.max_feats));   // (the host checks: a block's steps fit the LDS table)
        const int32_t* gsteps = a.steps + (size_t)grec[1] * IB_STEP_INTS;
        const int feat_base = grec[2];
        const int32_t* in_cols = a.in_cols + grec[3];
        __syncthreads();            // the previous block is complete in both waves: its LDS is free, its h / x stores are visible
        lap(6);

        // ---- what earlier blocks of the super-block add to this block's rows, for the pair's own 16 sample rows
        {
            int g = 0;
            for (int l = 0; l <= LL; ++l) {
                const int row0 = grec[12 + 4 * l], n_rows = grec[13 + 4 * l], kb = grec[14 + 4 * l], ke = grec[15 + 4 * l];
                if (ke <= kb || n_rows <= 0) continue;
                const float* W = l < LL ? a.w[l] : a.wout;
                const int64_t ldw = l < LL ? a.ldw[l] : a.ldwout;
                float* dst = l < LL ? sa.z_extra[l] : sa.zout_extra;
                const int64_t ldd = l < LL ? a.ldz[l] : a.ldzout;
                for (int t0 = 0; t0 < n_rows; t0 += 16 * IB_PROD_TILES, ++g) {
                    if (g % (2 * PAIRS) != wave_in_wg) continue;
                    const int nr = min(n_rows - t0, 16 * IB_PROD_TILES);
                    if (l == 0)
                        ib_product_group<true, PAIRS>(W, ldw, row0 + t0, nr, (nr + 15) >> 4, a.xpad, a.ldxpad, kb, ke, a.in_cols, dst, ldd,
                                                      wg_row0, a.B, lane);
                    else if (sa.ws[l] != nullptr)
                        ib_product_group_split<PAIRS>(sa.ws[l], sa.ldws[l], sa.ws_inv[l][0], row0 + t0, nr, (nr + 15) >> 4, a.h[l - 1],
                                                      a.ldh[l - 1], sa.hs_inv[l], kb, ke, dst, ldd, wg_row0, a.B, lane);
                    else
                        ib_product_group<false, PAIRS>(W, ldw, row0 + t0, nr, (nr + 15) >> 4, a.h[l - 1], a.ldh[l - 1], kb, ke, nullptr, dst,
                                                       ldd, wg_row0, a.B, lane);
                }
            }
        }
        lap(0);
        float* stg = stg0;
        float* zs = zs0;
        int parity = 0;
        if (consumer) {
            // (caches and stages; the tables behind them are written whole, by the loader wave meanwhile)
            const int zero_total = (int)(ftab - cache) & ~3;
            for (int j = lane * 4; j < zero_total; j += 256) *(ib_f4_alias*)(cache + j) = ib_f4{0.f, 0.f, 0.f, 0.f};
            if (lane * 4 + zero_total < lds_total && lane < 1) for (int j = zero_total; j < (int)(ftab - cache); ++j) cache[j] = 0.f;
            __builtin_amdgcn_wave_barrier();
            if constexpr (SD) {
                if (lane < Q4_ROWS)
                    for (int l = 1; l <= LL; ++l) {
                        const float inv = sa.hs_inv[l][r16];                  // rows of panel h[l - 1], read by layer l
                        scl[l * Q4_ROWS + lane] = inv * sa.ws_inv[l][0];
                        scl[(IB_MAX_LAYERS + 1 + l - 1) * Q4_ROWS + lane] = 1.0f / inv;
                    }
                __builtin_amdgcn_wave_barrier();
            }
            for (int l = 0; l < LL; ++l) {
                const float* hr = a.h[l] + r * a.ldh[l] + grec[4 + l];
                float* cl = cache + (size_t)l * a.cache_len * Q4_ROWS;
                const int n_old = grec[8 + l];
                if constexpr (SD) {
                    const float sc = scl[(IB_MAX_LAYERS + 1 + l) * Q4_ROWS + s];
                    for (int j = part; j < n_old; j += 4) sd_store_act(cache_b + (size_t)l * layer_bytes, j, s, hr[j], sc);
                } else {
                    for (int j = part; j < n_old; j += 4) cl[j * Q4_ROWS + s] = hr[j];
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            // the loader wave is done with its share of the products: the block's tables meanwhile
            const int n_feat = min(grec[32], a.max_feats);
            for (int i = lane; i < n_feat; i += 64) {
                const int fi = feat_base + i;
                const int sel = a.feat_sel[fi], e0 = a.feat_in[fi];
                int* ft = reinterpret_cast<int*>(ftab) + 12 * i;
                ft[0] = sel; ft[1] = a.feat_cols[fi]; ft[2] = e0; ft[3] = in_cols[e0]; ft[4] = a.feat_per[fi];
                if constexpr (KIND == 1 || KIND == 3) {
                    ftab[12 * i + 5] = sp0.x0[sel]; ftab[12 * i + 6] = sp0.xf[sel];
                    ftab[12 * i + 7] = sp0.y0[sel]; ftab[12 * i + 8] = sp0.yf[sel];
                }
            }
            for (int i = part; i < n_feat; i += 4) ytab[i * Q4_ROWS + s] = a.y[r * a.ldy + a.feat_sel[feat_base + i]];
            for (int i = lane; i < IB_BLK_INTS; i += 64) lrec[i] = grec[i];
            for (int i = lane; i < n_steps * IB_STEP_INTS; i += 64) lsteps[i] = gsteps[i];
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();            // the products are stored (the loader's stages read them), the cache is initialised
        lap(1);
        const LdsInts rec{lrec}, steps{lsteps};         // from here on the records are read from LDS
        // slabs of the pre-activations: those of the super-block GEMMs, plus the one just written where this block has one
        int z_slabs[IB_MAX_LAYERS];
#pragma unroll
        for (int l = 0; l < IB_MAX_LAYERS; ++l) z_slabs[l] = l < LL ? a.z_slabs[l] + (rec[15 + 4 * l] > rec[14 + 4 * l] ? 1 : 0) : 0;
        const int zout_slabs = a.zout_slabs + (rec[15 + 4 * LL] > rec[14 + 4 * LL] ? 1 : 0);

        int oj = 0;
        auto fill_out = [&](int step, int f, int buf) __attribute__((always_inline)) {
            const LdsInts sr = steps + step * IB_STEP_INTS;
            const int o_row0 = sr[4 * IB_MAX_LAYERS], o_nd = sr[4 * IB_MAX_LAYERS + 1], o_kb = sr[4 * IB_MAX_LAYERS + 2],
                      o_ke = sr[4 * IB_MAX_LAYERS + 3];
            float* os = ostg0 + (size_t)buf * o_floats;
            float* oz = ozs0 + (size_t)buf * o_floats;
            if constexpr (KIND == 2) {
                if constexpr (SD) stage_rows_split(reinterpret_cast<char*>(os), IB_STAGE_ROWS, sa.ws[LL], sa.ldws[LL], o_row0 + f, 1, a.mb_dim, o_kb, o_ke, lane);
                else stage_rows(os, gstride, a.wout, a.ldwout, o_row0 + f, 1, a.mb_dim, o_kb, o_ke, lane);
                stage_z16(oz, a.zout, a.ldzout, wave_row0, a.B, o_row0 + f, 1, a.mb_dim, zout_slabs, a.zout_slab_stride, lane);
            } else {
                const int o_P = KIND == 3 ? a.spg[sr[4 * IB_MAX_LAYERS + 5]].P : a.P;
                if constexpr (SD) {
                    if (o_nd == 1 && o_P <= 32) stage_rows_split_dma(reinterpret_cast<char*>(os), IB_STAGE_ROWS, sa.ws[LL], sa.ldws[LL], o_row0 + f, o_P, o_kb, o_ke, lane);
                    else stage_rows_split(reinterpret_cast<char*>(os), IB_STAGE_ROWS, sa.ws[LL], sa.ldws[LL], o_row0 + f, o_nd, o_P, o_kb, o_ke, lane);
                } else stage_rows(os, gstride, a.wout, a.ldwout, o_row0 + f, o_nd, o_P, o_kb, o_ke, lane);
                stage_z16(oz, a.zout, a.ldzout, wave_row0, a.B, o_row0 + f, o_nd, o_P, zout_slabs, a.zout_slab_stride, lane);
            }
        };
        auto fill_next_out = [&](int step, int f) __attribute__((always_inline)) {
            const int df = KIND == 2 ? a.mb_dim : 1;
            int ns = step, nf = f + df;
            while (ns < n_steps && nf >= steps[ns * IB_STEP_INTS + 4 * IB_MAX_LAYERS + 1]) { ++ns; nf = 0; }
            if (ns < n_steps) fill_out(ns, nf, (oj + 1) & 1);
        };
        if (loader) {                               // the first output stage: before anything is handed over
            int ns = 0;
            while (ns < n_steps && steps[ns * IB_STEP_INTS + 4 * IB_MAX_LAYERS + 1] <= 0) ++ns;
            if (ns < n_steps) fill_out(ns, 0, 0);
        }

        for (int st_i = 0; st_i < n_steps; ++st_i) {
            const LdsInts st = steps + st_i * IB_STEP_INTS;
            // ---- hidden units of this degree, layer by layer
            for (int l = 0; l < LL; ++l) {
                const int row0 = st[4 * l], n = st[4 * l + 1], kb = st[4 * l + 2], ke = st[4 * l + 3];
                const int c0l = rec[4 + l];
                float* cl = cache + (size_t)l * a.cache_len * Q4_ROWS;
                float* h16 = a.h[l] + r16 * a.ldh[l];
                const float* act = l == 0 ? xc : cache + ((size_t)(l - 1) * a.cache_len + (kb - rec[4 + l - 1])) * Q4_ROWS;
                const int len = ib_round8(l == 0 ? ke : ke - kb);
                for (int ub = row0; ub < row0 + n; ub += HROWS) {
                    const int nb = min(HROWS, row0 + n - ub);
#ifndef TFEP_PROBE_NO_HSTAGE          // (timing probe, wrong results: the chain without the loader's hidden-layer fetches)
                    if (loader) {
                        if (l == 0) stage_gather(stg, gstride, a.w[0], a.ldw[0], ub, nb, in_cols, ke, lane);
                        else if constexpr (SD) stage_rows_split_dma(reinterpret_cast<char*>(stg), HROWS, sa.ws[l], sa.ldws[l], ub, nb, kb, ke, lane);
                        else stage_rows(stg, gstride, a.w[l], a.ldw[l], ub, 1, nb, kb, ke, lane);
                        stage_z16(zs, a.z[l], a.ldz[l], wave_row0, a.B, ub, 1, nb, z_slabs[l], a.z_slab_stride[l], lane);
                    }
#endif
                    lap(6);
                    handover();
                    lap(2);
                    if (consumer) {
                        if constexpr (SD) {
                            char* const act_out = cache_b + (size_t)l * layer_bytes;
                            const float sc_out = scl[(IB_MAX_LAYERS + 1 + l) * Q4_ROWS + (lane & 15)];
                            for (int u0 = ub; u0 < ub + nb; u0 += 16) {
                                const int nu = min(16, ub + nb - u0);
                                if (l == 0)
                                    hidden_mfma16_l0_sd(stg, gstride, u0 - ub, zs, act, len, nu, u0 - c0l, act_out, sc_out, h16 + u0, live16, lane);
                                else
                                    hidden_mfma16_sd(reinterpret_cast<const char*>(stg), gb_h, u0 - ub, zs,
                                                     cache_b + (size_t)(l - 1) * layer_bytes + (size_t)((kb - rec[4 + l - 1]) >> 3) * (Q4_ROWS * 32),
                                                     (ke - kb + 7) >> 3, nu, scl[l * Q4_ROWS + (lane & 15)], u0 - c0l, act_out, sc_out, h16 + u0,
                                                     live16, lane);
                            }
                        } else {
                        for (int u0 = ub; u0 < ub + nb; u0 += 16)
                            hidden_mfma16(stg, gstride, u0 - ub, zs, act, len, min(16, ub + nb - u0), cl + (size_t)(u0 - c0l) * Q4_ROWS,
                                          h16 + u0, live16, lane);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    lap(3);
                    parity ^= 1;
                    stg = stg0 + (size_t)parity * hstride;
                    zs = zs0 + (size_t)parity * hstride;
                }
            }
            // ---- parameters and transformer inverse of this degree's features
            const int out_row0 = st[4 * IB_MAX_LAYERS], n_d = st[4 * IB_MAX_LAYERS + 1];
            const int okb = st[4 * IB_MAX_LAYERS + 2], oke = st[4 * IB_MAX_LAYERS + 3], floc = st[4 * IB_MAX_LAYERS + 4];
            (void)out_row0;
            const int* const fti = reinterpret_cast<const int*>(ftab);
            const float* cp = cache + ((size_t)(LL - 1) * a.cache_len + (okb - rec[4 + LL - 1])) * Q4_ROWS;
            const char* cp_sd = cache_b + (size_t)(LL - 1) * layer_bytes + (size_t)((okb - rec[4 + LL - 1]) >> 3) * (Q4_ROWS * 32);
            (void)cp_sd;
            const int olen = ib_round8(oke - okb);
            auto emit = [&](int fi, float xv) {                 // fi: the feature's position in the block
                const int col = fti[12 * fi + 1], e0 = fti[12 * fi + 2], icol = fti[12 * fi + 3];
                float in0 = xv, in1 = 0.f;
                const bool per = fti[12 * fi + 4] != 0;
                if (per) sincosf((xv - a.emb_lower) * a.emb_scale, &in1, &in0);
                xc[e0 * Q4_ROWS + s] = in0;
                if (per) xc[(e0 + 1) * Q4_ROWS + s] = in1;
                if (writer) {
                    a.x[r * a.ldx + col] = xv;
                    a.xpad[r * a.ldxpad + icol] = in0;
                    if (per) a.xpad[r * a.ldxpad + icol + 1] = in1;
                }
            };
            if constexpr (KIND == 2) {
                const int dim = a.mb_dim;
                for (int f = 0; f < n_d; f += dim) {
                    handover();                                             // output stage oj was filled a feature ago
                    if (loader) fill_next_out(st_i, f);
                    if (consumer) {
                        float acc[IB_MAX_P];
                        if constexpr (SD)
                            out_dot_mfma16_sd(acc, reinterpret_cast<const char*>(ostg0 + (size_t)(oj & 1) * o_floats), gb_o, ozs0 + (size_t)(oj & 1) * o_floats, cp_sd, pb,
                                              (oke - okb + 7) >> 3, dim, scl[LL * Q4_ROWS + (lane & 15)], lane);
                        else
                        out_dot_mfma16(acc, ostg0 + (size_t)(oj & 1) * o_floats, gstride, ozs0 + (size_t)(oj & 1) * o_floats, cp, pb, olen, dim, lane);
                        double yv[MOEBIUS_MAX_DIM], wv[MOEBIUS_MAX_DIM], xv[MOEBIUS_MAX_DIM];
#pragma unroll
                        for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
                            if (i < dim) {
                                yv[i] = (double)ytab[(floc + f + i) * Q4_ROWS + s];
                                wv[i] = (double)(-acc[i]);
                            }
                        ldj_acc += moebius_vector(yv, wv, dim, a.mb_max_radius, a.mb_unit_sphere, xv);
#pragma unroll
                        for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
                            if (i < dim) emit(floc + f + i, (float)xv[i]);
                        __builtin_amdgcn_wave_barrier();
                    }
                    ++oj;
                }
            } else {
            const SplineArgs& spa = KIND == 3 ? a.spg[st[4 * IB_MAX_LAYERS + 5]] : a.sp;      // (wave uniform)
            const int nP = SPEC > 0 ? 25 : (KIND == 3 ? spa.P : a.P);
            for (int f = 0; f < n_d; ++f) {
                float prm[IB_MAX_P];
                lap(3);
                handover();                                                 // output stage oj was filled a feature ago
                lap(4);
                if (!consumer) {                     // the loader fetches the NEXT feature's rows beside this one's dot
                    fill_next_out(st_i, f);
                    ++oj;
                    continue;
                }
                if constexpr (SD)
                    out_dot_mfma16_sd(prm, reinterpret_cast<const char*>(ostg0 + (size_t)(oj & 1) * o_floats), gb_o, ozs0 + (size_t)(oj & 1) * o_floats, cp_sd, pb,
                                      (oke - okb + 7) >> 3, nP, scl[LL * Q4_ROWS + (lane & 15)], lane);
                else
                out_dot_mfma16(prm, ostg0 + (size_t)(oj & 1) * o_floats, gstride, ozs0 + (size_t)(oj & 1) * o_floats, cp, pb, olen, nP, lane);
                lap(5);
                const float yv = ytab[(floc + f) * Q4_ROWS + s];
                const float* const ftf = ftab + 12 * (floc + f);
                float xv;
                if constexpr (KIND == 0) {                                  // affine.py:361-363
                    xv = (yv - prm[0]) * expf(-prm[1]);
                    ldj_acc -= (double)prm[1];
                } else {
                    SplineFlags fl = spa.f;
                    if constexpr (SPEC > 0) {
                        fl.K = 8; fl.circular = SPEC == 2; fl.identity = false; fl.learn_lower = false; fl.learn_upper = false;
                    }
                    const int K = fl.K;
                    if (KIND == 3 && K == 0) {                              // a plain shift member (affine.py:366-456): log-det 0
                        xv = yv - prm[0];
                    } else {
                    float w[8], hh[8], sraw[9];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        w[k] = k < K ? prm[k] : 0.f;
                        hh[k] = k < K ? prm[K + k] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j <= 8; ++j) {
                        sraw[j] = 0.f;
                        if (j <= K) {
                            const int pi = spline_slope_param(j, K, fl.circular, fl.identity);
                            if (pi >= 0) sraw[j] = prm[pi];
                        }
                    }
                    float last = 0.f, last2 = 0.f;
                    if (fl.circular || fl.learn_lower || fl.learn_upper) last = prm[nP - 1];
                    if (fl.learn_lower && fl.learn_upper) last2 = prm[nP - 2];
                    double ld;
#ifdef TFEP_PROBE_NO_SPLINE          // (timing probe, wrong results: the chain without the fp64 spline inverse)
                    xv = yv * 0.5f + 1e-3f * (w[0] + hh[1] + sraw[2] + last + last2);
                    ld = 0.0;
#else
                    xv = (float)rq_spline_inverse_selects<8, (SPEC > 0), (SPEC > 0)>(w, hh, sraw, last, last2, fl, ftf[5], ftf[6], ftf[7], ftf[8], yv, &ld, part);
#endif
                    ldj_acc -= ld;
                    }
                }
                const int col = fti[12 * (floc + f) + 1];
                const int e0 = fti[12 * (floc + f) + 2], icol = fti[12 * (floc + f) + 3];
                float in0 = xv, in1 = 0.f;
                const bool per = fti[12 * (floc + f) + 4] != 0;
                if (per) sincosf((xv - a.emb_lower) * a.emb_scale, &in1, &in0);
                xc[e0 * Q4_ROWS + s] = in0;
                if (per) xc[(e0 + 1) * Q4_ROWS + s] = in1;
                if (writer) {
                    a.x[r * a.ldx + col] = xv;
                    a.xpad[r * a.ldxpad + icol] = in0;
                    if (per) a.xpad[r * a.ldxpad + icol + 1] = in1;
                }
                __builtin_amdgcn_wave_barrier();
                ++oj;
            }
            }
        }
    }
    if (consumer && writer) a.ldj[row] = (float)((double)a.ldj[row] + ldj_acc);
    if (DIAG && threadIdx.x == (sa.diag_who ? 64 : 0)) {
        lap(6);
#pragma unroll
        for (int i = 0; i < 7; ++i) atomicAdd(&g_ib_cycles[i], dg[i]);
        atomicAdd(&g_ib_cycles[7], __builtin_readcyclecounter() - t_kernel0);
        atomicAdd(&g_ib_cycles[8], 1ull);
    }
}

}  // namespace tfep

using namespace tfep;

extern "C" {

int tfep_inverse_block_step_ints(void) { return IB_STEP_INTS; }
int tfep_inverse_block_record_ints(void) { return IB_BLK_INTS; }

/* TFEP_DIAG_INVERSE=1: per-phase cycle totals of the consumer waves of the super-block kernel (kind 1); out[9]; reads and clears. */
int tfep_diag_inverse_cycles(unsigned long long* out) {
    TFEP_REQUIRE(out != nullptr, "diag_inverse_cycles: NULL");
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ib_cycles), 9 * sizeof(unsigned long long));
    if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "hipMemcpyFromSymbol: %s", hipGetErrorString(e));
    unsigned long long zero[12] = {};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_ib_cycles), zero, sizeof(zero));
    if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "hipMemcpyToSymbol: %s", hipGetErrorString(e));
    return TFEP_OK;
}

int64_t tfep_inverse_block_lds_bytes(int n_layers, int cache_len, int max_feats) {
    if (n_layers < 1 || cache_len < 0 || max_feats < 0) return -1;
    return (int64_t)(ib_lds_floats(n_layers, cache_len, max_feats) * sizeof(float));
}

int64_t tfep_inverse_block_lds_bytes_paired(int n_layers, int cache_len, int max_feats) {
    if (n_layers < 1 || cache_len < 0 || max_feats < 0) return -1;
    return (int64_t)(ib_lds_floats_q4_paired(n_layers, cache_len, max_feats) * sizeof(float));
}

int64_t tfep_inverse_block_lds_bytes_rows(int n_layers, int cache_len, int max_feats, int rows_per_wave) {
    if (n_layers < 1 || cache_len < 0 || max_feats < 0 || (rows_per_wave != 16 && rows_per_wave != 64)) return -1;
    return (int64_t)((rows_per_wave == 16 ? ib_lds_floats_q4(n_layers, cache_len, max_feats)
                                          : ib_lds_floats(n_layers, cache_len, max_feats)) * sizeof(float));
}

int tfep_inverse_block(const tfep_inverse_block_desc* d, void* stream) {
    TFEP_REQUIRE(d != nullptr, "inverse_block: NULL descriptor");
    TFEP_REQUIRE(d->B >= 0 && d->n_steps >= 0 && d->n_blocks >= 0, "inverse_block: negative size");
    if (d->B == 0 || (d->n_steps == 0 && d->n_blocks == 0)) return TFEP_OK;
    TFEP_REQUIRE(d->n_layers >= 1 && d->n_layers <= IB_MAX_LAYERS, "inverse_block: 1..%d hidden layers", IB_MAX_LAYERS);
    TFEP_REQUIRE(d->kind >= 0 && d->kind <= 3,
                 "inverse_block: kind must be 0 (affine), 1 (spline), 2 (Moebius) or 3 (splines of several layouts)");
    TFEP_REQUIRE(d->x && d->xpad && d->y && d->zout && d->wout && d->log_det_J && d->steps && d->feat_cols && d->feat_sel,
                 "inverse_block: NULL pointer");
    InverseBlockArgs a = {};
    a.B = d->B; a.L = d->n_layers; a.n_steps = d->n_steps; a.kind = d->kind;
    a.x = d->x; a.ldx = d->ldx; a.xpad = d->xpad; a.ldxpad = d->ldxpad; a.y = d->y; a.ldy = d->ldy;
    for (int l = 0; l < d->n_layers; ++l) {
        TFEP_REQUIRE(d->h[l] && d->z[l] && d->w[l], "inverse_block: NULL layer pointer");
        a.h[l] = d->h[l]; a.ldh[l] = d->ldh[l]; a.z[l] = d->z[l]; a.ldz[l] = d->ldz[l]; a.w[l] = d->w[l]; a.ldw[l] = d->ldw[l];
        a.c0[l] = d->cache_col0[l]; a.n_old[l] = d->cache_n_old[l];
        a.z_slabs[l] = d->z_slabs[l] > 1 ? d->z_slabs[l] : 1; a.z_slab_stride[l] = d->z_slab_stride[l];
        TFEP_REQUIRE(a.n_old[l] >= 0 && a.n_old[l] <= d->cache_len, "inverse_block: cache too small");
    }
    a.zout = d->zout; a.ldzout = d->ldzout; a.ldj = d->log_det_J;
    a.zout_slabs = d->zout_slabs > 1 ? d->zout_slabs : 1; a.zout_slab_stride = d->zout_slab_stride; a.wout = d->wout; a.ldwout = d->ldwout;
    TFEP_REQUIRE(d->feat_in && d->feat_periodic && d->in_cols, "inverse_block: NULL pointer");
    a.steps = d->steps; a.feat_cols = d->feat_cols; a.feat_sel = d->feat_sel;
    a.feat_in = d->feat_in; a.feat_per = d->feat_periodic; a.in_cols = d->in_cols;
    a.emb_lower = d->emb_lower;
    a.emb_scale = (float)(2.0 * 3.14159265358979323846 / ((double)d->emb_upper - (double)d->emb_lower));
    TFEP_REQUIRE(d->cache_len >= 0 && d->max_feats >= 0, "inverse_block: negative LDS size");
    a.cache_len = ib_round4(d->cache_len); a.max_feats = ib_round4(d->max_feats);
    a.stage_gstride = 8 * ib_stage_cols(d->cache_len, d->max_feats);
    for (int l = 0; l < d->n_layers; ++l)            // the weight stage is filled with 16-byte loads
        TFEP_REQUIRE(((uintptr_t)d->w[l] & 15) == 0 && d->ldw[l] % 4 == 0 && d->cache_col0[l] % 4 == 0,
                     "inverse_block: packed weights / cache_col0 of layer %d not aligned to 4 floats", l);
    TFEP_REQUIRE(((uintptr_t)d->wout & 15) == 0 && d->ldwout % 4 == 0, "inverse_block: packed output weights not aligned to 4 floats");
    if (d->kind == 1) {
        int rc = make_spline_args(d->spline, &a.sp);
        if (rc) return rc;
        TFEP_REQUIRE(a.sp.f.K <= 8, "inverse_block: at most 8 spline bins");
        a.P = a.sp.P;
    } else if (d->kind == 3) {
        // members of a mixed transformer: d->spline is an array of n_spline_groups descriptors, the step records carry
        // the member of their features (checked on the host side of the caller: it is device memory here)
        TFEP_REQUIRE(d->spline && d->n_spline_groups >= 1 && d->n_spline_groups <= IB_MAX_GROUPS,
                     "inverse_block: 1..%d spline groups", IB_MAX_GROUPS);
        a.P = 0;
        for (int g = 0; g < d->n_spline_groups; ++g) {
            if (d->spline[g].n_bins == 0) {            // a plain shift member: one parameter, x = y - parameter
                a.spg[g] = SplineArgs{};
                a.spg[g].f.K = 0;
                a.spg[g].P = 1;
                a.P = a.P > 1 ? a.P : 1;
                continue;
            }
            int rc = make_spline_args(&d->spline[g], &a.spg[g]);
            if (rc) return rc;
            TFEP_REQUIRE(a.spg[g].f.K <= 8, "inverse_block: at most 8 spline bins");
            a.P = a.spg[g].P > a.P ? a.spg[g].P : a.P;
        }
        for (int g = d->n_spline_groups; g < IB_MAX_GROUPS; ++g) a.spg[g] = a.spg[0];      // (a stray member id stays in bounds)
    } else if (d->kind == 2) {
        TFEP_REQUIRE(d->moebius_dim >= 1 && d->moebius_dim <= MOEBIUS_MAX_DIM, "inverse_block: Moebius dimension=%d unsupported (1..%d)",
                     d->moebius_dim, MOEBIUS_MAX_DIM);
        a.mb_dim = d->moebius_dim; a.mb_unit_sphere = d->moebius_unit_sphere; a.mb_max_radius = d->moebius_max_radius;
        a.P = 1;
    } else {
        a.P = 2;
    }
    TFEP_REQUIRE(a.P <= IB_MAX_P, "inverse_block: too many parameters per feature");
    TFEP_REQUIRE(d->rows_per_wave == 0 || d->rows_per_wave == 16 || d->rows_per_wave == 64,
                 "inverse_block: rows_per_wave must be 64 (or 0: one sample row per lane) or 16 (four lanes per row)");
    if (d->n_blocks > 0) {
        // ---- super-block launch: every block of the list in one launch of the paired 16-row kernel
        TFEP_REQUIRE(d->rows_per_wave == 16 && d->paired != 0,
                     "inverse_block: a super-block launch (n_blocks > 0) needs rows_per_wave = 16 and paired = 1");
        TFEP_REQUIRE(d->blocks != nullptr && d->zout_extra != nullptr, "inverse_block: super-block launch without block records / extra slab");
        // (n_steps of a super-block launch: the most steps any of its blocks has -- their records are kept in LDS beside the
        // per-feature tables, sized for max_feats of them)
        TFEP_REQUIRE(d->n_steps >= 1 && d->n_steps <= (int)ib_round4(d->max_feats),
                     "inverse_block: super-block launch: n_steps (the most steps of a block, %d) must be 1 .. max_feats rounded up to 4 (%d)",
                     d->n_steps, (int)ib_round4(d->max_feats));
        InverseSuperArgs sa = {};
        a.n_steps = 0;
        a.cache_len = ib_round8(d->cache_len);          // whole k-groups of 8 (the split activation cache)
        a.stage_gstride = 8 * ib_stage_cols_q4(d->cache_len, d->max_feats);
        for (int l = 0; l < d->n_layers; ++l) {
            TFEP_REQUIRE(d->z_extra[l] != nullptr, "inverse_block: super-block launch without the extra slab of layer %d", l);
            TFEP_REQUIRE(d->ldh[l] % 4 == 0 && ((uintptr_t)d->h[l] & 15) == 0, "inverse_block: h[%d] not aligned to 4 floats", l);
            sa.z_extra[l] = d->z_extra[l];
            a.z_slabs[l] = d->z_slabs[l] > 0 ? d->z_slabs[l] : 0;         // slabs of the super-block GEMMs; the records add the extra one
        }
        a.zout_slabs = d->zout_slabs > 0 ? d->zout_slabs : 0;
        sa.zout_extra = d->zout_extra;
        for (int l = 1; l <= d->n_layers; ++l) {
            if (d->ws[l] == nullptr) continue;
            TFEP_REQUIRE(d->ws_inv_scale[l] && d->h_inv_scale[l], "inverse_block: split products of layer %d need ws_inv_scale and h_inv_scale", l);
            TFEP_REQUIRE(((uintptr_t)d->ws[l] & 15) == 0 && d->ldws[l] % 8 == 0, "inverse_block: split pack of layer %d not aligned", l);
            sa.ws[l] = (const float*)d->ws[l]; sa.ldws[l] = d->ldws[l]; sa.ws_inv[l] = d->ws_inv_scale[l]; sa.hs_inv[l] = d->h_inv_scale[l];
        }
        sa.n_blocks = d->n_blocks;
        sa.blocks = d->blocks;
        const size_t lds_s = ib_lds_floats_q4_paired(d->n_layers, d->cache_len, d->max_feats) * sizeof(float);
        TFEP_REQUIRE(lds_s <= 160 * 1024, "inverse_block: the pair needs %zu bytes of LDS (> 160 KiB)", lds_s);
        a.lds_floats = (int)(lds_s / sizeof(float));
        sa.a = a;
        static const bool diag = getenv("TFEP_DIAG_INVERSE") != nullptr && atoi(getenv("TFEP_DIAG_INVERSE")) != 0;
        sa.diag_who = diag && atoi(getenv("TFEP_DIAG_INVERSE")) == 2;      // (2: the loader wave's view of the same phases)
        // waves_per_workgroup = 4: two pairs per workgroup (they share the weight fetches of the products); 0 / 2: one
        const bool two = d->waves_per_workgroup == 4;
        TFEP_REQUIRE(d->waves_per_workgroup == 0 || d->waves_per_workgroup == 2 || two,
                     "inverse_block: a super-block launch takes waves_per_workgroup 0 / 2 (one pair) or 4 (two pairs)");
        TFEP_REQUIRE(!two || 2 * lds_s <= 160 * 1024, "inverse_block: two pairs need %zu bytes of LDS (> 160 KiB)", 2 * lds_s);
        // split dots: with the split pack of every layer 1 .. L (and not switched off: TFEP_INV_SPLIT_DOTS=0)
        static const bool sd_env = getenv("TFEP_INV_SPLIT_DOTS") == nullptr || atoi(getenv("TFEP_INV_SPLIT_DOTS")) != 0;
        bool sd = sd_env;
        for (int l = 1; l <= d->n_layers; ++l) sd = sd && sa.ws[l] != nullptr;
        void (*skernel)(InverseSuperArgs);
#define TFEP_SK(KIND_, DIAG_, PAIRS_) (sd ? inverse_superblock_kernel<KIND_, DIAG_, PAIRS_, true> : inverse_superblock_kernel<KIND_, DIAG_, PAIRS_, false>)
        if (two)
            skernel = d->kind == 0 ? TFEP_SK(0, false, 2) : d->kind == 1 ? (diag ? TFEP_SK(1, true, 2) : TFEP_SK(1, false, 2))
                      : d->kind == 2 ? TFEP_SK(2, false, 2) : TFEP_SK(3, false, 2);
        else
            skernel = d->kind == 0 ? TFEP_SK(0, false, 1) : d->kind == 1 ? (diag ? TFEP_SK(1, true, 1) : TFEP_SK(1, false, 1))
                      : d->kind == 2 ? TFEP_SK(2, false, 1) : TFEP_SK(3, false, 1);
#undef TFEP_SK
        // the 8-bin plain / circular layouts as compile-time constants (split dots, no diagnostics)
        static const bool spec_env = getenv("TFEP_INV_SPEC") == nullptr || atoi(getenv("TFEP_INV_SPEC")) != 0;
        int spec = 0;
        if (spec_env && sd && !diag && d->kind == 1 && a.sp.f.K == 8 && !a.sp.f.identity && !a.sp.f.learn_lower && !a.sp.f.learn_upper && a.sp.P == 25)
            spec = a.sp.f.circular ? 2 : 1;
        if (spec == 1) skernel = two ? inverse_superblock_kernel<1, false, 2, true, 1> : inverse_superblock_kernel<1, false, 1, true, 1>;
        if (spec == 2) skernel = two ? inverse_superblock_kernel<1, false, 2, true, 2> : inverse_superblock_kernel<1, false, 1, true, 2>;
        const size_t lds_wg = (two ? 2 : 1) * lds_s;
        static size_t lds_attr_s[24][TFEP_MAX_DEVICES] = {};
        size_t& attr = lds_attr_s[spec > 0 ? 20 + (spec - 1) * 2 + (two ? 1 : 0)
                                           : (d->kind == 1 && diag ? 4 : d->kind) + (two ? 5 : 0) + (sd ? 10 : 0)][current_device_slot()];
        if (lds_wg > attr) {
            hipError_t e = hipFuncSetAttribute((const void*)skernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wg);
            if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "hipFuncSetAttribute(LDS=%zu): %s", lds_wg, hipGetErrorString(e));
            attr = lds_wg;
        }
        const long long n_pairs = (d->B + Q4_ROWS - 1) / Q4_ROWS;
        skernel<<<(unsigned)(two ? (n_pairs + 1) / 2 : n_pairs), two ? 256 : 128, lds_wg, (hipStream_t)stream>>>(sa);
        return check_launch("inverse_superblock_kernel");
    }
    const bool q4 = d->rows_per_wave == 16;
    size_t lds = (q4 ? ib_lds_floats_q4(d->n_layers, d->cache_len, d->max_feats)
                           : ib_lds_floats(d->n_layers, d->cache_len, d->max_feats)) * sizeof(float);
    a.lds_floats = (int)(lds / sizeof(float));
    TFEP_REQUIRE(lds <= 160 * 1024, "inverse_block: block needs %zu bytes of LDS (> 160 KiB)", lds);
    static size_t lds_attr_on[12][TFEP_MAX_DEVICES] = {};      // per kernel and device: a process may drive several GPUs
    const bool pair = q4 && d->paired != 0;
    size_t& lds_attr = lds_attr_on[d->kind + (q4 ? 4 : 0) + (pair ? 4 : 0)][current_device_slot()];
    void (*kernel)(InverseBlockArgs) =
        pair ? (d->kind == 0 ? inverse_block_q4_kernel<0, true> : d->kind == 1 ? inverse_block_q4_kernel<1, true>
                : d->kind == 2 ? inverse_block_q4_kernel<2, true> : inverse_block_q4_kernel<3, true>)
        : q4 ? (d->kind == 0 ? inverse_block_q4_kernel<0, false> : d->kind == 1 ? inverse_block_q4_kernel<1, false>
                : d->kind == 2 ? inverse_block_q4_kernel<2, false> : inverse_block_q4_kernel<3, false>)
             : (d->kind == 0 ? inverse_block_kernel<0> : d->kind == 1 ? inverse_block_kernel<1>
                : d->kind == 2 ? inverse_block_kernel<2> : inverse_block_kernel<3>);
    const int rows = q4 ? Q4_ROWS : 64;
    if (q4) a.stage_gstride = 8 * ib_stage_cols_q4(d->cache_len, d->max_feats);
    int wpw = d->waves_per_workgroup > 1 ? d->waves_per_workgroup : 1;
    TFEP_REQUIRE(wpw == 1 || (q4 && (wpw == 2 || wpw == 4 || wpw == 8)), "inverse_block: waves_per_workgroup must be 1, 2, 4 or 8 (16-row layout)");
    TFEP_REQUIRE(!d->paired || (q4 && wpw == 1), "inverse_block: paired needs rows_per_wave = 16 and one pair per workgroup");
    const size_t lds_wave = lds;
    lds = pair ? ib_lds_floats_q4_paired(d->n_layers, d->cache_len, d->max_feats) * sizeof(float) : lds_wave * wpw;
    if (pair) {
        wpw = 2;                                        // the pair: consumer + loader on the same rows
        a.lds_floats = (int)(lds / sizeof(float));
    }
    TFEP_REQUIRE(lds <= 160 * 1024, "inverse_block: %d waves per workgroup need %zu bytes of LDS (> 160 KiB)", wpw, lds);
    if (lds > lds_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e));
        lds_attr = lds;
    }
    const long long n_waves = (d->B + rows - 1) / rows;
    kernel<<<(unsigned)(pair ? n_waves : (n_waves + wpw - 1) / wpw), 64 * wpw, lds, (hipStream_t)stream>>>(a);
    return check_launch("inverse_block_kernel");
}

}  // extern "C"
