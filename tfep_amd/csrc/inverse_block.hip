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
//   the masked packed weights are read with wave-uniform addresses (scalar loads, broadcast);
//   fp32 FMA -- the arithmetic is ~20 kFLOP per row per block, nothing for the matrix cores to do;
//   the transformer inverse is spline.h's rq_spline_element (same code as the stand-alone kernel) or the affine map.
#include "common.h"
#include "spline.h"

namespace tfep {

constexpr int IB_MAX_LAYERS = 4;
constexpr int IB_STEP_INTS = 4 * IB_MAX_LAYERS + 6;
constexpr int IB_MAX_P = 32;

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
    SplineArgs sp;
};

__device__ inline float elu_ib(float v) { return v > 0.f ? v : expm1f(v); }

// sum of the split-K slabs of one pre-activation (fixed order: deterministic)
__device__ __forceinline__ float slab_sum(const float* p, int slabs, int64_t stride) {
    float v = p[0];
    for (int s = 1; s < slabs; ++s) v += p[s * stride];
    return v;
}

// acc[g] += sum_{j in [kb, ke)} w[(row0 + g * row_stride) * ldw + j] * cache[(j - c0) * 64 + lane],  g < G.
// G independent accumulators share every LDS read: a single wave has no other source of instruction-level
// parallelism, and a one-accumulator loop runs at the LDS + scalar-load latency per FMA.
template <int G>
__device__ __forceinline__ void dot_rows(float (&acc)[G], const float* __restrict__ w, int64_t ldw, int row0, int row_stride,
                                         int n_valid, const float* __restrict__ cp, int c0, int kb, int ke, int lane) {
    const float* wr[G];
#pragma unroll
    for (int g = 0; g < G; ++g) wr[g] = w + (int64_t)(row0 + (g < n_valid ? g : 0) * row_stride) * ldw;
    int j = kb;
    for (; j + 4 <= ke; j += 4) {
        const float h0 = cp[(j - c0) * 64 + lane], h1 = cp[(j + 1 - c0) * 64 + lane];
        const float h2 = cp[(j + 2 - c0) * 64 + lane], h3 = cp[(j + 3 - c0) * 64 + lane];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            acc[g] = fmaf(wr[g][j], h0, acc[g]);
            acc[g] = fmaf(wr[g][j + 1], h1, acc[g]);
            acc[g] = fmaf(wr[g][j + 2], h2, acc[g]);
            acc[g] = fmaf(wr[g][j + 3], h3, acc[g]);
        }
    }
    for (; j < ke; ++j) {
        const float h0 = cp[(j - c0) * 64 + lane];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = fmaf(wr[g][j], h0, acc[g]);
    }
}

// Step record: per layer l  [row0, n, kb, ke]: units [row0, row0 + n) of layer l are computed from the inputs
//   l == 0: the first `ke` conditioner-input entries of the block (in_cols order);  l >= 1: packed columns [kb, ke) of layer l - 1
// then [out_row0, n_d, out_kb, out_ke, feat_off, 0].
__global__ void __launch_bounds__(64) inverse_block_kernel(InverseBlockArgs a) {
    extern __shared__ float cache[];              // [L][cache_len][64] hidden activations, then [max_feats][64] x values
    const int lane = threadIdx.x;
    const int row = blockIdx.x * 64 + lane;
    const bool live = row < a.B;
    const int64_t r = live ? row : 0;             // dead lanes shadow row 0 and store nothing
    float* xc = cache + (size_t)a.L * a.cache_len * 64;

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
            const float* zr = a.z[l] + r * a.ldz[l];
            float* hr = a.h[l] + r * a.ldh[l];
            for (int u0 = row0; u0 < row0 + n; u0 += 8) {
                const int nu = min(8, row0 + n - u0);
                float acc[8];
#pragma unroll
                for (int g = 0; g < 8; ++g)                                     // earlier blocks + bias
                    acc[g] = g < nu ? slab_sum(zr + u0 + g, a.z_slabs[l], a.z_slab_stride[l]) : 0.f;
                if (l == 0) {
                    for (int j = 0; j < ke; ++j) {
                        const float xv = xc[j * 64 + lane];
                        const int col = a.in_cols[j];
#pragma unroll
                        for (int g = 0; g < 8; ++g)
                            acc[g] = fmaf(a.w[0][(int64_t)(u0 + (g < nu ? g : 0)) * a.ldw[0] + col], xv, acc[g]);
                    }
                } else {
                    dot_rows<8>(acc, a.w[l], a.ldw[l], u0, 1, nu, cache + (size_t)(l - 1) * a.cache_len * 64, a.c0[l - 1],
                                kb, ke, lane);
                }
#pragma unroll
                for (int g = 0; g < 8; ++g)
                    if (g < nu) {
                        const float hv = elu_ib(acc[g]);
                        cl[(u0 + g - a.c0[l]) * 64 + lane] = hv;
                        if (live) hr[u0 + g] = hv;
                    }
            }
        }
        // ---- parameters and transformer inverse of this degree's features
        const int out_row0 = st[4 * IB_MAX_LAYERS], n_d = st[4 * IB_MAX_LAYERS + 1];
        const int okb = st[4 * IB_MAX_LAYERS + 2], oke = st[4 * IB_MAX_LAYERS + 3], foff = st[4 * IB_MAX_LAYERS + 4];
        const float* cp = cache + (size_t)(a.L - 1) * a.cache_len * 64;
        const int c0p = a.c0[a.L - 1];
        const float* zo = a.zout + r * a.ldzout;
        for (int f = 0; f < n_d; ++f) {
            float prm[IB_MAX_P];
#pragma unroll
            for (int p0 = 0; p0 < IB_MAX_P; p0 += 8) {
                float acc[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) acc[g] = 0.f;
                if (p0 < a.P) {                                         // wave-uniform
                    const int np = min(8, a.P - p0);
#pragma unroll
                    for (int g = 0; g < 8; ++g)
                        if (g < np) acc[g] = slab_sum(zo + out_row0 + (p0 + g) * n_d + f, a.zout_slabs, a.zout_slab_stride);
                    dot_rows<8>(acc, a.wout, a.ldwout, out_row0 + p0 * n_d + f, n_d, np, cp, c0p, okb, oke, lane);
                }
#pragma unroll
                for (int g = 0; g < 8; ++g) prm[p0 + g] = acc[g];
            }
            const int sel = a.feat_sel[foff + f], col = a.feat_cols[foff + f];
            const float yv = a.y[r * a.ldy + sel];
            float xv;
            if (a.kind == 0) {                                          // affine.py:361-363
                xv = (yv - prm[0]) * expf(-prm[1]);
                ldj_acc -= (double)prm[1];
            } else {
                const SplineFlags& fl = a.sp.f;
                const int K = fl.K;
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
                if (fl.circular || fl.learn_lower || fl.learn_upper) last = prm[a.P - 1];
                if (fl.learn_lower && fl.learn_upper) last2 = prm[a.P - 2];
                double ld;
                xv = (float)rq_spline_element<8, true>(w, hh, sraw, last, last2, fl, a.sp.x0[sel], a.sp.xf[sel],
                                                       a.sp.y0[sel], a.sp.yf[sel], yv, &ld);
                ldj_acc -= ld;
            }
            // the new feature becomes conditioner input: itself, or (cos, sin) under a periodic embedding
            // (same arithmetic as periodic_embedding_kernel, mafembed.py:112-145)
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
    if (live) a.ldj[row] = (float)((double)a.ldj[row] + ldj_acc);
}

}  // namespace tfep

using namespace tfep;

extern "C" {

int tfep_inverse_block_step_ints(void) { return IB_STEP_INTS; }

int tfep_inverse_block(const tfep_inverse_block_desc* d, void* stream) {
    TFEP_REQUIRE(d != nullptr, "inverse_block: NULL descriptor");
    TFEP_REQUIRE(d->B >= 0 && d->n_steps >= 0, "inverse_block: negative size");
    if (d->B == 0 || d->n_steps == 0) return TFEP_OK;
    TFEP_REQUIRE(d->n_layers >= 1 && d->n_layers <= IB_MAX_LAYERS, "inverse_block: 1..%d hidden layers", IB_MAX_LAYERS);
    TFEP_REQUIRE(d->kind == 0 || d->kind == 1, "inverse_block: kind must be 0 (affine) or 1 (spline)");
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
    a.cache_len = d->cache_len; a.max_feats = d->max_feats;
    if (d->kind == 1) {
        int rc = make_spline_args(d->spline, &a.sp);
        if (rc) return rc;
        TFEP_REQUIRE(a.sp.f.K <= 8, "inverse_block: at most 8 spline bins");
        a.P = a.sp.P;
    } else {
        a.P = 2;
    }
    TFEP_REQUIRE(a.P <= IB_MAX_P, "inverse_block: too many parameters per feature");
    const size_t lds = ((size_t)a.L * a.cache_len + a.max_feats) * 64 * sizeof(float);
    TFEP_REQUIRE(lds <= 160 * 1024, "inverse_block: block needs %zu bytes of LDS (> 160 KiB)", lds);
    static size_t lds_attr_on[TFEP_MAX_DEVICES] = {};          // per device: a process may drive several GPUs
    size_t& lds_attr = lds_attr_on[current_device_slot()];
    if (lds > lds_attr) {
        hipError_t e = hipFuncSetAttribute((const void*)inverse_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "hipFuncSetAttribute(LDS=%zu): %s", lds, hipGetErrorString(e));
        lds_attr = lds;
    }
    inverse_block_kernel<<<(unsigned)((d->B + 63) / 64), 64, lds, (hipStream_t)stream>>>(a);
    return check_launch("inverse_block_kernel");
}

}  // extern "C"
