// Standalone (unfused) transformer kernels: affine, volume-preserving shift, RQ spline,
// Moebius, periodic embedding, column gather/scatter.  HBM-bound elementwise work with a
// per-sample log|det J| reduction: one wavefront owns one sample (row), its 64 lanes walk the
// features with unit stride (coalesced 256-B segments per parameter row) and the log-derivative
// is summed in fp64 with a wave butterfly -- no atomics, bit-reproducible.
#include "common.h"
#include "spline.h"
#include "moebius.h"

#include <stdarg.h>

namespace tfep {

std::string& last_error() {
    static thread_local std::string s;
    return s;
}

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

constexpr int ROWS_PER_BLOCK = 4;   // 4 waves = 256 threads

__device__ inline void store_ldj(float* ldj, int b, double total, int accumulate) {
    if ((threadIdx.x & 63) == 0) {
        if (accumulate)
            ldj[b] = (float)((double)ldj[b] + total);
        else
            ldj[b] = (float)total;
    }
}

// ---------------------------------------------------------------- affine (affine.py:321-323, :361-363)
template <bool INVERSE>
__global__ void __launch_bounds__(256) affine_kernel(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ params, tfep_param_layout L,
                                                     float* __restrict__ y, int64_t ldy, float* __restrict__ ldj,
                                                     int accumulate, int B, int D) {
    const int b = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (int64_t)b * ldx;
    const float* pr = params + (int64_t)b * L.ld;
    float* yr = y + (int64_t)b * ldy;
    double acc = 0.0;
    for (int f = lane; f < D; f += 64) {
        const float shift = pr[f * L.stride_f];
        const float ls = pr[L.stride_p + f * L.stride_f];
        const float v = xr[f];
        if (INVERSE)
            yr[f] = (v - shift) * expf(-ls);
        else
            yr[f] = v * expf(ls) + shift;
        acc += (double)ls;
    }
    acc = wave_sum(acc);
    if (ldj) store_ldj(ldj, b, INVERSE ? -acc : acc, accumulate);
}

// ---------------------------------------------------------------- volume preserving shift (affine.py:366-456)
__global__ void __launch_bounds__(256) volpres_kernel(const float* __restrict__ x, int64_t ldx,
                                                      const float* __restrict__ shift, int64_t ldp,
                                                      const int32_t* __restrict__ periodic, float lower,
                                                      float upper, float sign, float* __restrict__ y,
                                                      int64_t ldy, int B, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * D) return;
    const int b = (int)(i / D), f = (int)(i % D);
    float v = x[(int64_t)b * ldx + f] + sign * shift[(int64_t)b * ldp + f];
    if (periodic && periodic[f]) {
        // float32 `%` with Python semantics, then + lower (affine.py:409, :454)
        const float period = upper - lower;
        float r = fmodf(v, period);
        if (r != 0.f && ((r < 0.f) != (period < 0.f))) r += period;
        v = r + lower;
    }
    y[(int64_t)b * ldy + f] = v;
}

// ---------------------------------------------------------------- RQ spline (spline.py)
template <int KMAX, bool INVERSE>
__global__ void __launch_bounds__(256) spline_kernel(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ params, tfep_param_layout L,
                                                     SplineArgs a, float* __restrict__ y, int64_t ldy,
                                                     float* __restrict__ ldj, int accumulate, int B, int D) {
    const int b = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (int64_t)b * ldx;
    const float* pr = params + (int64_t)b * L.ld;
    float* yr = y + (int64_t)b * ldy;
    const int K = a.f.K;
    double acc = 0.0;
    for (int f = lane; f < D; f += 64) {
        const float* pf = pr + f * L.stride_f;
        float w[KMAX], h[KMAX], sraw[KMAX + 1];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            w[k] = 0.f;
            h[k] = 0.f;
            if (k < K) {
                w[k] = pf[k * L.stride_p];
                h[k] = pf[(K + k) * L.stride_p];
            }
        }
#pragma unroll
        for (int j = 0; j <= KMAX; ++j) {
            sraw[j] = 0.f;
            if (j <= K) {
                const int pi = spline_slope_param(j, K, a.f.circular, a.f.identity);
                if (pi >= 0) sraw[j] = pf[pi * L.stride_p];
            }
        }
        float last = 0.f, last2 = 0.f;
        if (a.f.circular || a.f.learn_lower || a.f.learn_upper) last = pf[(a.P - 1) * L.stride_p];
        if (a.f.learn_lower && a.f.learn_upper) last2 = pf[(a.P - 2) * L.stride_p];
        double ld;
        const double out = rq_spline_element<KMAX, INVERSE>(w, h, sraw, last, last2, a.f, a.x0[f], a.xf[f],
                                                            a.y0[f], a.yf[f], xr[f], &ld);
        yr[f] = (float)out;
        acc += ld;
    }
    acc = wave_sum(acc);
    if (ldj) store_ldj(ldj, b, INVERSE ? -acc : acc, accumulate);
}

// ---------------------------------------------------------------- Moebius (moebius.py:374-478)
// One lane per d-vector; the map and its closed-form log|det J| are in moebius.h.  DIM > 0: the dimension is a compile-time
// constant (d = 2, 3: the loops of moebius_vector unroll to d steps instead of MOEBIUS_MAX_DIM predicated ones; the
// d = 2 vector is one 8-byte load / store) -- the same arithmetic in the same order, bit for bit; DIM = 0: any d.
template <int DIM>
__global__ void __launch_bounds__(256) moebius_kernel(const float* __restrict__ x, int64_t ldx,
                                                      const float* __restrict__ params, int64_t ldp, int dim_rt,
                                                      float max_radius, int unit_sphere, float sign,
                                                      float* __restrict__ y, int64_t ldy, float* __restrict__ ldj,
                                                      int accumulate, int B, int D, uint32_t* __restrict__ y_split = nullptr,
                                                      int64_t ld_split = 0, float* __restrict__ y_inv_scale = nullptr) {
    const int b = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int dim = DIM > 0 ? DIM : dim_rt;
    const int nvec = D / dim;
    const float* xr = x + (int64_t)b * ldx;
    const float* pr = params + (int64_t)b * ldp;
    float* yr = y + (int64_t)b * ldy;
    // (8-byte accesses for d = 2 when the three rows start on 8-byte boundaries: wave uniform)
    const bool vec2 = DIM == 2 && (((uintptr_t)xr | (uintptr_t)pr | (uintptr_t)yr) & 7u) == 0;
    double acc = 0.0;
    for (int v = lane; v < nvec; v += 64) {
        double xv[MOEBIUS_MAX_DIM], wv[MOEBIUS_MAX_DIM], yv[MOEBIUS_MAX_DIM];
        if (vec2) {
            const float2 xx = reinterpret_cast<const float2*>(xr)[v], pp = reinterpret_cast<const float2*>(pr)[v];
            xv[0] = (double)xx.x; xv[1] = (double)xx.y;
            wv[0] = (double)(sign * pp.x); wv[1] = (double)(sign * pp.y);
        } else {
#pragma unroll
            for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
                if (i < dim) {
                    xv[i] = (double)xr[v * dim + i];
                    wv[i] = (double)(sign * pr[v * dim + i]);
                }
        }
        acc += moebius_vector(xv, wv, dim, max_radius, unit_sphere, yv);
        if (vec2) {
            reinterpret_cast<float2*>(yr)[v] = float2{(float)yv[0], (float)yv[1]};
            if (y_split) {
                // the same row as split-f16 halves for the next layer's GEMM (tfep_split_rows' format: per 8 features 8 hi then
                // 8 lo halves), with the scale of the bound |y| <= 1 of a unit-sphere map: 2^14
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const float t0 = (float)yv[0] * 16384.0f, t1 = (float)yv[1] * 16384.0f;
                const _Float16 h0 = (_Float16)t0, h1 = (_Float16)t1;
                const h2 hi = {h0, h1}, lo = {(_Float16)(t0 - (float)h0), (_Float16)(t1 - (float)h1)};
                uint32_t* grp = y_split + (int64_t)b * ld_split + (v >> 2) * 8;        // 8 words = 32 bytes per group of 8 features
                grp[v & 3] = __builtin_bit_cast(uint32_t, hi);
                grp[4 + (v & 3)] = __builtin_bit_cast(uint32_t, lo);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MOEBIUS_MAX_DIM; ++i)
                if (i < dim) yr[v * dim + i] = (float)yv[i];
        }
    }
    if (y_split) {
        // padding groups up to the row's k-tile stay zero (the caller clears the buffer once); the row's 1/scale
        if (lane == 0) y_inv_scale[b] = 1.0f / 16384.0f;
    }
    acc = wave_sum(acc);
    if (ldj) store_ldj(ldj, b, acc, accumulate);
}

// ---------------------------------------------------------------- periodic embedding (mafembed.py:112-145)
__global__ void __launch_bounds__(256) periodic_embedding_kernel(const float* __restrict__ x, int64_t ldx,
                                                                 const int32_t* __restrict__ pidx, int n_per,
                                                                 const int32_t* __restrict__ nidx, int n_non,
                                                                 float lower, float scale,
                                                                 float* __restrict__ out, int64_t ldo, int B) {
    const int n_out = n_non + n_per;     // one thread per (row, source feature)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * n_out) return;
    const int b = (int)(i / n_out), j = (int)(i % n_out);
    if (j < n_non) {
        out[(int64_t)b * ldo + j] = x[(int64_t)b * ldx + nidx[j]];
    } else {
        const int q = j - n_non;
        const float t = (x[(int64_t)b * ldx + pidx[q]] - lower) * scale;
        float s, c;
        sincosf(t, &s, &c);
        out[(int64_t)b * ldo + n_non + 2 * q] = c;
        out[(int64_t)b * ldo + n_non + 2 * q + 1] = s;
    }
}

// ---------------------------------------------------------------- column gather / scatter
template <bool SCATTER>
__global__ void __launch_bounds__(256) columns_kernel(const float* __restrict__ src, int64_t lds,
                                                      const int32_t* __restrict__ idx, int n_idx,
                                                      float* __restrict__ dst, int64_t ldd, int B) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * n_idx) return;
    const int b = (int)(i / n_idx), j = (int)(i % n_idx);
    if (SCATTER)
        dst[(int64_t)b * ldd + idx[j]] = src[(int64_t)b * lds + j];
    else
        dst[(int64_t)b * ldd + j] = src[(int64_t)b * lds + idx[j]];
}

static inline unsigned row_blocks(int B) { return (unsigned)((B + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK); }

template <bool INVERSE>
static int launch_spline(const float* x, int64_t ldx, const float* params, tfep_param_layout L,
                         const tfep_spline_desc* desc, float* y, int64_t ldy, float* ldj, int accumulate,
                         int B, int D, void* stream) {
    SplineArgs a;
    int rc = make_spline_args(desc, &a);
    if (rc) return rc;
    TFEP_REQUIRE(B >= 0 && D >= 0, "spline: negative size");
    if (B == 0) return TFEP_OK;
    TFEP_REQUIRE(x && params && y, "spline: x/params/y must be non-NULL");
    hipStream_t s = (hipStream_t)stream;
    if (a.f.K <= 8)
        spline_kernel<8, INVERSE><<<row_blocks(B), 256, 0, s>>>(x, ldx, params, L, a, y, ldy, ldj, accumulate, B, D);
    else if (a.f.K <= 16)
        spline_kernel<16, INVERSE><<<row_blocks(B), 256, 0, s>>>(x, ldx, params, L, a, y, ldy, ldj, accumulate, B, D);
    else
        spline_kernel<32, INVERSE><<<row_blocks(B), 256, 0, s>>>(x, ldx, params, L, a, y, ldy, ldj, accumulate, B, D);
    return check_launch("spline_kernel");
}

}  // namespace tfep

using namespace tfep;

extern "C" {

int tfep_hip_abi_version(void) { return TFEP_HIP_ABI_VERSION; }
const char* tfep_last_error(void) { return last_error().c_str(); }

int tfep_affine_forward(const float* x, int64_t ldx, const float* params, tfep_param_layout layout, float* y,
                        int64_t ldy, float* log_det_J, int accumulate, int B, int D, void* stream) {
    TFEP_REQUIRE(B >= 0 && D >= 0, "affine: negative size");
    if (B == 0) return TFEP_OK;
    TFEP_REQUIRE(x && params && y, "affine: x/params/y must be non-NULL");
    affine_kernel<false><<<row_blocks(B), 256, 0, (hipStream_t)stream>>>(x, ldx, params, layout, y, ldy, log_det_J,
                                                                         accumulate, B, D);
    return check_launch("affine_kernel");
}

int tfep_affine_inverse(const float* y, int64_t ldy, const float* params, tfep_param_layout layout, float* x,
                        int64_t ldx, float* log_det_J, int accumulate, int B, int D, void* stream) {
    TFEP_REQUIRE(B >= 0 && D >= 0, "affine: negative size");
    if (B == 0) return TFEP_OK;
    TFEP_REQUIRE(x && params && y, "affine: x/params/y must be non-NULL");
    affine_kernel<true><<<row_blocks(B), 256, 0, (hipStream_t)stream>>>(y, ldy, params, layout, x, ldx, log_det_J,
                                                                        accumulate, B, D);
    return check_launch("affine_kernel");
}

int tfep_volume_preserving_shift(const float* x, int64_t ldx, const float* shift, int64_t ldp,
                                 const int32_t* periodic_mask, float lower, float upper, int sign, float* y,
                                 int64_t ldy, int B, int D, void* stream) {
    TFEP_REQUIRE(sign == 1 || sign == -1, "volume_preserving_shift: sign must be +1 or -1");
    if ((int64_t)B * D == 0) return TFEP_OK;
    TFEP_REQUIRE(x && shift && y, "volume_preserving_shift: x/shift/y must be non-NULL");
    const int64_t n = (int64_t)B * D;
    volpres_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, ldx, shift, ldp, periodic_mask,
                                                                                 lower, upper, (float)sign, y, ldy, B, D);
    return check_launch("volpres_kernel");
}

int tfep_spline_n_parameters_per_feature(const tfep_spline_desc* d) {
    if (!d) return fail(TFEP_ERR_INVALID_ARGUMENT, "spline descriptor is NULL");
    return spline_n_params(d->n_bins, d->circular != 0, d->identity_boundary_slopes != 0, d->learn_lower_bound != 0,
                           d->learn_upper_bound != 0);
}

int tfep_spline_forward(const float* x, int64_t ldx, const float* params, tfep_param_layout layout,
                        const tfep_spline_desc* desc, float* y, int64_t ldy, float* log_det_J, int accumulate, int B,
                        int D, void* stream) {
    return launch_spline<false>(x, ldx, params, layout, desc, y, ldy, log_det_J, accumulate, B, D, stream);
}

int tfep_spline_inverse(const float* y, int64_t ldy, const float* params, tfep_param_layout layout,
                        const tfep_spline_desc* desc, float* x, int64_t ldx, float* log_det_J, int accumulate, int B,
                        int D, void* stream) {
    return launch_spline<true>(y, ldy, params, layout, desc, x, ldx, log_det_J, accumulate, B, D, stream);
}

int tfep_moebius_forward(const float* x, int64_t ldx, const float* params, int64_t ldp, int dimension,
                         float max_radius, int unit_sphere, int sign, float* y, int64_t ldy, float* log_det_J,
                         int accumulate, int B, int D, void* stream) {
    TFEP_REQUIRE(B == 0 || (x && params && y), "moebius: x/params/y must be non-NULL");
    TFEP_REQUIRE(dimension >= 1 && dimension <= MOEBIUS_MAX_DIM, "moebius: dimension=%d unsupported (1..%d)", dimension,
                 MOEBIUS_MAX_DIM);
    TFEP_REQUIRE(D % dimension == 0, "moebius: n_features=%d is not a multiple of dimension=%d", D, dimension);
    TFEP_REQUIRE(sign == 1 || sign == -1, "moebius: sign must be +1 or -1");
    if (B == 0) return TFEP_OK;
    auto kernel = dimension == 2 ? moebius_kernel<2> : dimension == 3 ? moebius_kernel<3> : moebius_kernel<0>;
    kernel<<<row_blocks(B), 256, 0, (hipStream_t)stream>>>(x, ldx, params, ldp, dimension, max_radius, unit_sphere, (float)sign, y,
                                                           ldy, log_det_J, accumulate, B, D, nullptr, 0, nullptr);
    return check_launch("moebius_kernel");
}

int tfep_moebius_forward_split_out(const float* x, int64_t ldx, const float* params, int64_t ldp, float max_radius, float* y, int64_t ldy,
                                   float* log_det_J, int accumulate, void* y_split, int64_t ld_split, float* y_inv_scale, int B, int D,
                                   void* stream) {
    TFEP_REQUIRE(B == 0 || (x && params && y && y_split && y_inv_scale), "moebius_split_out: NULL pointer");
    TFEP_REQUIRE(D % 2 == 0 && D % 8 == 0, "moebius_split_out: n_features=%d must be a multiple of 8 (whole split groups of 2-vectors)", D);
    TFEP_REQUIRE(ld_split >= D && ld_split % 8 == 0 && ((uintptr_t)y_split & 15) == 0, "moebius_split_out: split rows too short / not aligned");
    TFEP_REQUIRE(((uintptr_t)x & 7) == 0 && ((uintptr_t)params & 7) == 0 && ((uintptr_t)y & 7) == 0 && ldx % 2 == 0 && ldp % 2 == 0 && ldy % 2 == 0,
                 "moebius_split_out: x, params and y rows must start on 8-byte boundaries");
    if (B == 0) return TFEP_OK;
    moebius_kernel<2><<<row_blocks(B), 256, 0, (hipStream_t)stream>>>(x, ldx, params, ldp, 2, max_radius, 1, 1.0f, y, ldy, log_det_J, accumulate, B,
                                                                      D, (uint32_t*)y_split, ld_split, y_inv_scale);
    return check_launch("moebius_kernel");
}

int tfep_periodic_embedding(const float* x, int64_t ldx, const int32_t* periodic_indices, int n_periodic,
                            const int32_t* nonperiodic_indices, int n_nonperiodic, float lower, float upper,
                            float* out, int64_t ldo, int B, void* stream) {
    TFEP_REQUIRE(B == 0 || (x && out), "periodic_embedding: x/out must be non-NULL");
    TFEP_REQUIRE(n_periodic == 0 || periodic_indices, "periodic_embedding: periodic_indices is NULL");
    TFEP_REQUIRE(n_nonperiodic == 0 || nonperiodic_indices, "periodic_embedding: nonperiodic_indices is NULL");
    const int64_t n = (int64_t)B * (n_periodic + n_nonperiodic);
    if (n == 0) return TFEP_OK;
    const float scale = (float)(2.0 * 3.14159265358979323846 / ((double)upper - (double)lower));
    periodic_embedding_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        x, ldx, periodic_indices, n_periodic, nonperiodic_indices, n_nonperiodic, lower, scale, out, ldo, B);
    return check_launch("periodic_embedding_kernel");
}

int tfep_gather_columns(const float* src, int64_t lds, const int32_t* idx, int n_idx, float* dst, int64_t ldd, int B,
                        void* stream) {
    const int64_t n = (int64_t)B * n_idx;
    if (n == 0) return TFEP_OK;
    TFEP_REQUIRE(src && dst && idx, "gather_columns: NULL pointer");
    columns_kernel<false><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, lds, idx, n_idx, dst, ldd, B);
    return check_launch("gather_columns");
}

int tfep_scatter_columns(const float* src, int64_t lds, const int32_t* idx, int n_idx, float* dst, int64_t ldd, int B,
                         void* stream) {
    const int64_t n = (int64_t)B * n_idx;
    if (n == 0) return TFEP_OK;
    TFEP_REQUIRE(src && dst && idx, "scatter_columns: NULL pointer");
    columns_kernel<true><<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, lds, idx, n_idx, dst, ldd, B);
    return check_launch("scatter_columns");
}

}  // extern "C"
