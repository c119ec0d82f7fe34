// TFEP loss / free-energy estimator sufficient statistics (reference loss.py:125-140,
// analysis/estimator.py:73-86).  One pass over the local shard; the 9 fp64 statistics are what
// the multi-GPU path all-reduces (max + rescaled sums) -- see tfep_hip.h.
//
// Two kernels: per-block partials (online max / rescaled-sum, fp64) then a single-block
// combine.  No atomics -> bit-reproducible for a fixed N.
#include "common.h"

namespace tfep {

struct Stats {
    double count, sum_r;
    double m_w, s_w, s_wr;
    double m_e, s_e;
    double m_b, s_b;
};

__device__ inline void online_add(double& m, double& s, double v) {
    // s = sum exp(v_i - m), m = running max.  -inf values contribute nothing.
    if (v == -INFINITY) return;
    if (v > m) {
        s = s * exp(m - v) + 1.0;
        m = v;
    } else {
        s += exp(v - m);
    }
}

__device__ inline void online_add2(double& m, double& s, double& sr, double v, double r) {
    if (v == -INFINITY) return;
    if (v > m) {
        const double c = exp(m - v);
        s = s * c + 1.0;
        sr = sr * c + r;
        m = v;
    } else {
        const double e = exp(v - m);
        s += e;
        sr += e * r;
    }
}

__device__ inline void combine(double& m, double& s, double m2, double s2) {
    if (m2 == -INFINITY) return;
    if (m == -INFINITY) {
        m = m2;
        s = s2;
        return;
    }
    const double mm = fmax(m, m2);
    s = s * exp(m - mm) + s2 * exp(m2 - mm);
    m = mm;
}

__device__ inline void combine2(double& m, double& s, double& sr, double m2, double s2, double sr2) {
    if (m2 == -INFINITY) return;
    if (m == -INFINITY) {
        m = m2;
        s = s2;
        sr = sr2;
        return;
    }
    const double mm = fmax(m, m2);
    const double c1 = exp(m - mm), c2 = exp(m2 - mm);
    s = s * c1 + s2 * c2;
    sr = sr * c1 + sr2 * c2;
    m = mm;
}

__device__ inline Stats stats_identity() {
    Stats s;
    s.count = 0.0;
    s.sum_r = 0.0;
    s.m_w = -INFINITY;
    s.s_w = 0.0;
    s.s_wr = 0.0;
    s.m_e = -INFINITY;
    s.s_e = 0.0;
    s.m_b = -INFINITY;
    s.s_b = 0.0;
    return s;
}

__device__ inline void stats_merge(Stats& a, const Stats& b) {
    a.count += b.count;
    a.sum_r += b.sum_r;
    combine2(a.m_w, a.s_w, a.s_wr, b.m_w, b.s_w, b.s_wr);
    combine(a.m_e, a.s_e, b.m_e, b.s_e);
    combine(a.m_b, a.s_b, b.m_b, b.s_b);
}

__device__ inline Stats stats_shfl_xor(const Stats& s, int off) {
    Stats o;
    o.count = __shfl_xor(s.count, off, 64);
    o.sum_r = __shfl_xor(s.sum_r, off, 64);
    o.m_w = __shfl_xor(s.m_w, off, 64);
    o.s_w = __shfl_xor(s.s_w, off, 64);
    o.s_wr = __shfl_xor(s.s_wr, off, 64);
    o.m_e = __shfl_xor(s.m_e, off, 64);
    o.s_e = __shfl_xor(s.s_e, off, 64);
    o.m_b = __shfl_xor(s.m_b, off, 64);
    o.s_b = __shfl_xor(s.s_b, off, 64);
    return o;
}

__device__ inline void block_reduce_and_store(Stats st, double* out) {
    __shared__ Stats sh[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Stats o = stats_shfl_xor(st, off);
        stats_merge(st, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = st;
    __syncthreads();
    if (threadIdx.x == 0) {
        Stats t = sh[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) stats_merge(t, sh[w]);
        out[0] = t.count;
        out[1] = t.sum_r;
        out[2] = t.m_w;
        out[3] = t.s_w;
        out[4] = t.s_wr;
        out[5] = t.m_e;
        out[6] = t.s_e;
        out[7] = t.m_b;
        out[8] = t.s_b;
    }
}

__global__ void __launch_bounds__(256) tfep_reduce_partial_kernel(const float* __restrict__ uB,
                                                                  const float* __restrict__ ldj,
                                                                  const float* __restrict__ uA,
                                                                  const float* __restrict__ lw,
                                                                  const float* __restrict__ bias, double inv_kT,
                                                                  int ignore_nan, int N, double* __restrict__ partial) {
    Stats st = stats_identity();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        // float32 arithmetic for r, like the reference tensors (loss.py:125-129)
        float rf = uB[i];
        if (ldj) rf = rf - ldj[i];
        if (uA) rf = rf - uA[i];
        const double r = (double)rf;
        const bool isn = isnan(rf);
        if (!(ignore_nan && isn)) {
            st.count += 1.0;
            st.sum_r += r;
        }
        if (lw) {
            // softmax over the whole batch (loss.py:133); NaN r skipped by nansum when ignore_nan
            const double v = (double)lw[i];
            if (ignore_nan && isn)
                online_add(st.m_w, st.s_w, v);
            else
                online_add2(st.m_w, st.s_w, st.s_wr, v, r);
        }
        double e = -r * inv_kT;
        if (bias) {
            const double bb = (double)bias[i] * inv_kT;
            online_add(st.m_b, st.s_b, bb);
            e += bb;
        }
        if (isn) {
            st.m_e = NAN;      // logsumexp propagates NaN (estimator.py:86)
        } else if (!isnan(st.m_e)) {
            online_add(st.m_e, st.s_e, e);
        }
    }
    block_reduce_and_store(st, partial + (int64_t)blockIdx.x * 9);
}

__global__ void __launch_bounds__(256) tfep_reduce_final_kernel(const double* __restrict__ partial, int n_partial,
                                                                double* __restrict__ out) {
    Stats st = stats_identity();
    for (int i = threadIdx.x; i < n_partial; i += blockDim.x) {
        Stats o;
        const double* p = partial + (int64_t)i * 9;
        o.count = p[0];
        o.sum_r = p[1];
        o.m_w = p[2];
        o.s_w = p[3];
        o.s_wr = p[4];
        o.m_e = p[5];
        o.s_e = p[6];
        o.m_b = p[7];
        o.s_b = p[8];
        stats_merge(st, o);
    }
    block_reduce_and_store(st, out);
}


// ------------------------------------------------------------------------------------------
// Bootstrap of the FEP estimator (reference analysis/bootstrap.py:185-262 with statistic = fep_estimator):
// one workgroup per resample; the resampled data are never materialised.
//   standard:  a_j = -work[idx[r][j]] / kT (+ bias[idx[r][j]] / kT),
//              dF_r = -kT (logsumexp_j a_j - log S)                     (no bias)
//              dF_r = -kT (logsumexp_j a_j - logsumexp_j bias_j / kT)   (bias: log_softmax of the resampled biases)
//   Bayesian:  idx == NULL, weights (R, S) >= 0 summing to 1:  dF_r = -kT logsumexp_j (-work[j]/kT + log weights[r][j])
// Online (max, rescaled sum) in fp64 per thread, merged across the workgroup.
// ------------------------------------------------------------------------------------------
struct MaxSum {
    double m, s;
};
__device__ inline void ms_add(MaxSum& a, double v) {
    if (v == -INFINITY) return;
    if (v <= a.m) {
        a.s += exp(v - a.m);
    } else {
        a.s = a.s * exp(a.m - v) + 1.0;
        a.m = v;
    }
}
__device__ inline MaxSum ms_merge(MaxSum a, MaxSum b) {
    if (b.s == 0.0) return a;
    if (a.s == 0.0) return b;
    MaxSum r;
    r.m = fmax(a.m, b.m);
    r.s = a.s * exp(a.m - r.m) + b.s * exp(b.m - r.m);
    return r;
}
__device__ inline MaxSum ms_block_reduce(MaxSum v, MaxSum* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        MaxSum o;
        o.m = __shfl_xor(v.m, off, 64);
        o.s = __shfl_xor(v.s, off, 64);
        v = ms_merge(v, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    MaxSum r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = ms_merge(r, sh[w]);
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(256) bootstrap_fep_kernel(const float* __restrict__ work, const float* __restrict__ bias,
                                                            const int64_t* __restrict__ idx, const float* __restrict__ weights,
                                                            int64_t n_data, int64_t S, double inv_kT, double kT,
                                                            double* __restrict__ out) {
    __shared__ MaxSum sh[4];
    const int64_t r = blockIdx.x;
    MaxSum a = {-INFINITY, 0.0}, b = {-INFINITY, 0.0};
    for (int64_t j = threadIdx.x; j < S; j += blockDim.x) {
        const int64_t i = idx ? idx[r * S + j] : j;
        if (i < 0 || i >= n_data) continue;                       // never for torch.randint(0, n_data) indices
        double v = -(double)work[i] * inv_kT;
        if (weights) v += log((double)weights[r * S + j]);
        if (bias) {
            const double bb = (double)bias[i] * inv_kT;
            v += bb;
            ms_add(b, bb);
        }
        ms_add(a, v);
    }
    a = ms_block_reduce(a, sh);
    if (bias) b = ms_block_reduce(b, sh);
    if (threadIdx.x == 0) {
        double lse = a.m + log(a.s);
        if (bias)
            lse -= b.m + log(b.s);
        else if (!weights)
            lse -= log((double)S);
        out[r] = -kT * lse;
    }
}

}  // namespace tfep

using namespace tfep;

extern "C" {

int tfep_tfep_reduce_workspace_doubles(int N) {
    int blocks = (N + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return blocks * 9;
}

int tfep_tfep_reduce(const float* target_potentials, const float* log_det_J, const float* ref_potentials,
                     const float* log_weights, const float* bias, float kT, int ignore_nan, int N,
                     double* workspace, double* out, void* stream) {
    TFEP_REQUIRE(target_potentials && out && workspace, "tfep_reduce: NULL pointer");
    TFEP_REQUIRE(N >= 0, "tfep_reduce: negative N");
    TFEP_REQUIRE(kT > 0.f, "tfep_reduce: kT must be positive");
    int blocks = (N + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipStream_t s = (hipStream_t)stream;
    tfep_reduce_partial_kernel<<<blocks, 256, 0, s>>>(target_potentials, log_det_J, ref_potentials, log_weights, bias,
                                                      1.0 / (double)kT, ignore_nan, N, workspace);
    tfep_reduce_final_kernel<<<1, 256, 0, s>>>(workspace, blocks, out);
    return check_launch("tfep_reduce");
}

int tfep_bootstrap_fep(const float* work, const float* bias, const int64_t* indices, const float* weights,
                       int64_t n_data, int64_t n_resamples, int64_t sample_size, float kT, double* out, void* stream) {
    TFEP_REQUIRE(n_resamples >= 0 && sample_size >= 0 && n_data >= 0, "bootstrap_fep: negative size");
    if (n_resamples == 0) return TFEP_OK;
    TFEP_REQUIRE(work && out, "bootstrap_fep: NULL pointer");
    TFEP_REQUIRE(kT > 0.f, "bootstrap_fep: kT must be positive");
    TFEP_REQUIRE(indices || sample_size <= n_data, "bootstrap_fep: without indices sample_size must be <= n_data");
    TFEP_REQUIRE(!(bias && weights), "bootstrap_fep: Bayesian weights are not supported with biased data");
    TFEP_REQUIRE(n_resamples <= 0x7fffffffLL, "bootstrap_fep: too many resamples");
    bootstrap_fep_kernel<<<(unsigned)n_resamples, 256, 0, (hipStream_t)stream>>>(work, bias, indices, weights, n_data, sample_size,
                                                                              1.0 / (double)kT, (double)kT, out);
    return check_launch("bootstrap_fep_kernel");
}

}  // extern "C"
