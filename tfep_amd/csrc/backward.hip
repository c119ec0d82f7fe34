// Backward (training) kernels of the flow path: transformer gradients, weight-norm gradient,
// layout helpers for the backward GEMMs.  Reference semantics: the reference relies on eager
// autograd through spline.py / affine.py / mafembed.py and on MaskedLinearFunc.backward
// (masked.py:279-302) + the weight-norm gradient hooks (masked.py:401-402, :429).
//
// Launch shape as the forward transformer kernels: one wavefront per sample row, lanes walk the
// features (coalesced per parameter row); all derivative arithmetic in fp64.
#include "common.h"
#include "spline.h"
#include "fp64_fast.h"

namespace tfep {

constexpr int ROWS_PER_BLOCK_B = 4;
static inline unsigned row_blocks_b(int B) { return (unsigned)((B + ROWS_PER_BLOCK_B - 1) / ROWS_PER_BLOCK_B); }

// ---------------------------------------------------------------- transpose (LDS tiled, 32 x 32)
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ in, int64_t ld_in, int R, int C,
                                                        float* __restrict__ out, int64_t ld_out) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        tile[ty + i][tx] = (r < R && c < C) ? in[(int64_t)r * ld_in + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;                  // out[c][r]
        if (c < C && r < R) out[(int64_t)c * ld_out + r] = tile[tx][ty + i];
    }
}

// out[c] (+)= sum_r in[r, c]   (bias gradient, masked.py:299-300)
// A workgroup owns 64 columns; its 16 waves each sum a contiguous sixteenth of the rows (a wave reads 256 contiguous
// bytes per row) with 8 independent fp64 partial sums per thread, and the slices are added in a fixed order through
// LDS: four times the loads in flight per column and four times the workgroups of a thread-per-column pass (which left
// most CUs idle on the (1024 x 800) .. (1024 x 3200) gradients of a small flow: 48 us per call whatever the size).
constexpr int CS_COLS = 64, CS_SLICES = 16;          // (16 row slices: 4 left the 4.9 GB gradient pass at 2.3 TB/s)

// the slices' partial sums in a fixed pairwise order (both kernels below: the same bits)
__device__ __forceinline__ double slice_tree_sum(const double (&part)[CS_SLICES][CS_COLS], int cx) {
    double t[CS_SLICES];
#pragma unroll
    for (int q = 0; q < CS_SLICES; ++q) t[q] = part[q][cx];
#pragma unroll
    for (int w = 1; w < CS_SLICES; w *= 2)
#pragma unroll
        for (int q = 0; q + w < CS_SLICES; q += 2 * w) t[q] += t[q + w];
    return t[0];
}

__global__ void __launch_bounds__(CS_COLS * CS_SLICES) colsum_kernel(const float* __restrict__ in, int64_t ld, int R, int C,
                                                                    float* __restrict__ out, int accumulate) {
    __shared__ double part[CS_SLICES][CS_COLS];
    const int cx = threadIdx.x & (CS_COLS - 1), sl = threadIdx.x / CS_COLS;
    const int c = blockIdx.x * CS_COLS + cx;
    const int per = (R + CS_SLICES - 1) / CS_SLICES;
    const int r1 = min(R, (sl + 1) * per);
    // 8 independent partial sums: 8 row loads in flight per thread instead of a load-add dependency chain
    double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (c < C) {
        int r = sl * per;
        for (; r + 8 <= r1; r += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] += (double)in[(int64_t)(r + j) * ld + c];
        }
        for (; r < r1; ++r) p[0] += (double)in[(int64_t)r * ld + c];
    }
    part[sl][cx] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    __syncthreads();
    if (sl == 0 && c < C) {
        const double s = slice_tree_sum(part, cx);
        out[c] = accumulate ? (float)((double)out[c] + s) : (float)s;
    }
}

// The same pass also returning max_r |in[r, c]| (the row scale of the transposed split operand of grad_weight).
__global__ void __launch_bounds__(CS_COLS * CS_SLICES) colsum_absmax_kernel(const float* __restrict__ in, int64_t ld, int R, int C,
                                                                           float* __restrict__ out, int accumulate,
                                                                           float* __restrict__ amax) {
    __shared__ double part[CS_SLICES][CS_COLS];
    __shared__ float pmax[CS_SLICES][CS_COLS];
    const int cx = threadIdx.x & (CS_COLS - 1), sl = threadIdx.x / CS_COLS;
    const int c = blockIdx.x * CS_COLS + cx;
    const int per = (R + CS_SLICES - 1) / CS_SLICES;
    const int r1 = min(R, (sl + 1) * per);
    double p[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        int r = sl * per;
        for (; r + 8 <= r1; r += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = in[(int64_t)(r + j) * ld + c];
                p[j] += (double)v;
                m[j] = fmaxf(m[j], fabsf(v));
            }
        }
        for (; r < r1; ++r) {
            const float v = in[(int64_t)r * ld + c];
            p[0] += (double)v;
            m[0] = fmaxf(m[0], fabsf(v));
        }
    }
    part[sl][cx] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));       // colsum_kernel's order
    pmax[sl][cx] = fmaxf(fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3])), fmaxf(fmaxf(m[4], m[5]), fmaxf(m[6], m[7])));
    __syncthreads();
    if (sl == 0 && c < C) {
        const double s = slice_tree_sum(part, cx);
        out[c] = accumulate ? (float)((double)out[c] + s) : (float)s;
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < CS_SLICES; ++q) mx = fmaxf(mx, pmax[q][cx]);
        amax[c] = mx;
    }
}

// ---------------------------------------------------------------- affine backward (affine.py:321-323)
__global__ void __launch_bounds__(256) affine_backward_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ params, tfep_param_layout L,
                                                              const float* __restrict__ gy, int64_t ldgy,
                                                              const float* __restrict__ gldj,
                                                              float* __restrict__ gparams, tfep_param_layout GL,
                                                              float* __restrict__ gx, int64_t ldgx, int B, int D) {
    const int b = blockIdx.x * ROWS_PER_BLOCK_B + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const float gl = gldj ? gldj[b] : 0.f;
    for (int f = lane; f < D; f += 64) {
        const float ls = params[(int64_t)b * L.ld + L.stride_p + f * L.stride_f];
        const float g = gy[(int64_t)b * ldgy + f];
        const float e = expf(ls);
        const float xv = x[(int64_t)b * ldx + f];
        gparams[(int64_t)b * GL.ld + f * GL.stride_f] = g;                                   // d/d shift
        gparams[(int64_t)b * GL.ld + GL.stride_p + f * GL.stride_f] = g * xv * e + gl;       // d/d log_scale
        gx[(int64_t)b * ldgx + f] = g * e;
    }
}

// ---------------------------------------------------------------- RQ spline backward
struct SplineArgsB {
    const float *x0, *xf, *y0, *yf;
    SplineFlags f;
    int P;
};

// sigmoid on the softmax's exponential (both signs through exp of a non-positive argument) and the hardware-seeded
// reciprocal: ~1e-11 relative, a fifth of the instructions of exp() + an IEEE division
__device__ inline double sigmoid_d(double z) {
    if (z > 20.0) return 1.0;
    const double e = exp_nonpos(-fabs(z));
    const double r = fast_rcp64(1.0 + e);
    return z >= 0.0 ? r : e * r;
}

// Reverse mode through rq_spline_element (spline.h), FORWARD direction, all variants.
template <int KMAX>
__device__ inline void rq_spline_backward(const float (&w)[KMAX], const float (&h)[KMAX], const float (&sraw)[KMAX + 1],
                                          float last, float last2, const SplineFlags& f, float x0f, float xff,
                                          float y0f, float yff, float vin, double gy, double gl, double (&guw)[KMAX],
                                          double (&guh)[KMAX], double (&gus)[KMAX + 1], double* glast, double* glast2,
                                          double* gxin) {
    const int K = f.K;
    const double mb = (double)f.min_bin;
    // domain (spline.py:384-410), as in rq_spline_element
    double x0 = x0f, y0 = y0f;
    double W = (double)xff - (double)x0f - K * mb, H = (double)yff - (double)y0f - K * mb;
    const bool learn = f.learn_lower || f.learn_upper;
    if (learn) {
        const double scale = exp((double)last);
        W *= scale;
        H *= scale;
        if (f.learn_lower && f.learn_upper) {
            x0 += (double)last2;
            y0 += (double)last2;
        } else if (f.learn_lower) {
            x0 = (double)xff - W - K * mb;
            y0 = (double)yff - H - K * mb;
        }
    }
    double gx0 = 0.0, gy0 = 0.0, gWt = 0.0, gHt = 0.0;   // grads w.r.t. x0', y0' and extra direct terms of W, H
    double v = vin;
    if (f.circular) v = py_mod(v - x0 + (double)last, (double)xff - x0) + x0;

    float mw = -INFINITY, mh = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) {
            mw = fmaxf(mw, w[k]);
            mh = fmaxf(mh, h[k]);
        }
    double pw[KMAX], ph[KMAX];
    double sw = 0.0, sh = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        pw[k] = 0.0;
        ph[k] = 0.0;
        if (k < K) {
            pw[k] = exp_nonpos((double)w[k] - (double)mw);
            ph[k] = exp_nonpos((double)h[k] - (double)mh);
            sw += pw[k];
            sh += ph[k];
        }
    }
    {
        const double isw = fast_rcp64(sw), ish = fast_rcp64(sh);       // (sums >= 1: the largest term is exp(0))
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            pw[k] *= isw;
            ph[k] *= ish;
        }
    }

    double kx = x0, ky = y0, bw = 0.0, bh = 0.0;
    int kbin = -1;
    bool found = false;
    const bool lower_tail = !(v > x0);
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K && !found) {
            const double wk = pw[k] * W + mb, hk = ph[k] * H + mb;
            if (v > kx + wk) {
                kx += wk;
                ky += hk;
            } else {
                found = true;
                bw = wk;
                bh = hk;
                kbin = k;
            }
        }

    double gw[KMAX], gh[KMAX], gd[KMAX + 1];      // grads wrt widths, heights, slopes (values)
    // the (at most two) knots whose slope enters the element, and the derivative of their softplus: only these raw slopes
    // get a gradient, so only their sigmoids are evaluated (there were K + 1 of them, each an exp and a division)
    int ja = -1, jb2 = -1;
    double sga = 0.0, sgb = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        gw[k] = 0.0;
        gh[k] = 0.0;
        gd[k] = 0.0;
    }
    gd[KMAX] = 0.0;
    double gv;

    auto slope = [&](float r) { return (double)(softplus_f(r + f.slope_offset) + f.min_slope); };
    if (lower_tail || !found) {
        // y = y_b + d (v - x_b), ld = log d; the boundary knot is a constant of the softmax outputs
        float rs = sraw[0];
        int jb = 0;
        if (!lower_tail) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k == K - 1) rs = sraw[k + 1];
            jb = K;
        }
        const double d = slope(rs);
        const double bx = lower_tail ? x0 : kx;
        const double gdb = gy * (v - bx) + gl * fast_rcp64(d);
        ja = jb;
        sga = sigmoid_d((double)rs + (double)f.slope_offset);
        // boundary knot: (x0', y0') below, (x0' + W + K mb, y0' + H + K mb) above
        gx0 = -gy * d;
        gy0 = gy;
        if (!lower_tail) {
            gWt = -gy * d;
            gHt = gy;
        }
#pragma unroll
        for (int j = 0; j <= KMAX; ++j)
            if (j == jb) gd[j] = gdb;
        gv = gy * d;
    } else {
        float rs0 = sraw[0], rs1 = sraw[0];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k == kbin) {
                rs0 = sraw[k];
                rs1 = sraw[k + 1];
            }
        const double dk = slope(rs0), dk1 = slope(rs1);
        ja = kbin;
        jb2 = kbin + 1;
        sga = sigmoid_d((double)rs0 + (double)f.slope_offset);
        sgb = sigmoid_d((double)rs1 + (double)f.slope_offset);
        // four reciprocals (hardware seed + Newton, fp64_fast.h) instead of sixteen IEEE divisions
        const double ibw = fast_rcp64(bw);
        const double s = bh * ibw, t = dk1 + dk - 2.0 * s;
        const double eps = (v - kx) * ibw, om = 1.0 - eps, e1 = eps * om;
        const double A = s * eps * eps + dk * e1;
        const double Dn = s + t * e1;
        const double Q = dk1 * eps * eps + 2.0 * s * e1 + dk * om * om;
        const double iDn = fast_rcp64(Dn), iQ = fast_rcp64(Q), is = fast_rcp64(s);
        const double dy_dA = bh * iDn, dy_dDn = -bh * A * iDn * iDn;
        const double gs = gy * (dy_dA * eps * eps + dy_dDn * (1.0 - 2.0 * e1)) +
                          gl * (2.0 * is + 2.0 * e1 * iQ - 2.0 * (1.0 - 2.0 * e1) * iDn);
        const double geps = gy * (dy_dA * (2.0 * s * eps + dk * (1.0 - 2.0 * eps)) + dy_dDn * t * (1.0 - 2.0 * eps)) +
                            gl * ((2.0 * dk1 * eps + 2.0 * s * (1.0 - 2.0 * eps) - 2.0 * dk * om) * iQ -
                                  2.0 * t * (1.0 - 2.0 * eps) * iDn);
        const double gdk = gy * (dy_dA * e1 + dy_dDn * e1) + gl * (om * om * iQ - 2.0 * e1 * iDn);
        const double gdk1 = gy * (dy_dDn * e1) + gl * (eps * eps * iQ - 2.0 * e1 * iDn);
        const double gh_bin = gy * A * iDn + gs * ibw;
        const double gw_bin = -gs * s * ibw - geps * eps * ibw;
        const double gxk = -geps * ibw;
        gv = geps * ibw;
        gx0 = gxk;
        gy0 = gy;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < kbin) {
                gw[k] = gxk;
                gh[k] = gy;
            } else if (k == kbin) {
                gw[k] = gw_bin;
                gh[k] = gh_bin;
            }
        }
#pragma unroll
        for (int j = 0; j <= KMAX; ++j) {
            if (j == kbin) gd[j] = gdk;
            if (j == kbin + 1) gd[j] = gdk1;
        }
    }

    // softmax backward: w_k = p_k W + mb
    double dotw = 0.0, doth = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        dotw += pw[k] * gw[k];
        doth += ph[k] * gh[k];
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        guw[k] = pw[k] * W * (gw[k] - dotw);
        guh[k] = ph[k] * H * (gh[k] - doth);
    }
    // softplus backward
#pragma unroll
    for (int j = 0; j <= KMAX; ++j) gus[j] = j == ja ? gd[j] * sga : (j == jb2 ? gd[j] * sgb : 0.0);
    *glast = f.circular ? gv : 0.0;
    *glast2 = 0.0;
    if (learn) {
        // W = W0 e^last, H = H0 e^last; w_k = p_k W + mb
        double gW = dotw + gWt, gH = doth + gHt;
        if (f.learn_lower && f.learn_upper) {
            *glast2 = gx0 + gy0;
        } else if (f.learn_lower) {      // x0' = xf - W - K mb, y0' = yf - H - K mb
            gW -= gx0;
            gH -= gy0;
        }
        *glast = gW * W + gH * H;
    }
    *gxin = gv;
}

template <int KMAX>
__global__ void __launch_bounds__(256, 3) spline_backward_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ params, tfep_param_layout L,
                                                              SplineArgsB a, const float* __restrict__ gy, int64_t ldgy,
                                                              const float* __restrict__ gldj,
                                                              float* __restrict__ gparams, tfep_param_layout GL,
                                                              float* __restrict__ gx, int64_t ldgx, int B, int D) {
    // Feature-major parameters (stride_p = 1, stride_f = P: the training step's layout): a lane's P values sit 4 P bytes from
    // its neighbour's, so element loads / stores touch 64 different cache lines per instruction.  The 64 P contiguous floats of
    // a batch of 64 features go through LDS instead: whole 256-byte rows in and out, the lanes read / write their own P
    // values at an odd pitch (P + 1 when P is even: conflict free).
    extern __shared__ float sb_stage[];
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.x * ROWS_PER_BLOCK_B + wave;
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int K = a.f.K;
    const int P = a.P;
    const bool staged = KMAX <= 16 && L.stride_p == 1 && L.stride_f == P && GL.stride_p == 1 && GL.stride_f == P && P <= 3 * KMAX + 3;
    const int pitch = P | 1;
    float* stage = sb_stage + wave * 64 * (3 * KMAX + 4);
    const double gl = gldj ? (double)gldj[b] : 0.0;
    for (int f0 = 0; f0 < D; f0 += 64) {
        const int f = f0 + lane;
        const int nf = min(64, D - f0);
        const bool live = f < D;
        const float* pf = params + (int64_t)b * L.ld + f * L.stride_f;
        float* gp = gparams + (int64_t)b * GL.ld + f * GL.stride_f;
        int sp = L.stride_p, gsp = GL.stride_p;                     // strides of this lane's own parameter reads / writes
        if (staged) {
            const float* src = params + (int64_t)b * L.ld + (int64_t)f0 * P;
            for (int i = lane; i < nf * P; i += 64) stage[(i / P) * pitch + i % P] = src[i];
            __builtin_amdgcn_wave_barrier();
            pf = stage + lane * pitch;
            gp = stage + lane * pitch;
            sp = gsp = 1;
        }
        if (live) {
            float w[KMAX], h[KMAX], sraw[KMAX + 1];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                w[k] = 0.f;
                h[k] = 0.f;
                if (k < K) {
                    w[k] = pf[k * sp];
                    h[k] = pf[(K + k) * sp];
                }
            }
#pragma unroll
            for (int j = 0; j <= KMAX; ++j) {
                sraw[j] = 0.f;
                if (j <= K) {
                    const int pi = spline_slope_param(j, K, a.f.circular, a.f.identity);
                    if (pi >= 0) sraw[j] = pf[pi * sp];
                }
            }
            const bool has_last = a.f.circular || a.f.learn_lower || a.f.learn_upper;
            const bool has_last2 = a.f.learn_lower && a.f.learn_upper;
            const float last = has_last ? pf[(P - 1) * sp] : 0.f;
            const float last2 = has_last2 ? pf[(P - 2) * sp] : 0.f;
            double guw[KMAX], guh[KMAX], gus[KMAX + 1], glast, glast2, gxin;
            rq_spline_backward<KMAX>(w, h, sraw, last, last2, a.f, a.x0[f], a.xf[f], a.y0[f], a.yf[f],
                                     x[(int64_t)b * ldx + f], (double)gy[(int64_t)b * ldgy + f], gl, guw, guh, gus, &glast,
                                     &glast2, &gxin);
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    gp[k * gsp] = (float)guw[k];
                    gp[(K + k) * gsp] = (float)guh[k];
                }
            // slopes: knot K of a circular spline shares the parameter of knot 0; identity boundary
            // slopes have no parameter.
            double g0 = gus[0];
#pragma unroll
            for (int j = 0; j <= KMAX; ++j)
                if (j == K && a.f.circular && !a.f.identity) g0 += gus[j];
#pragma unroll
            for (int j = 0; j <= KMAX; ++j) {
                if (j > K) continue;
                if (a.f.circular && !a.f.identity && j == K) continue;
                const int pi = spline_slope_param(j, K, a.f.circular, a.f.identity);
                if (pi >= 0) gp[pi * gsp] = (float)(j == 0 ? g0 : gus[j]);
            }
            if (has_last) gp[(P - 1) * gsp] = (float)glast;
            if (has_last2) gp[(P - 2) * gsp] = (float)glast2;
            gx[(int64_t)b * ldgx + f] = (float)gxin;
        }
        if (staged) {
            __builtin_amdgcn_wave_barrier();
            float* dst = gparams + (int64_t)b * GL.ld + (int64_t)f0 * P;
            for (int i = lane; i < nf * P; i += 64) dst[i] = stage[(i / P) * pitch + i % P];
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------- weight-norm backward
// W[o,i] = M[o,i] v[o,i] g[o]/n_o, n_o = ||v[o,:]||.  Given gW (packed coordinates):
//   gg[o]   = sum_i gW M v / n
//   gv[o,i] = M (g/n) gW - g v / n^3 * sum_j gW M v
// with the reference's hooks: gv = 0 where M == 0, gg = 0 for fully-masked rows (masked.py:401-402, :429).
// Without weight norm (g == NULL): gweight = gW o M (masked.py:293-297).
__global__ void __launch_bounds__(256) weight_norm_backward_kernel(const float* __restrict__ gw_packed, int64_t ldw,
                                                                   const float* __restrict__ v,
                                                                   const float* __restrict__ g,
                                                                   const float* __restrict__ mask, int N, int K,
                                                                   const int32_t* __restrict__ row_of_out,
                                                                   const int32_t* __restrict__ col_of_in,
                                                                   float* __restrict__ gv, float* __restrict__ gg) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= N) return;
    const int lane = threadIdx.x & 63;
    const float* vr = v + (int64_t)o * K;
    const float* mr = mask ? mask + (int64_t)o * K : nullptr;
    const float* gr = gw_packed + (int64_t)(row_of_out ? row_of_out[o] : o) * ldw;
    float* gvr = gv + (int64_t)o * K;
    if (!g) {
        for (int i = lane; i < K; i += 64) {
            const float gwv = gr[col_of_in ? col_of_in[i] : i];
            gvr[i] = (mr && mr[i] == 0.f) ? 0.f : gwv;
        }
        return;
    }
    double ss = 0.0, dot = 0.0, msum = 0.0;
    for (int i = lane; i < K; i += 64) {
        const double vv = vr[i];
        ss += vv * vv;
        const bool live = !mr || mr[i] != 0.f;
        if (live) {
            dot += (double)gr[col_of_in ? col_of_in[i] : i] * vv;
            msum += 1.0;
        }
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    msum = wave_sum(msum);
    const double n = sqrt(ss);
    const bool dead = msum == 0.0 || n == 0.0;
    const double gg_o = dead ? 0.0 : dot / n;
    const double go = (double)g[o];
    for (int i = lane; i < K; i += 64) {
        const bool live = !mr || mr[i] != 0.f;
        double out = 0.0;
        if (live && !dead) out = go / n * (double)gr[col_of_in ? col_of_in[i] : i] - go * (double)vr[i] * dot / (n * n * n);
        gvr[i] = (float)out;
    }
    if (lane == 0) gg[o] = (float)gg_o;
}

// The same for a layer whose mask rows are prefixes of its packed columns (col_cut, see tfep_masked_weight_prepare_split):
// one workgroup per output row, v staged in LDS (16-byte loads), the packed gradient read over its live prefix only,
// coalesced; the permuted results are assembled in LDS, in place of v, and leave as whole rows.  Against the kernel above
// (v, mask and the 4-byte gather of gW twice each): 11 instead of ~31 GB for the cfg2 output layer.
constexpr int WNB_THREADS = 512;
__global__ void __launch_bounds__(WNB_THREADS) weight_norm_backward_prefix_kernel(const float* __restrict__ gw_packed, int64_t ldw,
                                                                                  const float* __restrict__ v, const float* __restrict__ g,
                                                                                  int N, int K, const int32_t* __restrict__ row_of_out,
                                                                                  const int32_t* __restrict__ in_of_col,
                                                                                  const int32_t* __restrict__ col_cut,
                                                                                  float* __restrict__ gv, float* __restrict__ gg) {
    extern __shared__ float srow[];               // v[o, :], then the row of results
    __shared__ double red[2][WNB_THREADS / 64];
    const int o = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* vr = v + (int64_t)o * K;
    {
        const int head = min(K, (int)(((16u - (uint32_t)((uintptr_t)vr & 15u)) & 15u) >> 2));
        const int n4 = (K - head) >> 2;
        const float4* v4 = reinterpret_cast<const float4*>(vr + head);
        if (tid < head) srow[tid] = vr[tid];
        for (int i0 = tid; i0 < n4; i0 += 8 * WNB_THREADS) {
            float4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * WNB_THREADS;
                q[u] = i < n4 ? v4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * WNB_THREADS;
                if (i < n4) {
                    float* d = srow + head + 4 * i;
                    d[0] = q[u].x; d[1] = q[u].y; d[2] = q[u].z; d[3] = q[u].w;
                }
            }
        }
        for (int i = head + 4 * n4 + tid; i < K; i += WNB_THREADS) srow[i] = vr[i];
    }
    const int cut = min(col_cut[o], K);
    const float* gr = gw_packed + (int64_t)(row_of_out ? row_of_out[o] : o) * ldw;
    float* gvr = gv + (int64_t)o * K;
    // The live prefix of the packed gradient row and its permutation entries: 4 columns per thread and step, every step's
    // 16-byte loads issued up front (8 steps cover 16 384 columns) -- they were fetched element by element, twice, each
    // pass a chain of dependent L2 round trips (4.1 ms for the cfg2 output layer at 2.7 TB/s).
    constexpr int WNB_U = 8;
    // the live prefix of a packed gradient row is held in registers: WNB_U 16-byte loads per thread.  The host requires
    // in_features * 4 <= 64 KB (the row staged in LDS), i.e. at most 16 384 columns -- tie the two together
    static_assert(WNB_U * WNB_THREADS * 4 >= 16384, "weight_norm_backward_prefix: the register tile must cover the largest staged row");
    const bool vec = ((uintptr_t)gr & 15u) == 0 && (!in_of_col || ((uintptr_t)in_of_col & 15u) == 0);
    float4 gq[WNB_U];
    int4 iq[WNB_U];
#pragma unroll
    for (int u = 0; u < WNB_U; ++u) {
        const int c4 = (tid + u * WNB_THREADS) * 4;
        gq[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        iq[u] = make_int4(c4, c4 + 1, c4 + 2, c4 + 3);
        if (c4 < cut) {
            if (vec && c4 + 4 <= K) {
                gq[u] = *reinterpret_cast<const float4*>(gr + c4);
                if (in_of_col) iq[u] = *reinterpret_cast<const int4*>(in_of_col + c4);
            } else {
                float* ge = reinterpret_cast<float*>(&gq[u]);
                int* ie = reinterpret_cast<int*>(&iq[u]);
                for (int j = 0; j < 4; ++j)
                    if (c4 + j < cut) {
                        ge[j] = gr[c4 + j];
                        if (in_of_col) ie[j] = in_of_col[c4 + j];
                    }
            }
        }
    }
    __syncthreads();                              // v[o, :] is in LDS
    double ss = 0.0, dot = 0.0;
    if (g) {
        for (int i = tid; i < K; i += WNB_THREADS) ss += (double)srow[i] * (double)srow[i];
#pragma unroll
        for (int u = 0; u < WNB_U; ++u) {
            const int c4 = (tid + u * WNB_THREADS) * 4;
            const float ge[4] = {gq[u].x, gq[u].y, gq[u].z, gq[u].w};
            const int ie[4] = {iq[u].x, iq[u].y, iq[u].z, iq[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c4 + j < cut) dot += (double)ge[j] * (double)srow[ie[j]];
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        if (lane == 0) { red[0][wave] = ss; red[1][wave] = dot; }
    }
    __syncthreads();
    double n = 1.0, go = 0.0, gg_o = 0.0;
    bool dead = false;
    if (g) {
        ss = 0.0; dot = 0.0;
#pragma unroll
        for (int w = 0; w < WNB_THREADS / 64; ++w) { ss += red[0][w]; dot += red[1][w]; }     // fixed order: every thread the same bits
        n = sqrt(ss);
        dead = cut == 0 || n == 0.0;
        gg_o = dead ? 0.0 : dot / n;
        go = (double)g[o];
    }
    // the live results in registers (each reads its own v), then the row is cleared and they are scattered into it: the
    // masked entries need no permutation lookup
    const double a = (g && !dead) ? go / n : 0.0, b = (g && !dead) ? go * dot / (n * n * n) : 0.0;
    float4 res[WNB_U];
#pragma unroll
    for (int u = 0; u < WNB_U; ++u) {
        const int c4 = (tid + u * WNB_THREADS) * 4;
        const float ge[4] = {gq[u].x, gq[u].y, gq[u].z, gq[u].w};
        const int ie[4] = {iq[u].x, iq[u].y, iq[u].z, iq[u].w};
        float re[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c4 + j < cut) re[j] = g ? (float)(a * (double)ge[j] - b * (double)srow[ie[j]]) : ge[j];
        res[u] = make_float4(re[0], re[1], re[2], re[3]);
    }
    __syncthreads();                              // every read of v is done
    for (int i = tid; i < K; i += WNB_THREADS) srow[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < WNB_U; ++u) {
        const int c4 = (tid + u * WNB_THREADS) * 4;
        const float re[4] = {res[u].x, res[u].y, res[u].z, res[u].w};
        const int ie[4] = {iq[u].x, iq[u].y, iq[u].z, iq[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c4 + j < cut) srow[ie[j]] = re[j];
    }
    __syncthreads();
    {
        // the row of results out in 16-byte stores (head / body / tail like the load of v)
        const int head = min(K, (int)(((16u - (uint32_t)((uintptr_t)gvr & 15u)) & 15u) >> 2));
        const int n4 = (K - head) >> 2;
        float4* o4 = reinterpret_cast<float4*>(gvr + head);
        if (tid < head) gvr[tid] = srow[tid];
        for (int i = tid; i < n4; i += WNB_THREADS) {
            const float* sp = srow + head + 4 * i;
            o4[i] = make_float4(sp[0], sp[1], sp[2], sp[3]);
        }
        for (int i = head + 4 * n4 + tid; i < K; i += WNB_THREADS) gvr[i] = srow[i];
    }
    if (g && tid == 0) gg[o] = (float)gg_o;
}

// ---------------------------------------------------------------- periodic embedding backward
// out = [x_non..., cos t, sin t, ...], t = (x - lower) * scale  ->  gx[p] = (-sin t g_cos + cos t g_sin) * scale
__global__ void __launch_bounds__(256) periodic_embedding_backward_kernel(const float* __restrict__ x, int64_t ldx,
                                                                          const int32_t* __restrict__ pidx, int n_per,
                                                                          const int32_t* __restrict__ nidx, int n_non,
                                                                          float lower, float scale,
                                                                          const float* __restrict__ gout, int64_t ldg,
                                                                          float* __restrict__ gx, int64_t ldgx, int B) {
    const int n_src = n_non + n_per;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * n_src) return;
    const int b = (int)(i / n_src), j = (int)(i % n_src);
    if (j < n_non) {
        gx[(int64_t)b * ldgx + nidx[j]] = gout[(int64_t)b * ldg + j];
    } else {
        const int q = j - n_non;
        const float t = (x[(int64_t)b * ldx + pidx[q]] - lower) * scale;
        float sn, cs;
        sincosf(t, &sn, &cs);
        const float gc = gout[(int64_t)b * ldg + n_non + 2 * q], gs = gout[(int64_t)b * ldg + n_non + 2 * q + 1];
        gx[(int64_t)b * ldgx + pidx[q]] = (-sn * gc + cs * gs) * scale;
    }
}

// ---------------------------------------------------------------- Moebius VJP (moebius.py:374-478)
// Reverse mode through the closed-form forward of moebius_kernel (transformers.hip); one lane per vector.
constexpr int MOEBIUS_MAX_DIM_B = 8;

__global__ void __launch_bounds__(256) moebius_backward_kernel(const float* __restrict__ x, int64_t ldx,
                                                               const float* __restrict__ params, int64_t ldp, int dim,
                                                               float max_radius, int unit_sphere, float sign,
                                                               const float* __restrict__ gy, int64_t ldgy,
                                                               const float* __restrict__ gldj,
                                                               float* __restrict__ gparams, int64_t ldgp,
                                                               float* __restrict__ gx, int64_t ldgx, int B, int D) {
    const int b = blockIdx.x * ROWS_PER_BLOCK_B + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int nvec = D / dim;
    const double gl = gldj ? (double)gldj[b] : 0.0;
    for (int v = lane; v < nvec; v += 64) {
        double xv[MOEBIUS_MAX_DIM_B], wv[MOEBIUS_MAX_DIM_B], uv[MOEBIUS_MAX_DIM_B], dv[MOEBIUS_MAX_DIM_B],
            yv[MOEBIUS_MAX_DIM_B], gyv[MOEBIUS_MAX_DIM_B], xb[MOEBIUS_MAX_DIM_B], ub[MOEBIUS_MAX_DIM_B],
            db[MOEBIUS_MAX_DIM_B];
        double nw2 = 0.0, nx2 = 0.0;
#pragma unroll
        for (int i = 0; i < MOEBIUS_MAX_DIM_B; ++i)
            if (i < dim) {
                xv[i] = (double)x[(int64_t)b * ldx + v * dim + i];
                wv[i] = (double)(sign * params[(int64_t)b * ldp + v * dim + i]);
                gyv[i] = (double)gy[(int64_t)b * ldgy + v * dim + i];
                nw2 += wv[i] * wv[i];
                nx2 += xv[i] * xv[i];
            }
        const double nw = sqrt(nw2), nx = sqrt(nx2);
        const double s = (double)max_radius / (1.0 + nw) * (unit_sphere ? 1.0 : nx);
        double nu2 = 0.0, Dd = 0.0;
#pragma unroll
        for (int i = 0; i < MOEBIUS_MAX_DIM_B; ++i)
            if (i < dim) {
                uv[i] = s * wv[i];
                dv[i] = xv[i] - uv[i];
                nu2 += uv[i] * uv[i];
                Dd += dv[i] * dv[i];
            }
        const double N = (unit_sphere ? 1.0 : nx2) - nu2;
        const double c = N / Dd;
        double xd = 0.0, dy = 0.0, xy = 0.0;
#pragma unroll
        for (int i = 0; i < MOEBIUS_MAX_DIM_B; ++i)
            if (i < dim) {
                yv[i] = c * dv[i] - uv[i];
                xd += xv[i] * dv[i];
                dy += dv[i] * yv[i];
                xy += xv[i] * yv[i];
            }
        // ---- reverse
        double cb, Db = 0.0;
        double ybar[MOEBIUS_MAX_DIM_B];
#pragma unroll
        for (int i = 0; i < MOEBIUS_MAX_DIM_B; ++i) {
            ybar[i] = i < dim ? gyv[i] : 0.0;
            xb[i] = 0.0;
            db[i] = 0.0;
        }
        if (unit_sphere) {
            cb = gl * dim / c;
        } else {
            // ld = (dim-1) log|c| - 2 log|x| + log|q|,  q = x.y - 2 (x.d)(d.y)/D
            const double q = xy - 2.0 * xd * dy / Dd;
            const double qb = gl / q;
#pragma unroll
            for (int i = 0; i < MOEBIUS_MAX_DIM_B; ++i)
                if (i < dim) {
                    ybar[i] += qb * (xv[i] - 2.0 * xd * dv[i] / Dd);
                    xb[i] += qb * (yv[i] - 2.0 * dy * dv[i] / Dd) - gl * 2.0 * xv[i] / nx2;
                    db[i] += qb * (-2.0 / Dd) * (dy * xv[i] + xd * yv[i]);
                }
            Db += qb * 2.0 * xd * dy / (Dd * Dd);
            cb = gl * (dim - 1) / c;
        }
        double yd = 0.0;
#pragma unroll
        for (int i = 0; i < MOEBIUS_MAX_DIM_B; ++i)
            if (i < dim) yd += ybar[i] * dv[i];
        cb += yd;
        const double Nb = cb / Dd;
        Db += -cb * N / (Dd * Dd);
        double sb = 0.0;
#pragma unroll
        for (int i = 0; i < MOEBIUS_MAX_DIM_B; ++i)
            if (i < dim) {
                db[i] += c * ybar[i] + 2.0 * Db * dv[i];
                ub[i] = -ybar[i] - db[i] - 2.0 * Nb * uv[i];
                xb[i] += db[i] + (unit_sphere ? 0.0 : 2.0 * Nb * xv[i]);
                sb += ub[i] * wv[i];
            }
        const double nwb = -sb * s / (1.0 + nw);
        const double nxb = unit_sphere ? 0.0 : sb * s / nx;
#pragma unroll
        for (int i = 0; i < MOEBIUS_MAX_DIM_B; ++i)
            if (i < dim) {
                double wb = s * ub[i];
                if (nw > 0.0) wb += nwb * wv[i] / nw;
                if (!unit_sphere) xb[i] += nxb * xv[i] / nx;
                gparams[(int64_t)b * ldgp + v * dim + i] = (float)(sign * wb);
                gx[(int64_t)b * ldgx + v * dim + i] = (float)xb[i];
            }
    }
}

// dst[b, c] = src[b, c]
__global__ void __launch_bounds__(256) copy_2d_kernel(const float* __restrict__ src, int64_t lds,
                                                      float* __restrict__ dst, int64_t ldd, int B, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C) return;
    const int b = (int)(i / C), c = (int)(i % C);
    dst[(int64_t)b * ldd + c] = src[(int64_t)b * lds + c];
}

// out[b, c] += in[b, c]   (gx = direct + through-the-conditioner)
__global__ void __launch_bounds__(256) add_inplace_kernel(const float* __restrict__ in, int64_t ldi,
                                                          float* __restrict__ out, int64_t ldo, int B, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C) return;
    const int b = (int)(i / C), c = (int)(i % C);
    out[(int64_t)b * ldo + c] += in[(int64_t)b * ldi + c];
}

}  // namespace tfep

using namespace tfep;

extern "C" {

int tfep_transpose(const float* in, int64_t ld_in, int R, int C, float* out, int64_t ld_out, void* stream) {
    TFEP_REQUIRE(R >= 0 && C >= 0 && ld_in >= C && ld_out >= R, "transpose: bad sizes");
    if (R == 0 || C == 0) return TFEP_OK;
    TFEP_REQUIRE(in && out, "transpose: NULL pointer");
    dim3 grid((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32));
    transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(in, ld_in, R, C, out, ld_out);
    return check_launch("transpose_kernel");
}

int tfep_column_sums_absmax(const float* in, int64_t ld, int R, int C, float* out, int accumulate, float* absmax, void* stream) {
    TFEP_REQUIRE(R >= 0 && C >= 0, "column_sums_absmax: bad sizes");
    if (C == 0) return TFEP_OK;
    TFEP_REQUIRE(out && absmax && (in || R == 0), "column_sums_absmax: NULL pointer");
    colsum_absmax_kernel<<<(unsigned)((C + CS_COLS - 1) / CS_COLS), CS_COLS * CS_SLICES, 0, (hipStream_t)stream>>>(in, ld, R, C, out, accumulate, absmax);
    return check_launch("colsum_absmax_kernel");
}

int tfep_column_sums(const float* in, int64_t ld, int R, int C, float* out, int accumulate, void* stream) {
    TFEP_REQUIRE(R >= 0 && C >= 0, "column_sums: bad sizes");
    if (C == 0) return TFEP_OK;
    TFEP_REQUIRE(out && (in || R == 0), "column_sums: NULL pointer");
    colsum_kernel<<<(unsigned)((C + CS_COLS - 1) / CS_COLS), CS_COLS * CS_SLICES, 0, (hipStream_t)stream>>>(in, ld, R, C, out, accumulate);
    return check_launch("colsum_kernel");
}

int tfep_add_inplace(const float* in, int64_t ld_in, float* out, int64_t ld_out, int B, int C, void* stream) {
    const int64_t n = (int64_t)B * C;
    if (n <= 0) return TFEP_OK;
    TFEP_REQUIRE(in && out, "add_inplace: NULL pointer");
    add_inplace_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(in, ld_in, out, ld_out, B, C);
    return check_launch("add_inplace_kernel");
}

int tfep_affine_backward(const float* x, int64_t ldx, const float* params, tfep_param_layout layout, const float* gy,
                         int64_t ldgy, const float* g_log_det_J, float* gparams, tfep_param_layout glayout, float* gx,
                         int64_t ldgx, int B, int D, void* stream) {
    TFEP_REQUIRE(B >= 0 && D >= 0, "affine_backward: negative size");
    if (B == 0 || D == 0) return TFEP_OK;
    TFEP_REQUIRE(x && params && gy && gparams && gx, "affine_backward: NULL pointer");
    affine_backward_kernel<<<row_blocks_b(B), 256, 0, (hipStream_t)stream>>>(x, ldx, params, layout, gy, ldgy, g_log_det_J,
                                                                            gparams, glayout, gx, ldgx, B, D);
    return check_launch("affine_backward_kernel");
}

int tfep_spline_backward(const float* x, int64_t ldx, const float* params, tfep_param_layout layout,
                         const tfep_spline_desc* d, const float* gy, int64_t ldgy, const float* g_log_det_J,
                         float* gparams, tfep_param_layout glayout, float* gx, int64_t ldgx, int B, int D, void* stream) {
    TFEP_REQUIRE(d != nullptr, "spline_backward: descriptor is NULL");
    TFEP_REQUIRE(d->n_bins >= 1 && d->n_bins <= 32, "spline_backward: n_bins=%d unsupported (1..32)", d->n_bins);
    TFEP_REQUIRE(B >= 0 && D >= 0, "spline_backward: negative size");
    if (B == 0 || D == 0) return TFEP_OK;
    TFEP_REQUIRE(x && params && gy && gparams && gx && d->x0 && d->xf && d->y0 && d->yf, "spline_backward: NULL pointer");
    SplineArgsB a;
    a.x0 = d->x0; a.xf = d->xf; a.y0 = d->y0; a.yf = d->yf;
    a.f.K = d->n_bins; a.f.circular = d->circular != 0; a.f.identity = d->identity_boundary_slopes != 0;
    a.f.learn_lower = d->learn_lower_bound != 0; a.f.learn_upper = d->learn_upper_bound != 0;
    a.f.min_bin = d->min_bin_size; a.f.min_slope = d->min_slope;
    a.f.slope_offset = (float)log(exp(1.0 - (double)d->min_slope) - 1.0);
    a.P = spline_n_params(a.f.K, a.f.circular, a.f.identity, a.f.learn_lower, a.f.learn_upper);
    hipStream_t s = (hipStream_t)stream;
    if (a.f.K <= 8)
        spline_backward_kernel<8><<<row_blocks_b(B), 256, 4 * 64 * (3 * 8 + 4) * sizeof(float), s>>>(x, ldx, params, layout, a, gy, ldgy, g_log_det_J, gparams, glayout, gx, ldgx, B, D);
    else if (a.f.K <= 16)
        spline_backward_kernel<16><<<row_blocks_b(B), 256, 4 * 64 * (3 * 16 + 4) * sizeof(float), s>>>(x, ldx, params, layout, a, gy, ldgy, g_log_det_J, gparams, glayout, gx, ldgx, B, D);
    else
        spline_backward_kernel<32><<<row_blocks_b(B), 256, 0, s>>>(x, ldx, params, layout, a, gy, ldgy, g_log_det_J, gparams, glayout, gx, ldgx, B, D);
    return check_launch("spline_backward_kernel");
}

int tfep_moebius_backward(const float* x, int64_t ldx, const float* params, int64_t ldp, int dimension,
                          float max_radius, int unit_sphere, int sign, const float* gy, int64_t ldgy,
                          const float* g_log_det_J, float* gparams, int64_t ldgp, float* gx, int64_t ldgx, int B, int D,
                          void* stream) {
    TFEP_REQUIRE(dimension >= 1 && dimension <= MOEBIUS_MAX_DIM_B, "moebius_backward: dimension=%d unsupported", dimension);
    TFEP_REQUIRE(D % dimension == 0 && (sign == 1 || sign == -1), "moebius_backward: bad arguments");
    if (B <= 0 || D <= 0) return TFEP_OK;
    TFEP_REQUIRE(x && params && gy && gparams && gx, "moebius_backward: NULL pointer");
    moebius_backward_kernel<<<row_blocks_b(B), 256, 0, (hipStream_t)stream>>>(x, ldx, params, ldp, dimension, max_radius,
                                                                             unit_sphere, (float)sign, gy, ldgy, g_log_det_J,
                                                                             gparams, ldgp, gx, ldgx, B, D);
    return check_launch("moebius_backward_kernel");
}

int tfep_copy_2d(const float* src, int64_t lds, float* dst, int64_t ldd, int B, int C, void* stream) {
    const int64_t n = (int64_t)B * C;
    if (n <= 0) return TFEP_OK;
    TFEP_REQUIRE(src && dst, "copy_2d: NULL pointer");
    copy_2d_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, lds, dst, ldd, B, C);
    return check_launch("copy_2d_kernel");
}

int tfep_weight_norm_backward(const float* gw_packed, int64_t ldw, const float* weight_v, const float* weight_g,
                              const float* mask, int out_features, int in_features, const int32_t* row_of_out,
                              const int32_t* col_of_in, float* grad_v, float* grad_g, void* stream) {
    TFEP_REQUIRE(out_features >= 0 && in_features >= 0, "weight_norm_backward: negative size");
    if (out_features == 0 || in_features == 0) return TFEP_OK;
    TFEP_REQUIRE(gw_packed && weight_v && grad_v && (!weight_g || grad_g), "weight_norm_backward: NULL pointer");
    weight_norm_backward_kernel<<<(unsigned)((out_features + 3) / 4), 256, 0, (hipStream_t)stream>>>(
        gw_packed, ldw, weight_v, weight_g, mask, out_features, in_features, row_of_out, col_of_in, grad_v, grad_g);
    return check_launch("weight_norm_backward_kernel");
}

int tfep_weight_norm_backward_prefix(const float* gw_packed, int64_t ldw, const float* weight_v, const float* weight_g,
                                     int out_features, int in_features, const int32_t* row_of_out, const int32_t* in_of_col,
                                     const int32_t* col_cut, float* grad_v, float* grad_g, void* stream) {
    TFEP_REQUIRE(out_features >= 0 && in_features >= 0, "weight_norm_backward_prefix: negative size");
    if (out_features == 0 || in_features == 0) return TFEP_OK;
    TFEP_REQUIRE(gw_packed && weight_v && grad_v && col_cut && (!weight_g || grad_g), "weight_norm_backward_prefix: NULL pointer");
    TFEP_REQUIRE(ldw >= in_features, "weight_norm_backward_prefix: packed gradient rows too short");
    TFEP_REQUIRE((size_t)in_features * 4 <= 64 * 1024, "weight_norm_backward_prefix: a row of %d weights does not fit the LDS stage",
                 in_features);
    weight_norm_backward_prefix_kernel<<<(unsigned)out_features, WNB_THREADS, (size_t)in_features * 4, (hipStream_t)stream>>>(
        gw_packed, ldw, weight_v, weight_g, out_features, in_features, row_of_out, in_of_col, col_cut, grad_v, grad_g);
    return check_launch("weight_norm_backward_prefix_kernel");
}

int tfep_periodic_embedding_backward(const float* x, int64_t ldx, const int32_t* periodic_indices, int n_periodic,
                                     const int32_t* nonperiodic_indices, int n_nonperiodic, float lower, float upper,
                                     const float* gout, int64_t ldg, float* gx, int64_t ldgx, int B, void* stream) {
    const int64_t n = (int64_t)B * (n_periodic + n_nonperiodic);
    if (n == 0) return TFEP_OK;
    TFEP_REQUIRE(x && gout && gx, "periodic_embedding_backward: NULL pointer");
    const float scale = (float)(2.0 * 3.14159265358979323846 / ((double)upper - (double)lower));
    periodic_embedding_backward_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        x, ldx, periodic_indices, n_periodic, nonperiodic_indices, n_nonperiodic, lower, scale, gout, ldg, gx, ldgx, B);
    return check_launch("periodic_embedding_backward_kernel");
}

}  // extern "C"
