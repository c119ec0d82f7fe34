// Vector-instruction issue rates on gfx950, in cycles per wave-instruction per SIMD, for the instruction kinds the
// config-5 edge kernel is made of -- alone and beside fp16 MFMAs, at 1 / 2 / 4 waves per SIMD.  Every workgroup times its
// own loop with s_memtime (shader clock) and writes cycles per loop pass.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

using f4 = __attribute__((ext_vector_type(4))) float;
using u4 = __attribute__((ext_vector_type(4))) uint32_t;
using f2 = __attribute__((ext_vector_type(2))) float;

#define REP8(x) x x x x x x x x
#define REP16(x) REP8(x) REP8(x)

// kind: 0 v_fma_f32 (16 independent), 1 v_pk_fma_f32 (16 independent pairs), 2 v_exp_f32, 3 v_cvt_pk_f16_f32,
//       4 v_fma_mixlo_f16, 5 mfma 16x16x32 f16 only (16, 4 accumulators), 6 mfma + 32 v_fma interleaved (2 per MFMA),
//       7 mfma + 16 v_pk_fma interleaved (1 per MFMA), 8 v_mul_f32 dependent chain (16 long), 9 v_rcp_f32,
//       10 mfma + 64 v_fma (4 per MFMA), 11 v_pk_mul_f32
template <int KIND>
__global__ void probe(float* out, long long* cyc, int iters) {
    float a[16];
    f2 p[16];
    f4 acc[4];
    u4 wa, wb;
    const float s = 1.0f + 1e-7f * threadIdx.x, t = 1e-9f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = 1.0f + 0.001f * i + threadIdx.x * 1e-6f; p[i] = f2{a[i], a[i] + 0.5f}; }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    wa = u4{0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    wb = u4{0x38003800u, 0x38003800u + threadIdx.x, 0x38003800u, 0x38003800u};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(t));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 15]), "v"(p[(i + 2) & 15]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define CVT(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
#define MIX(i) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]" : "+v"(a[i]) : "v"(s));
#define MFMA(i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(i) & 3]) : "v"(wa), "v"(wb));
#define ALL16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
        if (KIND == 0) { ALL16(FMA) }
        if (KIND == 1) { ALL16(PKFMA) }
        if (KIND == 2) { ALL16(EXP) }
        if (KIND == 3) { ALL16(CVT) }
        if (KIND == 4) { ALL16(MIX) }
        if (KIND == 5) { ALL16(MFMA) }
#define MF2(i) MFMA(i) FMA(i) FMA((i + 8) & 15)
        if (KIND == 6) { ALL16(MF2) }
#define MP1(i) MFMA(i) PKFMA(i)
        if (KIND == 7) { ALL16(MP1) }
        if (KIND == 8) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(s));
        }
        if (KIND == 9) { ALL16(RCP) }
#define MF4(i) MFMA(i) FMA(i) FMA((i + 4) & 15) FMA((i + 8) & 15) FMA((i + 12) & 15)
        if (KIND == 10) { ALL16(MF4) }
        if (KIND == 11) { ALL16(PKMUL) }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += a[i] + p[i].x + p[i].y;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float g_last_ms = 0.f;
template <int KIND>
static double run(int waves_per_simd, int iters) {
    const int blocks = 256, threads = 256 * waves_per_simd;
    float* out; long long* cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 100);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    g_last_ms = ms;
    static long long h[256];
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (int i = 0; i < blocks; ++i) m += (double)h[i];
    hipFree(out); hipFree(cyc);
    (void)m;
    return g_last_ms * 1e-3 * 2.37e9 / iters;      // wall time as cycles at 2.37 GHz per loop pass: s_memtime does not tick
                                                    // at wall rate when several waves share a SIMD (calibration line)
}

int main() {
    const char* names[] = {"16 v_fma_f32", "16 v_pk_fma_f32", "16 v_exp_f32", "16 v_cvt_pk_f16_f32", "16 v_fma_mixlo_f16",
                           "16 mfma_16x16x32_f16", "16 mfma + 32 v_fma", "16 mfma + 16 v_pk_fma", "16 dependent v_mul",
                           "16 v_rcp_f32", "16 mfma + 64 v_fma", "16 v_pk_mul_f32"};
    for (int w : {1, 2, 4}) {
        double r[12];
        r[0] = run<0>(w, 20000); r[1] = run<1>(w, 20000); r[2] = run<2>(w, 20000); r[3] = run<3>(w, 20000);
        r[4] = run<4>(w, 20000); r[5] = run<5>(w, 20000); r[6] = run<6>(w, 20000); r[7] = run<7>(w, 20000);
        r[8] = run<8>(w, 20000); r[9] = run<9>(w, 20000); r[10] = run<10>(w, 20000); r[11] = run<11>(w, 20000);
        {   // calibrate the tick: ticks of the MFMA-only loop against its wall time
            const double ticks = run<5>(w, 200000) * 200000;
            printf("%d waves/SIMD  s_memtime rate: %.3f GHz (ticks / wall time of a 200000-pass MFMA loop, %.2f ms)\n", w,
                   ticks / (g_last_ms * 1e-3) / 1e9, g_last_ms);
        }
        for (int k = 0; k < 12; ++k)
            printf("%d waves/SIMD  %-24s %8.1f cycles (wall) per pass of ALL the waves of a SIMD = %6.2f per wave's 16-group\n", w,
                   names[k], r[k], r[k] / w);
    }
    return 0;
}
