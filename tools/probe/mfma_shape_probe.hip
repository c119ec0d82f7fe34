// Which fp16 MFMA shape gives more sustained throughput at the board power limit: v_mfma_f32_16x16x32_f16 (what the split
// GEMM uses) or v_mfma_f32_32x32x16_f16 (half the operand-register and LDS-fragment reads per flop)?  Pure register
// loop, 4 waves per workgroup, one workgroup per SIMD set as in the split GEMM (1 wave / SIMD), operands rotate through 4
// pseudo-random fragments so the datapath toggles like on real data.  Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline f16x8 frag(uint32_t seed, float scale) {
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        seed = seed * 1664525u + 1013904223u;
        v[j] = (_Float16)(((int)(seed >> 9) % 2048 - 1024) * scale);      // ~uniform in [-1, 1) * 1024 * scale
    }
    return v;
}

template <int ZERO>
__global__ void __launch_bounds__(256, 1) probe16(float* out, int iters) {
    constexpr int NACC = 48;                                             // 48 x 4 = 192 accumulator registers
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a[r] = ZERO ? frag(0, 0.f) : frag(threadIdx.x * 8 + r, 1e-3f);
        b[r] = ZERO ? frag(0, 0.f) : frag(threadIdx.x * 8 + 4 + r, 1e-3f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)      // inline asm pins the accumulators in place (the builtin made the compiler
                                            // rename half of them and pad with s_nop: 1702 instead of ~2400 TFLOP/s on zeros)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ZERO>
__global__ void __launch_bounds__(256, 1) probe32(float* out, int iters) {
    constexpr int NACC = 12;                                             // 12 x 16 = 192 accumulator registers
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    f16x8 a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        a[r] = ZERO ? frag(0, 0.f) : frag(threadIdx.x * 8 + r, 1e-3f);
        b[r] = ZERO ? frag(0, 0.f) : frag(threadIdx.x * 8 + 4 + r, 1e-3f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)                                // same flops per iteration as probe16
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i + rep) & 3]), "v"(b[(i >> 2) & 3]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" {

// flops of one launch: both shapes do 48 x 16384 = 24 x 32768 flops per wave and iteration
double probe_flops(int blocks, int iters) { return (double)blocks * 4.0 * iters * 48.0 * 16384.0; }

// shape: 16 or 32; zero: all-zero operands.  Returns milliseconds of `launches` back-to-back launches, < 0 on error.
float probe_run(int shape, int zero, int blocks, int iters, int launches) {
    float* out = nullptr;
    if (hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)) != hipSuccess) return -1.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto go = [&]() {
        if (shape == 16) { if (zero) probe16<1><<<blocks, 256>>>(out, iters); else probe16<0><<<blocks, 256>>>(out, iters); }
        else { if (zero) probe32<1><<<blocks, 256>>>(out, iters); else probe32<0><<<blocks, 256>>>(out, iters); }
    };
    go();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) go();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const bool ok = hipGetLastError() == hipSuccess;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(out);
    return ok ? ms : -2.f;
}

}
