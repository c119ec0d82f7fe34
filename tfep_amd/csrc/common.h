// Shared host/device helpers for libtfep_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/tfep_hip.h"

namespace tfep {

// Thread-local last-error string returned by tfep_last_error().
std::string& last_error();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TFEP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return TFEP_OK;
}

#define TFEP_REQUIRE(cond, ...)                                            \
    do {                                                                   \
        if (!(cond)) return ::tfep::fail(TFEP_ERR_INVALID_ARGUMENT, __VA_ARGS__); \
    } while (0)

constexpr int WAVE = 64;

// Sum over the 64 lanes of a wavefront (butterfly; every lane gets the total).
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ inline double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Python / torch `%` for floats: result has the sign of the divisor.
__device__ inline double py_mod(double a, double b) {
    double r = fmod(a, b);
    if (r != 0.0 && ((r < 0.0) != (b < 0.0))) r += b;
    return r;
}

}  // namespace tfep
