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

// Slot of the current HIP device for per-device one-time state (kernel attributes).
constexpr int TFEP_MAX_DEVICES = 64;
inline int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev % TFEP_MAX_DEVICES;
}

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
// For a positive divisor (a period) the remainder is formed with one floor and one fma: a - floor(a/b) * b is exactly
// representable whenever |a| >= b (it is a multiple of ulp(b) below 2b), so the fma returns it exactly, and for
// |a| < b it is a or a + b rounded once -- the same values fmod() + the sign fix-up produce, without fmod's
// data-dependent reduction loop (which serialised the lanes of the circular-spline epilogues).  A quotient that rounds
// across an integer leaves r just outside [0, b): one exact-sum correction brings it back.
__device__ inline double py_mod(double a, double b) {
    if (b > 0.0 && fabs(a) < 1e300) {
        const double q = floor(a / b);
        double r = fma(-q, b, a);
        if (r < 0.0) r += b;
        if (r >= b) r -= b;
        return r;
    }
    double r = fmod(a, b);
    if (r != 0.0 && ((r < 0.0) != (b < 0.0))) r += b;
    return r;
}

}  // namespace tfep
