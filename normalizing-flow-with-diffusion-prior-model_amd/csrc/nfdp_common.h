// Shared host/device helpers for the nfdp gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/nfdp.h"

void nfdp_set_error(const char* fmt, ...);

#define NFDP_REQUIRE(cond, code, ...)   \
    do {                                \
        if (!(cond)) {                  \
            nfdp_set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

#define NFDP_CHECK_LAUNCH(what)                                                   \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            nfdp_set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
            return NFDP_EHIP;                                                     \
        }                                                                         \
    } while (0)

static inline hipStream_t as_stream(nfdp_stream_t s) { return (hipStream_t)s; }
static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// Grid cap for grid-stride HBM-bound kernels: 256 CUs x 8 blocks (cdna_hip_programming.md G11).
static inline int stream_grid(long work_items, int block) {
    long g = cdiv(work_items, block);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

#ifdef __HIPCC__
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#endif
