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

// Measurement switches exist only in -DNFDP_DIAG builds (tools/build_variant.sh): environment reads for A/B runs, the NFDP_ABLATE_* /
// *_STAMPS timing-only code paths.  The shipped library reads no environment variable and contains none of those paths.
#ifdef NFDP_DIAG
#include <stdlib.h>
#define NFDP_DIAG_ENV(name) getenv(name)
#else
#define NFDP_DIAG_ENV(name) ((const char*)nullptr)
#if defined(NFDP_ABLATE_NOSTREAM) || defined(NFDP_ABLATE_DMA4) || defined(NFDP_ABLATE_MFMA16) || defined(NFDP_ABLATE_NOMFMA) || \
    defined(NFDP_FUSED_STAMPS) || defined(NFDP_FUSED_STAMPS_FINE) || defined(NFDP_FUSED_CLOCK) || defined(NFDP_BWD_ABLATE_LOADS) || \
    defined(NFDP_BWD_ABLATE_STORES) || defined(NFDP_BWD_ABLATE_STORE_ADDR) || defined(NFDP_W2_ABLATE) || defined(NFDP_CONV_ABLATE_SPLIT) || defined(NFDP_LX_ABLATE) || defined(NFDP_LX_STAMPS)
#error "timing-only / stamping code paths need -DNFDP_DIAG (tools/build_variant.sh adds it): they must not reach the shipped library"
#endif
#endif

// "once per DEVICE" flag for launch-site set-up (hipFuncSetAttribute is per device): `static NfdpPerDevice f; if (!f) { ...; f = true; }`
struct NfdpPerDevice {
    unsigned long long mask = 0;
    static int dev() {
        int d = 0;
        (void)hipGetDevice(&d);
        return d & 63;
    }
    bool operator!() const { return !((mask >> dev()) & 1ull); }
    NfdpPerDevice& operator=(bool v) {
        if (v) mask |= 1ull << dev();
        return *this;
    }
};

// hipFuncSetAttribute with its result checked: a kernel that needs > 64 KB of dynamic LDS and did not get the opt-in fails at launch with
// a generic error; the cause is recorded here (thread-local error text, and once on stderr).
static inline bool nfdp_func_attr(const void* fn, hipFuncAttribute attr, int value) {
    const hipError_t e = hipFuncSetAttribute(fn, attr, value);
    if (e == hipSuccess) return true;
    nfdp_set_error("hipFuncSetAttribute(%d, %d) failed: %s", (int)attr, value, hipGetErrorString(e));
    fprintf(stderr, "nfdp: hipFuncSetAttribute(%d, %d) failed: %s\n", (int)attr, value, hipGetErrorString(e));
    return false;
}

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
