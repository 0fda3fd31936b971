// ActNorm and the invertible 1x1 convolution on NHWC fp32 tensors, standalone and fused.
//
// Reference arithmetic:  ActNorm.transform/invert  normalizing_flow/transforms.py:80, :93
//                        InvConv2d.transform/invert normalizing_flow/transforms.py:132, :144
//                        StepFlow ordering          normalizing_flow/glow.py:46-47, :61-62
//
// Both ops are HBM-bound (2*C*4 bytes per pixel, C..C^2 FMAs).  Design for gfx950:
//   * affine-only kernels stream float4 per lane (16 B/lane coalesced), exp(scale) staged once per block in LDS;
//   * the per-pixel CxC product stages a 256-pixel tile through LDS with coalesced float4 global loads, gives
//     each lane one pixel (row stride C+1 words -> conflict-free ds_read_b32), keeps the pixel's C values in
//     VGPRs and reads W through the scalar cache (uniform addresses -> s_load, W lives in SGPRs), so the VALU
//     does one v_fma per MAC and nothing else; results go back through the same LDS tile and leave with
//     coalesced float4 stores.  ActNorm is folded in as a pre-affine (forward) or post-affine (inverse), so a
//     flow head costs one HBM round trip instead of two.
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;

enum Mode { kMatvec = 0, kPreAffine = 1, kPostAffine = 2 };

// ----------------------------------------------------------------------------- affine-only (ActNorm alone)
template <bool INV, bool VEC4>
__global__ void __launch_bounds__(kBlock) actnorm_kernel(const float* __restrict__ x, float* __restrict__ y, long total,
                                                         int C, const float* __restrict__ scale,
                                                         const float* __restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [C] exp(+-scale), [C] bias
    float* es = smem;
    float* bs = smem + C;
    for (int c = threadIdx.x; c < C; c += kBlock) {
        es[c] = expf(INV ? -scale[c] : scale[c]);
        bs[c] = bias[c];
    }
    __syncthreads();
    if (VEC4) {
        const long n4 = total >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        float4* y4 = reinterpret_cast<float4*>(y);
        for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
            const int c = (int)((i << 2) % C);
            float4 v = x4[i];
            if (!INV) {
                v.x = es[c] * (v.x + bs[c]);
                v.y = es[c + 1] * (v.y + bs[c + 1]);
                v.z = es[c + 2] * (v.z + bs[c + 2]);
                v.w = es[c + 3] * (v.w + bs[c + 3]);
            } else {
                v.x = __fsub_rn(__fmul_rn(v.x, es[c]), bs[c]);
                v.y = __fsub_rn(__fmul_rn(v.y, es[c + 1]), bs[c + 1]);
                v.z = __fsub_rn(__fmul_rn(v.z, es[c + 2]), bs[c + 2]);
                v.w = __fsub_rn(__fmul_rn(v.w, es[c + 3]), bs[c + 3]);
            }
            y4[i] = v;
        }
    } else {
        for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
            const int c = (int)(i % C);
            const float v = x[i];
            y[i] = INV ? __fsub_rn(__fmul_rn(v, es[c]), bs[c]) : es[c] * (v + bs[c]);
        }
    }
}

// ----------------------------------------------------------------------------- per-pixel CxC product, C templated
// One workgroup = 64 pixels x 4 waves.  Lane = pixel, wave = quarter of the output channels: every lane keeps its
// pixel's C inputs in VGPRs (conflict-free LDS row reads, stride C+1 words) and produces C/4 outputs with W rows
// fetched through the scalar cache (wave-uniform addresses -> s_load, SGPR operands: one v_fma per MAC).  Splitting
// by output channel instead of giving a lane the whole CxC product quarters the serial FMA chain and quadruples
// the number of workgroups (128 at the 4x4 level of a B=512 batch instead of 32).  Tiles are staged through LDS so
// global loads/stores are coalesced float4.  Workgroup 0 also emits the flow's two batch-independent log-det
// terms (transforms.py:81, :131) so no extra launch is needed for them.
constexpr int kTilePx = 64;

template <int C, int MODE>
__global__ void __launch_bounds__(kBlock) pixel_linear_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              long npix, const float* __restrict__ w,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ logabsdet, float hw,
                                                              float* __restrict__ ldj_const_out) {
    constexpr int LD = C + 1;
    constexpr int OPW = (C + 3) / 4;                 // outputs per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                               // [64][LD]
    float* yout = smem + kTilePx * LD;               // [64][LD]
    float* es = yout + kTilePx * LD;                 // [C]
    float* bs = es + C;                              // [C]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (MODE != kMatvec) {
        for (int c = threadIdx.x; c < C; c += kBlock) {
            es[c] = expf(MODE == kPostAffine ? -scale[c] : scale[c]);
            bs[c] = bias[c];
        }
    }
    if (ldj_const_out != nullptr && blockIdx.x == 0 && wave == 0) {
        double sacc = 0.0;
        for (int c = lane; c < C; c += 64) sacc += (double)scale[c];
        sacc = wave_sum_d(sacc);
        if (lane == 0) {
            ldj_const_out[0] = __fmul_rn(hw, (float)sacc);
            ldj_const_out[1] = logabsdet ? __fmul_rn(hw, logabsdet[0]) : 0.0f;
        }
    }
    const long ntiles = (npix + kTilePx - 1) / kTilePx;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long p0 = t * kTilePx;
        const int np = (int)((npix - p0) < kTilePx ? (npix - p0) : kTilePx);
        const int nval = np * C;
        __syncthreads();                             // previous tile's stores done; es/bs visible
        const float* src = x + p0 * C;
        if (C % 4 == 0) {
            const float4* src4 = reinterpret_cast<const float4*>(src);
            for (int i = threadIdx.x; i < (nval >> 2); i += kBlock) {
                const float4 v = src4[i];
                const int e = i << 2;
                float* d = xin + (e / C) * LD + (e % C);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int e = threadIdx.x; e < nval; e += kBlock) xin[(e / C) * LD + (e % C)] = src[e];
        }
        __syncthreads();
        if (lane < np) {
            const float* row = xin + lane * LD;
            float v[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                v[c] = row[c];
                if (MODE == kPreAffine) v[c] = es[c] * (v[c] + bs[c]);
            }
#pragma unroll
            for (int j = 0; j < OPW; ++j) {
                const int o = wave * OPW + j;        // wave-uniform
                if (o < C) {
                    const float* wr = w + o * C;     // uniform address -> scalar loads
                    float acc = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) acc = fmaf(wr[c], v[c], acc);
                    if (MODE == kPostAffine) acc = __fsub_rn(__fmul_rn(acc, es[o]), bs[o]);
                    yout[lane * LD + o] = acc;
                }
            }
        }
        __syncthreads();
        float* dst = y + p0 * C;
        if (C % 4 == 0) {
            float4* dst4 = reinterpret_cast<float4*>(dst);
            for (int i = threadIdx.x; i < (nval >> 2); i += kBlock) {
                const int e = i << 2;
                const float* q = yout + (e / C) * LD + (e % C);
                dst4[i] = make_float4(q[0], q[1], q[2], q[3]);
            }
        } else {
            for (int e = threadIdx.x; e < nval; e += kBlock) dst[e] = yout[(e / C) * LD + (e % C)];
        }
    }
}

// Any C (runtime): one thread per output element; x row re-read through L1, W through L2.  Correctness path for
// channel counts without a templated instance (the MFMA 1x1 conv serves large C on the fast path).
template <int MODE>
__global__ void __launch_bounds__(kBlock) pixel_linear_generic_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                      long total, int C, const float* __restrict__ w,
                                                                      const float* __restrict__ scale,
                                                                      const float* __restrict__ bias) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int o = (int)(i % C);
        const long p = i / C;
        const float* xr = x + p * C;
        const float* wr = w + (long)o * C;
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) {
            float v = xr[c];
            if (MODE == kPreAffine) v = expf(scale[c]) * (v + bias[c]);
            acc = fmaf(wr[c], v, acc);
        }
        if (MODE == kPostAffine) acc = __fsub_rn(__fmul_rn(acc, expf(-scale[o])), bias[o]);
        y[i] = acc;
    }
}

// Wide C (64, 96, 192: the deeper levels of 4- and 5-level models) on the exact-fp32 matrix cores.  As one FMA chain per lane
// (pixel_linear_kernel) a lane walks C/4 x C serial FMAs: 25 us for the 2048 pixels of a C = 96 level, 130 us for the 512 pixels of a
// C = 192 level.  Per pixel the product is a C x C matrix-vector product; v_mfma_f32_32x32x2_f32 takes exactly this shape: A = W (rows =
// output channel, 2 input channels per instruction, this lane's elements straight from global memory), B = the 32-pixel tile (lane =
// pixel, lane half = which of the two input channels, from LDS, ActNorm folded into the read), C/2 chained instructions per 32 x 32
// outputs.  A workgroup owns 32 pixels; wave r computes output rows 32 r .. 32 r + 31 (C/32 waves); all reads of the tile happen
// before any write, so in place is safe.  Products and sums are fp32, input channels ascending (as the FMA chain).
template <int C, int MODE>
__global__ void __launch_bounds__(64 * (C / 32)) pixel_linear_mfma_kernel(const float* __restrict__ x, float* __restrict__ y, long npix,
                                                                        const float* __restrict__ w, const float* __restrict__ scale,
                                                                        const float* __restrict__ bias,
                                                                        const float* __restrict__ logabsdet, float hw,
                                                                        float* __restrict__ ldj_const_out) {
    static_assert(C % 32 == 0, "whole 32-row tiles");
    constexpr int NW = C / 32, NT = 64 * NW, LD = C + 1, KP = C / 2;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                                 // [32][LD]
    float* es = smem + 32 * LD;                      // [C]
    float* bs = es + C;                              // [C]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long p0 = (long)blockIdx.x * 32;
    const int np = (int)((npix - p0) < 32 ? (npix - p0) : 32);
    if (MODE != kMatvec)
        for (int c = tid; c < C; c += NT) {
            es[c] = expf(MODE == kPostAffine ? -scale[c] : scale[c]);
            bs[c] = bias[c];
        }
    if (ldj_const_out != nullptr && blockIdx.x == 0 && wave == 0) {
        double sacc = 0.0;
        for (int c = lane; c < C; c += 64) sacc += (double)scale[c];
        sacc = wave_sum_d(sacc);
        if (lane == 0) {
            ldj_const_out[0] = __fmul_rn(hw, (float)sacc);
            ldj_const_out[1] = logabsdet ? __fmul_rn(hw, logabsdet[0]) : 0.0f;
        }
    }
    // this lane's W elements first (global latency under the tile load): row 32 wave + (lane & 31), input channels 2 j + (lane >> 5)
    const int m0 = lane & 31, kh = lane >> 5;
    // (16-byte loads along the lane's row, both lane halves the same addresses: a quarter of the memory instructions of 4-byte loads at
    // stride 8; the half then picks its two of every four elements)
    const float4* wrow4 = reinterpret_cast<const float4*>(w + (long)(32 * wave + m0) * C);
    float a[KP];
#pragma unroll
    for (int i = 0; i < C / 4; ++i) {
        const float4 v = wrow4[i];
        a[2 * i] = kh ? v.y : v.x;
        a[2 * i + 1] = kh ? v.w : v.z;
    }
    const float4* src4 = reinterpret_cast<const float4*>(x + p0 * C);
    for (int i = tid; i < 32 * C / 4; i += NT) {
        const int e = i * 4, r = e / C, c = e % C;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < np) v = src4[i];
        float* d = X + r * LD + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* xrow = X + m0 * LD + kh;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        float b = xrow[2 * j];
        if (MODE == kPreAffine) b = es[2 * j + kh] * (b + bs[2 * j + kh]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b, acc, 0, 0, 0);
    }
    __syncthreads();                                 // every wave has read the tile
    float* orow = X + m0 * LD + 32 * wave + 4 * kh;  // D row held by (register r, lane half): (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = (r & 3) + 8 * (r >> 2);
        float v = acc[r];
        if (MODE == kPostAffine) v = __fsub_rn(__fmul_rn(v, es[32 * wave + 4 * kh + o]), bs[32 * wave + 4 * kh + o]);
        orow[o] = v;
    }
    __syncthreads();
    float4* dst4 = reinterpret_cast<float4*>(y + p0 * C);
    for (int i = tid; i < np * C / 4; i += NT) {
        const int e = i * 4;
        const float* q = X + (e / C) * LD + (e % C);
        dst4[i] = make_float4(q[0], q[1], q[2], q[3]);
    }
}

template <int C, int MODE>
int launch_mfma(const float* x, float* y, long npix, const float* w, const float* scale, const float* bias, const float* lad, float hw,
                float* cst, hipStream_t s) {
    const size_t lds = ((size_t)32 * (C + 1) + 2 * C) * sizeof(float);
    hipLaunchKernelGGL((pixel_linear_mfma_kernel<C, MODE>), dim3((unsigned)cdiv(npix, 32)), dim3(64 * (C / 32)), lds, s, x, y, npix, w,
                       scale, bias, lad, hw, cst);
    return NFDP_OK;
}

template <int C, int MODE>
int launch_templated(const float* x, float* y, long npix, const float* w, const float* scale, const float* bias,
                     const float* lad, float hw, float* cst, hipStream_t s) {
    const size_t lds = ((size_t)2 * kTilePx * (C + 1) + 2 * C) * sizeof(float);
    const long ntiles = cdiv(npix, kTilePx);
    const int grid = (int)(ntiles < 4096 ? ntiles : 4096);
    if (lds > 65536) {                               // C = 192 (last level of a 5-level, 3-channel model): 99 KB, opted into once
        static NfdpPerDevice attr_set;
        if (!attr_set) {
            nfdp_func_attr(reinterpret_cast<const void*>(&pixel_linear_kernel<C, MODE>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
    }
    hipLaunchKernelGGL((pixel_linear_kernel<C, MODE>), dim3(grid), dim3(kBlock), lds, s, x, y, npix, w, scale, bias, lad, hw,
                       cst);
    return NFDP_OK;
}

template <int MODE>
int dispatch(const char* what, const float* x, float* y, long npix, int C, const float* w, const float* scale,
             const float* bias, nfdp_stream_t stream, const float* lad = nullptr, float hw = 0.f, float* cst = nullptr) {
    NFDP_REQUIRE(x && y && w, NFDP_EINVAL, "%s: null pointer", what);
    NFDP_REQUIRE(MODE == kMatvec || (scale && bias), NFDP_EINVAL, "%s: null scale/bias", what);
    NFDP_REQUIRE(npix >= 0 && C > 0, NFDP_ESHAPE, "%s: bad shape npix=%ld C=%d", what, npix, C);
    if (npix == 0) return NFDP_OK;
    hipStream_t s = as_stream(stream);
    const bool al16 = ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w) & 15) == 0);
    switch (C) {
        case 4: launch_templated<4, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); break;
        case 8: launch_templated<8, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); break;
        case 12: launch_templated<12, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); break;
        case 16: launch_templated<16, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); break;
        case 24: launch_templated<24, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); break;
        case 48: launch_templated<48, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); break;
        // deeper models (L = 4, 5: CelebA 128x128 is L = 5): C = 2^(l+2) * {1, 3}
        case 32: launch_templated<32, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); break;
        // wide C: the exact-fp32 matrix cores (16-byte aligned tensors; else the FMA-chain kernel)
        case 64: if (al16) launch_mfma<64, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); else launch_templated<64, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); break;
        case 96: if (al16) launch_mfma<96, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); else launch_templated<96, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); break;
        case 192: if (al16) launch_mfma<192, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); else launch_templated<192, MODE>(x, y, npix, w, scale, bias, lad, hw, cst, s); break;
        default: {
            NFDP_REQUIRE(x != y, NFDP_EINVAL, "%s: in-place needs a templated C (got %d)", what, C);
            NFDP_REQUIRE(cst == nullptr, NFDP_EUNSUPPORTED, "%s: fused log-det constants need a templated C (got %d)", what, C);
            const long total = npix * C;
            hipLaunchKernelGGL((pixel_linear_generic_kernel<MODE>), dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, s,
                               x, y, total, C, w, scale, bias);
        }
    }
    NFDP_CHECK_LAUNCH(what);
    return NFDP_OK;
}

template <bool INV>
int launch_actnorm(const char* what, const float* x, float* y, long npix, int C, const float* scale, const float* bias,
                   nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y && scale && bias, NFDP_EINVAL, "%s: null pointer", what);
    NFDP_REQUIRE(npix >= 0 && C > 0 && C <= 8192, NFDP_ESHAPE, "%s: bad shape npix=%ld C=%d", what, npix, C);
    if (npix == 0) return NFDP_OK;
    const long total = npix * C;
    const size_t lds = 2 * (size_t)C * sizeof(float);
    const bool vec = (C % 4 == 0) && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL((actnorm_kernel<INV, true>), dim3(stream_grid(total >> 2, kBlock)), dim3(kBlock), lds,
                           as_stream(stream), x, y, total, C, scale, bias);
    else
        hipLaunchKernelGGL((actnorm_kernel<INV, false>), dim3(stream_grid(total, kBlock)), dim3(kBlock), lds,
                           as_stream(stream), x, y, total, C, scale, bias);
    NFDP_CHECK_LAUNCH(what);
    return NFDP_OK;
}

}  // namespace

extern "C" int nfdp_actnorm_fwd_f32(const float* x, float* y, long npix, int C, const float* scale, const float* bias,
                                    nfdp_stream_t stream) {
    return launch_actnorm<false>("actnorm_fwd", x, y, npix, C, scale, bias, stream);
}

extern "C" int nfdp_actnorm_inv_f32(const float* y, float* x, long npix, int C, const float* scale, const float* bias,
                                    nfdp_stream_t stream) {
    return launch_actnorm<true>("actnorm_inv", y, x, npix, C, scale, bias, stream);
}

extern "C" int nfdp_invconv_apply_f32(const float* x, float* y, long npix, int C, const float* w, nfdp_stream_t stream) {
    return dispatch<kMatvec>("invconv_apply", x, y, npix, C, w, nullptr, nullptr, stream);
}

extern "C" int nfdp_actnorm_invconv_fwd_f32(const float* x, float* y, long npix, int C, const float* scale,
                                            const float* bias, const float* w, nfdp_stream_t stream) {
    return dispatch<kPreAffine>("actnorm_invconv_fwd", x, y, npix, C, w, scale, bias, stream);
}

extern "C" int nfdp_flow_head_fwd_f32(const float* x, float* y, long npix, int C, const float* scale, const float* bias,
                                      const float* w, const float* logabsdet, float hw, float* ldj_const_out,
                                      nfdp_stream_t stream) {
    NFDP_REQUIRE(ldj_const_out, NFDP_EINVAL, "flow_head_fwd: null pointer");
    return dispatch<kPreAffine>("flow_head_fwd", x, y, npix, C, w, scale, bias, stream, logabsdet, hw, ldj_const_out);
}

extern "C" int nfdp_invconv_actnorm_inv_f32(const float* y, float* x, long npix, int C, const float* scale,
                                            const float* bias, const float* winv, nfdp_stream_t stream) {
    return dispatch<kPostAffine>("invconv_actnorm_inv", y, x, npix, C, winv, scale, bias, stream);
}
