// Index-reshuffle kernels: NCHW<->NHWC boundary conversion, Squeeze / unsqueeze, channel-slice copies,
// pre/post-processing.  Pure data movement (bit-exact); every kernel is one thread per OUTPUT element so the
// stores are fully coalesced and the gathers hit L1/L2 (the tensors are 12 KB per image).
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int C, int HW) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        int c = (int)(i % C);
        long bp = i / C;
        int p = (int)(bp % HW);
        long b = bp / HW;
        y[i] = x[(b * C + c) * HW + p];
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int C, int HW) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        int p = (int)(i % HW);
        long bc = i / HW;
        int c = (int)(bc % C);
        long b = bc / C;
        y[i] = x[(b * HW + p) * C + c];
    }
}

// y[b,h,w,4c+2h1+w1] = x[b,2h+h1,2w+w1,c]     (reference transforms.py:226)
// One thread per output float4 = the four sub-pixels of one channel: four independent 4-byte loads (neighbouring
// lanes read neighbouring channels of the same four pixels, so every fetched line is consumed by the wave) and one
// coalesced 16-byte store.  `total4` = B*(H/2)*(W/2)*C.
__global__ void squeeze_nhwc_kernel(const float* __restrict__ x, int x_ld, float4* __restrict__ y, long total4, int C, int H,
                                    int W) {
    const int h2 = H / 2, w2 = W / 2;
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total4; i += (long)gridDim.x * kBlock) {
        const int c = (int)(i % C);
        long r = i / C;
        const int w = (int)(r % w2);
        r /= w2;
        const int h = (int)(r % h2);
        const long b = r / h2;
        const float* p = x + ((b * H + 2 * h) * W + 2 * w) * (long)x_ld + c;
        const long row = (long)W * x_ld;
        y[i] = make_float4(p[0], p[x_ld], p[row], p[row + x_ld]);
    }
}

// x[b,2h+c1,2w+c2,c] = y[b,h,w,4c+2c1+c2]     (reference transforms.py:238); one thread per y float4 (mirror of the above).
__global__ void unsqueeze_nhwc_kernel(const float4* __restrict__ y, float* __restrict__ x, long total4, int C, int H, int W) {
    const int w2 = W / 2, h2 = H / 2;
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total4; i += (long)gridDim.x * kBlock) {
        const int c = (int)(i % C);
        long r = i / C;
        const int w = (int)(r % w2);
        r /= w2;
        const int h = (int)(r % h2);
        const long b = r / h2;
        const float4 v = y[i];
        float* p = x + ((b * H + 2 * h) * W + 2 * w) * (long)C + c;
        const long row = (long)W * C;
        p[0] = v.x; p[C] = v.y; p[row] = v.z; p[row + C] = v.w;
    }
}

__global__ void copy_channels_kernel(const float* __restrict__ src, int src_ld, float* __restrict__ dst, int dst_ld,
                                     long total, int n) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        int c = (int)(i % n);
        long p = i / n;
        dst[p * dst_ld + c] = src[p * src_ld + c];
    }
}

// utils.py:189-196 + trainer.py:155.  fp32 ops in the reference's order:
//   v = (u8/255)*255 -> floor(v / 2^(8-n_bits)) -> / n_bins - 0.5 -> + noise / n_bins
// (u8/255)*255 is not always exactly u8 in fp32, so it is reproduced literally.
__global__ void preprocess_kernel(const uint8_t* __restrict__ u8, const float* __restrict__ noise, float* __restrict__ y,
                                  long total, int C, int HW, int n_bits) {
    const float n_bins = (float)(1 << n_bits);
    const float div = (float)(1 << (8 - n_bits));
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        int c = (int)(i % C);
        long bp = i / C;
        int p = (int)(bp % HW);
        long b = bp / HW;
        long src = (b * C + c) * HW + p;
        float v = __fmul_rn(__fdiv_rn((float)u8[src], 255.0f), 255.0f);
        if (n_bits < 8) v = floorf(__fdiv_rn(v, div));
        v = __fsub_rn(__fdiv_rn(v, n_bins), 0.5f);
        if (noise) v = __fadd_rn(v, __fdiv_rn(noise[src], n_bins));
        y[i] = v;
    }
}

// utils.py:189-196 for a float batch in [0,1]
__global__ void preprocess_f32_kernel(const float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ y,
                                      long n, int n_bits) {
    const float n_bins = (float)(1 << n_bits);
    const float div = (float)(1 << (8 - n_bits));
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        float v = __fmul_rn(x[i], 255.0f);
        if (n_bits < 8) v = floorf(__fdiv_rn(v, div));
        v = __fsub_rn(__fdiv_rn(v, n_bins), 0.5f);
        if (noise) v = __fadd_rn(v, __fdiv_rn(noise[i], n_bins));
        y[i] = v;
    }
}

// utils.py:210
__global__ void postprocess_kernel(const float* __restrict__ x, uint8_t* __restrict__ u8, long total, int C, int HW,
                                   float n_bins) {
    const float mul = 256.0f / n_bins;
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        int p = (int)(i % HW);
        long bc = i / HW;
        int c = (int)(bc % C);
        long b = bc / C;
        float v = x[(b * HW + p) * C + c];
        v = __fmul_rn(floorf(__fmul_rn(__fadd_rn(v, 0.5f), n_bins)), mul);
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        u8[i] = (uint8_t)v;
    }
}


// Explicit im2col of a NARROW NHWC tensor for a 3x3 "same" conv: out[p][tap*C + c] = bf16(x[p + shift(tap)][c]) (0 outside the
// image), columns 9C..Kpad zero.  For the C/2 -> 512 input conv of the coupling network the 9C-wide matrix is a tenth of the
// 512-channel tensors it meets, and it turns the conv and its weight gradient into plain GEMMs with K-contiguous rows.
__global__ void im2col3x3_bf16_kernel(const float* __restrict__ x, int x_ld, int H, int W, int C, __bf16* __restrict__ out,
                                      int Kpad, long nchunks) {
    const int per = Kpad / 8, K = 9 * C;
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < nchunks; i += (long)gridDim.x * kBlock) {
        const long p = i / per;
        const int k0 = (int)(i - p * per) * 8;
        const long b = p / ((long)H * W);
        const int r = (int)(p - b * H * W);
        const int y = r / W, xx = r - y * W;
        bf16x8 v;
        int tap = k0 / C, c = k0 - tap * C;                        // one division per thread, then walk (tap, c)
        const float* row = x + ((b * H + y) * W + xx) * (long)x_ld;   // centre pixel
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = 0.f;
            if (k0 + j < K) {
                const int dy = (tap * 11 >> 5) - 1, dx = tap - 3 * (tap * 11 >> 5) - 1;   // tap / 3, tap % 3 for tap < 9
                if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(xx + dx) < (unsigned)W) f = row[(dy * W + dx) * (long)x_ld + c];
            }
            v[j] = (__bf16)f;
            if (++c == C) { c = 0; ++tap; }
        }
        *reinterpret_cast<bf16x8*>(out + i * 8) = v;
    }
}

}  // namespace

#define LAUNCH_1D(kernel, total, ...)                                                              \
    hipLaunchKernelGGL(kernel, dim3(stream_grid((total), kBlock)), dim3(kBlock), 0, as_stream(stream), __VA_ARGS__)

extern "C" int nfdp_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int H, int W, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y, NFDP_EINVAL, "nchw_to_nhwc: null pointer");
    NFDP_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, NFDP_ESHAPE, "nchw_to_nhwc: bad shape");
    long total = (long)B * C * H * W;
    if (total == 0) return NFDP_OK;
    LAUNCH_1D(nchw_to_nhwc_kernel, total, x, y, total, C, H * W);
    NFDP_CHECK_LAUNCH("nchw_to_nhwc");
    return NFDP_OK;
}

extern "C" int nfdp_nhwc_to_nchw_f32(const float* x, float* y, int B, int C, int H, int W, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y, NFDP_EINVAL, "nhwc_to_nchw: null pointer");
    NFDP_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, NFDP_ESHAPE, "nhwc_to_nchw: bad shape");
    long total = (long)B * C * H * W;
    if (total == 0) return NFDP_OK;
    LAUNCH_1D(nhwc_to_nchw_kernel, total, x, y, total, C, H * W);
    NFDP_CHECK_LAUNCH("nhwc_to_nchw");
    return NFDP_OK;
}

extern "C" int nfdp_squeeze_nhwc_f32(const float* x, int x_ld, float* y, int B, int C, int H, int W, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y, NFDP_EINVAL, "squeeze: null pointer");
    NFDP_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && x_ld >= C, NFDP_ESHAPE, "squeeze: bad shape");
    NFDP_REQUIRE(H % 2 == 0 && W % 2 == 0, NFDP_ESHAPE, "squeeze: H=%d W=%d must be even", H, W);
    long total = (long)B * C * H * W;
    if (total == 0) return NFDP_OK;
    NFDP_REQUIRE(((uintptr_t)y & 15) == 0, NFDP_EINVAL, "squeeze: y must be 16-byte aligned");
    LAUNCH_1D(squeeze_nhwc_kernel, total / 4, x, x_ld, reinterpret_cast<float4*>(y), total / 4, C, H, W);
    NFDP_CHECK_LAUNCH("squeeze");
    return NFDP_OK;
}

extern "C" int nfdp_unsqueeze_nhwc_f32(const float* y, float* x, int B, int C4, int h, int w, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y, NFDP_EINVAL, "unsqueeze: null pointer");
    NFDP_REQUIRE(B >= 0 && C4 > 0 && h > 0 && w > 0, NFDP_ESHAPE, "unsqueeze: bad shape");
    NFDP_REQUIRE(C4 % 4 == 0, NFDP_ESHAPE, "unsqueeze: C=%d must be divisible by 4", C4);
    long total = (long)B * C4 * h * w;
    if (total == 0) return NFDP_OK;
    NFDP_REQUIRE(((uintptr_t)y & 15) == 0, NFDP_EINVAL, "unsqueeze: y must be 16-byte aligned");
    LAUNCH_1D(unsqueeze_nhwc_kernel, total / 4, reinterpret_cast<const float4*>(y), x, total / 4, C4 / 4, 2 * h, 2 * w);
    NFDP_CHECK_LAUNCH("unsqueeze");
    return NFDP_OK;
}

extern "C" int nfdp_copy_channels_f32(const float* src, int src_ld, float* dst, int dst_ld, long npix, int n,
                                      nfdp_stream_t stream) {
    NFDP_REQUIRE(src && dst, NFDP_EINVAL, "copy_channels: null pointer");
    NFDP_REQUIRE(npix >= 0 && n > 0 && src_ld >= n && dst_ld >= n, NFDP_ESHAPE, "copy_channels: bad shape");
    long total = npix * n;
    if (total == 0) return NFDP_OK;
    LAUNCH_1D(copy_channels_kernel, total, src, src_ld, dst, dst_ld, total, n);
    NFDP_CHECK_LAUNCH("copy_channels");
    return NFDP_OK;
}

// torch.cat((a, b), dim=channel) of two pixel-strided NHWC tensors in one launch (the U-Net's skip connections,
// denoising_diffusion_pytorch.py:478-486): float4 per thread when all channel counts, strides and pointers allow.
template <bool VEC>
__global__ void concat_channels_kernel(const float* __restrict__ a, int a_ld, int ca, const float* __restrict__ b, int b_ld, int cb,
                                       float* __restrict__ out, int out_ld, long total) {
    constexpr int V = VEC ? 4 : 1;
    const int c = (ca + cb) / V;
    for (long e = blockIdx.x * (long)kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const long p = e / c;
        const int j = (int)(e - p * c) * V;
        const float* src = j < ca ? a + p * a_ld + j : b + p * b_ld + (j - ca);
        float* dst = out + p * out_ld + j;
        if (VEC) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
        else *dst = *src;
    }
}

extern "C" int nfdp_concat_channels_f32(const float* a, int a_ld, int ca, const float* b, int b_ld, int cb, float* out, int out_ld,
                                        long npix, nfdp_stream_t stream) {
    NFDP_REQUIRE(a && b && out, NFDP_EINVAL, "concat_channels: null pointer");
    NFDP_REQUIRE(npix >= 0 && ca > 0 && cb > 0 && a_ld >= ca && b_ld >= cb && out_ld >= ca + cb, NFDP_ESHAPE,
                 "concat_channels: bad shape");
    if (npix == 0) return NFDP_OK;
    const bool vec = ((ca | cb | a_ld | b_ld | out_ld) & 3) == 0 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0;
    if (vec) {
        const long total = npix * ((ca + cb) / 4);
        LAUNCH_1D(concat_channels_kernel<true>, total, a, a_ld, ca, b, b_ld, cb, out, out_ld, total);
    } else {
        const long total = npix * (ca + cb);
        LAUNCH_1D(concat_channels_kernel<false>, total, a, a_ld, ca, b, b_ld, cb, out, out_ld, total);
    }
    NFDP_CHECK_LAUNCH("concat_channels");
    return NFDP_OK;
}

extern "C" int nfdp_preprocess_u8_nchw_to_nhwc_f32(const uint8_t* u8, const float* noise_nchw, float* y, int B, int C,
                                                   int H, int W, int n_bits, nfdp_stream_t stream) {
    NFDP_REQUIRE(u8 && y, NFDP_EINVAL, "preprocess: null pointer");
    NFDP_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, NFDP_ESHAPE, "preprocess: bad shape");
    NFDP_REQUIRE(n_bits >= 1 && n_bits <= 8, NFDP_EINVAL, "preprocess: n_bits=%d outside [1,8]", n_bits);
    long total = (long)B * C * H * W;
    if (total == 0) return NFDP_OK;
    LAUNCH_1D(preprocess_kernel, total, u8, noise_nchw, y, total, C, H * W, n_bits);
    NFDP_CHECK_LAUNCH("preprocess");
    return NFDP_OK;
}

extern "C" int nfdp_preprocess_f32(const float* x, const float* noise, float* y, long n, int n_bits, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y, NFDP_EINVAL, "preprocess_f32: null pointer");
    NFDP_REQUIRE(n >= 0, NFDP_ESHAPE, "preprocess_f32: bad size");
    NFDP_REQUIRE(n_bits >= 1 && n_bits <= 8, NFDP_EINVAL, "preprocess_f32: n_bits=%d outside [1,8]", n_bits);
    if (n == 0) return NFDP_OK;
    LAUNCH_1D(preprocess_f32_kernel, n, x, noise, y, n, n_bits);
    NFDP_CHECK_LAUNCH("preprocess_f32");
    return NFDP_OK;
}

extern "C" int nfdp_postprocess_nhwc_to_u8_nchw(const float* x, uint8_t* u8, int B, int C, int H, int W, float n_bins,
                                                nfdp_stream_t stream) {
    NFDP_REQUIRE(x && u8, NFDP_EINVAL, "postprocess: null pointer");
    NFDP_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && n_bins > 0, NFDP_ESHAPE, "postprocess: bad shape");
    long total = (long)B * C * H * W;
    if (total == 0) return NFDP_OK;
    LAUNCH_1D(postprocess_kernel, total, x, u8, total, C, H * W, n_bins);
    NFDP_CHECK_LAUNCH("postprocess");
    return NFDP_OK;
}

extern "C" int nfdp_im2col3x3_bf16(const float* x, int x_ld, void* out, int B, int H, int W, int C, int Kpad, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && out, NFDP_EINVAL, "im2col3x3: null pointer");
    NFDP_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && x_ld >= C && Kpad >= 9 * C && Kpad % 8 == 0, NFDP_ESHAPE,
                 "im2col3x3: bad shape C=%d x_ld=%d Kpad=%d", C, x_ld, Kpad);
    NFDP_REQUIRE(((uintptr_t)out & 15) == 0, NFDP_ESHAPE, "im2col3x3: out must be 16-byte aligned");
    const long nchunks = (long)B * H * W * (Kpad / 8);
    if (nchunks == 0) return NFDP_OK;
    LAUNCH_1D(im2col3x3_bf16_kernel, nchunks, x, x_ld, H, W, C, reinterpret_cast<__bf16*>(out), Kpad, nchunks);
    NFDP_CHECK_LAUNCH("im2col3x3");
    return NFDP_OK;
}
