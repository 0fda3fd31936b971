// Backward kernels of the flow path (training step: loss.backward() of normalizing_flow/trainer.py:163-164), exact fp32.
//
// The reference differentiates through torch autograd; the derivatives restated here are those of
//   ActNorm            y = e^s (x + b)                                  transforms.py:80
//   Conv2dActNorm+ReLU y = relu(e^s (conv(x) + b))                      utils.py:67-69, :83-87
//   ZeroConv2d         y = (conv(x) + b) e^{3 logs}                     utils.py:43-44
//   AffineCoupling     s = sigmoid(h_ls + 2), y_b = (x_b + h_t) s, ldj += sum log(s + 1e-6)   transforms.py:179-185
//   IsotropicGaussian  logp += sum -0.5 (log 2pi + 2 logs + (z - mean)^2 e^{-2 logs})          prior.py:36-37
//   conv2d             weight gradient (data gradient = the forward conv kernel on flipped/transposed weights)
//
// Design for gfx950:
//  * weight gradient dW[o][tap,ci] = sum_p dy[p][o] x[p(+)tap][ci] is a GEMM whose reduction runs over PIXELS, the slow
//    dimension of both NHWC operands.  v_mfma_f32_32x32x2_f32 takes exactly that shape from memory: lane l supplies
//    A[l%32][l/32] and B[l/32][l%32], i.e. 32 consecutive channels of pixel p0 + l/32 - one coalesced 128-B row per
//    half-wave, no LDS transpose, exact fp32 products.  A wave owns up to 4x4 accumulator tiles (128 x 128 outputs, all
//    256 AGPRs) and a slice of the pixels; the slices are summed by a second kernel in a fixed order (deterministic).
//  * everything else is HBM-bound: one pass that writes the input gradient and accumulates the per-channel parameter
//    gradients (fp64 partials per workgroup, fixed-order finish).
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr int drow32(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ----------------------------------------------------------------------------- conv weight gradient
struct WgradParams {
    const float* x;      // [B,H,W,Cin] pixel stride x_ld
    const float* dy;     // [B,H,W,Cout] pixel stride dy_ld
    float* ws;           // [ksplit][Mpad][Npad] partial sums
    int x_ld, dy_ld, B, H, W, Cin, Cout, k, N;   // N = k*k*Cin
    int mblocks, nblocks, ksplit, Mpad, Npad;
    long npix, px_per_split;
};

template <int MT, int NT>
__global__ void __launch_bounds__(64) wgrad_kernel(const WgradParams p) {
    const int lane = threadIdx.x, c32 = lane & 31, kk = lane >> 5;
    int bid = blockIdx.x;
    const int nb = bid % p.nblocks; bid /= p.nblocks;
    const int mb = bid % p.mblocks;
    const int ks = bid / p.mblocks;
    const int pad = p.k >> 1, HW = p.H * p.W;

    int m_idx[MT];
    bool m_ok[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        m_idx[i] = (mb * MT + i) * 32 + c32;
        m_ok[i] = m_idx[i] < p.Cout;
    }
    int n_ci[NT], n_dy[NT], n_dx[NT];
    bool n_ok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (nb * NT + j) * 32 + c32;
        n_ok[j] = n < p.N;
        const int tap = n_ok[j] ? n / p.Cin : 0;
        n_ci[j] = n_ok[j] ? n % p.Cin : 0;
        n_dy[j] = tap / p.k - pad;
        n_dx[j] = tap % p.k - pad;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const long p_begin = (long)ks * p.px_per_split;
    const long p_end = (p_begin + p.px_per_split) < p.npix ? (p_begin + p.px_per_split) : p.npix;
    for (long p0 = p_begin; p0 < p_end; p0 += 2) {
        const long px = p0 + kk;
        const bool pv = px < p_end;
        const long b = px / HW;
        const int r = (int)(px - b * HW);
        const int y = r / p.W, xx = r - y * p.W;
        float a[MT], bv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = (pv && m_ok[i]) ? p.dy[px * p.dy_ld + m_idx[i]] : 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int yy = y + n_dy[j], xc = xx + n_dx[j];
            const bool ok = pv && n_ok[j] && (unsigned)yy < (unsigned)p.H && (unsigned)xc < (unsigned)p.W;
            bv[j] = ok ? p.x[((b * p.H + yy) * p.W + xc) * p.x_ld + n_ci[j]] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[j], acc[i][j], 0, 0, 0);
    }
    float* out = p.ws + (size_t)ks * p.Mpad * p.Npad;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (mb * MT + i) * 32 + drow32(r, kk);
                const int n = (nb * NT + j) * 32 + c32;
                out[(size_t)m * p.Npad + n] = acc[i][j][r];
            }
}

// dw[o][ci][ky][kx] (OIHW) = sum over the pixel slices, in slice order
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int ksplit, int Mpad, int Npad,
                                    int Cout, int Cin, int k) {
    const long total = (long)Cout * Cin * k * k;
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int kx = (int)(i % k);
        long r = i / k;
        const int ky = (int)(r % k);
        r /= k;
        const int ci = (int)(r % Cin);
        const int o = (int)(r / Cin);
        const int n = (ky * k + kx) * Cin + ci;
        float s = 0.f;
        for (int q = 0; q < ksplit; ++q) s += ws[((size_t)q * Mpad + o) * Npad + n];
        dw[i] = s;
    }
}

// wt[i][o][ky][kx] = w[o][i][k-1-ky][k-1-kx]: the forward conv kernel on wt is the data gradient of conv(., w)
__global__ void flip_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int O, int I, int k) {
    const long total = (long)O * I * k * k;
    for (long e = blockIdx.x * (long)kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const int kx = (int)(e % k);
        long r = e / k;
        const int ky = (int)(r % k);
        r /= k;
        const int o = (int)(r % O);
        const int i = (int)(r / O);
        wt[e] = w[(((long)o * I + i) * k + (k - 1 - ky)) * k + (k - 1 - kx)];
    }
}

// ----------------------------------------------------------------------------- scale / ReLU backward + channel sums
// dc[p][c] = dy[p][c] * [y[p][c] > 0 if relu] * exp(f * mul[c])          (dc nullable; mul nullable = factor 1)
// part[blk][0][c] = sum_p dy*y (masked)      part[blk][1][c] = sum_p dc   (y nullable: first sum skipped)
template <int CT>
__global__ void __launch_bounds__(kBlock) scale_bwd_kernel(const float* __restrict__ dy, int dy_ld, const float* __restrict__ y,
                                                          int y_ld, const float* __restrict__ mul, float f, int relu,
                                                          long npix, int C, float* __restrict__ dc, int dc_ld,
                                                          double* __restrict__ part, long rows_per_block) {
    constexpr int RT = kBlock / CT;
    constexpr int MAXC = 4;
    __shared__ double red[2][RT][CT];
    const int tx = threadIdx.x % CT, ty = threadIdx.x / CT;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = (r0 + rows_per_block) < npix ? (r0 + rows_per_block) : npix;
    for (int cb = 0; cb < C; cb += CT * MAXC) {
        double s_mul[MAXC], s_add[MAXC];
        float fac[MAXC];
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            s_mul[q] = 0.0; s_add[q] = 0.0;
            const int c = cb + q * CT + tx;
            fac[q] = (mul != nullptr && c < C) ? expf(__fmul_rn(f, mul[c])) : 1.0f;
        }
        for (long r = r0 + ty; r < r1; r += RT) {
#pragma unroll
            for (int q = 0; q < MAXC; ++q) {
                const int c = cb + q * CT + tx;
                if (c < C) {
                    float g = dy[r * dy_ld + c];
                    float yv = 0.f;
                    if (y != nullptr) {
                        yv = y[r * y_ld + c];
                        if (relu && !(yv > 0.f)) g = 0.f;
                        s_mul[q] += (double)__fmul_rn(g, yv);
                    }
                    const float d = __fmul_rn(g, fac[q]);
                    if (dc != nullptr) dc[r * dc_ld + c] = d;
                    s_add[q] += (double)d;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < MAXC; ++q) {
            __syncthreads();
            red[0][ty][tx] = s_mul[q];
            red[1][ty][tx] = s_add[q];
            __syncthreads();
            const int c = cb + q * CT + tx;
            if (ty == 0 && c < C) {
                double a = 0.0, b = 0.0;
                for (int t = 0; t < RT; ++t) { a += red[0][t][tx]; b += red[1][t][tx]; }
                part[((size_t)blockIdx.x * 2 + 0) * C + c] = a;
                part[((size_t)blockIdx.x * 2 + 1) * C + c] = b;
            }
        }
    }
}

__global__ void scale_bwd_finish_kernel(const double* __restrict__ part, int nblk, int C, float f, float* __restrict__ dmul,
                                        float* __restrict__ dadd) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int q = 0; q < nblk; ++q) {
        a += part[((size_t)q * 2 + 0) * C + c];
        b += part[((size_t)q * 2 + 1) * C + c];
    }
    if (dmul) dmul[c] = (float)(a * (double)f);
    if (dadd) dadd[c] = (float)b;
}

// ----------------------------------------------------------------------------- coupling backward
__device__ __forceinline__ float sigmoid_p2(float log_s) {
    const float v = __fadd_rn(log_s, 2.0f);
    return 1.0f / (1.0f + expf(-v));
}

// x: tensor BEFORE the coupling (x_a | x_b), h = (log_s | t), dy: gradient w.r.t. (y_a | y_b), gl[b]: gradient w.r.t.
// the per-sample log-det increment.  dx = (dy_a | dy_b s);  dh = (ds s (1-s) | dy_b s),  ds = dy_b (x_b + t) + gl/(s+1e-6)
__global__ void coupling_bwd_kernel(const float* __restrict__ x, const float* __restrict__ h, const float* __restrict__ dy,
                                    const double* __restrict__ gl, long total, int HW, int C, float* __restrict__ dx,
                                    float* __restrict__ dh) {
    const int Ch = C >> 1;
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int c = (int)(i % Ch);
        const long pxl = i / Ch;
        const long row = pxl * C;
        const float g = gl ? (float)gl[pxl / HW] : 0.f;
        const float s = sigmoid_p2(h[row + c]);
        const float t = h[row + Ch + c];
        const float dyb = dy[row + Ch + c];
        const float dt = __fmul_rn(dyb, s);
        const float ds = __fadd_rn(__fmul_rn(dyb, __fadd_rn(x[row + Ch + c], t)), __fdiv_rn(g, __fadd_rn(s, 1e-6f)));
        dx[row + c] = dy[row + c];
        dx[row + Ch + c] = dt;
        dh[row + Ch + c] = dt;
        dh[row + c] = __fmul_rn(__fmul_rn(ds, s), __fsub_rn(1.0f, s));
    }
}

// ----------------------------------------------------------------------------- Gaussian log-density backward
// dz = gl (-(z-m) e^{-2 ls});  dmean = -dz;  dlogs = gl (-1 + (z-m)^2 e^{-2 ls});  dh is always per element [B,HW,2Cz]
template <int H_MODE>
__global__ void gauss_logp_bwd_kernel(const float* __restrict__ z, int z_ld, const float* __restrict__ h,
                                      const double* __restrict__ gl, long total, int HW, int Cz, float* __restrict__ dz,
                                      float* __restrict__ dh) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int c = (int)(i % Cz);
        const long pxl = i / Cz;
        const float g = (float)gl[pxl / HW];
        const float mean = H_MODE == 0 ? h[pxl * 2 * Cz + c] : h[c];
        const float logs = H_MODE == 0 ? h[pxl * 2 * Cz + Cz + c] : h[Cz + c];
        const float d = __fsub_rn(z[pxl * z_ld + c], mean);
        const float e = expf(__fmul_rn(-2.0f, logs));
        const float de = __fmul_rn(d, e);
        dz[i] = __fmul_rn(g, -de);
        dh[pxl * 2 * Cz + c] = __fmul_rn(g, de);
        dh[pxl * 2 * Cz + Cz + c] = __fmul_rn(g, __fadd_rn(-1.0f, __fmul_rn(d, de)));
    }
}

template <int MT, int NT>
void launch_wgrad(const WgradParams& p, hipStream_t s) {
    hipLaunchKernelGGL((wgrad_kernel<MT, NT>), dim3(p.mblocks * p.nblocks * p.ksplit), dim3(64), 0, s, p);
}

}  // namespace

// tile shape: up to 4x4 accumulator tiles per wave
static void wgrad_tiles(int Cout, int N, int* MT, int* NT) {
    const int mt = (int)cdiv(Cout, 32), nt = (int)cdiv(N, 32);
    *MT = mt >= 4 ? 4 : (mt >= 2 ? 2 : 1);
    *NT = nt >= 4 ? 4 : (nt >= 2 ? 2 : 1);
}

static void wgrad_plan(int B, int H, int W, int Cin, int Cout, int k, WgradParams* p) {
    int MT, NT;
    p->N = k * k * Cin;
    wgrad_tiles(Cout, p->N, &MT, &NT);
    p->mblocks = (int)cdiv(Cout, 32 * MT);
    p->nblocks = (int)cdiv(p->N, 32 * NT);
    p->Mpad = p->mblocks * 32 * MT;
    p->Npad = p->nblocks * 32 * NT;
    p->npix = (long)B * H * W;
    // ~2048 waves on the chip (256 CUs x 4 SIMDs x 2), at least 64 pixels per slice, at most 256 slices
    long want = 2048 / ((long)p->mblocks * p->nblocks);
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    long per = cdiv(p->npix, want);
    if (per < 64) per = 64;
    per = (per + 1) & ~1L;
    p->px_per_split = per;
    p->ksplit = (int)cdiv(p->npix, per);
}

extern "C" size_t nfdp_conv_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int k) {
    WgradParams p;
    wgrad_plan(B, H, W, Cin, Cout, k, &p);
    return (size_t)p.ksplit * p.Mpad * p.Npad * sizeof(float);
}

extern "C" int nfdp_conv_wgrad_f32(const float* x, int x_ld, const float* dy, int dy_ld, int B, int H, int W, int Cin, int Cout,
                                   int k, float* dw, void* ws, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && dy && dw && ws, NFDP_EINVAL, "conv_wgrad: null pointer");
    NFDP_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && k >= 1 && k <= 7 && (k & 1), NFDP_ESHAPE,
                 "conv_wgrad: bad shape B=%d H=%d W=%d Cin=%d Cout=%d k=%d", B, H, W, Cin, Cout, k);
    NFDP_REQUIRE(x_ld >= Cin && dy_ld >= Cout, NFDP_ESHAPE, "conv_wgrad: pixel stride smaller than the channel count");
    WgradParams p;
    wgrad_plan(B, H, W, Cin, Cout, k, &p);
    p.x = x; p.dy = dy; p.ws = reinterpret_cast<float*>(ws);
    p.x_ld = x_ld; p.dy_ld = dy_ld; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.k = k;
    int MT, NT;
    wgrad_tiles(Cout, p.N, &MT, &NT);
    hipStream_t s = as_stream(stream);
    if (MT == 4 && NT == 4) launch_wgrad<4, 4>(p, s);
    else if (MT == 4 && NT == 2) launch_wgrad<4, 2>(p, s);
    else if (MT == 4 && NT == 1) launch_wgrad<4, 1>(p, s);
    else if (MT == 2 && NT == 4) launch_wgrad<2, 4>(p, s);
    else if (MT == 2 && NT == 2) launch_wgrad<2, 2>(p, s);
    else if (MT == 2 && NT == 1) launch_wgrad<2, 1>(p, s);
    else if (MT == 1 && NT == 4) launch_wgrad<1, 4>(p, s);
    else if (MT == 1 && NT == 2) launch_wgrad<1, 2>(p, s);
    else launch_wgrad<1, 1>(p, s);
    const long total = (long)Cout * Cin * k * k;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, s, p.ws, dw, p.ksplit, p.Mpad,
                       p.Npad, Cout, Cin, k);
    NFDP_CHECK_LAUNCH("conv_wgrad");
    return NFDP_OK;
}

extern "C" int nfdp_conv_flip_weights_f32(const float* w, float* wt, int O, int I, int k, nfdp_stream_t stream) {
    NFDP_REQUIRE(w && wt, NFDP_EINVAL, "conv_flip_weights: null pointer");
    NFDP_REQUIRE(O > 0 && I > 0 && k > 0, NFDP_ESHAPE, "conv_flip_weights: bad shape");
    const long total = (long)O * I * k * k;
    hipLaunchKernelGGL(flip_weights_kernel, dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream), w, wt, O, I, k);
    NFDP_CHECK_LAUNCH("conv_flip_weights");
    return NFDP_OK;
}

static int scale_bwd_blocks(long npix) {
    long g = cdiv(npix, 64);
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" size_t nfdp_scale_bwd_ws_bytes(long npix, int C) { return (size_t)scale_bwd_blocks(npix) * 2 * C * sizeof(double); }

extern "C" int nfdp_scale_bwd_f32(const float* dy, int dy_ld, const float* y, int y_ld, const float* mul, float f, int relu,
                                  long npix, int C, float* dc, int dc_ld, float* dmul, float* dadd, void* ws,
                                  nfdp_stream_t stream) {
    NFDP_REQUIRE(dy && ws, NFDP_EINVAL, "scale_bwd: null pointer");
    NFDP_REQUIRE(npix > 0 && C > 0 && C <= 4096 && dy_ld >= C, NFDP_ESHAPE, "scale_bwd: bad shape npix=%ld C=%d", npix, C);
    NFDP_REQUIRE((y == nullptr || y_ld >= C) && (dc == nullptr || dc_ld >= C), NFDP_ESHAPE, "scale_bwd: bad pixel stride");
    const int nblk = scale_bwd_blocks(npix);
    const long rows = cdiv(npix, nblk);
    double* part = reinterpret_cast<double*>(ws);
    hipStream_t s = as_stream(stream);
    if (C > 128)
        hipLaunchKernelGGL((scale_bwd_kernel<256>), dim3(nblk), dim3(kBlock), 0, s, dy, dy_ld, y, y_ld, mul, f, relu, npix, C, dc,
                           dc_ld, part, rows);
    else if (C > 32)
        hipLaunchKernelGGL((scale_bwd_kernel<64>), dim3(nblk), dim3(kBlock), 0, s, dy, dy_ld, y, y_ld, mul, f, relu, npix, C, dc,
                           dc_ld, part, rows);
    else
        hipLaunchKernelGGL((scale_bwd_kernel<16>), dim3(nblk), dim3(kBlock), 0, s, dy, dy_ld, y, y_ld, mul, f, relu, npix, C, dc,
                           dc_ld, part, rows);
    hipLaunchKernelGGL(scale_bwd_finish_kernel, dim3((int)cdiv(C, kBlock)), dim3(kBlock), 0, s, part, nblk, C, f, dmul, dadd);
    NFDP_CHECK_LAUNCH("scale_bwd");
    return NFDP_OK;
}

extern "C" int nfdp_coupling_bwd_f32(const float* x, const float* h, const float* dy, const double* gl, int B, int HW, int C,
                                     float* dx, float* dh, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && h && dy && dx && dh, NFDP_EINVAL, "coupling_bwd: null pointer");
    NFDP_REQUIRE(B >= 0 && HW > 0 && C > 0 && C % 2 == 0, NFDP_ESHAPE, "coupling_bwd: bad shape");
    const long total = (long)B * HW * (C / 2);
    if (total == 0) return NFDP_OK;
    hipLaunchKernelGGL(coupling_bwd_kernel, dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream), x, h, dy, gl,
                       total, HW, C, dx, dh);
    NFDP_CHECK_LAUNCH("coupling_bwd");
    return NFDP_OK;
}

extern "C" int nfdp_gauss_logp_bwd_f32(const float* z, int z_ld, const float* h, int h_mode, const double* gl, int B, int HW,
                                       int Cz, float* dz, float* dh, nfdp_stream_t stream) {
    NFDP_REQUIRE(z && h && gl && dz && dh, NFDP_EINVAL, "gauss_logp_bwd: null pointer");
    NFDP_REQUIRE(B >= 0 && HW > 0 && Cz > 0 && z_ld >= Cz, NFDP_ESHAPE, "gauss_logp_bwd: bad shape");
    NFDP_REQUIRE(h_mode == 0 || h_mode == 1, NFDP_EINVAL, "gauss_logp_bwd: h_mode=%d", h_mode);
    const long total = (long)B * HW * Cz;
    if (total == 0) return NFDP_OK;
    if (h_mode == 0)
        hipLaunchKernelGGL((gauss_logp_bwd_kernel<0>), dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream), z,
                           z_ld, h, gl, total, HW, Cz, dz, dh);
    else
        hipLaunchKernelGGL((gauss_logp_bwd_kernel<1>), dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream), z,
                           z_ld, h, gl, total, HW, Cz, dz, dh);
    NFDP_CHECK_LAUNCH("gauss_logp_bwd");
    return NFDP_OK;
}
