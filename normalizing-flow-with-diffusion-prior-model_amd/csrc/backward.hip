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
    const bool pointwise = p.k == 1;                          // 1x1: no taps, no pixel decomposition (64-bit div/mod per step)
#pragma unroll 4                                             // loads of several k-steps in flight (small tiles are latency-bound)
    for (long p0 = p_begin; p0 < p_end; p0 += 2) {
        const long px = p0 + kk;
        const bool pv = px < p_end;
        float a[MT], bv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = (pv && m_ok[i]) ? p.dy[px * p.dy_ld + m_idx[i]] : 0.f;
        if (pointwise) {
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = (pv && n_ok[j]) ? p.x[px * p.x_ld + n_ci[j]] : 0.f;
        } else {
            const long b = px / HW;
            const int r = (int)(px - b * HW);
            const int y = r / p.W, xx = r - y * p.W;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int yy = y + n_dy[j], xc = xx + n_dx[j];
                const bool ok = pv && n_ok[j] && (unsigned)yy < (unsigned)p.H && (unsigned)xc < (unsigned)p.W;
                bv[j] = ok ? p.x[((b * p.H + yy) * p.W + xc) * p.x_ld + n_ci[j]] : 0.f;
            }
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

// dw[o][ci][ky][kx] (OIHW) = sum over the pixel slices.  32 outputs x 8 lanes per workgroup: lane g sums slices g, g+8, ...
// and the 8 lane sums are combined in lane order (fixed order -> deterministic; a tiny CxC output with 2048 slices would
// otherwise be one workgroup walking them serially).
__global__ void __launch_bounds__(kBlock) wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int ksplit,
                                                             int Mpad, int Npad, int Cout, int Cin, int k) {
    __shared__ float red[8][32];
    const long total = (long)Cout * Cin * k * k;
    const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
    for (long i0 = (long)blockIdx.x * 32; i0 < total; i0 += (long)gridDim.x * 32) {
        const long i = i0 + el;
        float s = 0.f;
        if (i < total) {
            const int kx = (int)(i % k);
            long r = i / k;
            const int ky = (int)(r % k);
            r /= k;
            const int ci = (int)(r % Cin);
            const int o = (int)(r / Cin);
            const int n = (ky * k + kx) * Cin + ci;
            for (int q = g; q < ksplit; q += 8) s += ws[((size_t)q * Mpad + o) * Npad + n];
        }
        __syncthreads();
        red[g][el] = s;
        __syncthreads();
        if (g == 0 && i < total) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += red[j][el];
            dw[i] = t;
        }
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
        // 4 rows per trip, all loads first: one row at a time leaves a single pair of 4-byte loads in flight per thread
        constexpr int UR = 4;
        for (long rb = r0 + ty; rb < r1; rb += (long)RT * UR) {
            float gv[UR][MAXC], yv[UR][MAXC];
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const long r = rb + (long)u * RT;
#pragma unroll
                for (int q = 0; q < MAXC; ++q) {
                    const int c = cb + q * CT + tx;
                    const bool ok = r < r1 && c < C;
                    gv[u][q] = ok ? dy[r * dy_ld + c] : 0.f;
                    yv[u][q] = (ok && y != nullptr) ? y[r * y_ld + c] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const long r = rb + (long)u * RT;
#pragma unroll
                for (int q = 0; q < MAXC; ++q) {
                    const int c = cb + q * CT + tx;
                    if (r < r1 && c < C) {
                        float g = gv[u][q];
                        if (y != nullptr) {
                            if (relu && !(yv[u][q] > 0.f)) g = 0.f;
                            s_mul[q] += (double)__fmul_rn(g, yv[u][q]);
                        }
                        const float d = __fmul_rn(g, fac[q]);
                        if (dc != nullptr) dc[r * dc_ld + c] = d;
                        s_add[q] += (double)d;
                    }
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

// One workgroup per 32 channels: 8 lanes of partials per channel (lane g takes blocks g, g+8, ...), combined in lane order.
// (With one thread per channel the whole partial buffer - tens of MB for C = 512 - went through C/256 = 2 CUs.)
__global__ void __launch_bounds__(kBlock) scale_bwd_finish_kernel(const double* __restrict__ part, int nblk, int C, float f,
                                                                 float* __restrict__ dmul, float* __restrict__ dadd) {
    __shared__ double red[2][8][32];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int q = g; q < nblk; q += 8) {
            a += part[((size_t)q * 2 + 0) * C + c];
            b += part[((size_t)q * 2 + 1) * C + c];
        }
    red[0][g][cl] = a;
    red[1][g][cl] = b;
    __syncthreads();
    if (g == 0 && c < C) {
        double sa = 0.0, sb = 0.0;
        for (int t = 0; t < 8; ++t) { sa += red[0][t][cl]; sb += red[1][t][cl]; }
        if (dmul) dmul[c] = (float)(sa * (double)f);
        if (dadd) dadd[c] = (float)sb;
    }
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

// ----------------------------------------------------------------------------- conv weight gradient, bf16 MFMA
// Same GEMM (reduction over pixels), operands rounded to bf16, fp32 accumulation: v_mfma_f32_32x32x16_bf16 wants 8
// consecutive k (= pixels) per lane for one row/column (= channel), i.e. the TRANSPOSE of the NHWC tiles.  The fp32 tiles
// are read with coalesced float4 loads (4 channels of one pixel), converted, and written to LDS transposed
// ([channel][pixel], 16-bit scalar stores), so both MFMA operands are plain ds_read_b128.  A workgroup owns a 128 x 128
// block of dW and a slice of the pixels (32 per stage, register-staged double buffering); 4 waves as 2 x 2, each 2 x 2
// accumulator tiles.  Requires Cin % 4 == 0 (the 4 values of a float4 of the im2col row share a tap).
constexpr int WB_BM = 128, WB_BN = 128, WB_KC = 32, WB_LD = WB_KC + 8;     // LDS row stride in bf16 elements (80 B)

__global__ void __launch_bounds__(kBlock) wgrad_bf16_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) __bf16 At[2][WB_BM][WB_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bt[2][WB_BN][WB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
    const int nb = bid % p.nblocks; bid /= p.nblocks;
    const int mb = bid % p.mblocks;
    const int ks = bid / p.mblocks;
    const int pad = p.k >> 1, HW = p.H * p.W;
    const long p_begin = (long)ks * p.px_per_split;
    const long p_end = (p_begin + p.px_per_split) < p.npix ? (p_begin + p.px_per_split) : p.npix;

    // this thread's 4 float4 of each operand tile: element e = (tid + 256 i) * 4 -> pixel e / 128, channel e % 128
    int a_m[4], b_ci[4], b_dy[4], b_dx[4];
    bool a_ok[4], b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c0 = ((tid + kBlock * i) * 4) % WB_BM;
        a_m[i] = mb * WB_BM + c0;
        a_ok[i] = a_m[i] < p.Cout;                          // Cout % 4 == 0 is required: the float4 is in or out as a whole
        const int n = nb * WB_BN + c0;
        b_ok[i] = n < p.N;
        const int tap = b_ok[i] ? n / p.Cin : 0;
        b_ci[i] = b_ok[i] ? n % p.Cin : 0;
        b_dy[i] = tap / p.k - pad;
        b_dx[i] = tap % p.k - pad;
    }
    float4 ra[4], rb[4];
    auto load_stage = [&](long p0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px_l = ((tid + kBlock * i) * 4) / WB_BM;             // 0..31
            const long px = p0 + px_l;
            const bool pv = px < p_end;
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pv && a_ok[i]) ra[i] = *reinterpret_cast<const float4*>(p.dy + px * p.dy_ld + a_m[i]);
            if (pv && b_ok[i]) {
                const long b = px / HW;
                const int r = (int)(px - b * HW);
                const int y = r / p.W + b_dy[i], xx = r % p.W + b_dx[i];
                if ((unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
                    rb[i] = *reinterpret_cast<const float4*>(p.x + ((b * p.H + y) * p.W + xx) * p.x_ld + b_ci[i]);
            }
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = (tid + kBlock * i) * 4;
            const int px_l = e / WB_BM, c0 = e % WB_BM;
            At[buf][c0 + 0][px_l] = (__bf16)ra[i].x; At[buf][c0 + 1][px_l] = (__bf16)ra[i].y;
            At[buf][c0 + 2][px_l] = (__bf16)ra[i].z; At[buf][c0 + 3][px_l] = (__bf16)ra[i].w;
            Bt[buf][c0 + 0][px_l] = (__bf16)rb[i].x; Bt[buf][c0 + 1][px_l] = (__bf16)rb[i].y;
            Bt[buf][c0 + 2][px_l] = (__bf16)rb[i].z; Bt[buf][c0 + 3][px_l] = (__bf16)rb[i].w;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int c32 = lane & 31, g = lane >> 5;
    load_stage(p_begin);
    store_stage(0);
    __syncthreads();
    int buf = 0;
    for (long p0 = p_begin; p0 < p_end; p0 += WB_KC) {
        const bool more = p0 + WB_KC < p_end;
        if (more) load_stage(p0 + WB_KC);                  // global loads of the next stage fly under the MFMAs
#pragma unroll
        for (int kstep = 0; kstep < WB_KC / 16; ++kstep) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(&At[buf][wm * 64 + i * 32 + c32][kstep * 16 + 8 * g]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(&Bt[buf][wn * 64 + j * 32 + c32][kstep * 16 + 8 * g]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* out = p.ws + (size_t)ks * p.Mpad * p.Npad;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb * WB_BM + wm * 64 + i * 32 + drow32(r, g);
                const int n = nb * WB_BN + wn * 64 + j * 32 + c32;
                out[(size_t)m * p.Npad + n] = acc[i][j][r];
            }
}

// ----------------------------------------------------------------------------- tap shift-sum (narrow 3x3 convs)
// A 3x3 conv with few output channels spends its time staging im2col rows.  Computed instead as a pointwise GEMM
// T[p][tap*Co + co] = sum_ci x[p][ci] W[co][ci][tap] (a wide 1x1 conv: fast) followed by
//   out[p][co] = act( (sum_tap T[p + d(tap)][tap*Co + co] + add[co]) * exp(f * mul[co]) ),  zero outside the image,
// which is this kernel (the same decomposition as the fused flow-step kernel's GEMM3 + shift-sum).
__global__ void tap_shift_sum_kernel(const float* __restrict__ T, float* __restrict__ out, long total, int H, int W, int Co,
                                     const float* __restrict__ add, const float* __restrict__ mul, float f, int relu) {
    const int ld = 9 * Co;
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int co = (int)(i % Co);
        const long px = i / Co;
        const int xx = (int)(px % W);
        const int y = (int)((px / W) % H);
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if ((unsigned)(y + dy) < (unsigned)H && (unsigned)(xx + dx) < (unsigned)W)
                acc += T[(px + (long)dy * W + dx) * ld + tap * Co + co];
        }
        if (add) acc = __fadd_rn(acc, add[co]);
        if (mul) acc = __fmul_rn(acc, expf(__fmul_rn(f, mul[co])));
        if (relu) acc = fmaxf(acc, 0.f);
        out[i] = acc;
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
    // ~2048 waves on the chip (256 CUs x 4 SIMDs x 2), at least 64 pixels per slice; small outputs (the CxC gradient of
    // the 1x1 flow convolution) get up to 2048 slices: one wave walks its slice serially, a load latency per step
    long want = 2048 / ((long)p->mblocks * p->nblocks);
    if (want < 1) want = 1;
    const long cap = ((long)p->Mpad * p->Npad <= 64 * 64) ? 2048 : 256;
    if (want > cap) want = cap;
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
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(stream_grid(total, 32)), dim3(kBlock), 0, s, p.ws, dw, p.ksplit, p.Mpad,
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
    long g = cdiv(npix, 32);
    if (g > 2048) g = 2048;
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
    hipLaunchKernelGGL(scale_bwd_finish_kernel, dim3((int)cdiv(C, 32)), dim3(kBlock), 0, s, part, nblk, C, f, dmul, dadd);
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

static void wgrad_bf16_plan(int B, int H, int W, int Cin, int Cout, int k, WgradParams* p) {
    p->N = k * k * Cin;
    p->mblocks = (int)cdiv(Cout, WB_BM);
    p->nblocks = (int)cdiv(p->N, WB_BN);
    p->Mpad = p->mblocks * WB_BM;
    p->Npad = p->nblocks * WB_BN;
    p->npix = (long)B * H * W;
    long want = 1024 / ((long)p->mblocks * p->nblocks);     // ~4 workgroups per CU
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    long per = cdiv(p->npix, want);
    per = cdiv(per, WB_KC) * WB_KC;
    if (per < 4 * WB_KC) per = 4 * WB_KC;
    p->px_per_split = per;
    p->ksplit = (int)cdiv(p->npix, per);
}

extern "C" int nfdp_conv_wgrad_bf16_supported(int Cin, int Cout) { return Cin % 4 == 0 && Cout % 4 == 0; }

extern "C" size_t nfdp_conv_wgrad_bf16_ws_bytes(int B, int H, int W, int Cin, int Cout, int k) {
    WgradParams p;
    wgrad_bf16_plan(B, H, W, Cin, Cout, k, &p);
    return (size_t)p.ksplit * p.Mpad * p.Npad * sizeof(float);
}

extern "C" int nfdp_conv_wgrad_bf16(const float* x, int x_ld, const float* dy, int dy_ld, int B, int H, int W, int Cin, int Cout,
                                    int k, float* dw, void* ws, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && dy && dw && ws, NFDP_EINVAL, "conv_wgrad_bf16: null pointer");
    NFDP_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && k >= 1 && k <= 7 && (k & 1), NFDP_ESHAPE,
                 "conv_wgrad_bf16: bad shape B=%d H=%d W=%d Cin=%d Cout=%d k=%d", B, H, W, Cin, Cout, k);
    NFDP_REQUIRE(nfdp_conv_wgrad_bf16_supported(Cin, Cout), NFDP_EUNSUPPORTED,
                 "conv_wgrad_bf16: Cin=%d and Cout=%d must be multiples of 4 (use nfdp_conv_wgrad_f32)", Cin, Cout);
    NFDP_REQUIRE(x_ld >= Cin && dy_ld >= Cout && x_ld % 4 == 0 && dy_ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0,
                 NFDP_ESHAPE, "conv_wgrad_bf16: operands must be 16-byte aligned with pixel strides that are multiples of 4");
    WgradParams p;
    wgrad_bf16_plan(B, H, W, Cin, Cout, k, &p);
    p.x = x; p.dy = dy; p.ws = reinterpret_cast<float*>(ws);
    p.x_ld = x_ld; p.dy_ld = dy_ld; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.k = k;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(wgrad_bf16_kernel, dim3(p.mblocks * p.nblocks * p.ksplit), dim3(kBlock), 0, s, p);
    const long total = (long)Cout * Cin * k * k;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(stream_grid(total, 32)), dim3(kBlock), 0, s, p.ws, dw, p.ksplit, p.Mpad,
                       p.Npad, Cout, Cin, k);
    NFDP_CHECK_LAUNCH("conv_wgrad_bf16");
    return NFDP_OK;
}

extern "C" int nfdp_tap_shift_sum_f32(const float* T, float* out, int B, int H, int W, int Co, const float* add, const float* mul,
                                      float mul_exp, int relu, nfdp_stream_t stream) {
    NFDP_REQUIRE(T && out, NFDP_EINVAL, "tap_shift_sum: null pointer");
    NFDP_REQUIRE(B >= 0 && H > 0 && W > 0 && Co > 0, NFDP_ESHAPE, "tap_shift_sum: bad shape");
    const long total = (long)B * H * W * Co;
    if (total == 0) return NFDP_OK;
    hipLaunchKernelGGL(tap_shift_sum_kernel, dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream), T, out, total, H, W,
                       Co, add, mul, mul_exp, relu);
    NFDP_CHECK_LAUNCH("tap_shift_sum");
    return NFDP_OK;
}
