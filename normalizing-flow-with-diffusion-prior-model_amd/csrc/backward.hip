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
    const void* x;       // [B,H,W,Cin] pixel stride x_ld   (fp32; the bf16 kernel also takes bf16)
    const void* dy;      // [B,H,W,Cout] pixel stride dy_ld
    float* ws;           // [ksplit][Mpad][Npad] partial sums
    int a_blocked = 0, b_blocked = 0;   // wgrad_bf16_dma_kernel: dy / x is a 512-channel tensor in the fused kernels' blocked layout
    float* ws_db = nullptr;   // optional [ksplit][Mpad]: per-slice column sums of dy (wgrad_bf16_dma_kernel, n-block 0)
    int x_ld, dy_ld, B, H, W, Cin, Cout, k, N;   // N = k*k*Cin
    int mblocks, nblocks, ksplit, Mpad, Npad;
    long npix, px_per_split;
};

template <int MT, int NT>
__global__ void __launch_bounds__(64) wgrad_kernel(const WgradParams p) {
    const float* __restrict__ xf = reinterpret_cast<const float*>(p.x);
    const float* __restrict__ dyf = reinterpret_cast<const float*>(p.dy);
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
        for (int i = 0; i < MT; ++i) a[i] = (pv && m_ok[i]) ? dyf[px * p.dy_ld + m_idx[i]] : 0.f;
        if (pointwise) {
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = (pv && n_ok[j]) ? xf[px * p.x_ld + n_ci[j]] : 0.f;
        } else {
            const long b = px / HW;
            const int r = (int)(px - b * HW);
            const int y = r / p.W, xx = r - y * p.W;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int yy = y + n_dy[j], xc = xx + n_dx[j];
                const bool ok = pv && n_ok[j] && (unsigned)yy < (unsigned)p.H && (unsigned)xc < (unsigned)p.W;
                bv[j] = ok ? xf[((b * p.H + yy) * p.W + xc) * p.x_ld + n_ci[j]] : 0.f;
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

// Where a reduced element (m = dy channel, n = (tap, x channel) column of the padded [Mpad][Npad] slices) lands in dw, and whether
// it is added to what is there (gradient accumulation straight into the parameter's .grad):
//   dst = m*sm + (n % D)*sd + (n / D)*st + off   for m < Cout, n < N.
// form 0: dw[o][ci][ky][kx] (OIHW) of the conv itself.  form 1 (k = 1, x = im2col rows of a c-channel tensor): the 3x3 weight
// dw[o][c][3][3].  form 2 (k = 1, x = im2col rows of the c-channel OUTPUT gradient, dy = the conv's input): dw[c][Cout][3][3] with the
// taps mirrored (the zero conv's weight gradient from the im2col matrix of its output gradient).
struct WgradOut { int D, N; long sm, sd, st, off; int acc; };

static bool wgrad_out_map(int form, int c, int Cout, int Cin, int k, int acc, WgradOut* o) {
    o->acc = acc;
    if (form == 0) { o->D = Cin; o->N = k * k * Cin; o->sm = (long)Cin * k * k; o->sd = k * k; o->st = 1; o->off = 0; return true; }
    if (k != 1 || c <= 0 || 9 * c > Cin) return false;
    o->D = c; o->N = 9 * c;
    if (form == 1) { o->sm = 9L * c; o->sd = 9; o->st = 1; o->off = 0; return true; }
    if (form == 2) { o->sm = 9; o->sd = 9L * Cout; o->st = -1; o->off = 8; return true; }
    return false;
}

__device__ __forceinline__ void wgrad_put(float* __restrict__ dw, const WgradOut& o, int m, int n, float v) {
    float* d = dw + (m * o.sm + (n % o.D) * o.sd + (n / o.D) * o.st + o.off);
    *d = o.acc ? *d + v : v;
}

// Sum over the pixel slices.  256/LPO outputs x LPO lanes per workgroup: lane g sums slices g, g+LPO, ... and the lane sums are
// combined in lane order (fixed order -> deterministic; a tiny CxC output with 2048 slices would otherwise be one workgroup walking
// them serially).  LPO = 64 for the tiny outputs, 8 otherwise.
template <int LPO>
__global__ void __launch_bounds__(kBlock) wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int ksplit,
                                                             int Mpad, int Npad, int Cout, WgradOut o) {
    constexpr int OPW = kBlock / LPO;
    __shared__ float red[LPO][OPW];
    const long total = (long)Cout * o.N;
    const int el = threadIdx.x % OPW, g = threadIdx.x / OPW;
    for (long i0 = (long)blockIdx.x * OPW; i0 < total; i0 += (long)gridDim.x * OPW) {
        const long i = i0 + el;
        const int m = (int)(i / o.N), n = (int)(i % o.N);
        float s = 0.f;
        if (i < total)
            for (int q = g; q < ksplit; q += LPO) s += ws[((size_t)q * Mpad + m) * Npad + n];
        __syncthreads();
        red[g][el] = s;
        __syncthreads();
        if (g == 0 && i < total) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < LPO; ++j) t += red[j][el];
            wgrad_put(dw, o, m, n, t);
        }
    }
}

// The same sum for the larger outputs, walking the padded [Mpad][Npad] slices as they lie in memory: a wave reads 1 KB
// contiguous per slice (float4 per lane), the 4 waves take slices w, w+4, ... and are combined in wave order.
__global__ void __launch_bounds__(kBlock) wgrad_reduce_vec_kernel(const float* __restrict__ ws, float* __restrict__ dw, int ksplit,
                                                                 int Mpad, int Npad, int Cout, WgradOut o) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long slice = (long)Mpad * Npad;
    const long e0 = ((long)blockIdx.x * 64 + lane) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e0 < slice) {
        int q = wave;
        for (; q + 28 < ksplit; q += 32) {                        // 8 loads in flight per lane
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(ws + (size_t)(q + 4 * u) * slice + e0);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; q < ksplit; q += 4) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)q * slice + e0);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && e0 < slice) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) { t[0] += red[w][lane].x; t[1] += red[w][lane].y; t[2] += red[w][lane].z; t[3] += red[w][lane].w; }
        const int m = (int)(e0 / Npad), n0 = (int)(e0 % Npad);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m < Cout && n0 + j < o.N) wgrad_put(dw, o, m, n0 + j, t[j]);
    }
}

static void launch_wgrad_reduce(const float* ws, float* dw, int ksplit, int Mpad, int Npad, int Cout, const WgradOut& o, hipStream_t s) {
    const long slice = (long)Mpad * Npad, total = (long)Cout * o.N;
    if (slice >= 16384 && Npad % 4 == 0)
        hipLaunchKernelGGL(wgrad_reduce_vec_kernel, dim3((int)cdiv(slice, 256)), dim3(kBlock), 0, s, ws, dw, ksplit, Mpad, Npad, Cout, o);
    else if (total <= 4096 && ksplit >= 256)
        hipLaunchKernelGGL((wgrad_reduce_kernel<64>), dim3(stream_grid(total, 4)), dim3(kBlock), 0, s, ws, dw, ksplit, Mpad, Npad, Cout, o);
    else
        hipLaunchKernelGGL((wgrad_reduce_kernel<8>), dim3(stream_grid(total, 32)), dim3(kBlock), 0, s, ws, dw, ksplit, Mpad, Npad, Cout, o);
}

// The slice sum for a layer y = act((conv(x, W) + b) * exp(s)) whose raw-output gradient dc is the GEMM's dy operand, one workgroup per
// output row m (= channel): besides dW[m][:] it yields the layer's ActNorm gradients without another pass over the activations -
//   db[m] = sum_p dc[p][m]                        (column sums from wgrad_bf16_dma_kernel's ones-fragment MFMAs, ws_db)
//   ds[m] = sum_p dy*y = sum_p dc[p][m] (u[p][m] + b[m]) = sum_n W[m][n] dW[m][n] + b[m] db[m]      (u = conv output = W x)
// Fixed summation order (slices in order per thread, LDS tree over the threads): deterministic.
__device__ __forceinline__ float wgrad_get(const float* __restrict__ w, const WgradOut& o, int m, int n) {
    return w[m * o.sm + (n % o.D) * o.sd + (n / o.D) * o.st + o.off];
}

__device__ __forceinline__ void wgrad_reduce_row(const float* __restrict__ ws, const float* __restrict__ ws_db, float* __restrict__ dw,
                                                 int ksplit, int Mpad, int Npad, const WgradOut& o, const float* __restrict__ w,
                                                 const float* __restrict__ bias, float* __restrict__ dscale, float* __restrict__ dbias,
                                                 int m, float4* red, float* sdot, float* sdb) {
    // One workgroup per output row.  Thread t owns float4 column t % (Npad/4) and the slices sg, sg + ng, ... (sg = t / (Npad/4), ng =
    // 256 / (Npad/4) slice groups), 8 loads in flight per thread: the kernel streams ksplit * Npad floats per row and is otherwise
    // latency-bound (the first version, 4 loads in flight per lane and a serial bias-gradient loop, took 27-40 us per launch).
    const int t = threadIdx.x;
    const int nc4 = Npad / 4, ng = kBlock / nc4 > 0 ? kBlock / nc4 : 1;
    const int c = t % nc4, sg = t / nc4;
    const long slice = (long)Mpad * Npad;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sg < ng && t < ng * nc4) {
        const float* col = ws + (size_t)m * Npad + 4 * c;
        int q = sg;
        for (; q + 7 * ng < ksplit; q += 8 * ng) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(col + (size_t)(q + u * ng) * slice);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        for (; q < ksplit; q += ng) {
            const float4 v = *reinterpret_cast<const float4*>(col + (size_t)q * slice);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    red[t] = a;
    const bool affine = w != nullptr;                     // (uniform)
    float db = 0.f;
    if (affine)
        for (int q = t; q < ksplit; q += kBlock) db += ws_db[(size_t)q * Mpad + m];
    sdb[t] = db;
    __syncthreads();
    float dot = 0.f;
    if (t < nc4) {
        float4 s4 = red[t];
        for (int gq = 1; gq < ng; ++gq) { const float4 v = red[t + gq * nc4]; s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; }
        const float tv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * t + e < o.N) {
                if (affine) dot = fmaf(wgrad_get(w, o, m, 4 * t + e), tv[e], dot);
                wgrad_put(dw, o, m, 4 * t + e, tv[e]);
            }
    }
    if (!affine) return;
    sdot[t] = dot;
    __syncthreads();
    for (int sft = kBlock / 2; sft > 0; sft >>= 1) {              // fixed tree: deterministic
        if (t < sft) { sdot[t] += sdot[t + sft]; sdb[t] += sdb[t + sft]; }
        __syncthreads();
    }
    if (t == 0) {
        const float dbt = sdb[0];
        const float ds = fmaf(bias[m], dbt, sdot[0]);
        dscale[m] = o.acc ? dscale[m] + ds : ds;
        dbias[m] = o.acc ? dbias[m] + dbt : dbt;
    }
}

__global__ void __launch_bounds__(kBlock) wgrad_reduce_rows_kernel(const float* __restrict__ ws, const float* __restrict__ ws_db,
                                                                  float* __restrict__ dw, int ksplit, int Mpad, int Npad, WgradOut o,
                                                                  const float* __restrict__ w, const float* __restrict__ bias,
                                                                  float* __restrict__ dscale, float* __restrict__ dbias) {
    __shared__ float4 red[kBlock];
    __shared__ float sdot[kBlock], sdb[kBlock];
    wgrad_reduce_row(ws, ws_db, dw, ksplit, Mpad, Npad, o, w, bias, dscale, dbias, blockIdx.x, red, sdot, sdb);
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
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(__bf16* p, float v) { *p = (__bf16)v; }

// TDY / TY / TDC: fp32 or bf16 storage of the incoming gradient, the saved output and the produced gradient (bf16 training
// mode keeps the 512-channel hidden tensors and their gradients in bf16: they are bf16 MFMA operands anyway)
template <int CT, typename TDY, typename TY, typename TDC>
__global__ void __launch_bounds__(kBlock) scale_bwd_kernel(const TDY* __restrict__ dy, int dy_ld, const TY* __restrict__ y,
                                                          int y_ld, const float* __restrict__ mul, float f, int relu,
                                                          long npix, int C, TDC* __restrict__ dc, int dc_ld,
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
                    gv[u][q] = ok ? ldf(dy + r * dy_ld + c) : 0.f;
                    yv[u][q] = (ok && y != nullptr) ? ldf(y + r * y_ld + c) : 0.f;
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
                        if (dc != nullptr) stf(dc + r * dc_ld + c, d);
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

// Vector form for C % 8 == 0 (the 512-channel hidden tensors): a thread owns 8 consecutive channels of a pixel, i.e. 16-byte
// (bf16) or 2 x 16-byte (fp32) accesses.  The scalar kernel above issues one 4- or 2-byte access per lane and is bound by
// the number of memory instructions, not by HBM: halving the bytes (bf16 storage) left its time unchanged.
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const __bf16* p, float (&v)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)a[j];
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void st8(__bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = a;
}

template <int TPR, typename TDY, typename TY, typename TDC>
__global__ void __launch_bounds__(kBlock) scale_bwd_vec_kernel(const TDY* __restrict__ dy, int dy_ld, const TY* __restrict__ y,
                                                              int y_ld, const float* __restrict__ mul, float f, int relu,
                                                              long npix, int C, TDC* __restrict__ dc, int dc_ld,
                                                              double* __restrict__ part, long rows_per_block) {
    constexpr int RT = kBlock / TPR;
    constexpr int UR = 4;
    __shared__ double red[2][RT][TPR * 8];
    const int tx = threadIdx.x % TPR, ty = threadIdx.x / TPR;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = (r0 + rows_per_block) < npix ? (r0 + rows_per_block) : npix;
    const int C8 = C >> 3;
    for (int cb = 0; cb < C8; cb += TPR) {
        const int cg = cb + tx;
        const bool cok = cg < C8;
        const int c = cg * 8;
        double s_mul[8], s_add[8];
        float fac[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s_mul[j] = 0.0; s_add[j] = 0.0;
            fac[j] = (mul != nullptr && cok) ? expf(__fmul_rn(f, mul[c + j])) : 1.0f;
        }
        if (cok)
            for (long rb = r0 + ty; rb < r1; rb += (long)RT * UR) {
                float gv[UR][8], yv[UR][8];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const long r = rb + (long)u * RT;
                    if (r < r1) {
                        ld8(dy + r * dy_ld + c, gv[u]);
                        if (y != nullptr) ld8(y + r * y_ld + c, yv[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    const long r = rb + (long)u * RT;
                    if (r < r1) {
                        float d[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float g = gv[u][j];
                            if (y != nullptr) {
                                if (relu && !(yv[u][j] > 0.f)) g = 0.f;
                                s_mul[j] += (double)__fmul_rn(g, yv[u][j]);
                            }
                            d[j] = __fmul_rn(g, fac[j]);
                        }
                        if (dc != nullptr) st8(dc + r * dc_ld + c, d);
#pragma unroll
                        for (int j = 0; j < 8; ++j) s_add[j] += (double)d[j];
                    }
                }
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[0][ty][tx * 8 + j] = s_mul[j];
            red[1][ty][tx * 8 + j] = s_add[j];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 2 * TPR * 8; idx += kBlock) {
            const int which = idx / (TPR * 8), col = idx % (TPR * 8);
            const int ch = cb * 8 + col;
            if (ch < C) {
                double a = 0.0;
                for (int t = 0; t < RT; ++t) a += red[which][t][col];
                part[((size_t)blockIdx.x * 2 + which) * C + ch] = a;
            }
        }
    }
}

// One workgroup per 8 channels: 32 lanes of partials per channel (lane g takes blocks g, g+32, ...), combined in lane order.
// (With one thread per channel the whole partial buffer - MBs for C = 512 - went through C/256 = 2 CUs; the fused GEMM epilogue
// writes one partial row per 128 pixels, 8 MB at level 0, so the reads are spread over C/8 = 64 workgroups.)
__global__ void __launch_bounds__(kBlock) scale_bwd_finish_kernel(const double* __restrict__ part, int nblk, int C, float f,
                                                                 float* __restrict__ dmul, float* __restrict__ dadd, int acc) {
    __shared__ double red[2][32][8];
    const int cl = threadIdx.x & 7, g = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int q = g; q < nblk; q += 32) {
            a += part[((size_t)q * 2 + 0) * C + c];
            b += part[((size_t)q * 2 + 1) * C + c];
        }
    red[0][g][cl] = a;
    red[1][g][cl] = b;
    __syncthreads();
    if (g == 0 && c < C) {
        double sa = 0.0, sb = 0.0;
        for (int t = 0; t < 32; ++t) { sa += red[0][t][cl]; sb += red[1][t][cl]; }
        if (dmul) dmul[c] = (float)(sa * (double)f) + (acc ? dmul[c] : 0.f);       // acc: add to the parameter's .grad in place
        if (dadd) dadd[c] = (float)sb + (acc ? dadd[c] : 0.f);
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

// ----------------------------------------------------------------------------- link between two flow steps of the backward pass
// Between the reverse coupling network of flow i and that of flow i-1 the backward pass runs, per pixel, the flow head of i backwards
// (1x1 conv, ActNorm: transforms.py:132, :80) and the affine coupling of i-1 backwards (transforms.py:179-185) with the scale
// backward of its ZeroConv2d (utils.py:43-44) - seven launches at their floor in the per-op path (ActNorm recompute, W^T product,
// two scale backwards + two finishes, coupling backward).  Here: one launch, one thread per pixel with its C-channel rows in registers,
//   a   = e^s (xpre + b)                 (ActNorm output of flow i: written out for the 1x1 conv's weight gradient)
//   da  = W^T dx_in,   dy = da e^s       (dy = dx_in when there is no head: first call of a level; `wt` is W itself, row-major)
//   part: sum da a (-> dscale), sum da e^s (-> dbias)
//   coupling of i-1 on (ypost, hout, dy, gl) as coupling_bwd_kernel -> dx_out, dh;  dc3 = dh e^{f logs3}
//   part: sum dh hout (-> dlogs3 / f), sum dc3 (-> dbias3)                            (no coupling: dx_out = dy, last call of a level)
// and per-workgroup column sums in fp64 (fixed order: deterministic), reduced for all calls of a level by ONE finish launch.
struct LinkParams {
    const float* dx_in; const float* xpre; const float* wt; const float* scale; const float* bias; float* a_out;
    const float* ypost; const float* hout; const double* gl; const float* logs3; float f3;
    float* dx_out; float* dc3_out; double* part;
    long npix; int HW;
    float* part_w;       // [blocks][C*C] fp32: per-workgroup partial of the 1x1 conv's weight gradient (head calls), or null
};

// LB pixels (threads) per workgroup: 64 (one wave) - with 256 the 8x8 / 4x4 levels of a B = 512 step have 128 / 32 workgroups, most of
// the chip idle, and a workgroup's serial parts (the C x C product per pixel, the column sums over its rows, the weight-gradient MFMAs)
// set the launch time: 23 / 32 us against 17 / 25 us.
template <int C, int LB>
__global__ void __launch_bounds__(LB) flow_link_bwd_kernel(const LinkParams p) {
    constexpr int Ch = C / 2, TS = C + 1;
    __shared__ __attribute__((aligned(16))) float wt_s[C * C];
    __shared__ float es_s[C], bs_s[C], fac_s[C];
    extern __shared__ __attribute__((aligned(16))) float link_dyn[];       // tile [256][C+1] | tile2 [256][C+1] (ActNorm output rows)
    float* tile = link_dyn;
    float* tile2 = link_dyn + LB * TS;
    const int tid = threadIdx.x;
    const long px = (long)blockIdx.x * LB + tid;
    const bool live = px < p.npix;
    const bool head = p.xpre != nullptr, cpl = p.ypost != nullptr;
    if (head) {
        for (int i = tid; i < C * C; i += LB) wt_s[i] = p.wt[i];
        for (int c = tid; c < C; c += LB) { es_s[c] = expf(p.scale[c]); bs_s[c] = p.bias[c]; }
    }
    if (cpl)
        for (int c = tid; c < C; c += LB) fac_s[c] = expf(__fmul_rn(p.f3, p.logs3[c]));
    __syncthreads();
    // column sums over the workgroup's 256 pixels, one quantity at a time through a [256][C+1] tile (odd row stride: conflict-free)
    double* out = p.part + (size_t)blockIdx.x * 4 * C;
    auto colsum = [&](int q, const float (&v)[C]) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < C; ++c) tile[tid * TS + c] = live ? v[c] : 0.f;
        __syncthreads();
        if (tid < C) {
            double sum = 0.0;
            for (int r = 0; r < LB; ++r) sum += (double)tile[r * TS + tid];
            out[q * C + tid] = sum;
        }
    };
    float dy[C];
    if (head) {
        // da = W^T dx_in: the thread's dx row goes through its own LDS tile row so the product can walk the input channels with a
        // run-time index (a fully unrolled C x C product keeps every weight live: 8 KB of scratch per lane at C = 48)
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) v = *reinterpret_cast<const float4*>(p.dx_in + px * C + c);
            float* d = tile + tid * TS + c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        float da[C];
#pragma unroll
        for (int c = 0; c < C; ++c) da[c] = 0.f;
#pragma unroll 2
        for (int k = 0; k < C; ++k) {
            const float xv = tile[tid * TS + k];
            const float* wr = wt_s + k * C;                       // W[k][:] (row-major 1x1 weight): broadcast float4 reads
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                const float4 w4 = *reinterpret_cast<const float4*>(wr + c);
                da[c] = fmaf(w4.x, xv, da[c]); da[c + 1] = fmaf(w4.y, xv, da[c + 1]);
                da[c + 2] = fmaf(w4.z, xv, da[c + 2]); da[c + 3] = fmaf(w4.w, xv, da[c + 3]);
            }
        }
        float q0[C];
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) xv = *reinterpret_cast<const float4*>(p.xpre + px * C + c);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            float av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float es = es_s[c + j];
                av[j] = __fmul_rn(es, __fadd_rn(xs[j], bs_s[c + j]));
                dy[c + j] = __fmul_rn(da[c + j], es);
                q0[c + j] = __fmul_rn(da[c + j], av[j]);
                tile2[tid * TS + c + j] = av[j];
            }
            if (live && p.a_out) *reinterpret_cast<float4*>(p.a_out + px * C + c) = make_float4(av[0], av[1], av[2], av[3]);
        }
        if (p.part_w) {
            // The 1x1 conv's weight gradient dW[k][c] = sum_p dx_in[p][k] a[p][c] (transforms.py:132) of this workgroup's 256 pixels on
            // the exact-fp32 matrix cores: both row sets are in LDS (tile: dx_in, zero rows for absent pixels; tile2: a), a wave takes
            // 64 pixels, 2 per v_mfma_f32_32x32x2_f32 (lane = channel, lane half = pixel: consecutive LDS words per half-wave).  The
            // four wave partials are summed through LDS; nfdp_flow_link_finish_f32 sums the workgroups in fp64.  (Before: a separate
            // split-K GEMM + reduce per flow on the side stream, 96 launches and 2.2 ms of kernel time per step.)
            constexpr int MT = (C + 31) / 32;
            __syncthreads();
            const int lane = tid & 63, wave = tid >> 6, m0 = lane & 31, kk = lane >> 5;
            f32x16 accw[MT][MT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accw[i][j][r] = 0.f;
#pragma unroll 4
            for (int st = 0; st < 32; ++st) {
                const int prow = (64 * wave + 2 * st + kk) * TS;
                float av_[MT], bv_[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int chn = 32 * i + m0;
                    av_[i] = chn < C ? tile[prow + chn] : 0.f;
                    bv_[i] = chn < C ? tile2[prow + chn] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j) accw[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_[i], bv_[j], accw[i][j], 0, 0, 0);
            }
            __syncthreads();                                       // every wave is done reading the two tiles
            constexpr int NW = LB / 64;
            float* redw = tile;                                    // [NW waves][C*C], over the two tiles
            static_assert(NW * C * C <= 2 * LB * TS, "wave partials alias the tiles");
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * i + drow32(r, kk), col = 32 * j + m0;
                        if (row < C && col < C) redw[wave * C * C + row * C + col] = accw[i][j][r];
                    }
            __syncthreads();
            float* pw = p.part_w + (size_t)blockIdx.x * C * C;
            for (int e = tid; e < C * C; e += LB) {
                float v = redw[e];
#pragma unroll
                for (int wv = 1; wv < NW; ++wv) v += redw[wv * C * C + e];
                pw[e] = v;
            }
        }
        colsum(0, q0);
        colsum(1, dy);
    } else {
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) v = *reinterpret_cast<const float4*>(p.dx_in + px * C + c);
            dy[c] = v.x; dy[c + 1] = v.y; dy[c + 2] = v.z; dy[c + 3] = v.w;
        }
        if (tid < C) { out[tid] = 0.0; out[C + tid] = 0.0; }
    }
    if (cpl) {
        const float g = (p.gl && live) ? (float)p.gl[px / p.HW] : 0.f;
        float xb[Ch], hv[C];
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) v = *reinterpret_cast<const float4*>(p.hout + px * C + c);
            hv[c] = v.x; hv[c + 1] = v.y; hv[c + 2] = v.z; hv[c + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < Ch; c += (Ch % 4 == 0 ? 4 : 2)) {
            if constexpr (Ch % 4 == 0) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) v = *reinterpret_cast<const float4*>(p.ypost + px * C + Ch + c);
                xb[c] = v.x; xb[c + 1] = v.y; xb[c + 2] = v.z; xb[c + 3] = v.w;
            } else {
                float2 v = make_float2(0.f, 0.f);
                if (live) v = *reinterpret_cast<const float2*>(p.ypost + px * C + Ch + c);
                xb[c] = v.x; xb[c + 1] = v.y;
            }
        }
        float q2[C], dc3[C];
#pragma unroll
        for (int c = 0; c < Ch; ++c) {
            const float sg = sigmoid_p2(hv[c]);
            const float t = hv[Ch + c];
            const float dyb = dy[Ch + c];
            const float dt = __fmul_rn(dyb, sg);
            const float ds = __fadd_rn(__fmul_rn(dyb, __fadd_rn(xb[c], t)), __fdiv_rn(g, __fadd_rn(sg, 1e-6f)));
            const float dls = __fmul_rn(__fmul_rn(ds, sg), __fsub_rn(1.0f, sg));
            dy[Ch + c] = dt;                                   // dy now holds dx_out = (dy_a | dy_b s)
            dc3[c] = __fmul_rn(dls, fac_s[c]);
            dc3[Ch + c] = __fmul_rn(dt, fac_s[Ch + c]);
            q2[c] = __fmul_rn(dls, hv[c]);
            q2[Ch + c] = __fmul_rn(dt, hv[Ch + c]);
        }
        if (live) {
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                *reinterpret_cast<float4*>(p.dx_out + px * C + c) = make_float4(dy[c], dy[c + 1], dy[c + 2], dy[c + 3]);
                *reinterpret_cast<float4*>(p.dc3_out + px * C + c) = make_float4(dc3[c], dc3[c + 1], dc3[c + 2], dc3[c + 3]);
            }
        }
        colsum(2, q2);
        colsum(3, dc3);
    } else {
        if (live) {
#pragma unroll
            for (int c = 0; c < C; c += 4)
                *reinterpret_cast<float4*>(p.dx_out + px * C + c) = make_float4(dy[c], dy[c + 1], dy[c + 2], dy[c + 3]);
        }
        if (tid < C) { out[2 * C + tid] = 0.0; out[3 * C + tid] = 0.0; }
    }
}

// The same launch with TPP threads per pixel (C = 24: 2, C = 48: 4; one wave per channel part, 64 pixels per workgroup as above, so the
// partial buffers keep their layout).  With one thread per pixel the C = 48 launch of a B = 512 step is 128 one-wave workgroups whose serial
// parts - 2304 FMAs of the C x C product per lane, four 32x32x2 MFMA tiles over 32 steps, four column sums of 64 dependent LDS reads - set
// its 25 us (35 us beside the weight-gradient stream); here a thread forms C / TPP outputs of the product (the same k-ascending FMA chain
// per output), a wave takes one MFMA tile over all 64 pixels (the same step order per tile), and the four column sums run side by side,
// one per wave, after the rest (the same 64-row fp64 sums): bit-identical results.
template <int C, int TPP>
__global__ void __launch_bounds__(64 * TPP) flow_link_bwd_split_kernel(const LinkParams p) {
    constexpr int Ch = C / 2, TS = C + 1, CPT = C / TPP, CQ = Ch / TPP, NT = 64 * TPP;
    static_assert(C % TPP == 0 && Ch % TPP == 0 && CPT % 4 == 0 && CQ % 2 == 0, "channel parts");
    extern __shared__ __attribute__((aligned(16))) float link_dyn[];
    float* wt_s = link_dyn;                               // [C][C]
    float* es_s = wt_s + C * C;
    float* bs_s = es_s + C;
    float* fac_s = bs_s + C;
    float* tq = fac_s + C;                                // four [64][TS] tiles: dx_in -> q0 | a -> (free) | dy | q2, then dc3 in tile 1
    float* T0 = tq, *T1 = tq + 64 * TS, *T2 = tq + 2 * 64 * TS, *T3 = tq + 3 * 64 * TS;
    const int tid = threadIdx.x, pl = tid & 63;
    const int part = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long px = (long)blockIdx.x * 64 + pl;
    const bool live = px < p.npix;
    const bool head = p.xpre != nullptr, cpl = p.ypost != nullptr;
    if (head) {
        for (int i = tid; i < C * C; i += NT) wt_s[i] = p.wt[i];
        for (int c = tid; c < C; c += NT) { es_s[c] = expf(p.scale[c]); bs_s[c] = p.bias[c]; }
    }
    if (cpl)
        for (int c = tid; c < C; c += NT) fac_s[c] = expf(__fmul_rn(p.f3, p.logs3[c]));
    const int cs = part * CPT;                            // this thread's channels of the head phase
    double* out = p.part + (size_t)blockIdx.x * 4 * C;
    if (head) {
        // the pixel's dx_in row into tile 0 (every part its CPT channels)
#pragma unroll
        for (int c = 0; c < CPT; c += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) v = *reinterpret_cast<const float4*>(p.dx_in + px * C + cs + c);
            float* d = T0 + pl * TS + cs + c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        float da[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) da[c] = 0.f;
#pragma unroll 2
        for (int k = 0; k < C; ++k) {
            const float xv = T0[pl * TS + k];
            const float* wr = wt_s + k * C + cs;
#pragma unroll
            for (int c = 0; c < CPT; c += 4) {
                const float4 w4 = *reinterpret_cast<const float4*>(wr + c);
                da[c] = fmaf(w4.x, xv, da[c]); da[c + 1] = fmaf(w4.y, xv, da[c + 1]);
                da[c + 2] = fmaf(w4.z, xv, da[c + 2]); da[c + 3] = fmaf(w4.w, xv, da[c + 3]);
            }
        }
        float q0[CPT];
#pragma unroll
        for (int c = 0; c < CPT; c += 4) {
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) xv = *reinterpret_cast<const float4*>(p.xpre + px * C + cs + c);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
            float av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float es = es_s[cs + c + j];
                av[j] = __fmul_rn(es, __fadd_rn(xs[j], bs_s[cs + c + j]));
                const float dyv = __fmul_rn(da[c + j], es);
                q0[c + j] = __fmul_rn(da[c + j], av[j]);
                T1[pl * TS + cs + c + j] = av[j];
                T2[pl * TS + cs + c + j] = live ? dyv : 0.f;
            }
            if (live && p.a_out) *reinterpret_cast<float4*>(p.a_out + px * C + cs + c) = make_float4(av[0], av[1], av[2], av[3]);
        }
        __syncthreads();
        if (p.part_w) {
            // dW[k][c] = sum_p dx_in[p][k] a[p][c] of the workgroup's 64 pixels: tile (i, j) of the MT x MT row tiles on wave t % TPP, over
            // all 64 pixels in the step order of flow_link_bwd_kernel<C, 64>
            constexpr int MT = (C + 31) / 32;
            const int lane = tid & 63, m0 = lane & 31, kk = lane >> 5;
            float* pw = p.part_w + (size_t)blockIdx.x * C * C;
            for (int t = part; t < MT * MT; t += TPP) {
                const int ti = t / MT, tj = t % MT;
                f32x16 accw;
#pragma unroll
                for (int r = 0; r < 16; ++r) accw[r] = 0.f;
                const int ca = 32 * ti + m0, cb = 32 * tj + m0;
#pragma unroll 4
                for (int st = 0; st < 32; ++st) {
                    const int prow = (2 * st + kk) * TS;
                    const float av_ = ca < C ? T0[prow + ca] : 0.f;
                    const float bv_ = cb < C ? T1[prow + cb] : 0.f;
                    accw = __builtin_amdgcn_mfma_f32_32x32x2f32(av_, bv_, accw, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * ti + drow32(r, kk), col = 32 * tj + m0;
                    if (row < C && col < C) pw[row * C + col] = accw[r];
                }
            }
            __syncthreads();                                   // every wave is done reading tiles 0 and 1
        }
#pragma unroll
        for (int c = 0; c < CPT; ++c) T0[pl * TS + cs + c] = live ? q0[c] : 0.f;        // column-sum quantity 0
    } else {
        // no head: dy = dx_in (tile 2 for the coupling's reads), quantities 0 and 1 are zero
#pragma unroll
        for (int c = 0; c < CPT; c += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) v = *reinterpret_cast<const float4*>(p.dx_in + px * C + cs + c);
            float* d = T2 + pl * TS + cs + c;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
    }
    // ---- coupling of the previous flow: this thread's CQ channel pairs (c, Ch + c)
    const int c0 = part * CQ;
    if (cpl) {
        const float g = (p.gl && live) ? (float)p.gl[px / p.HW] : 0.f;
        float hl[CQ], ht[CQ], xb[CQ];
#pragma unroll
        for (int c = 0; c < CQ; c += 2) {
            float2 a2 = make_float2(0.f, 0.f), b2 = a2, x2 = a2;
            if (live) {
                a2 = *reinterpret_cast<const float2*>(p.hout + px * C + c0 + c);
                b2 = *reinterpret_cast<const float2*>(p.hout + px * C + Ch + c0 + c);
                x2 = *reinterpret_cast<const float2*>(p.ypost + px * C + Ch + c0 + c);
            }
            hl[c] = a2.x; hl[c + 1] = a2.y; ht[c] = b2.x; ht[c + 1] = b2.y; xb[c] = x2.x; xb[c + 1] = x2.y;
        }
        float dxa[CQ], dxb[CQ], dc3a[CQ], dc3b[CQ];
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            const float sg = sigmoid_p2(hl[c]);
            const float t = ht[c];
            const float dya = T2[pl * TS + c0 + c], dyb = T2[pl * TS + Ch + c0 + c];
            const float dt = __fmul_rn(dyb, sg);
            const float ds = __fadd_rn(__fmul_rn(dyb, __fadd_rn(xb[c], t)), __fdiv_rn(g, __fadd_rn(sg, 1e-6f)));
            const float dls = __fmul_rn(__fmul_rn(ds, sg), __fsub_rn(1.0f, sg));
            dxa[c] = dya; dxb[c] = dt;
            dc3a[c] = __fmul_rn(dls, fac_s[c0 + c]);
            dc3b[c] = __fmul_rn(dt, fac_s[Ch + c0 + c]);
            T3[pl * TS + c0 + c] = live ? __fmul_rn(dls, hl[c]) : 0.f;                  // quantity 2
            T3[pl * TS + Ch + c0 + c] = live ? __fmul_rn(dt, ht[c]) : 0.f;
        }
        __syncthreads();                                       // tile 1 (a) is free: quantity 3 goes there
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
            T1[pl * TS + c0 + c] = live ? dc3a[c] : 0.f;
            T1[pl * TS + Ch + c0 + c] = live ? dc3b[c] : 0.f;
        }
        if (live) {
#pragma unroll
            for (int c = 0; c < CQ; c += 2) {
                *reinterpret_cast<float2*>(p.dx_out + px * C + c0 + c) = make_float2(dxa[c], dxa[c + 1]);
                *reinterpret_cast<float2*>(p.dx_out + px * C + Ch + c0 + c) = make_float2(dxb[c], dxb[c + 1]);
                *reinterpret_cast<float2*>(p.dc3_out + px * C + c0 + c) = make_float2(dc3a[c], dc3a[c + 1]);
                *reinterpret_cast<float2*>(p.dc3_out + px * C + Ch + c0 + c) = make_float2(dc3b[c], dc3b[c + 1]);
            }
        }
    } else if (live) {                                         // last call of a level: dx_out = dy
#pragma unroll
        for (int c = 0; c < CPT; c += 4) {
            const float* d = T2 + pl * TS + cs + c;
            *reinterpret_cast<float4*>(p.dx_out + px * C + cs + c) = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
    __syncthreads();
    // ---- the four column sums (fp64, rows in order), one quantity per wave (TPP = 2: two each)
    for (int q = part; q < 4; q += TPP) {
        const bool has = q < 2 ? head : cpl;
        const float* T = q == 0 ? T0 : (q == 1 ? T2 : (q == 2 ? T3 : T1));
        if (pl < C) {
            double sum = 0.0;
            if (has)
                for (int r = 0; r < 64; ++r) sum += (double)T[r * TS + pl];
            out[q * C + pl] = sum;
        }
    }
}

// One launch for a whole level: call k's partials [nblk][4][C] -> the four gradients it feeds (pointers may be null).
constexpr int kMaxLinkCalls = 33;
struct LinkFinish {
    float* dscale[kMaxLinkCalls]; float* dbias[kMaxLinkCalls]; float* dlogs3[kMaxLinkCalls]; float* dbias3[kMaxLinkCalls];
    float* dw[kMaxLinkCalls];
    int n, nblk, C, acc; float f3;
    long call_doubles;   // doubles between the partial buffers of consecutive calls
    int wblocks;         // workgroups per call for the 1x1 conv's weight gradient: ceil(C*C / 16)
};
__global__ void __launch_bounds__(kBlock) flow_link_finish_kernel(const double* __restrict__ part, const LinkFinish f) {
    __shared__ double red[kBlock];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= 4 * f.n) {
        // dW[e] (+)= sum over the workgroups' fp32 partials (behind the call's fp64 column sums), 16 elements x 16 groups per workgroup
        // (64 x 4: every thread walked a quarter of the partials - 512 of the 2048 of a 16x16 level - in a launch of ~100 workgroups)
        const int wb = blockIdx.x - 4 * f.n, call = wb / f.wblocks, e = (wb % f.wblocks) * 16 + (tid & 15), g = tid >> 4;
        float* dst = f.dw[call];
        if (dst == nullptr) return;
        const float* src = reinterpret_cast<const float*>(part + (size_t)call * f.call_doubles + (size_t)f.nblk * 4 * f.C);
        const int cc = f.C * f.C;
        double sum = 0.0;
        if (e < cc)
            for (int b = g; b < f.nblk; b += 16) sum += (double)src[(size_t)b * cc + e];
        red[tid] = sum;
        __syncthreads();
        if (g == 0 && e < cc) {
            double t = 0.0;
            for (int k = 0; k < 16; ++k) t += red[16 * k + tid];     // fixed order: deterministic
            const float v = (float)t;
            dst[e] = f.acc ? dst[e] + v : v;
        }
        return;
    }
    const int call = blockIdx.x / 4, q = blockIdx.x % 4;
    float* dst = q == 0 ? f.dscale[call] : (q == 1 ? f.dbias[call] : (q == 2 ? f.dlogs3[call] : f.dbias3[call]));
    if (dst == nullptr) return;
    const double* src = part + (size_t)call * f.call_doubles + (size_t)q * f.C;
    // groups of NCH threads (the channel count rounded up to 16 / 32 / 64): channel = tid % NCH, workgroup partials g, g + G, ...
    // (with 64-thread groups a C = 12 level had 48 of 256 threads at work, each walking 512 of its 2048 partials: 100 us per launch)
    const int NCH = f.C <= 16 ? 16 : (f.C <= 32 ? 32 : 64), G = kBlock / NCH;
    const int c = tid % NCH, g = tid / NCH;
    double sum = 0.0;
    if (c < f.C)
        for (int b = g; b < f.nblk; b += G) sum += src[(size_t)b * 4 * f.C + c];
    red[tid] = sum;
    __syncthreads();
    if (g == 0 && c < f.C) {
        double t = 0.0;
        for (int k = 0; k < G; ++k) t += red[k * NCH + c];      // fixed order: deterministic
        const float v = (float)(q == 2 ? t * (double)f.f3 : t);
        dst[c] = f.acc ? dst[c] + v : v;
    }
}

// per call: [blocks][4][C] fp64 column sums, then [blocks][C*C] fp32 partials of the 1x1 conv's weight gradient (rounded up to 8 bytes)
// pixels per workgroup of the link kernel (see flow_link_bwd_kernel); the partial buffers are laid out per workgroup, so the three
// entry points below derive it from npix alike
static int link_block(long npix) {
    static const int forced = [] { const char* e = NFDP_DIAG_ENV("NFDP_LINK_BLOCK"); return e ? atoi(e) : 0; }();   // A/B measurements
    if (forced == 64 || forced == 256) return forced;
    (void)npix;
    return 64;             // measured (rocprofv3, B = 512): 18.7 / 16.8 / 25.2 us per launch at the three levels against 22.7 / 23.0 / 32.0 with 256
}
template <int C, int LB>
static void launch_flow_link_lb(hipStream_t s, const LinkParams& p) {
    constexpr int kDyn = 2 * LB * (C + 1) * (int)sizeof(float);
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&flow_link_bwd_kernel<C, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, kDyn);
        attr_set = true;
    }
    hipLaunchKernelGGL((flow_link_bwd_kernel<C, LB>), dim3((unsigned)cdiv(p.npix, LB)), dim3(LB), kDyn, s, p);
}
static thread_local int g_link_split = 1;                 // test hook (thread-local): 0 = one thread per pixel at every C
extern "C" void nfdp_debug_set_link_split(int v) { g_link_split = v; }
template <int C, int TPP>
static void launch_flow_link_split(hipStream_t s, const LinkParams& p) {
    constexpr int kDyn = (C * C + 3 * C + 4 * 64 * (C + 1)) * (int)sizeof(float);
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&flow_link_bwd_split_kernel<C, TPP>), hipFuncAttributeMaxDynamicSharedMemorySize, kDyn);
        attr_set = true;
    }
    hipLaunchKernelGGL((flow_link_bwd_split_kernel<C, TPP>), dim3((unsigned)cdiv(p.npix, 64)), dim3(64 * TPP), kDyn, s, p);
}
template <int C>
static void launch_flow_link(hipStream_t s, const LinkParams& p) {
    if (link_block(p.npix) == 64) {
        if constexpr (C == 48) { if (g_link_split) { launch_flow_link_split<48, 4>(s, p); return; } }
        if constexpr (C == 24) { if (g_link_split) { launch_flow_link_split<24, 2>(s, p); return; } }
        launch_flow_link_lb<C, 64>(s, p);
    } else launch_flow_link_lb<C, 256>(s, p);
}

extern "C" size_t nfdp_flow_link_bwd_part_bytes(long npix, int C) {
    const size_t nblk = (size_t)cdiv(npix, link_block(npix));
    return nblk * 4 * C * sizeof(double) + ((nblk * C * C * sizeof(float) + 7) & ~(size_t)7);
}

extern "C" int nfdp_flow_link_bwd_f32(const float* dx_in, const float* xpre, const float* wt, const float* scale, const float* bias,
                                      float* a_out, const float* ypost, const float* hout, const double* gl, const float* logs3, float f3,
                                      float* dx_out, float* dc3_out, void* part, long npix, int HW, int C, nfdp_stream_t stream) {
    NFDP_REQUIRE(dx_in && dx_out && part, NFDP_EINVAL, "flow_link_bwd: null pointer");
    NFDP_REQUIRE(!xpre || (wt && scale && bias), NFDP_EINVAL, "flow_link_bwd: head given without wt / scale / bias");
    NFDP_REQUIRE(!ypost || (hout && logs3 && dc3_out), NFDP_EINVAL, "flow_link_bwd: coupling given without hout / logs3 / dc3_out");
    NFDP_REQUIRE(xpre || ypost, NFDP_EINVAL, "flow_link_bwd: neither a head nor a coupling");
    NFDP_REQUIRE(npix >= 0 && HW > 0, NFDP_ESHAPE, "flow_link_bwd: bad shape");
    NFDP_REQUIRE((((uintptr_t)dx_in | (uintptr_t)xpre | (uintptr_t)a_out | (uintptr_t)ypost | (uintptr_t)hout | (uintptr_t)dx_out |
                   (uintptr_t)dc3_out) & 15) == 0, NFDP_ESHAPE, "flow_link_bwd: tensors must be 16-byte aligned");
    if (npix == 0) return NFDP_OK;
    LinkParams p{dx_in, xpre, wt, scale, bias, a_out, ypost, hout, gl, logs3, f3, dx_out, dc3_out, reinterpret_cast<double*>(part), npix, HW,
                 nullptr};
    const size_t nblk = (size_t)cdiv(npix, link_block(npix));
    if (xpre) p.part_w = reinterpret_cast<float*>(p.part + nblk * 4 * C);
    hipStream_t s = as_stream(stream);
    switch (C) {
        case 4: launch_flow_link<4>(s, p); break;
        case 8: launch_flow_link<8>(s, p); break;
        case 16: launch_flow_link<16>(s, p); break;
        case 12: launch_flow_link<12>(s, p); break;
        case 24: launch_flow_link<24>(s, p); break;
        case 48: launch_flow_link<48>(s, p); break;
        default: nfdp_set_error("flow_link_bwd: unsupported C=%d", C); return NFDP_EUNSUPPORTED;
    }
    NFDP_CHECK_LAUNCH("flow_link_bwd");
    return NFDP_OK;
}

extern "C" int nfdp_flow_link_finish_f32(const void* part, int n_calls, long npix, int C, float f3, float* const* dscale,
                                         float* const* dbias, float* const* dlogs3, float* const* dbias3, float* const* dw,
                                         int accumulate, nfdp_stream_t stream) {
    NFDP_REQUIRE(part && dscale && dbias && dlogs3 && dbias3 && dw, NFDP_EINVAL, "flow_link_finish: null pointer");
    NFDP_REQUIRE(n_calls >= 0 && n_calls <= kMaxLinkCalls && C > 0 && C <= 64 && npix >= 0, NFDP_ESHAPE, "flow_link_finish: bad shape");
    if (n_calls == 0 || npix == 0) return NFDP_OK;
    LinkFinish f;
    bool any_dw = false;
    for (int i = 0; i < n_calls; ++i) {
        f.dscale[i] = dscale[i]; f.dbias[i] = dbias[i]; f.dlogs3[i] = dlogs3[i]; f.dbias3[i] = dbias3[i]; f.dw[i] = dw[i];
        any_dw = any_dw || dw[i] != nullptr;
    }
    f.n = n_calls; f.nblk = (int)cdiv(npix, link_block(npix)); f.C = C; f.acc = accumulate; f.f3 = f3;
    f.call_doubles = (long)(nfdp_flow_link_bwd_part_bytes(npix, C) / sizeof(double));
    f.wblocks = (C * C + 15) / 16;
    hipLaunchKernelGGL(flow_link_finish_kernel, dim3(4 * n_calls + (any_dw ? n_calls * f.wblocks : 0)), dim3(kBlock), 0, as_stream(stream),
                       reinterpret_cast<const double*>(part), f);
    NFDP_CHECK_LAUNCH("flow_link_finish");
    return NFDP_OK;
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
// consecutive k (= pixels) per lane for one row/column (= channel), i.e. the TRANSPOSE of the NHWC tiles.  The tiles are
// staged in LDS as they lie in memory ([pixel][channel], 8-byte stores of 4 channels) and read with gfx950's transposing
// ds_read_b64_tr_b16: per 16-lane group a 4-pixel x 16-channel block comes back channel-major, two reads make one
// 8-pixel operand.  Row stride 320 B puts the 4 pixel rows of a 32-lane half on disjoint 64-byte bank groups (conflict-free).
// A workgroup owns a 128 x 128 block of dW and a slice of the pixels (32 per stage, register-staged double buffering);
// 4 waves as 2 x 2, each 2 x 2 accumulator tiles.  Requires Cin % 4 == 0 (4 staged values of the im2col row share a tap).
constexpr int WB_BM = 128, WB_BN = 128, WB_KC = 32, WB_RS = WB_BM + 32;    // LDS row stride in bf16 elements (320 B)
typedef short s16x4 __attribute__((ext_vector_type(4)));

// 8 consecutive pixels (rows r0..r0+7 of a [pixel][channel] image) of this lane's channel: `p` already holds the lane's
// gather address (row q = (lane%16)/4, columns 4*(lane%4).. of its group's 16-channel block)
__device__ __forceinline__ bf16x8 tr_read8(const __bf16* p) {
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * WB_RS));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// XB / DB: the x / dy operand is stored as bf16 (loaded 4 values = 8 bytes at a time, no conversion) instead of fp32
template <bool XB, bool DB>
__global__ void __launch_bounds__(kBlock) wgrad_bf16_kernel(const WgradParams p) {
    __shared__ __attribute__((aligned(16))) __bf16 At[2][WB_KC][WB_RS];
    __shared__ __attribute__((aligned(16))) __bf16 Bt[2][WB_KC][WB_RS];
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
    bf16x4 ra[4], rb[4];                                       // staged already rounded to bf16
    auto load4 = [](const void* base, long off, bool is_bf16) -> bf16x4 {
        bf16x4 r;
        if (is_bf16) {
            r = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(base) + off);
        } else {
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
            r[0] = (__bf16)v.x; r[1] = (__bf16)v.y; r[2] = (__bf16)v.z; r[3] = (__bf16)v.w;
        }
        return r;
    };
    auto load_stage = [&](long p0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px_l = ((tid + kBlock * i) * 4) / WB_BM;             // 0..31
            const long px = p0 + px_l;
            const bool pv = px < p_end;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ra[i][j] = (__bf16)0.f; rb[i][j] = (__bf16)0.f; }
            if (pv && a_ok[i]) ra[i] = load4(p.dy, px * p.dy_ld + a_m[i], DB);
            if (pv && b_ok[i]) {
                const long b = px / HW;
                const int r = (int)(px - b * HW);
                const int y = r / p.W + b_dy[i], xx = r % p.W + b_dx[i];
                if ((unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
                    rb[i] = load4(p.x, ((b * p.H + y) * p.W + xx) * p.x_ld + b_ci[i], XB);
            }
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = (tid + kBlock * i) * 4;
            const int px_l = e / WB_BM, c0 = e % WB_BM;
            *reinterpret_cast<bf16x4*>(&At[buf][px_l][c0]) = ra[i];
            *reinterpret_cast<bf16x4*>(&Bt[buf][px_l][c0]) = rb[i];
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
    // transposing-read gather address of this lane inside a 16-pixel x 32-channel operand block
    const int tr_off = (8 * g + ((lane & 15) >> 2)) * WB_RS + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
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
            for (int i = 0; i < 2; ++i) af[i] = tr_read8(&At[buf][kstep * 16][wm * 64 + i * 32] + tr_off);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = tr_read8(&Bt[buf][kstep * 16][wn * 64 + j * 32] + tr_off);
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

// The same GEMM for pointwise convs whose operands are BOTH stored in bf16 (the 512 -> 512 conv of the coupling network and the
// im2col form of its input conv): no gather and no conversion, so both tiles go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: 1 KiB = 4 pixel rows x 128 channels per wave instruction, no VGPRs), 64 pixels per stage, double
// buffered like pw_gemm_bf16_kernel (conv.hip).  The LDS image is lane-linear with 256-byte rows, so the bank swizzle that
// makes the transposing reads conflict-free (16-byte chunk ^ ((row & 3) << 2 | (row >> 2) & 3)) is applied to the per-lane
// SOURCE address and again on the read.  Requires whole 64-pixel stages (npix and the slice length multiples of 64).
constexpr int WD_KC = 64;
constexpr int WD_TILE_BYTES = WD_KC * WB_BM * 2;              // 16 KiB per operand per stage

__device__ __forceinline__ int wd_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__global__ void __launch_bounds__(kBlock, 2) wgrad_bf16_dma_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(1024))) char wd_smem[];   // [2 stages][dy tile 16 KiB | x tile 16 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroup -> (pixel slice, output tile).  The mblocks x nblocks tiles of one pixel slice read the same operand rows (each A tile
    // nblocks times, each B tile mblocks times): they belong on ONE XCD, whose L2 then serves the re-reads.  Workgroups go to XCDs
    // round-robin (blockIdx % 8), so with the plain order the 16 tiles of a slice of the 512 x 512 gradient sat on all 8 XCDs and every
    // XCD fetched its own copies from the Infinity Cache / HBM (3x the unique bytes; the kernel spent 63 % of its wave cycles waiting).
    int bid = blockIdx.x, nb, mb, ks;
    if (p.ksplit % 8 == 0) {
        const int tiles = p.mblocks * p.nblocks, j = bid >> 3, t = j % tiles;
        ks = (bid & 7) + 8 * (j / tiles);
        mb = t / p.nblocks; nb = t % p.nblocks;
    } else {
        nb = bid % p.nblocks; bid /= p.nblocks;
        mb = bid % p.mblocks;
        ks = bid / p.mblocks;
    }
    const long p_begin = (long)ks * p.px_per_split;
    const long p_end = (p_begin + p.px_per_split) < p.npix ? (p_begin + p.px_per_split) : p.npix;
    const __bf16* __restrict__ dyg = reinterpret_cast<const __bf16*>(p.dy);
    const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(p.x);

    // this lane's source offsets for its wave's 4 pieces (4 pixel rows each) of each operand tile
    long src_a[4], src_b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ wd_swz(row);                  // logical 16-byte chunk (8 channels) stored at this lane's slot
        int ca = mb * WB_BM + ch * 8, cb = nb * WB_BN + ch * 8;
        if (ca > p.Cout - 8) ca = p.Cout - 8;                      // ragged last block: any valid address, those columns are dropped
        if (cb > p.N - 8) cb = p.N - 8;
        // blocked operand (32-pixel tiles of [hb][s][g][pixel][8 channels], 512 channels): the 16-byte unit of (pixel, 8-channel group
        // c8) sits at tile*16384 + (c8/4)*1024 + ((c8/2)%2)*512 + (c8%2)*256 + (pixel%32)*8 elements; a stage's 64 pixels are two whole
        // tiles, so the per-stage advance p0 * 512 is the same expression as for [pixel][512] rows
        auto blk = [](int row_, int c) { const int c8 = c >> 3; return (long)(row_ >> 5) * 16384 + (c8 >> 2) * 1024 + ((c8 >> 1) & 1) * 512 + (c8 & 1) * 256 + (row_ & 31) * 8; };
        src_a[i] = p.a_blocked ? blk(row, ca) : (long)row * p.dy_ld + ca;
        src_b[i] = p.b_blocked ? blk(row, cb) : (long)row * p.x_ld + cb;
    }
    auto stage = [&](int buf, long p0) {
        char* base = wd_smem + buf * 2 * WD_TILE_BYTES + wave * 4 * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dyg + p0 * p.dy_ld + src_a[i]),
                                             (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xg + p0 * p.x_ld + src_b[i]),
                                             (__attribute__((address_space(3))) void*)(base + WD_TILE_BYTES + i * 1024), 16, 0, 0);
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
    // transposing-read gather addresses (bytes inside an operand tile) for pixel sub-block 0 of a 16-pixel k-step: lane 4q+p of a
    // 16-lane group supplies row q, 8 bytes at channel 4p of its group's 16-channel half; the second read is 4 rows further
    const int tq = (lane & 15) >> 2, tp = lane & 3, hh = (lane >> 4) & 1;
    int off_a[2][2], off_b[2][2];                                  // [tile][r]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = 8 * g + 4 * r + tq;                     // + 16 * kstep: leaves wd_swz unchanged
            const int cha = (wm * 64 + i * 32) / 8 + 2 * hh + (tp >> 1), chb = (wn * 64 + i * 32) / 8 + 2 * hh + (tp >> 1);
            off_a[i][r] = row * 256 + ((cha ^ wd_swz(row)) << 4) + 8 * (tp & 1);
            off_b[i][r] = row * 256 + ((chb ^ wd_swz(row)) << 4) + 8 * (tp & 1);
        }
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    auto frag = [&](const char* tile, const int (&off)[2], int kstep) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + off[0] + kstep * 16 * 256));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + off[1] + kstep * 16 * 256));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    // optional column sums of dy (the bias gradient of the layer whose output gradient dy is): one extra MFMA per A fragment against
    // an all-ones B fragment, in the waves that own n-block 0's first column tile (every column of the result is the sum over pixels)
    const bool do_db = p.ws_db != nullptr && nb == 0 && wn == 0;
    f32x16 acc_db[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_db[i][r] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    stage(0, p_begin);
    __syncthreads();
    int cur = 0;
    for (long p0 = p_begin; p0 < p_end; p0 += WD_KC) {
        if (p0 + WD_KC < p_end) stage(cur ^ 1, p0 + WD_KC);
        const char* at = wd_smem + cur * 2 * WD_TILE_BYTES;
        const char* bt = at + WD_TILE_BYTES;
#pragma unroll
        for (int kstep = 0; kstep < WD_KC / 16; ++kstep) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = frag(at, off_a[i], kstep);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j] = frag(bt, off_b[j], kstep);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if (do_db) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc_db[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], ones, acc_db[i], 0, 0, 0);
            }
        }
        __syncthreads();                                           // drains the DMA of the next stage, frees buffer cur
        cur ^= 1;
    }
    if (do_db && (lane & 31) == 0) {
        float* dbo = p.ws_db + (size_t)ks * p.Mpad;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dbo[mb * WB_BM + wm * 64 + i * 32 + drow32(r, lane >> 5)] = acc_db[i][r];
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

// The wide form of that GEMM for the coupling network's 512-channel gradients (dy has 512 channels: every weight gradient of the network,
// 48 x 3 launches per training step).  wgrad_bf16_dma_kernel's 128 x 128 workgroup tiles (64 x 64 per wave) read one LDS fragment per
// MFMA - the LDS peak - and every operand tile 4 times from L2; measured 10 % MFMA-busy, 63 % of the wave cycles waiting.  Here a
// workgroup of 8 waves (two per SIMD, so one wave's DMA issue and waits sit under the other's MFMAs) owns 256 (dy channels) x TN (x
// channels, TN = 256 or 128) outputs: a wave's 128 x TN/4 tile takes 4 + TN/128 fragments per 4 * TN/128 MFMAs, and every operand row is
// read by 2 (not 4) workgroups.  Pixels arrive 32 per stage by LDS-DMA through a 4-slot ring (3 stages in flight, counted vmcnt + one raw
// barrier per stage).  The price is more pixel slices (256 CUs / 2..4 tiles) for the reduce kernel to sum.
constexpr int W2_KC = 32;                  // pixels per stage
#ifndef NFDP_W2_NS
#define NFDP_W2_NS 4
#endif
constexpr int W2_NS = NFDP_W2_NS, W2_DEPTH = W2_NS - 1;     // ring slots / stages in flight
constexpr int W2_THREADS = 512;
constexpr int W2_BM = 256;
template <int TN> constexpr int w2_slot_bytes() { return W2_KC * (W2_BM + TN) * 2; }

template <int N>
__device__ __forceinline__ void w2_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int TN, bool ABLK, bool BBLK>          // ABLK / BBLK: dy / x in the blocked layout
__device__ __forceinline__ void wgrad_wide_body(const WgradParams& p, int bid, char* w2_smem) {
    static_assert(TN == 256 || TN == 128, "x tile");
    // w2_smem: [W2_NS][dy tile 32 x 512 B | x tile 32 x 2*TN B]
    constexpr int SLOT = w2_slot_bytes<TN>(), A_BYTES = W2_KC * W2_BM * 2, PB = TN * 2;
    constexpr int NBP = TN == 256 ? 2 : 1;             // this wave's x pieces per stage (dy: always 2)
    constexpr int PPS = 2 + NBP;                       // LDS-DMA pieces per wave and stage
    constexpr int NJ = TN / 128;                       // 32-column tiles of a wave (its 128 x TN/4 outputs: 4 x NJ MFMA tiles)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    // workgroup -> (pixel slice, output tile): the tiles of one slice on one XCD (blockIdx % 8), whose L2 serves the second read of each row
    int nb, mb, ks;
    if (p.ksplit % 8 == 0) {
        const int tiles = p.mblocks * p.nblocks, j = bid >> 3, t = j % tiles;
        ks = (bid & 7) + 8 * (j / tiles);
        mb = t / p.nblocks; nb = t % p.nblocks;
    } else {
        nb = bid % p.nblocks; bid /= p.nblocks;
        mb = bid % p.mblocks;
        ks = bid / p.mblocks;
    }
    const long p_begin = (long)ks * p.px_per_split;
    const long p_end = (p_begin + p.px_per_split) < p.npix ? (p_begin + p.px_per_split) : p.npix;
    const int nstage = p_end > p_begin ? (int)((p_end - p_begin) / W2_KC) : 0;   // (an empty last slice still writes its zero tile)
    const __bf16* __restrict__ dyg = reinterpret_cast<const __bf16*>(p.dy);
    const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(p.x);

    // A blocked operand (32-pixel tiles of [hb][s][g][pixel][8 channels] = [64 chunks of 8 channels][32 pixels][16 B], 512 channels; a
    // stage is one tile) keeps that chunk-major order in LDS: its 256-channel half is 16 KiB of consecutive bytes, so a DMA piece is two
    // whole chunks = 1 KiB of consecutive global bytes.  (Gathering it into pixel rows, as wgrad_bf16_dma_kernel does, makes every wave
    // instruction touch 32 cache lines for 1 KiB: the texture path, not HBM or the matrix cores, then bounds the kernel - 136 us for the
    // level-0 512 x 512 gradient against 42 us of HBM time.)  Inside a chunk pixel p sits at slot p ^ ((chunk & 3) << 2): the four chunks a
    // transposing read covers then fall into four different 64-byte bank groups.
    long src_a[2], src_b[NBP];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (ABLK) {
            const int chunk = 2 * (2 * wave + i) + (lane >> 5), px = (lane & 31) ^ ((chunk & 3) << 2);
            src_a[i] = (long)(mb * 32 + chunk) * 256 + px * 8;
        } else {                                       // pixel rows: a piece = 2 rows of 256 channels (32 chunks of 16 B)
            const int row = 4 * wave + 2 * i + (lane >> 5);
            const int ch = (lane & 31) ^ wd_swz(row);
            src_a[i] = (long)row * p.dy_ld + mb * W2_BM + ch * 8;      // Cout == 512: always inside
        }
    }
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
        if (BBLK) {                                    // (TN == 256: a blocked x has 512 channels)
            const int chunk = 2 * (NBP * wave + i) + (lane >> 5), px = (lane & 31) ^ ((chunk & 3) << 2);
            src_b[i] = (long)(nb * 32 + chunk) * 256 + px * 8;
        } else {
            int row, ch;
            if (TN == 256) { row = 4 * wave + 2 * i + (lane >> 5); ch = (lane & 31) ^ wd_swz(row); }
            else { row = 4 * wave + (lane >> 4); ch = (lane & 15) ^ wd_swz(row); }      // a piece = 4 pixel rows of 128 channels
            int cb = nb * TN + ch * 8;
            if (cb > p.N - 8) cb = p.N - 8;            // ragged last block: any valid address, those columns are dropped
            src_b[i] = (long)row * p.x_ld + cb;
        }
    }
    auto stage = [&](int slot, long p0) {
        char* base = w2_smem + slot * SLOT;
#if defined(NFDP_W2_ABLATE) && NFDP_W2_ABLATE == 2     // timing-only build: the matrix work without the operand stream
        return;
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dyg + p0 * p.dy_ld + src_a[i]),
                                             (__attribute__((address_space(3))) void*)(base + (2 * wave + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NBP; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xg + p0 * p.x_ld + src_b[i]),
                                             (__attribute__((address_space(3))) void*)(base + A_BYTES + (NBP * wave + i) * 1024), 16, 0, 0);
    };

    f32x16 acc[4][NJ], acc_db;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc_db[r] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j][r] = 0.f;
    }
    const int c32 = lane & 31, g = lane >> 5;
    // transposing-read gather addresses inside an operand tile (see wgrad_bf16_dma_kernel; row pitch 512 B for dy, 2*TN B for x)
    const int tq = (lane & 15) >> 2, tp = lane & 3, hh = (lane >> 4) & 1;
    int off_a[4][2], off_b[NJ][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 8 * g + 4 * r + tq;                         // + 16 * kstep: leaves both swizzles unchanged
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cha = (wm * 128 + i * 32) / 8 + 2 * hh + (tp >> 1);
            off_a[i][r] = ABLK ? cha * 512 + ((row ^ ((cha & 3) << 2)) << 4) + 8 * (tp & 1)
                                      : row * (W2_BM * 2) + ((cha ^ wd_swz(row)) << 4) + 8 * (tp & 1);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int chb = (wn * (TN / 4) + j * 32) / 8 + 2 * hh + (tp >> 1);
            off_b[j][r] = BBLK ? chb * 512 + ((row ^ ((chb & 3) << 2)) << 4) + 8 * (tp & 1)
                               : row * PB + ((chb ^ wd_swz(row)) << 4) + 8 * (tp & 1);
        }
    }
    constexpr int KA = ABLK ? 256 : 16 * W2_BM * 2, KB = BBLK ? 256 : 16 * PB;   // bytes per 16-pixel k-step
    // The transposing reads are hand-issued (asm volatile) with hand-counted waits: a compiler-visible LDS read makes hipcc drain the
    // in-flight LDS-DMA first (s_waitcnt vmcnt(0): possible alias), which would serialise every stage's fetch with its MFMAs.
    // optional column sums of dy (the layer's bias gradient): in n-block 0 wave (wm, wn) also reads dy fragment wn of its row half and
    // multiplies it by an all-ones B fragment, so the four waves of a row half cover its four fragments with one extra MFMA per k-step
    const bool do_db = p.ws_db != nullptr && nb == 0;
    int off_d[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 8 * g + 4 * r + tq;
        const int cha = (wm * 128 + wn * 32) / 8 + 2 * hh + (tp >> 1);
        off_d[r] = ABLK ? cha * 512 + ((row ^ ((cha & 3) << 2)) << 4) + 8 * (tp & 1)
                               : row * (W2_BM * 2) + ((cha ^ wd_swz(row)) << 4) + 8 * (tp & 1);
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    const unsigned smem0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)w2_smem;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    struct Frags { s16x4 a[4][2], b[NJ][2], d[2]; };
    // request the fragments of k-step KS of the stage in ring slot `tile` (one address add per read and stage; the k-step and the x
    // tile's offset are immediates)
    auto issue = [&](Frags& f, unsigned tile, auto ksc, auto dbc) __attribute__((always_inline)) {
        constexpr int KS = decltype(ksc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 2; ++r)
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.a[i][r]) : "v"(tile + off_a[i][r]), "i"(KS * KA));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 2; ++r)
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.b[j][r]) : "v"(tile + off_b[j][r]), "i"(A_BYTES + KS * KB));
        const unsigned ad0 = tile + off_d[0], ad1 = tile + off_d[1];     // (named outside the constexpr branch: clang rejects the capture inside)
        if constexpr (decltype(dbc)::value) {
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.d[0]) : "v"(ad0), "i"(KS * KA));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.d[1]) : "v"(ad1), "i"(KS * KA));
        }
    };
    // all of f's reads are back once at most N younger LDS reads are outstanding (LDS returns in order)
    auto arrived = [&](Frags& f, auto nc, auto dbc) __attribute__((always_inline)) {
        constexpr int N = decltype(nc)::value;
        if constexpr (decltype(dbc)::value) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.d[0]), "+v"(f.d[1]) : "i"(N));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.a[i][0]), "+v"(f.a[i][1]) : "i"(N));
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.b[j][0]), "+v"(f.b[j][1]) : "i"(N));
    };
    auto cat = [](const s16x4& lo, const s16x4& hi) -> bf16x8 {
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma = [&](const Frags& f, auto dbc) __attribute__((always_inline)) {
        bf16x8 bfr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = cat(f.b[j][0], f.b[j][1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 af = cat(f.a[i][0], f.a[i][1]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (decltype(dbc)::value) acc_db = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cat(f.d[0], f.d[1]), ones, acc_db, 0, 0, 0);
    };

    // Software pipeline over the stages (two 16-pixel k-steps each): the fragments of the NEXT k-step are requested before the MFMAs of
    // the current one, across the stage boundary too - the barrier that publishes stage it+1 (and frees the slot of stage it-1 for stage
    // it+3) sits between the two k-steps of stage it, so no LDS latency and no barrier skew is exposed in front of an MFMA group.
    static_assert(W2_KC == 32 && W2_NS == 4, "pipeline written for two k-steps per stage and a 4-slot ring");
    auto run = [&](auto dbc) __attribute__((always_inline)) {
        constexpr int NR = 2 * (4 + NJ) + (decltype(dbc)::value ? 2 : 0);        // LDS reads per k-step
        constexpr std::integral_constant<int, 0> k0{};
        constexpr std::integral_constant<int, 1> k1{};
        Frags f0, f1;
        if (nstage >= 3) w2_wait_barrier<2 * PPS>();
        else if (nstage == 2) w2_wait_barrier<PPS>();
        else w2_wait_barrier<0>();
#if !(defined(NFDP_W2_ABLATE) && NFDP_W2_ABLATE == 1)
        issue(f0, smem0, k0, dbc);
#endif
        for (int it = 0; it < nstage; ++it) {
            const unsigned tile = smem0 + (it % W2_NS) * SLOT;
#if !(defined(NFDP_W2_ABLATE) && NFDP_W2_ABLATE == 1)     // (== 1: timing-only build, the operand stream without the matrix work)
            issue(f1, tile, k1, dbc);
            arrived(f0, std::integral_constant<int, NR>{}, dbc);
            mma(f0, dbc);
#endif
            if (it + 1 < nstage) {
                // this wave's pieces of stage it+1 have landed once only stage it+2's are outstanding; after the barrier every wave's
                // have, and every wave is past its reads of stage it-1
                if (it + 2 < nstage) w2_wait_barrier<PPS>();
                else w2_wait_barrier<0>();
                if (it + 3 < nstage) stage((it + 3) % W2_NS, p_begin + (long)(it + 3) * W2_KC);
#if !(defined(NFDP_W2_ABLATE) && NFDP_W2_ABLATE == 1)
                issue(f0, smem0 + ((it + 1) % W2_NS) * SLOT, k0, dbc);
                arrived(f1, std::integral_constant<int, NR>{}, dbc);
#endif
            } else {
#if !(defined(NFDP_W2_ABLATE) && NFDP_W2_ABLATE == 1)
                arrived(f1, std::integral_constant<int, 0>{}, dbc);
#endif
            }
#if !(defined(NFDP_W2_ABLATE) && NFDP_W2_ABLATE == 1)
            mma(f1, dbc);
#endif
        }
    };
#pragma unroll
    for (int s0 = 0; s0 < 3; ++s0)
        if (s0 < nstage) stage(s0, p_begin + (long)s0 * W2_KC);
    if (nstage > 0) {
        if (do_db) run(std::true_type{});
        else run(std::false_type{});
    }
    if (do_db && c32 == 0) {
        float* dbo = p.ws_db + (size_t)ks * p.Mpad;
#pragma unroll
        for (int r = 0; r < 16; ++r) dbo[mb * W2_BM + wm * 128 + wn * 32 + drow32(r, g)] = acc_db[r];
    }
    float* out = p.ws + (size_t)ks * p.Mpad * p.Npad;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb * W2_BM + wm * 128 + i * 32 + drow32(r, g);
                const int n = nb * TN + wn * (TN / 4) + j * 32 + c32;
                out[(size_t)m * p.Npad + n] = acc[i][j][r];
            }
}

template <int TN, bool ABLK, bool BBLK>
__global__ void __launch_bounds__(W2_THREADS, 1) wgrad_bf16_wide_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(1024))) char w2_dyn_smem[];
    wgrad_wide_body<TN, ABLK, BBLK>(p, blockIdx.x, w2_dyn_smem);
}

// The weight gradients of one flow's coupling network as ONE launch: up to three GEMMs over the same pixels (dW2: y1 x dc2; dW1: im2col
// rows x dc1; dW3: im2col rows of dc3 x y2), each with its own slice count, workgroups dealt to the GEMMs in order (the 256 x 256 tiles
// first: they take longest).  One launch per GEMM gave each 256 / tiles slices; together the 8 tiles of a level-0 flow share the chip,
// the slices are 1.5-2.7 x longer, and the fp32 partial tiles the reduce has to re-read shrink accordingly.
struct WgradMulti {
    WgradParams p[3];
    int first[4];        // first workgroup of each GEMM (multiples of 8: the XCD mapping of the body reads the low bits)
    int kind[3];         // 0: <256, blocked, blocked>  1: <256, blocked, rows>  2: <128, blocked, rows>
    int n;
};
__global__ void __launch_bounds__(W2_THREADS, 1) wgrad_bf16_multi_kernel(const WgradMulti m) {
    extern __shared__ __attribute__((aligned(1024))) char w2_dyn_smem[];
    const int b = blockIdx.x;
    const int q = (m.n > 2 && b >= m.first[2]) ? 2 : ((m.n > 1 && b >= m.first[1]) ? 1 : 0);
    const int bid = b - m.first[q];
    if (bid >= m.p[q].mblocks * m.p[q].nblocks * m.p[q].ksplit) return;      // padding up to the next multiple of 8 workgroups
    if (m.kind[q] == 0) wgrad_wide_body<256, true, true>(m.p[q], bid, w2_dyn_smem);
    else if (m.kind[q] == 1) wgrad_wide_body<256, true, false>(m.p[q], bid, w2_dyn_smem);
    else wgrad_wide_body<128, true, false>(m.p[q], bid, w2_dyn_smem);
}

// The slice sums of those GEMMs in one launch: one workgroup per output row of every GEMM (wgrad_reduce_rows_kernel's work; a GEMM
// without ActNorm parameters - the zero conv's - has w == nullptr and only writes dW).
struct ReduceMulti {
    const float* ws[3]; const float* ws_db[3]; float* dw[3];
    const float* w[3]; const float* bias[3]; float* dscale[3]; float* dbias[3];
    WgradOut o[3];
    int ksplit[3], Npad[3];
    int n;
};
__global__ void __launch_bounds__(kBlock) wgrad_reduce_multi_kernel(const ReduceMulti r) {
    __shared__ float4 red[kBlock];
    __shared__ float sdot[kBlock], sdb[kBlock];
    const int q = blockIdx.x / 512, m = blockIdx.x % 512;
    wgrad_reduce_row(r.ws[q], r.ws_db[q], r.dw[q], r.ksplit[q], 512, r.Npad[q], r.o[q], r.w[q], r.bias[q], r.dscale[q], r.dbias[q], m, red,
                     sdot, sdb);
}

// The same two launches for up to kWgradBatch flows of a level at once (blockIdx.y = flow): the GEMMs of ONE flow at the 8x8 / 4x4 levels of a
// B = 512 step cover 32768 / 8192 pixels - a fraction of the chip, 66 / 52 us per flow against 165 us for the 16x larger level-0 flow.  The
// flows' gradients are independent of each other, so the backward pass keeps their operands and runs all of them here, each GEMM with fewer
// and longer pixel slices.  Shapes and slice plans are shared (WgradMulti / ReduceMulti hold them once); only pointers differ per flow.
constexpr int kWgradBatch = 16;
struct WgradBatchPtrs {
    const void* x[kWgradBatch][3];
    const void* dy[kWgradBatch][3];
    float* ws[kWgradBatch][3];
    float* ws_db[kWgradBatch][3];
};
__global__ void __launch_bounds__(W2_THREADS, 1) wgrad_bf16_multi_batch_kernel(const WgradMulti m, const WgradBatchPtrs t) {
    extern __shared__ __attribute__((aligned(1024))) char w2_dyn_smem[];
    const int b = blockIdx.x, f = blockIdx.y;
    const int q = (m.n > 2 && b >= m.first[2]) ? 2 : ((m.n > 1 && b >= m.first[1]) ? 1 : 0);
    const int bid = b - m.first[q];
    if (bid >= m.p[q].mblocks * m.p[q].nblocks * m.p[q].ksplit) return;
    WgradParams p = m.p[q];
    p.x = t.x[f][q]; p.dy = t.dy[f][q]; p.ws = t.ws[f][q]; p.ws_db = t.ws_db[f][q];
    if (m.kind[q] == 0) wgrad_wide_body<256, true, true>(p, bid, w2_dyn_smem);
    else if (m.kind[q] == 1) wgrad_wide_body<256, true, false>(p, bid, w2_dyn_smem);
    else wgrad_wide_body<128, true, false>(p, bid, w2_dyn_smem);
}
struct ReduceBatch {                                  // (by value: 2.9 KB of the 4 KB kernel-argument space)
    const float* ws[kWgradBatch][3];
    const float* ws_db[kWgradBatch][3];
    float* dw[kWgradBatch][3];
    const float* w[kWgradBatch][3];
    const float* bias[kWgradBatch][3];
    float* dscale[kWgradBatch][3];
    float* dbias[kWgradBatch][3];
    WgradOut o[3];
    int ksplit[3], Npad[3];
};
__global__ void __launch_bounds__(kBlock) wgrad_reduce_multi_batch_kernel(const ReduceBatch r) {
    __shared__ float4 red[kBlock];
    __shared__ float sdot[kBlock], sdb[kBlock];
    const int q = blockIdx.x / 512, m = blockIdx.x % 512, f = blockIdx.y;
    wgrad_reduce_row(r.ws[f][q], r.ws_db[f][q], r.dw[f][q], r.ksplit[q], 512, r.Npad[q], r.o[q], r.w[f][q], r.bias[f][q], r.dscale[f][q],
                     r.dbias[f][q], m, red, sdot, sdb);
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

extern "C" int nfdp_conv_wgrad_f32_ex(const float* x, int x_ld, const float* dy, int dy_ld, int B, int H, int W, int Cin, int Cout,
                                      int k, float* dw, void* ws, int out_form, int out_c, int accumulate, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && dy && dw && ws, NFDP_EINVAL, "conv_wgrad: null pointer");
    WgradOut om;
    NFDP_REQUIRE(wgrad_out_map(out_form, out_c, Cout, Cin, k, accumulate, &om), NFDP_ESHAPE,
                 "conv_wgrad: output form %d needs k = 1 and 9*out_c <= Cin (k=%d out_c=%d Cin=%d)", out_form, k, out_c, Cin);
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
    launch_wgrad_reduce(p.ws, dw, p.ksplit, p.Mpad, p.Npad, Cout, om, s);
    NFDP_CHECK_LAUNCH("conv_wgrad");
    return NFDP_OK;
}

extern "C" int nfdp_conv_wgrad_f32(const float* x, int x_ld, const float* dy, int dy_ld, int B, int H, int W, int Cin, int Cout,
                                   int k, float* dw, void* ws, nfdp_stream_t stream) {
    return nfdp_conv_wgrad_f32_ex(x, x_ld, dy, dy_ld, B, H, W, Cin, Cout, k, dw, ws, 0, 0, 0, stream);
}

extern "C" int nfdp_conv_flip_weights_f32(const float* w, float* wt, int O, int I, int k, nfdp_stream_t stream) {
    NFDP_REQUIRE(w && wt, NFDP_EINVAL, "conv_flip_weights: null pointer");
    NFDP_REQUIRE(O > 0 && I > 0 && k > 0, NFDP_ESHAPE, "conv_flip_weights: bad shape");
    const long total = (long)O * I * k * k;
    hipLaunchKernelGGL(flip_weights_kernel, dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream), w, wt, O, I, k);
    NFDP_CHECK_LAUNCH("conv_flip_weights");
    return NFDP_OK;
}

// used by the fused scale/ReLU backward epilogue of the pointwise GEMM (conv.hip): same partial layout [blk][2][C]
void nfdp_internal_scale_bwd_finish(const double* part, int nblk, int C, float f, float* dmul, float* dadd, int acc, hipStream_t s) {
    hipLaunchKernelGGL(scale_bwd_finish_kernel, dim3((int)cdiv(C, 8)), dim3(kBlock), 0, s, part, nblk, C, f, dmul, dadd, acc);
}

static int scale_bwd_blocks(long npix) {
    long g = cdiv(npix, 32);
    if (g > 512) g = 512;            // 2 workgroups per CU; every block costs 2 C doubles of partials the finish kernel re-reads
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" size_t nfdp_scale_bwd_ws_bytes(long npix, int C) { return (size_t)scale_bwd_blocks(npix) * 2 * C * sizeof(double); }

template <typename TDY, typename TY, typename TDC>
static void launch_scale_bwd(const void* dy, int dy_ld, const void* y, int y_ld, const float* mul, float f, int relu, long npix,
                             int C, void* dc, int dc_ld, double* part, int nblk, long rows, hipStream_t s) {
    const TDY* a = reinterpret_cast<const TDY*>(dy);
    const TY* b = reinterpret_cast<const TY*>(y);
    TDC* c = reinterpret_cast<TDC*>(dc);
    const bool vec = C >= 64 && C % 8 == 0 && dy_ld % 8 == 0 && (y == nullptr || y_ld % 8 == 0) && (dc == nullptr || dc_ld % 8 == 0) &&
                     (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dc) & 15) == 0;
    if (vec && C > 512)
        hipLaunchKernelGGL((scale_bwd_vec_kernel<256, TDY, TY, TDC>), dim3(nblk), dim3(kBlock), 0, s, a, dy_ld, b, y_ld, mul, f, relu,
                           npix, C, c, dc_ld, part, rows);
    else if (vec && C > 128)
        hipLaunchKernelGGL((scale_bwd_vec_kernel<64, TDY, TY, TDC>), dim3(nblk), dim3(kBlock), 0, s, a, dy_ld, b, y_ld, mul, f, relu,
                           npix, C, c, dc_ld, part, rows);
    else if (vec)
        hipLaunchKernelGGL((scale_bwd_vec_kernel<16, TDY, TY, TDC>), dim3(nblk), dim3(kBlock), 0, s, a, dy_ld, b, y_ld, mul, f, relu,
                           npix, C, c, dc_ld, part, rows);
    else if (C > 128)
        hipLaunchKernelGGL((scale_bwd_kernel<256, TDY, TY, TDC>), dim3(nblk), dim3(kBlock), 0, s, a, dy_ld, b, y_ld, mul, f, relu, npix,
                           C, c, dc_ld, part, rows);
    else if (C > 32)
        hipLaunchKernelGGL((scale_bwd_kernel<64, TDY, TY, TDC>), dim3(nblk), dim3(kBlock), 0, s, a, dy_ld, b, y_ld, mul, f, relu, npix,
                           C, c, dc_ld, part, rows);
    else
        hipLaunchKernelGGL((scale_bwd_kernel<16, TDY, TY, TDC>), dim3(nblk), dim3(kBlock), 0, s, a, dy_ld, b, y_ld, mul, f, relu, npix,
                           C, c, dc_ld, part, rows);
}

extern "C" int nfdp_scale_bwd_ex(const void* dy, int dy_dtype, int dy_ld, const void* y, int y_dtype, int y_ld, const float* mul,
                                 float f, int relu, long npix, int C, void* dc, int dc_dtype, int dc_ld, float* dmul, float* dadd,
                                 void* ws, int accumulate, nfdp_stream_t stream) {
    NFDP_REQUIRE(dy && ws, NFDP_EINVAL, "scale_bwd: null pointer");
    NFDP_REQUIRE(npix > 0 && C > 0 && C <= 4096 && dy_ld >= C, NFDP_ESHAPE, "scale_bwd: bad shape npix=%ld C=%d", npix, C);
    NFDP_REQUIRE((y == nullptr || y_ld >= C) && (dc == nullptr || dc_ld >= C), NFDP_ESHAPE, "scale_bwd: bad pixel stride");
    const int nblk = scale_bwd_blocks(npix);
    const long rows = cdiv(npix, nblk);
    double* part = reinterpret_cast<double*>(ws);
    hipStream_t s = as_stream(stream);
    const int combo = (dy_dtype == NFDP_BF16 ? 4 : 0) | ((y && y_dtype == NFDP_BF16) ? 2 : 0) | ((dc && dc_dtype == NFDP_BF16) ? 1 : 0);
    switch (combo) {
        case 0: launch_scale_bwd<float, float, float>(dy, dy_ld, y, y_ld, mul, f, relu, npix, C, dc, dc_ld, part, nblk, rows, s); break;
        case 3: launch_scale_bwd<float, __bf16, __bf16>(dy, dy_ld, y, y_ld, mul, f, relu, npix, C, dc, dc_ld, part, nblk, rows, s); break;
        case 7: launch_scale_bwd<__bf16, __bf16, __bf16>(dy, dy_ld, y, y_ld, mul, f, relu, npix, C, dc, dc_ld, part, nblk, rows, s); break;
        default:
            nfdp_set_error("scale_bwd: unsupported dtype combination dy=%d y=%d dc=%d (f32/f32/f32, f32/bf16/bf16, bf16/bf16/bf16)",
                           dy_dtype, y_dtype, dc_dtype);
            return NFDP_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(scale_bwd_finish_kernel, dim3((int)cdiv(C, 8)), dim3(kBlock), 0, s, part, nblk, C, f, dmul, dadd, accumulate);
    NFDP_CHECK_LAUNCH("scale_bwd");
    return NFDP_OK;
}

extern "C" int nfdp_scale_bwd(const void* dy, int dy_dtype, int dy_ld, const void* y, int y_dtype, int y_ld, const float* mul,
                              float f, int relu, long npix, int C, void* dc, int dc_dtype, int dc_ld, float* dmul, float* dadd,
                              void* ws, nfdp_stream_t stream) {
    return nfdp_scale_bwd_ex(dy, dy_dtype, dy_ld, y, y_dtype, y_ld, mul, f, relu, npix, C, dc, dc_dtype, dc_ld, dmul, dadd, ws, 0, stream);
}

extern "C" int nfdp_scale_bwd_f32(const float* dy, int dy_ld, const float* y, int y_ld, const float* mul, float f, int relu,
                                  long npix, int C, float* dc, int dc_ld, float* dmul, float* dadd, void* ws,
                                  nfdp_stream_t stream) {
    return nfdp_scale_bwd(dy, NFDP_F32, dy_ld, y, NFDP_F32, y_ld, mul, f, relu, npix, C, dc, NFDP_F32, dc_ld, dmul, dadd, ws, stream);
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
    long want = 512 / ((long)p->mblocks * p->nblocks);      // ~2 workgroups per CU: every pixel slice costs a partial tile the reduce re-reads
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    long per = cdiv(p->npix, want);
    per = cdiv(per, 64) * 64;                              // whole stages of either kernel (32 / 64 pixels)
    if (per < 128) per = 128;
    p->px_per_split = per;
    p->ksplit = (int)cdiv(p->npix, per);
}

// The wide kernel's plan: dy has exactly 512 channels (two 256-row tiles), x up to 512 columns in tiles of 256 (128 when N <= 128);
// one workgroup per CU, so 256 / tiles pixel slices of whole 32-pixel stages.
static bool wgrad_wide_shape_ok(long npix, int N, int Cout, int k) {
    return k == 1 && Cout == 512 && N >= 8 && N <= 512 && N % 8 == 0 && npix % W2_KC == 0 && npix >= 2 * W2_KC;
}
static bool wgrad_wide_ok(long npix, int N, int Cout, int k) {          // (the single-GEMM entries: switchable for A/B measurements)
    static const bool off = [] { const char* e = NFDP_DIAG_ENV("NFDP_WGRAD_WIDE"); return e && e[0] == '0'; }();
    return !off && wgrad_wide_shape_ok(npix, N, Cout, k);
}
static int wgrad_wide_tn(int N) { return N > 128 ? 256 : 128; }
static void wgrad_wide_plan(long npix, int N, WgradParams* p) {
    const int tn = wgrad_wide_tn(N);
    p->N = N;
    p->mblocks = 2;
    p->nblocks = (int)cdiv(N, tn);
    p->Mpad = 512;
    p->Npad = p->nblocks * tn;
    p->npix = npix;
    long want = 256 / ((long)p->mblocks * p->nblocks);
    static const int want_env = [] { const char* e = NFDP_DIAG_ENV("NFDP_WGRAD_SLICES"); return e ? atoi(e) : 0; }();   // sweeps
    if (want > npix / (8 * W2_KC)) want = npix / (8 * W2_KC);     // at least 8 stages per slice: below that the partial tiles cost more than they spread
    if (want < 1) want = 1;
    if (want_env > 0) want = want_env;
    long per = cdiv(cdiv(npix, want), W2_KC) * W2_KC;
    if (per < 2 * W2_KC) per = 2 * W2_KC;
    p->px_per_split = per;
    p->ksplit = (int)cdiv(npix, per);
}
template <int TN, bool ABLK, bool BBLK>
static void launch_wgrad_wide_t(const WgradParams& p, hipStream_t s) {
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&wgrad_bf16_wide_kernel<TN, ABLK, BBLK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, W2_NS * w2_slot_bytes<TN>());
        attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_bf16_wide_kernel<TN, ABLK, BBLK>), dim3(p.mblocks * p.nblocks * p.ksplit), dim3(W2_THREADS),
                       W2_NS * w2_slot_bytes<TN>(), s, p);
}
static void launch_wgrad_wide(const WgradParams& p, hipStream_t s) {
    if (wgrad_wide_tn(p.N) == 256) {
        if (p.a_blocked && p.b_blocked) launch_wgrad_wide_t<256, true, true>(p, s);
        else if (p.a_blocked) launch_wgrad_wide_t<256, true, false>(p, s);
        else if (p.b_blocked) launch_wgrad_wide_t<256, false, true>(p, s);
        else launch_wgrad_wide_t<256, false, false>(p, s);
    } else {                                           // (a blocked x has 512 channels: never here)
        if (p.a_blocked) launch_wgrad_wide_t<128, true, false>(p, s);
        else launch_wgrad_wide_t<128, false, false>(p, s);
    }
}

extern "C" int nfdp_conv_wgrad_bf16_supported(int Cin, int Cout) { return Cin % 4 == 0 && Cout % 4 == 0; }

extern "C" size_t nfdp_conv_wgrad_bf16_ws_bytes(int B, int H, int W, int Cin, int Cout, int k) {
    WgradParams p;
    wgrad_bf16_plan(B, H, W, Cin, Cout, k, &p);
    size_t bytes = (size_t)p.ksplit * p.Mpad * (p.Npad + 1) * sizeof(float);  // + the per-slice column sums of the _affine entry
    if (wgrad_wide_ok((long)B * H * W, k * k * Cin, Cout, k)) {               // whichever kernel the launch picks
        WgradParams q;
        wgrad_wide_plan((long)B * H * W, Cin, &q);
        const size_t wide = (size_t)q.ksplit * q.Mpad * (q.Npad + 1) * sizeof(float);
        if (wide > bytes) bytes = wide;
    }
    return bytes;
}

extern "C" int nfdp_conv_wgrad_bf16_affine(const void* x, int x_ld, const void* dy, int dy_ld, long npix, int Cin, int Cout, float* dw,
                                           void* ws, int out_form, int out_c, int accumulate, int layout, const float* w,
                                           const float* bias, float* dscale, float* dbias, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && dy && dw && ws, NFDP_EINVAL, "conv_wgrad_bf16_affine: null pointer");
    const bool affine = w != nullptr;
    NFDP_REQUIRE(!affine || (bias && dscale && dbias), NFDP_EINVAL, "conv_wgrad_bf16_affine: w given without bias / dscale / dbias");
    WgradOut om;
    NFDP_REQUIRE(wgrad_out_map(out_form, out_c, Cout, Cin, 1, accumulate, &om) && !(affine && out_form == 2), NFDP_ESHAPE,
                 "conv_wgrad_bf16_affine: output form %d (out_c=%d Cin=%d; form 2 has no ActNorm gradients)", out_form, out_c, Cin);
    NFDP_REQUIRE(!affine || Cin <= 768, NFDP_EUNSUPPORTED, "conv_wgrad_bf16_affine: Cin=%d (at most 768)", Cin);
    NFDP_REQUIRE(!(layout & 1) || (Cin == 512 && x_ld == 512), NFDP_ESHAPE, "conv_wgrad_bf16_affine: a blocked x has 512 channels");
    NFDP_REQUIRE(!(layout & 2) || (Cout == 512 && dy_ld == 512), NFDP_ESHAPE, "conv_wgrad_bf16_affine: a blocked dy has 512 channels");
    NFDP_REQUIRE(npix > 0 && (npix % WD_KC == 0 || wgrad_wide_ok(npix, Cin, Cout, 1)) && npix < (1L << 31) && Cin >= 8 && Cout >= 8 && Cin % 8 == 0 && Cout % 8 == 0 && x_ld >= Cin &&
                     dy_ld >= Cout && x_ld % 8 == 0 && dy_ld % 8 == 0 && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0,
                 NFDP_EUNSUPPORTED, "conv_wgrad_bf16_affine: bf16 operands, 16-byte aligned rows, channels multiples of 8, pixels a multiple of %d (32 with 512 dy channels)",
                 WD_KC);
    WgradParams p;
    const bool wide = wgrad_wide_ok(npix, Cin, Cout, 1);
    if (wide) wgrad_wide_plan(npix, Cin, &p);
    else wgrad_bf16_plan(1, 1, (int)npix, Cin, Cout, 1, &p);
    p.x = x; p.dy = dy; p.ws = reinterpret_cast<float*>(ws);
    p.ws_db = affine ? p.ws + (size_t)p.ksplit * p.Mpad * p.Npad : nullptr;
    p.b_blocked = layout & 1; p.a_blocked = (layout >> 1) & 1;
    p.x_ld = x_ld; p.dy_ld = dy_ld; p.B = 1; p.H = 1; p.W = (int)npix; p.Cin = Cin; p.Cout = Cout; p.k = 1;
    hipStream_t s = as_stream(stream);
    if (wide) {
        launch_wgrad_wide(p, s);
    } else {
        static NfdpPerDevice attr_set;
        if (!attr_set) {
            nfdp_func_attr(reinterpret_cast<const void*>(&wgrad_bf16_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      4 * WD_TILE_BYTES);
            attr_set = true;
        }
        hipLaunchKernelGGL(wgrad_bf16_dma_kernel, dim3(p.mblocks * p.nblocks * p.ksplit), dim3(kBlock), 4 * WD_TILE_BYTES, s, p);
    }
    if (affine)
        hipLaunchKernelGGL(wgrad_reduce_rows_kernel, dim3(Cout), dim3(kBlock), 0, s, p.ws, p.ws_db, dw, p.ksplit, p.Mpad, p.Npad, om, w,
                           bias, dscale, dbias);
    else
        launch_wgrad_reduce(p.ws, dw, p.ksplit, p.Mpad, p.Npad, Cout, om, s);
    NFDP_CHECK_LAUNCH("conv_wgrad_bf16_affine");
    return NFDP_OK;
}

// ---- the weight gradients of one flow's coupling network in one GEMM launch + one reduce launch (see wgrad_bf16_multi_kernel)
static bool wgrad3_plan(long npix, const int* cin, int n, WgradParams* p, bool batched = false) {
    if (n < 1 || n > 3) return false;
    double denom = 0.0;
    double cost[3];
    for (int q = 0; q < n; ++q) {
        if (!wgrad_wide_shape_ok(npix, cin[q], 512, 1)) return false;
        const int tn = wgrad_wide_tn(cin[q]);
        static const double c256 = [] { const char* e = NFDP_DIAG_ENV("NFDP_WGRAD3_C256"); return e ? atof(e) : 0.87; }();   // (sweeps)
        static const double c128 = [] { const char* e = NFDP_DIAG_ENV("NFDP_WGRAD3_C128"); return e ? atof(e) : 0.6; }();
        cost[q] = tn == 256 ? c256 : c128;                 // measured us per 32-pixel stage of a workgroup
        denom += 2.0 * (double)cdiv(cin[q], tn) * cost[q];
    }
    const long cap = npix / (8 * W2_KC);                   // at least 8 stages per slice
    for (int q = 0; q < n; ++q) {
        const int tn = wgrad_wide_tn(cin[q]);
        long want = (long)(256.0 / denom * cost[q]) / 8 * 8;   // slices in proportion to the stage cost: the GEMMs finish together
        if (want < 8) want = 8;
        if (want > cap) want = cap >= 8 ? cap / 8 * 8 : (cap > 0 ? cap : 1);
        if (batched) {
            // a level's flows in one launch: a sixteenth of the slices each (the plan does NOT depend on how many flows a call carries,
            // so a flow's sums - and bits - are the same whether it travels with 15 others or, under activation recomputation, with 3)
            want = (long)(256.0 / denom * cost[q] / kWgradBatch + 0.5);
            if (want < 1) want = 1;
            if (want > cap && cap > 0) want = cap;
        }
        p[q].N = cin[q];
        p[q].mblocks = 2;
        p[q].nblocks = (int)cdiv(cin[q], tn);
        p[q].Mpad = 512;
        p[q].Npad = p[q].nblocks * tn;
        p[q].npix = npix;
        long per = cdiv(cdiv(npix, want), W2_KC) * W2_KC;
        if (per < 2 * W2_KC) per = 2 * W2_KC;
        p[q].px_per_split = per;
        p[q].ksplit = (int)cdiv(npix, per);
    }
    return true;
}

extern "C" size_t nfdp_coupling_wgrad3_ws_bytes(long npix, const int* cin, int n_gemm) {
    WgradParams p[3];
    if (!cin || !wgrad3_plan(npix, cin, n_gemm, p)) return 0;
    size_t bytes = 0;
    for (int q = 0; q < n_gemm; ++q) bytes += (size_t)p[q].ksplit * 512 * (p[q].Npad + 1) * sizeof(float);
    return bytes;
}

extern "C" int nfdp_coupling_wgrad3_bf16(const void* const* x, const int* x_ld, const int* x_blocked, const void* const* dy, const int* cin,
                                         float* const* dw, const int* out_form, const int* out_c, const float* const* w,
                                         const float* const* bias, float* const* dscale, float* const* dbias, int n_gemm, long npix,
                                         void* ws, int accumulate, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && x_ld && x_blocked && dy && cin && dw && out_form && out_c && w && bias && dscale && dbias && ws, NFDP_EINVAL,
                 "coupling_wgrad3: null pointer");
    WgradMulti m;
    NFDP_REQUIRE(wgrad3_plan(npix, cin, n_gemm, m.p), NFDP_EUNSUPPORTED,
                 "coupling_wgrad3: 1..3 GEMMs over a multiple of %d pixels, x channels multiples of 8 up to 512", W2_KC);
    ReduceMulti r;
    m.n = r.n = n_gemm;
    float* wsf = reinterpret_cast<float*>(ws);
    int blocks = 0;
    for (int q = 0; q < n_gemm; ++q) {
        WgradParams& p = m.p[q];
        NFDP_REQUIRE(x[q] && dy[q] && dw[q], NFDP_EINVAL, "coupling_wgrad3: null pointer in GEMM %d", q);
        const bool affine = w[q] != nullptr;
        NFDP_REQUIRE(!affine || (bias[q] && dscale[q] && dbias[q]), NFDP_EINVAL, "coupling_wgrad3: w given without bias / dscale / dbias");
        NFDP_REQUIRE(wgrad_out_map(out_form[q], out_c[q], 512, cin[q], 1, accumulate, &r.o[q]) && !(affine && out_form[q] == 2), NFDP_ESHAPE,
                     "coupling_wgrad3: output form %d of GEMM %d (out_c=%d Cin=%d)", out_form[q], q, out_c[q], cin[q]);
        NFDP_REQUIRE(!x_blocked[q] || (cin[q] == 512 && x_ld[q] == 512), NFDP_ESHAPE, "coupling_wgrad3: a blocked x has 512 channels");
        NFDP_REQUIRE(x_ld[q] >= cin[q] && x_ld[q] % 8 == 0 && (((uintptr_t)x[q] | (uintptr_t)dy[q]) & 15) == 0, NFDP_ESHAPE,
                     "coupling_wgrad3: 16-byte aligned bf16 rows");
        p.x = x[q]; p.dy = dy[q];
        p.ws = wsf;
        wsf += (size_t)p.ksplit * 512 * p.Npad;
        p.ws_db = affine ? wsf : nullptr;
        wsf += (size_t)p.ksplit * 512;
        p.a_blocked = 1; p.b_blocked = x_blocked[q] ? 1 : 0;
        p.x_ld = x_ld[q]; p.dy_ld = 512; p.B = 1; p.H = 1; p.W = (int)npix; p.Cin = cin[q]; p.Cout = 512; p.k = 1;
        m.kind[q] = x_blocked[q] ? 0 : (wgrad_wide_tn(cin[q]) == 256 ? 1 : 2);
        m.first[q] = blocks;
        blocks += p.mblocks * p.nblocks * p.ksplit;
        blocks = (blocks + 7) & ~7;                                   // the next GEMM starts on XCD 0 again
        r.ws[q] = p.ws; r.ws_db[q] = p.ws_db; r.dw[q] = dw[q]; r.w[q] = w[q]; r.bias[q] = bias[q]; r.dscale[q] = dscale[q]; r.dbias[q] = dbias[q];
        r.ksplit[q] = p.ksplit; r.Npad[q] = p.Npad;
    }
    m.first[n_gemm] = blocks;
    for (int q = n_gemm; q < 3; ++q) { m.kind[q] = 0; m.first[q + 1] = blocks; }
    hipStream_t s = as_stream(stream);
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&wgrad_bf16_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  W2_NS * w2_slot_bytes<256>());
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_bf16_multi_kernel, dim3(blocks), dim3(W2_THREADS), W2_NS * w2_slot_bytes<256>(), s, m);
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(512 * n_gemm), dim3(kBlock), 0, s, r);
    NFDP_CHECK_LAUNCH("coupling_wgrad3");
    return NFDP_OK;
}

// ---- ... and for several flows of a level at once (see wgrad_bf16_multi_batch_kernel).  Pointer tables are [flows][n_gemm] row-major
// HOST arrays; shapes (x_ld, x_blocked, cin, out_form, out_c) are shared by the flows.  ws: flows * nfdp_coupling_wgrad3_batch_ws_bytes.
extern "C" size_t nfdp_coupling_wgrad3_batch_ws_bytes(long npix, const int* cin, int n_gemm) {
    WgradParams p[3];
    if (!cin || !wgrad3_plan(npix, cin, n_gemm, p, true)) return 0;
    size_t bytes = 0;
    for (int q = 0; q < n_gemm; ++q) bytes += (size_t)p[q].ksplit * 512 * (p[q].Npad + 1) * sizeof(float);
    return bytes;
}

extern "C" int nfdp_coupling_wgrad3_batch_max_flows(void) { return kWgradBatch; }

extern "C" int nfdp_coupling_wgrad3_batch_bf16(const void* const* x, const int* x_ld, const int* x_blocked, const void* const* dy,
                                               const int* cin, float* const* dw, const int* out_form, const int* out_c,
                                               const float* const* w, const float* const* bias, float* const* dscale,
                                               float* const* dbias, int n_gemm, int flows, long npix, void* ws, int accumulate,
                                               nfdp_stream_t stream) {
    NFDP_REQUIRE(x && x_ld && x_blocked && dy && cin && dw && out_form && out_c && w && bias && dscale && dbias && ws, NFDP_EINVAL,
                 "coupling_wgrad3_batch: null pointer");
    NFDP_REQUIRE(flows >= 1 && flows <= kWgradBatch, NFDP_ESHAPE, "coupling_wgrad3_batch: 1..%d flows per call (%d)", kWgradBatch, flows);
    WgradMulti m;
    NFDP_REQUIRE(wgrad3_plan(npix, cin, n_gemm, m.p, true), NFDP_EUNSUPPORTED,
                 "coupling_wgrad3_batch: 1..3 GEMMs over a multiple of %d pixels, x channels multiples of 8 up to 512", W2_KC);
    WgradBatchPtrs t;
    ReduceBatch r;
    m.n = n_gemm;
    const size_t per_flow = nfdp_coupling_wgrad3_batch_ws_bytes(npix, cin, n_gemm) / sizeof(float);
    int blocks = 0;
    size_t off = 0;
    for (int q = 0; q < n_gemm; ++q) {
        WgradParams& p = m.p[q];
        bool affine = false;
        for (int f = 0; f < flows; ++f) {
            const int i = f * n_gemm + q;
            NFDP_REQUIRE(x[i] && dy[i] && dw[i], NFDP_EINVAL, "coupling_wgrad3_batch: null pointer in GEMM %d of flow %d", q, f);
            const bool aff = w[i] != nullptr;
            NFDP_REQUIRE(f == 0 || aff == affine, NFDP_EINVAL, "coupling_wgrad3_batch: GEMM %d has ActNorm parameters in some flows only", q);
            affine = aff;
            NFDP_REQUIRE(!aff || (bias[i] && dscale[i] && dbias[i]), NFDP_EINVAL, "coupling_wgrad3_batch: w given without bias / dscale / dbias");
            NFDP_REQUIRE((((uintptr_t)x[i] | (uintptr_t)dy[i]) & 15) == 0, NFDP_ESHAPE, "coupling_wgrad3_batch: 16-byte aligned bf16 rows");
            float* wsf = reinterpret_cast<float*>(ws) + (size_t)f * per_flow + off;
            t.x[f][q] = x[i]; t.dy[f][q] = dy[i]; t.ws[f][q] = wsf;
            t.ws_db[f][q] = aff ? wsf + (size_t)p.ksplit * 512 * p.Npad : nullptr;
            r.ws[f][q] = t.ws[f][q]; r.ws_db[f][q] = t.ws_db[f][q]; r.dw[f][q] = dw[i]; r.w[f][q] = w[i]; r.bias[f][q] = bias[i];
            r.dscale[f][q] = dscale[i]; r.dbias[f][q] = dbias[i];
        }
        NFDP_REQUIRE(wgrad_out_map(out_form[q], out_c[q], 512, cin[q], 1, accumulate, &r.o[q]) && !(affine && out_form[q] == 2), NFDP_ESHAPE,
                     "coupling_wgrad3_batch: output form %d of GEMM %d (out_c=%d Cin=%d)", out_form[q], q, out_c[q], cin[q]);
        NFDP_REQUIRE(!x_blocked[q] || (cin[q] == 512 && x_ld[q] == 512), NFDP_ESHAPE, "coupling_wgrad3_batch: a blocked x has 512 channels");
        NFDP_REQUIRE(x_ld[q] >= cin[q] && x_ld[q] % 8 == 0, NFDP_ESHAPE, "coupling_wgrad3_batch: x rows of whole 16-byte pieces");
        off += (size_t)p.ksplit * 512 * (p.Npad + 1);
        p.x = nullptr; p.dy = nullptr; p.ws = nullptr; p.ws_db = nullptr;          // (per flow: WgradBatchPtrs)
        p.a_blocked = 1; p.b_blocked = x_blocked[q] ? 1 : 0;
        p.x_ld = x_ld[q]; p.dy_ld = 512; p.B = 1; p.H = 1; p.W = (int)npix; p.Cin = cin[q]; p.Cout = 512; p.k = 1;
        m.kind[q] = x_blocked[q] ? 0 : (wgrad_wide_tn(cin[q]) == 256 ? 1 : 2);
        m.first[q] = blocks;
        blocks += p.mblocks * p.nblocks * p.ksplit;
        blocks = (blocks + 7) & ~7;
        r.ksplit[q] = p.ksplit; r.Npad[q] = p.Npad;
    }
    m.first[n_gemm] = blocks;
    for (int q = n_gemm; q < 3; ++q) { m.kind[q] = 0; m.first[q + 1] = blocks; }
    for (int f = 0; f < kWgradBatch; ++f)
        for (int q = 0; q < 3; ++q)
            if (f >= flows || q >= n_gemm) {
                t.x[f][q] = t.dy[f][q] = nullptr; t.ws[f][q] = t.ws_db[f][q] = nullptr;
                r.ws[f][q] = r.ws_db[f][q] = nullptr; r.dw[f][q] = nullptr; r.w[f][q] = r.bias[f][q] = nullptr;
                r.dscale[f][q] = r.dbias[f][q] = nullptr;
            }
    static_assert(sizeof(WgradMulti) + sizeof(WgradBatchPtrs) <= 4096 && sizeof(ReduceBatch) <= 4096, "kernel arguments are passed by value");
    hipStream_t s = as_stream(stream);
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&wgrad_bf16_multi_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                       W2_NS * w2_slot_bytes<256>());
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_bf16_multi_batch_kernel, dim3(blocks, flows), dim3(W2_THREADS), W2_NS * w2_slot_bytes<256>(), s, m, t);
    hipLaunchKernelGGL(wgrad_reduce_multi_batch_kernel, dim3(512 * n_gemm, flows), dim3(kBlock), 0, s, r);
    NFDP_CHECK_LAUNCH("coupling_wgrad3_batch");
    return NFDP_OK;
}

extern "C" int nfdp_conv_wgrad_bf16_ex(const void* x, int x_dtype, int x_ld, const void* dy, int dy_dtype, int dy_ld, int B, int H, int W,
                                       int Cin, int Cout, int k, float* dw, void* ws, int out_form, int out_c, int accumulate,
                                       nfdp_stream_t stream) {
    NFDP_REQUIRE(x && dy && dw && ws, NFDP_EINVAL, "conv_wgrad_bf16: null pointer");
    WgradOut om;
    NFDP_REQUIRE(wgrad_out_map(out_form, out_c, Cout, Cin, k, accumulate, &om), NFDP_ESHAPE,
                 "conv_wgrad_bf16: output form %d needs k = 1 and 9*out_c <= Cin (k=%d out_c=%d Cin=%d)", out_form, k, out_c, Cin);
    NFDP_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && k >= 1 && k <= 7 && (k & 1), NFDP_ESHAPE,
                 "conv_wgrad_bf16: bad shape B=%d H=%d W=%d Cin=%d Cout=%d k=%d", B, H, W, Cin, Cout, k);
    NFDP_REQUIRE(nfdp_conv_wgrad_bf16_supported(Cin, Cout), NFDP_EUNSUPPORTED,
                 "conv_wgrad_bf16: Cin=%d and Cout=%d must be multiples of 4 (use nfdp_conv_wgrad_f32)", Cin, Cout);
    const bool xb = x_dtype == NFDP_BF16, db = dy_dtype == NFDP_BF16;
    NFDP_REQUIRE(x_ld >= Cin && dy_ld >= Cout && x_ld % 4 == 0 && dy_ld % 4 == 0 && ((uintptr_t)x & (xb ? 7 : 15)) == 0 &&
                     ((uintptr_t)dy & (db ? 7 : 15)) == 0,
                 NFDP_ESHAPE, "conv_wgrad_bf16: operands must be aligned to 4 elements with pixel strides that are multiples of 4");
    WgradParams p;
    wgrad_bf16_plan(B, H, W, Cin, Cout, k, &p);
    p.x = x; p.dy = dy; p.ws = reinterpret_cast<float*>(ws); p.ws_db = nullptr;
    p.x_ld = x_ld; p.dy_ld = dy_ld; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.k = k;
    hipStream_t s = as_stream(stream);
    const dim3 grid(p.mblocks * p.nblocks * p.ksplit);
    const bool dma = xb && db && k == 1 && Cout >= 8 && Cin >= 8 && Cin % 8 == 0 && Cout % 8 == 0 && x_ld % 8 == 0 && dy_ld % 8 == 0 &&
                     (((uintptr_t)x | (uintptr_t)dy) & 15) == 0 && p.npix % WD_KC == 0;
    if (dma) {
        static NfdpPerDevice attr_set;
        if (!attr_set) {
            nfdp_func_attr(reinterpret_cast<const void*>(&wgrad_bf16_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      4 * WD_TILE_BYTES);
            attr_set = true;
        }
        hipLaunchKernelGGL(wgrad_bf16_dma_kernel, grid, dim3(kBlock), 4 * WD_TILE_BYTES, s, p);
    } else if (xb && db) hipLaunchKernelGGL((wgrad_bf16_kernel<true, true>), grid, dim3(kBlock), 0, s, p);
    else if (xb) hipLaunchKernelGGL((wgrad_bf16_kernel<true, false>), grid, dim3(kBlock), 0, s, p);
    else if (db) hipLaunchKernelGGL((wgrad_bf16_kernel<false, true>), grid, dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((wgrad_bf16_kernel<false, false>), grid, dim3(kBlock), 0, s, p);
    launch_wgrad_reduce(p.ws, dw, p.ksplit, p.Mpad, p.Npad, Cout, om, s);
    NFDP_CHECK_LAUNCH("conv_wgrad_bf16");
    return NFDP_OK;
}

extern "C" int nfdp_conv_wgrad_bf16(const void* x, int x_dtype, int x_ld, const void* dy, int dy_dtype, int dy_ld, int B, int H, int W,
                                    int Cin, int Cout, int k, float* dw, void* ws, nfdp_stream_t stream) {
    return nfdp_conv_wgrad_bf16_ex(x, x_dtype, x_ld, dy, dy_dtype, dy_ld, B, H, W, Cin, Cout, k, dw, ws, 0, 0, 0, stream);
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
