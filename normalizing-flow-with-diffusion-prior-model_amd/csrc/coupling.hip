// Affine-coupling epilogue, Gaussian log-density / sampling, and the bits-per-dim reduction.
// Reference: AffineCoupling.transform/invert  normalizing_flow/transforms.py:179-185, :196-200
//            IsotropicGaussian               normalizing_flow/prior.py:36-37, :49-50
//            GaussianPrior (zero-input conv) normalizing_flow/prior.py:79-82
//            calculate_loss                  normalizing_flow/utils.py:255-256
// All per-sample sums run in one workgroup per sample with fp64 partials and a fixed combine order
// (lane shuffle tree -> wave partials in LDS -> serial over the 4 waves), so results are deterministic.
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;
constexpr float kLog2Pi = 1.8378770664093453f;   // float(np.log(2*np.pi)), prior.py:24

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < kBlock / 64; ++i) s += red[i];
    return s;   // valid in thread 0
}

__device__ __forceinline__ float sigmoid_p2(float log_s) {   // torch.sigmoid(log_s + 2.0)
    const float v = __fadd_rn(log_s, 2.0f);
    return 1.0f / (1.0f + expf(-v));
}

template <bool INV>
__global__ void __launch_bounds__(kBlock) coupling_kernel(float* __restrict__ x, const float* __restrict__ h, int HW,
                                                          int C, double* __restrict__ ldj,
                                                          const float* __restrict__ ldj_const) {
    __shared__ double red[kBlock / 64];
    const int Ch = C >> 1;
    const long base = (long)blockIdx.x * HW * C;
    const int n = HW * Ch;
    double acc = 0.0;
    for (int e = threadIdx.x; e < n; e += kBlock) {
        const int p = e / Ch, c = e - p * Ch;
        const long row = base + (long)p * C;
        const float log_s = h[row + c];
        const float t = h[row + Ch + c];
        const float s = sigmoid_p2(log_s);
        const float xb = x[row + Ch + c];
        if (!INV) {
            x[row + Ch + c] = __fmul_rn(__fadd_rn(xb, t), s);                        // (x_b + t) * s
            acc += (double)logf(__fadd_rn(s, 1e-6f));
        } else {
            x[row + Ch + c] = __fsub_rn(__fdiv_rn(xb, __fadd_rn(s, 1e-6f)), t);       // y_b/(s+1e-6) - t
        }
    }
    if (!INV && ldj) {
        const double s = block_sum_d(acc, red);
        if (threadIdx.x == 0) {
            double v = ldj[blockIdx.x];
            if (ldj_const) v = (v + (double)ldj_const[0]) + (double)ldj_const[1];
            ldj[blockIdx.x] = v + s;
        }
    }
}

// Large images (64x64 ... levels of a 128x128 model): one workgroup per sample leaves most of the chip idle (B = 32), so the
// pixels of a sample are cut into gridDim.y chunks; the chunk sums of log(s + 1e-6) go to part[b][chunk] and
// coupling_ldj_finish_kernel adds them to ldj in chunk order (deterministic).
template <bool INV>
__global__ void __launch_bounds__(kBlock) coupling_chunk_kernel(float* __restrict__ x, const float* __restrict__ h, int HW, int C,
                                                                double* __restrict__ part) {
    __shared__ double red[kBlock / 64];
    const int Ch = C >> 1;
    const long base = (long)blockIdx.x * HW * C;
    const int n = HW * Ch;
    const int per = (n + gridDim.y - 1) / gridDim.y;
    const int e0 = blockIdx.y * per, e1 = (e0 + per) < n ? (e0 + per) : n;
    double acc = 0.0;
    for (int e = e0 + threadIdx.x; e < e1; e += kBlock) {
        const int p = e / Ch, c = e - p * Ch;
        const long row = base + (long)p * C;
        const float log_s = h[row + c];
        const float t = h[row + Ch + c];
        const float s = sigmoid_p2(log_s);
        const float xb = x[row + Ch + c];
        if (!INV) {
            x[row + Ch + c] = __fmul_rn(__fadd_rn(xb, t), s);
            acc += (double)logf(__fadd_rn(s, 1e-6f));
        } else {
            x[row + Ch + c] = __fsub_rn(__fdiv_rn(xb, __fadd_rn(s, 1e-6f)), t);
        }
    }
    if (!INV && part) {
        const double s = block_sum_d(acc, red);
        if (threadIdx.x == 0) part[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = s;
    }
}

// The ZeroConv2d's tap shift-sum (backward.hip: tap_shift_sum_kernel) and the coupling in ONE pass: T [B*H*W][9*C] holds the tap
// contributions of the narrow 3x3 output conv (n = tap*C + co), and
//   h[p][co] = (sum_tap T[p + d(tap)][tap*C + co] + add[co]) * exp(f * mul[co])   (same expression and tap order: bit-identical)
// is consumed at once - (log_s | t) never goes to memory, two launches and a pass over h less per flow (the tiled path of images larger
// than the whole-image kernels' 16x16).  A thread owns VEC consecutive channel pairs of one pixel: VEC-wide loads of T (16 bytes for
// C % 8 == 0).  Grid (B, chunks): chunk sums of log(s + 1e-6) go to part[b][chunk] when gridDim.y > 1 (coupling_ldj_finish_kernel adds
// them in chunk order), else straight into ldj[b].
template <bool INV, int VEC>
__global__ void __launch_bounds__(kBlock) tap_coupling_kernel(float* __restrict__ x, const float* __restrict__ T, int H, int W, int C,
                                                              const float* __restrict__ add, const float* __restrict__ mul, float f,
                                                              double* __restrict__ ldj, const float* __restrict__ ldj_const,
                                                              double* __restrict__ part) {
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ double red[kBlock / 64];
    const int Ch = C >> 1, G = Ch / VEC, HW = H * W, ld = 9 * C;
    const long base = (long)blockIdx.x * HW;
    const int n = HW * G;
    const int per = (n + gridDim.y - 1) / gridDim.y;
    const int e0 = blockIdx.y * per, e1 = (e0 + per) < n ? (e0 + per) : n;
    double acc = 0.0;
    for (int e = e0 + threadIdx.x; e < e1; e += kBlock) {
        const int p = e / G, c = (e - p * G) * VEC;
        const int xx = p % W, y = p / W;
        float a_ls[VEC], a_t[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { a_ls[j] = 0.f; a_t[j] = 0.f; }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const bool ok = (unsigned)(y + dy) < (unsigned)H && (unsigned)(xx + dx) < (unsigned)W;
            // (an out-of-image tap reads the pixel's own row and adds +0: x + 0 == x bit for bit, the sums start from +0)
            const float* row = T + (base + (ok ? p + (long)dy * W + dx : p)) * ld + tap * C + c;
            const vec_t u = *reinterpret_cast<const vec_t*>(row), v = *reinterpret_cast<const vec_t*>(row + Ch);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { a_ls[j] += ok ? u[j] : 0.f; a_t[j] += ok ? v[j] : 0.f; }
        }
        float* xb = x + (base + p) * C + Ch + c;
        vec_t xv = *reinterpret_cast<const vec_t*>(xb);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float log_s = __fmul_rn(__fadd_rn(a_ls[j], add[c + j]), expf(__fmul_rn(f, mul[c + j])));
            float t = __fmul_rn(__fadd_rn(a_t[j], add[Ch + c + j]), expf(__fmul_rn(f, mul[Ch + c + j])));
            // (opaque: the two-pass path rounds (log_s | t) to fp32 in memory between the two kernels; here hipcc otherwise folds the
            // e^{f mul} factor into the sigmoid's argument arithmetic and the results differ by an ulp)
            asm volatile("" : "+v"(log_s), "+v"(t));
            const float s = sigmoid_p2(log_s);
            if (!INV) {
                xv[j] = __fmul_rn(__fadd_rn(xv[j], t), s);
                acc += (double)logf(__fadd_rn(s, 1e-6f));
            } else {
                xv[j] = __fsub_rn(__fdiv_rn(xv[j], __fadd_rn(s, 1e-6f)), t);
            }
        }
        *reinterpret_cast<vec_t*>(xb) = xv;
    }
    if (!INV && (ldj || part)) {
        const double s = block_sum_d(acc, red);
        if (threadIdx.x == 0) {
            if (part) part[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = s;
            else {
                double v = ldj[blockIdx.x];
                if (ldj_const) v = (v + (double)ldj_const[0]) + (double)ldj_const[1];
                ldj[blockIdx.x] = v + s;
            }
        }
    }
}

// ----------------------------------------------------------------------------- tap sums + coupling + the adjoining flow head
// tap_coupling_kernel with the per-pixel C x C product of the ADJOINING flow head behind the coupling, while the pixel is on chip:
//   forward:  x <- W' (e^{s'} (y + b'))   with y the coupling's output and (s', b', W') the NEXT flow's ActNorm / 1x1 conv
//             (reference glow.py:40-45: actnorm -> invconv -> coupling; the first flow's head stays a launch of its own);
//   inverse:  x <- (W^-1 y) e^{-s} - b    with (s, b, W^-1) THIS flow's (transforms.py:86-93, :139-146).
// One launch and one pass over the flow state per flow instead of two.  Same expressions in the same order as pixel_linear_kernel
// (e^s (v + b) per element, then one fmaf chain per output over ascending input channels): bit-identical to the two launches.
// A workgroup walks groups of PPB whole pixels: G = (C/2)/VEC threads per pixel compute the coupling for VEC channels each (as
// tap_coupling_kernel), the finished pixels (both halves) meet in LDS, then the same thread produces the 2 VEC outputs that replace the
// values it loaded.  Workgroup (0, 0) also emits the head's two batch-independent log-det terms (forward), as pixel_linear_kernel does.
template <bool INV, int C>
__global__ void __launch_bounds__(kBlock) tap_coupling_head_kernel(float* __restrict__ x, const float* __restrict__ T, int H, int W,
                                                                   const float* __restrict__ add, const float* __restrict__ mul, float f,
                                                                   double* __restrict__ ldj, const float* __restrict__ ldj_const,
                                                                   double* __restrict__ part, const float* __restrict__ hscale,
                                                                   const float* __restrict__ hbias, const float* __restrict__ hw,
                                                                   const float* __restrict__ hlad, float* __restrict__ hconst_out) {
    constexpr int Ch = C / 2, VEC = Ch % 4 == 0 ? 4 : 2, G = Ch / VEC, PPB = kBlock / G, LD = C + 4;
    static_assert(C % 4 == 0 && C <= 48, "channel counts of the tiled levels");
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ __attribute__((aligned(16))) float xs[PPB * LD];
    __shared__ __attribute__((aligned(16))) float ws[C * C];
    __shared__ float es[C], bs[C];
    __shared__ double red[kBlock / 64];
    const int HW = H * W, ld = 9 * C;
    const long base = (long)blockIdx.x * HW;
    for (int i = threadIdx.x; i < C * C; i += kBlock) ws[i] = hw[i];
    for (int c = threadIdx.x; c < C; c += kBlock) {
        es[c] = expf(INV ? -hscale[c] : hscale[c]);
        bs[c] = hbias[c];
    }
    if (!INV && hconst_out != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
        double sacc = 0.0;
        for (int c = threadIdx.x; c < C; c += 64) sacc += (double)hscale[c];
        sacc = wave_sum_d(sacc);
        if (threadIdx.x == 0) {
            hconst_out[0] = __fmul_rn((float)HW, (float)sacc);
            hconst_out[1] = hlad ? __fmul_rn((float)HW, hlad[0]) : 0.0f;
        }
    }
    const int per = (HW + gridDim.y - 1) / gridDim.y;
    const int q0 = blockIdx.y * per, q1 = (q0 + per) < HW ? (q0 + per) : HW;
    const int pl = threadIdx.x / G, g = threadIdx.x - pl * G, c = g * VEC;
    double acc = 0.0;
    for (int q = q0; q < q1; q += PPB) {
        const int p = q + pl;
        const bool valid = pl < PPB && p < q1;
        __syncthreads();                                     // the previous group's rows are consumed (first pass: ws / es / bs staged)
        if (valid) {
            const int xx = p % W, y = p / W;
            float a_ls[VEC], a_t[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) { a_ls[j] = 0.f; a_t[j] = 0.f; }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                const bool ok = (unsigned)(y + dy) < (unsigned)H && (unsigned)(xx + dx) < (unsigned)W;
                const float* row = T + (base + (ok ? p + (long)dy * W + dx : p)) * ld + tap * C + c;
                const vec_t u = *reinterpret_cast<const vec_t*>(row), v = *reinterpret_cast<const vec_t*>(row + Ch);
#pragma unroll
                for (int j = 0; j < VEC; ++j) { a_ls[j] += ok ? u[j] : 0.f; a_t[j] += ok ? v[j] : 0.f; }
            }
            const float* xp = x + (base + p) * C + c;
            const vec_t xa = *reinterpret_cast<const vec_t*>(xp);
            vec_t xv = *reinterpret_cast<const vec_t*>(xp + Ch);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float log_s = __fmul_rn(__fadd_rn(a_ls[j], add[c + j]), expf(__fmul_rn(f, mul[c + j])));
                float t = __fmul_rn(__fadd_rn(a_t[j], add[Ch + c + j]), expf(__fmul_rn(f, mul[Ch + c + j])));
                asm volatile("" : "+v"(log_s), "+v"(t));     // (as tap_coupling_kernel: the two-pass rounding points)
                const float s = sigmoid_p2(log_s);
                if (!INV) {
                    xv[j] = __fmul_rn(__fadd_rn(xv[j], t), s);
                    acc += (double)logf(__fadd_rn(s, 1e-6f));
                } else {
                    xv[j] = __fsub_rn(__fdiv_rn(xv[j], __fadd_rn(s, 1e-6f)), t);
                }
            }
            float* r = xs + pl * LD;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float va = xa[j], vb = xv[j];
                asm volatile("" : "+v"(va), "+v"(vb));       // the flow state as the two launches hand it over in memory
                r[c + j] = INV ? va : es[c + j] * (va + bs[c + j]);
                r[Ch + c + j] = INV ? vb : es[Ch + c + j] * (vb + bs[Ch + c + j]);
            }
        }
        __syncthreads();
        if (valid) {
            float v[C];
            const float4* r4 = reinterpret_cast<const float4*>(xs + pl * LD);
#pragma unroll
            for (int i = 0; i < C / 4; ++i) {
                const float4 t4 = r4[i];
                v[4 * i] = t4.x; v[4 * i + 1] = t4.y; v[4 * i + 2] = t4.z; v[4 * i + 3] = t4.w;
            }
            float* xp = x + (base + p) * C + c;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                vec_t out;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int o = half * Ch + c + j;
                    const float4* w4 = reinterpret_cast<const float4*>(ws + o * C);
                    float a = 0.0f;
#pragma unroll
                    for (int i = 0; i < C / 4; ++i) {
                        const float4 wv = w4[i];
                        a = fmaf(wv.x, v[4 * i], a);
                        a = fmaf(wv.y, v[4 * i + 1], a);
                        a = fmaf(wv.z, v[4 * i + 2], a);
                        a = fmaf(wv.w, v[4 * i + 3], a);
                    }
                    if (INV) a = __fsub_rn(__fmul_rn(a, es[o]), bs[o]);
                    out[j] = a;
                }
                *reinterpret_cast<vec_t*>(xp + half * Ch) = out;
            }
        }
    }
    if (!INV && (ldj || part)) {
        const double s = block_sum_d(acc, red);
        if (threadIdx.x == 0) {
            if (part) part[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = s;
            else {
                double v = ldj[blockIdx.x];
                if (ldj_const) v = (v + (double)ldj_const[0]) + (double)ldj_const[1];
                ldj[blockIdx.x] = v + s;
            }
        }
    }
}

__global__ void coupling_ldj_finish_kernel(double* __restrict__ ldj, int B, const double* __restrict__ part, int nch,
                                           const float* __restrict__ ldj_const) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int q = 0; q < nch; ++q) s += part[(size_t)b * nch + q];
    double v = ldj[b];
    if (ldj_const) v = (v + (double)ldj_const[0]) + (double)ldj_const[1];
    ldj[b] = v + s;
}

// H_MODE 0: h [B,HW,2Cz]; 1: h [2Cz] broadcast
template <int H_MODE>
__global__ void __launch_bounds__(kBlock) gauss_logp_kernel(const float* __restrict__ z, int z_ld,
                                                            const float* __restrict__ h, int HW, int Cz,
                                                            double* __restrict__ logp) {
    __shared__ double red[kBlock / 64];
    const long zb = (long)blockIdx.x * HW * z_ld;
    const long hb = (long)blockIdx.x * HW * 2 * Cz;
    const int n = HW * Cz;
    double acc = 0.0;
    for (int e = threadIdx.x; e < n; e += kBlock) {
        const int p = e / Cz, c = e - p * Cz;
        const float zv = z[zb + (long)p * z_ld + c];
        const float mean = H_MODE == 0 ? h[hb + (long)p * 2 * Cz + c] : h[c];
        const float logs = H_MODE == 0 ? h[hb + (long)p * 2 * Cz + Cz + c] : h[Cz + c];
        const float d = __fsub_rn(zv, mean);
        // -0.5 * (Log2PI + 2*logs + d^2 * exp(-2*logs)), fp32 op by op as prior.py:36
        const float t = __fmul_rn(__fmul_rn(d, d), expf(__fmul_rn(-2.0f, logs)));
        const float lp = __fmul_rn(-0.5f, __fadd_rn(__fadd_rn(kLog2Pi, __fmul_rn(2.0f, logs)), t));
        acc += (double)lp;
    }
    const double s = block_sum_d(acc, red);
    if (threadIdx.x == 0) logp[blockIdx.x] += s;
}

template <int H_MODE>
__global__ void __launch_bounds__(kBlock) gauss_sample_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                                                              float* __restrict__ out, int out_ld, long total, int HW,
                                                              int Cz, float temperature) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int c = (int)(i % Cz);
        const long bp = i / Cz;
        const float mean = H_MODE == 0 ? h[bp * 2 * Cz + c] : h[c];
        const float logs = H_MODE == 0 ? h[bp * 2 * Cz + Cz + c] : h[Cz + c];
        // mean + (exp(logs) * T) * eps, prior.py:50
        out[bp * out_ld + c] = __fadd_rn(mean, __fmul_rn(__fmul_rn(expf(logs), temperature), eps[i]));
    }
}

__global__ void zeroconv_bias_only_kernel(const float* __restrict__ bias, const float* __restrict__ logs,
                                          float* __restrict__ h, int n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) h[i] = __fmul_rn(bias[i], expf(__fmul_rn(logs[i], 3.0f)));   // (0 + bias) * exp(logs*3), utils.py:44
}

// ldj[b] += (double)( hw * (float)sum_i v[i] )
__global__ void __launch_bounds__(kBlock) ldj_add_scaled_kernel(double* __restrict__ ldj, int B,
                                                                const float* __restrict__ v, int n_sum, float hw) {
    __shared__ double red[kBlock / 64];
    __shared__ float term;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_sum; i += kBlock) acc += (double)v[i];
    const double s = block_sum_d(acc, red);
    if (threadIdx.x == 0) term = __fmul_rn(hw, (float)s);
    __syncthreads();
    const double t = (double)term;
    for (int b = threadIdx.x; b < B; b += kBlock) ldj[b] += t;
}

// out[0] = hw*sum(scale) (transforms.py:81), out[1] = hw*logabsdet (transforms.py:131), both fp32
__global__ void __launch_bounds__(kBlock) flow_ldj_consts_kernel(const float* __restrict__ scale, int C,
                                                                 const float* __restrict__ logabsdet, float hw,
                                                                 float* __restrict__ out) {
    __shared__ double red[kBlock / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < C; i += kBlock) acc += (double)scale[i];
    const double s = block_sum_d(acc, red);
    if (threadIdx.x == 0) {
        out[0] = __fmul_rn(hw, (float)s);
        out[1] = __fmul_rn(hw, logabsdet[0]);
    }
}

__global__ void __launch_bounds__(kBlock) bpd_sum_kernel(const double* __restrict__ ldj, const double* __restrict__ logp,
                                                         int B, double n_bins, double n_pixel, double* __restrict__ out) {
    __shared__ double red[kBlock / 64];
    const double c = 1.4426950408889634 / n_pixel;        // np.log2(np.e) / n_pixel
    const double k = log(n_bins) * n_pixel;
    double acc = 0.0;
    for (int b = threadIdx.x; b < B; b += kBlock) {
        const double ll = ldj[b] + (logp ? logp[b] : 0.0);
        acc += (k - ll) * c;
    }
    const double s = block_sum_d(acc, red);
    if (threadIdx.x == 0) {
        out[0] = s;
        out[1] = (double)B;
    }
}

}  // namespace

extern "C" int nfdp_coupling_fwd_f32(float* x, const float* h, int B, int HW, int C, double* ldj, const float* ldj_const,
                                     nfdp_stream_t stream) {
    NFDP_REQUIRE(x && h, NFDP_EINVAL, "coupling_fwd: null pointer");
    NFDP_REQUIRE(B >= 0 && HW > 0 && C > 0, NFDP_ESHAPE, "coupling_fwd: bad shape");
    NFDP_REQUIRE(C % 2 == 0, NFDP_ESHAPE, "coupling_fwd: C=%d must be even", C);
    if (B == 0) return NFDP_OK;
    hipLaunchKernelGGL((coupling_kernel<false>), dim3(B), dim3(kBlock), 0, as_stream(stream), x, h, HW, C, ldj, ldj_const);
    NFDP_CHECK_LAUNCH("coupling_fwd");
    return NFDP_OK;
}

static int coupling_chunks(int B, int HW, int C) {
    const long n = (long)HW * (C / 2);
    if (B <= 0 || n < 16384 || B >= 512) return 0;               // one workgroup per sample is fine
    long want = 1024 / B, cap = n / 4096;                          // ~4 workgroups per CU, at least 4096 elements each
    if (want > cap) want = cap;
    return want < 2 ? 0 : (int)want;
}

extern "C" size_t nfdp_coupling_chunked_ws_bytes(int B, int HW, int C) {
    const int nch = coupling_chunks(B, HW, C);
    return nch ? (size_t)B * nch * sizeof(double) : 0;
}

extern "C" int nfdp_coupling_fwd_chunked_f32(float* x, const float* h, int B, int HW, int C, double* ldj, const float* ldj_const,
                                             void* ws, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && h, NFDP_EINVAL, "coupling_fwd_chunked: null pointer");
    NFDP_REQUIRE(B >= 0 && HW > 0 && C > 0 && C % 2 == 0, NFDP_ESHAPE, "coupling_fwd_chunked: bad shape");
    const int nch = coupling_chunks(B, HW, C);
    NFDP_REQUIRE(nch > 0 && (ws || !ldj), NFDP_EINVAL, "coupling_fwd_chunked: not a chunked geometry or missing workspace");
    hipLaunchKernelGGL((coupling_chunk_kernel<false>), dim3(B, nch), dim3(kBlock), 0, as_stream(stream), x, h, HW, C,
                       ldj ? reinterpret_cast<double*>(ws) : nullptr);
    if (ldj)
        hipLaunchKernelGGL(coupling_ldj_finish_kernel, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0, as_stream(stream), ldj, B,
                           reinterpret_cast<const double*>(ws), nch, ldj_const);
    NFDP_CHECK_LAUNCH("coupling_fwd_chunked");
    return NFDP_OK;
}

// chunks per sample of the tap+coupling pass: ~1024 thread-items (pixel x channel group) per workgroup, one chunk for small images
static int tap_vec(int C) { return (C / 2) % 4 == 0 ? 4 : ((C / 2) % 2 == 0 ? 2 : 1); }
static int tap_chunks(int B, int HW, int C) {
    const long n = (long)HW * ((C / 2) / tap_vec(C));
    if (B <= 0 || n <= 2048) return 1;
    long nch = (n + 1023) / 1024;
    return (int)(nch > 256 ? 256 : nch);
}

extern "C" size_t nfdp_tap_coupling_ws_bytes(int B, int HW, int C) {
    const int nch = tap_chunks(B, HW, C);
    return nch > 1 ? (size_t)B * nch * sizeof(double) : 0;
}

extern "C" int nfdp_tap_coupling_f32(float* x, const float* T, int B, int H, int W, int C, const float* add, const float* mul, float f,
                                     double* ldj, const float* ldj_const, void* ws, int inverse, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && T && add && mul, NFDP_EINVAL, "tap_coupling: null pointer");
    NFDP_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && C % 2 == 0, NFDP_ESHAPE, "tap_coupling: bad shape");
    if (B == 0) return NFDP_OK;
    const int nch = tap_chunks(B, H * W, C), vec = tap_vec(C);
    NFDP_REQUIRE(vec == 1 || ((((uintptr_t)x | (uintptr_t)T) & 15) == 0), NFDP_ESHAPE, "tap_coupling: tensors must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    double* part = nullptr;
    if (!inverse) {
        NFDP_REQUIRE(nch == 1 || ws || !ldj, NFDP_EINVAL, "tap_coupling: needs nfdp_tap_coupling_ws_bytes of workspace");
        part = (nch > 1 && ldj) ? reinterpret_cast<double*>(ws) : nullptr;
    }
    double* l = inverse ? nullptr : ldj;
    const float* lc = inverse ? nullptr : ldj_const;
#define NFDP_TC(INVV, V) hipLaunchKernelGGL((tap_coupling_kernel<INVV, V>), dim3(B, nch), dim3(kBlock), 0, s, x, T, H, W, C, add, mul, f, l, lc, part)
    if (inverse) { if (vec == 4) NFDP_TC(true, 4); else if (vec == 2) NFDP_TC(true, 2); else NFDP_TC(true, 1); }
    else { if (vec == 4) NFDP_TC(false, 4); else if (vec == 2) NFDP_TC(false, 2); else NFDP_TC(false, 1); }
#undef NFDP_TC
    if (part)
        hipLaunchKernelGGL(coupling_ldj_finish_kernel, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0, s, ldj, B, part, nch, ldj_const);
    NFDP_CHECK_LAUNCH("tap_coupling");
    return NFDP_OK;
}

// chunks per sample of the head-fused pass: 1-4 groups of PPB whole pixels per workgroup, ~768+ workgroups when the batch has that many groups (measured: 1568 one-group workgroups are slower than 800 of two)
static int tap_head_ppb(int C) { return kBlock / ((C / 2) / tap_vec(C)); }
static int tap_head_chunks(int B, int HW, int C) {
    const int ppb = tap_head_ppb(C);
    const long groups = (HW + ppb - 1) / ppb;
    long k = (long)B * groups / 768;
    k = k < 1 ? 1 : (k > 4 ? 4 : k);
    const long nch = (groups + k - 1) / k;
    return (int)(nch > 256 ? 256 : nch);
}

extern "C" int nfdp_tap_coupling_head_supported(int C) {
    return C == 4 || C == 8 || C == 12 || C == 16 || C == 24 || C == 32 || C == 48;
}

extern "C" size_t nfdp_tap_coupling_head_ws_bytes(int B, int HW, int C) {
    if (!nfdp_tap_coupling_head_supported(C)) return 0;
    const int nch = tap_head_chunks(B, HW, C);
    return (size_t)B * nch * sizeof(double);          // (one chunk: needed only by a launch that defers its log-det sum)
}

extern "C" int nfdp_tap_coupling_head_chunks(int B, int HW, int C) {
    return nfdp_tap_coupling_head_supported(C) ? tap_head_chunks(B, HW, C) : 0;
}

extern "C" int nfdp_tap_coupling_head_f32(float* x, const float* T, int B, int H, int W, int C, const float* add, const float* mul, float f,
                                          double* ldj, const float* ldj_const, void* ws, int inverse, const float* head_scale,
                                          const float* head_bias, const float* head_w, const float* head_logabsdet, float* head_const_out,
                                          int defer_ldj, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && T && add && mul && head_scale && head_bias && head_w, NFDP_EINVAL, "tap_coupling_head: null pointer");
    NFDP_REQUIRE(B >= 0 && H > 0 && W > 0, NFDP_ESHAPE, "tap_coupling_head: bad shape");
    NFDP_REQUIRE(nfdp_tap_coupling_head_supported(C), NFDP_EUNSUPPORTED, "tap_coupling_head: unsupported C=%d", C);
    if (B == 0) return NFDP_OK;
    NFDP_REQUIRE((((uintptr_t)x | (uintptr_t)T) & 15) == 0 || C % 8 != 0, NFDP_ESHAPE, "tap_coupling_head: tensors must be 16-byte aligned");
    NFDP_REQUIRE((((uintptr_t)x | (uintptr_t)T) & 7) == 0, NFDP_ESHAPE, "tap_coupling_head: tensors must be 8-byte aligned");
    const int nch = tap_head_chunks(B, H * W, C);
    hipStream_t s = as_stream(stream);
    double* part = nullptr;
    const bool defer = !inverse && ldj && defer_ldj;       // chunk sums stay in ws: nfdp_ldj_finish_level_f64 adds a level's launches up at once
    if (!inverse) {
        NFDP_REQUIRE((nch == 1 && !defer) || ws || !ldj, NFDP_EINVAL, "tap_coupling_head: needs nfdp_tap_coupling_head_ws_bytes of workspace");
        part = ((nch > 1 || defer) && ldj) ? reinterpret_cast<double*>(ws) : nullptr;
    }
    double* l = inverse ? nullptr : ldj;
    const float* lc = inverse ? nullptr : ldj_const;
#define NFDP_TCH(CC)                                                                                                                     \
    case CC:                                                                                                                             \
        if (inverse)                                                                                                                     \
            hipLaunchKernelGGL((tap_coupling_head_kernel<true, CC>), dim3(B, nch), dim3(kBlock), 0, s, x, T, H, W, add, mul, f, l, lc,  \
                               part, head_scale, head_bias, head_w, head_logabsdet, head_const_out);                                    \
        else                                                                                                                             \
            hipLaunchKernelGGL((tap_coupling_head_kernel<false, CC>), dim3(B, nch), dim3(kBlock), 0, s, x, T, H, W, add, mul, f, l, lc, \
                               part, head_scale, head_bias, head_w, head_logabsdet, head_const_out);                                    \
        break;
    switch (C) {
        NFDP_TCH(4) NFDP_TCH(8) NFDP_TCH(12) NFDP_TCH(16) NFDP_TCH(24) NFDP_TCH(32) NFDP_TCH(48)
        default: break;
    }
#undef NFDP_TCH
    if (part && !defer)
        hipLaunchKernelGGL(coupling_ldj_finish_kernel, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0, s, ldj, B, part, nch, ldj_const);
    NFDP_CHECK_LAUNCH("tap_coupling_head");
    return NFDP_OK;
}

// The deferred log-det sums of up to 32 launches (the flows of a level) in ONE launch, in launch order and with each launch's two
// constants, exactly as the launches' own finish kernels would have added them one after the other: bit-identical ldj.
struct LdjLevel {
    const double* part[32];
    const float* cst[32];
    int n, nch;
};
__global__ void ldj_finish_level_kernel(double* __restrict__ ldj, int B, const LdjLevel lv) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    double v = ldj[b];
    for (int f = 0; f < lv.n; ++f) {
        double s = 0.0;
        for (int q = 0; q < lv.nch; ++q) s += lv.part[f][(size_t)b * lv.nch + q];
        if (lv.cst[f]) v = (v + (double)lv.cst[f][0]) + (double)lv.cst[f][1];
        v = v + s;
    }
    ldj[b] = v;
}

extern "C" int nfdp_ldj_finish_level_f64(double* ldj, int B, const void* const* parts, const float* const* consts, int n, int nch,
                                         nfdp_stream_t stream) {
    NFDP_REQUIRE(ldj && parts && consts, NFDP_EINVAL, "ldj_finish_level: null pointer");
    NFDP_REQUIRE(B >= 0 && n >= 0 && nch >= 1, NFDP_ESHAPE, "ldj_finish_level: bad shape");
    if (B == 0 || n == 0) return NFDP_OK;
    for (int i = 0; i < n; i += 32) {
        LdjLevel lv;
        lv.n = n - i < 32 ? n - i : 32;
        lv.nch = nch;
        for (int f = 0; f < lv.n; ++f) {
            NFDP_REQUIRE(parts[i + f], NFDP_EINVAL, "ldj_finish_level: null partial-sum buffer");
            lv.part[f] = reinterpret_cast<const double*>(parts[i + f]);
            lv.cst[f] = consts[i + f];
        }
        hipLaunchKernelGGL(ldj_finish_level_kernel, dim3((B + kBlock - 1) / kBlock), dim3(kBlock), 0, as_stream(stream), ldj, B, lv);
    }
    NFDP_CHECK_LAUNCH("ldj_finish_level");
    return NFDP_OK;
}

extern "C" int nfdp_coupling_inv_f32(float* y, const float* h, int B, int HW, int C, nfdp_stream_t stream) {
    NFDP_REQUIRE(y && h, NFDP_EINVAL, "coupling_inv: null pointer");
    NFDP_REQUIRE(B >= 0 && HW > 0 && C > 0, NFDP_ESHAPE, "coupling_inv: bad shape");
    NFDP_REQUIRE(C % 2 == 0, NFDP_ESHAPE, "coupling_inv: C=%d must be even", C);
    if (B == 0) return NFDP_OK;
    const int nch = coupling_chunks(B, HW, C);
    if (nch)
        hipLaunchKernelGGL((coupling_chunk_kernel<true>), dim3(B, nch), dim3(kBlock), 0, as_stream(stream), y, h, HW, C, nullptr);
    else
        hipLaunchKernelGGL((coupling_kernel<true>), dim3(B), dim3(kBlock), 0, as_stream(stream), y, h, HW, C, nullptr, nullptr);
    NFDP_CHECK_LAUNCH("coupling_inv");
    return NFDP_OK;
}

extern "C" int nfdp_ldj_add_scaled_f32(double* ldj, int B, const float* v, int n_sum, float hw, nfdp_stream_t stream) {
    NFDP_REQUIRE(ldj && v, NFDP_EINVAL, "ldj_add_scaled: null pointer");
    NFDP_REQUIRE(B >= 0 && n_sum > 0, NFDP_ESHAPE, "ldj_add_scaled: bad shape");
    if (B == 0) return NFDP_OK;
    hipLaunchKernelGGL(ldj_add_scaled_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), ldj, B, v, n_sum, hw);
    NFDP_CHECK_LAUNCH("ldj_add_scaled");
    return NFDP_OK;
}

extern "C" int nfdp_flow_ldj_consts_f32(const float* scale, int C, const float* logabsdet, float hw, float* out2,
                                        nfdp_stream_t stream) {
    NFDP_REQUIRE(scale && logabsdet && out2, NFDP_EINVAL, "flow_ldj_consts: null pointer");
    NFDP_REQUIRE(C > 0, NFDP_ESHAPE, "flow_ldj_consts: bad C");
    hipLaunchKernelGGL(flow_ldj_consts_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), scale, C, logabsdet, hw, out2);
    NFDP_CHECK_LAUNCH("flow_ldj_consts");
    return NFDP_OK;
}

extern "C" int nfdp_gauss_logp_f32(const float* z, int z_ld, const float* h, int h_mode, int B, int HW, int Cz,
                                   double* logp, nfdp_stream_t stream) {
    NFDP_REQUIRE(z && h && logp, NFDP_EINVAL, "gauss_logp: null pointer");
    NFDP_REQUIRE(B >= 0 && HW > 0 && Cz > 0 && z_ld >= Cz, NFDP_ESHAPE, "gauss_logp: bad shape");
    NFDP_REQUIRE(h_mode == 0 || h_mode == 1, NFDP_EINVAL, "gauss_logp: h_mode=%d", h_mode);
    if (B == 0) return NFDP_OK;
    if (h_mode == 0)
        hipLaunchKernelGGL((gauss_logp_kernel<0>), dim3(B), dim3(kBlock), 0, as_stream(stream), z, z_ld, h, HW, Cz, logp);
    else
        hipLaunchKernelGGL((gauss_logp_kernel<1>), dim3(B), dim3(kBlock), 0, as_stream(stream), z, z_ld, h, HW, Cz, logp);
    NFDP_CHECK_LAUNCH("gauss_logp");
    return NFDP_OK;
}

extern "C" int nfdp_gauss_sample_f32(const float* h, int h_mode, const float* eps, float* out, int out_ld, int B, int HW,
                                     int Cz, float temperature, nfdp_stream_t stream) {
    NFDP_REQUIRE(h && eps && out, NFDP_EINVAL, "gauss_sample: null pointer");
    NFDP_REQUIRE(B >= 0 && HW > 0 && Cz > 0 && out_ld >= Cz, NFDP_ESHAPE, "gauss_sample: bad shape");
    NFDP_REQUIRE(h_mode == 0 || h_mode == 1, NFDP_EINVAL, "gauss_sample: h_mode=%d", h_mode);
    const long total = (long)B * HW * Cz;
    if (total == 0) return NFDP_OK;
    if (h_mode == 0)
        hipLaunchKernelGGL((gauss_sample_kernel<0>), dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream),
                           h, eps, out, out_ld, total, HW, Cz, temperature);
    else
        hipLaunchKernelGGL((gauss_sample_kernel<1>), dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream),
                           h, eps, out, out_ld, total, HW, Cz, temperature);
    NFDP_CHECK_LAUNCH("gauss_sample");
    return NFDP_OK;
}

extern "C" int nfdp_zeroconv_bias_only_f32(const float* bias, const float* logs, float* h, int n, nfdp_stream_t stream) {
    NFDP_REQUIRE(bias && logs && h, NFDP_EINVAL, "zeroconv_bias_only: null pointer");
    NFDP_REQUIRE(n > 0, NFDP_ESHAPE, "zeroconv_bias_only: bad n");
    hipLaunchKernelGGL(zeroconv_bias_only_kernel, dim3((int)cdiv(n, kBlock)), dim3(kBlock), 0, as_stream(stream), bias, logs,
                       h, n);
    NFDP_CHECK_LAUNCH("zeroconv_bias_only");
    return NFDP_OK;
}

extern "C" int nfdp_bpd_sum_f64(const double* ldj, const double* logp, int B, double n_bins, double n_pixel, double* out,
                                nfdp_stream_t stream) {
    NFDP_REQUIRE(ldj && out, NFDP_EINVAL, "bpd_sum: null pointer");
    NFDP_REQUIRE(B > 0 && n_bins > 0 && n_pixel > 0, NFDP_ESHAPE, "bpd_sum: bad arguments");
    hipLaunchKernelGGL(bpd_sum_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), ldj, logp, B, n_bins, n_pixel, out);
    NFDP_CHECK_LAUNCH("bpd_sum");
    return NFDP_OK;
}
