// U-Net building blocks of the diffusion prior's eps-predictor on NHWC fp32 tensors (the convolutions themselves
// run in conv.hip).  Reference: diffusion_prior/gaussian_diffusion.py
//   WeightStandardizedConv2d :183-197   LayerNorm :200-209      SinusoidalPosEmb :225-237   time_mlp :407-412
//   Block :260-276 (GroupNorm + scale/shift + SiLU)             ResnetBlock mlp :282-285
//   LinearAttention :305-333            Attention :336-358      Upsample :169-173
// These are HBM/latency-bound glue ops; each is one launch with fp64 statistics where the reference reduces.
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ double block_sum256(double v, double* red) {
    v = wave_sum_d(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return s;                                   // valid in every thread
}

// ----------------------------------------------------------------------------- weight standardisation (:189-197)
// one workgroup per output channel: w' = (w - mean) * rsqrt(var_biased + eps)
__global__ void __launch_bounds__(kBlock) ws_standardize_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                               int fan_in, float eps) {
    __shared__ double red[4];
    const float* wr = w + (long)blockIdx.x * fan_in;
    float* orow = out + (long)blockIdx.x * fan_in;
    double s = 0.0;
    for (int i = threadIdx.x; i < fan_in; i += kBlock) s += (double)wr[i];
    const double mean = block_sum256(s, red) / fan_in;
    double q = 0.0;
    for (int i = threadIdx.x; i < fan_in; i += kBlock) {
        const double d = (double)wr[i] - mean;
        q += d * d;
    }
    const double var = block_sum256(q, red) / fan_in;
    const float m = (float)mean, r = rsqrtf(__fadd_rn((float)var, eps));
    for (int i = threadIdx.x; i < fan_in; i += kBlock) orow[i] = __fmul_rn(__fsub_rn(wr[i], m), r);
}

// ----------------------------------------------------------------------------- GroupNorm + scale/shift + SiLU (:267-276)
// one workgroup per (sample, group); x,y: [B][HW][C]; ss: [B][2C] (scale | shift) or null
__global__ void __launch_bounds__(kBlock) groupnorm_silu_kernel(const float* __restrict__ x, float* __restrict__ y, int HW,
                                                               int C, int groups, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ ss, int ss_ld, float eps, int lds_cached,
                                                               const float* __restrict__ res) {
    __shared__ double red[4];
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups, n = HW * cpg;
    const long base = (long)b * HW * C + (long)g * cpg;
    constexpr int RC = 16;                                  // register-cached elements per thread
    if (n <= kBlock * RC && cpg % 4 == 0) {
        // one global read: each thread keeps its float4 pieces (4 consecutive channels of one pixel) in registers
        const int q = cpg / 4, nv = HW * q;
        float4 v[RC / 4];
        double s1 = 0.0;
#pragma unroll
        for (int k = 0; k < RC / 4; ++k) {
            const int e = threadIdx.x + k * kBlock;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < nv) {
                v[k] = *reinterpret_cast<const float4*>(x + base + (long)(e / q) * C + (e % q) * 4);
                s1 += (double)v[k].x + (double)v[k].y + (double)v[k].z + (double)v[k].w;
            }
        }
        // everything else the output pass reads is requested now, so its latency hides under the two block reductions: this
        // thread's 4 channels are the same for every k when q divides the block size (q = cpg/4 is 1, 2 or 4 for the models here)
        const bool same_c = (kBlock % q) == 0;
        const int cp = g * cpg + (threadIdx.x % q) * 4;
        float4 gpre = make_float4(0.f, 0.f, 0.f, 0.f), bpre = gpre, sspre = gpre, shpre = gpre, rpre[RC / 4];
        if (same_c) {
            gpre = *reinterpret_cast<const float4*>(gamma + cp);
            bpre = *reinterpret_cast<const float4*>(beta + cp);
            if (ss) {
                const float* sr = ss + (long)b * ss_ld;
                sspre = make_float4(sr[cp], sr[cp + 1], sr[cp + 2], sr[cp + 3]);
                shpre = make_float4(sr[C + cp], sr[C + cp + 1], sr[C + cp + 2], sr[C + cp + 3]);
            }
        }
        if (res) {
#pragma unroll
            for (int k = 0; k < RC / 4; ++k) {
                const int e = threadIdx.x + k * kBlock;
                rpre[k] = e < nv ? *reinterpret_cast<const float4*>(res + base + (long)(e / q) * C + (e % q) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const double mean_d = block_sum256(s1, red) / n;
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < RC / 4; ++k) {
            if (threadIdx.x + k * kBlock < nv) {
                const double d0 = v[k].x - mean_d, d1 = v[k].y - mean_d, d2 = v[k].z - mean_d, d3 = v[k].w - mean_d;
                s2 += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
        }
        const double var_d = block_sum256(s2, red) / n;
        const float m = (float)mean_d, r = (float)(1.0 / sqrt(var_d + (double)eps));
#pragma unroll
        for (int k = 0; k < RC / 4; ++k) {
            const int e = threadIdx.x + k * kBlock;
            if (e < nv) {
                const int c0 = g * cpg + (e % q) * 4;
                float o[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                const float gq[4] = {gpre.x, gpre.y, gpre.z, gpre.w}, bq[4] = {bpre.x, bpre.y, bpre.z, bpre.w};
                const float sq[4] = {sspre.x, sspre.y, sspre.z, sspre.w}, hq[4] = {shpre.x, shpre.y, shpre.z, shpre.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ga = same_c ? gq[j] : gamma[c0 + j], be = same_c ? bq[j] : beta[c0 + j];
                    float t = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(o[j], m), r), ga), be);
                    if (ss) {
                        const float sc = same_c ? sq[j] : ss[(long)b * ss_ld + c0 + j], sh = same_c ? hq[j] : ss[(long)b * ss_ld + C + c0 + j];
                        t = __fadd_rn(__fmul_rn(t, __fadd_rn(sc, 1.0f)), sh);
                    }
                    o[j] = t / (1.0f + expf(-t));
                }
                const long off = base + (long)(e / q) * C + (e % q) * 4;
                if (res) {                                         // ResnetBlock's  h + res_conv(x)  (:301), same fp32 add as a separate pass
                    const float4 rv = rpre[k];
                    o[0] = __fadd_rn(o[0], rv.x); o[1] = __fadd_rn(o[1], rv.y); o[2] = __fadd_rn(o[2], rv.z); o[3] = __fadd_rn(o[3], rv.w);
                }
                *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        return;
    }
    // larger groups (the 64x64 / 32x32 latent parts of the larger models): the group is staged in LDS when it fits (one global
    // read instead of three), float4 pieces, same fp64 statistics
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    if (lds_cached && cpg % 4 == 0) {
        const int q = cpg / 4, nv = HW * q;
        float4* g4 = reinterpret_cast<float4*>(gsm);
        double s1 = 0.0;
        for (int e = threadIdx.x; e < nv; e += kBlock) {
            const float4 v = *reinterpret_cast<const float4*>(x + base + (long)(e / q) * C + (e % q) * 4);
            g4[e] = v;
            s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        }
        const double mean_d = block_sum256(s1, red) / n;
        double s2 = 0.0;
        for (int e = threadIdx.x; e < nv; e += kBlock) {
            const float4 v = g4[e];
            const double d0 = v.x - mean_d, d1 = v.y - mean_d, d2 = v.z - mean_d, d3 = v.w - mean_d;
            s2 += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        const double var_d = block_sum256(s2, red) / n;
        const float m = (float)mean_d, r = (float)(1.0 / sqrt(var_d + (double)eps));
        for (int e = threadIdx.x; e < nv; e += kBlock) {
            const int c0 = g * cpg + (e % q) * 4;
            const float4 v = g4[e];
            float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(o[j], m), r), gamma[c0 + j]), beta[c0 + j]);
                if (ss) t = __fadd_rn(__fmul_rn(t, __fadd_rn(ss[(long)b * ss_ld + c0 + j], 1.0f)), ss[(long)b * ss_ld + C + c0 + j]);
                o[j] = t / (1.0f + expf(-t));
            }
            const long off = base + (long)(e / q) * C + (e % q) * 4;
            if (res) {
                const float4 rv = *reinterpret_cast<const float4*>(res + off);
                o[0] = __fadd_rn(o[0], rv.x); o[1] = __fadd_rn(o[1], rv.y); o[2] = __fadd_rn(o[2], rv.z); o[3] = __fadd_rn(o[3], rv.w);
            }
            *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    double s = 0.0;
    for (int e = threadIdx.x; e < n; e += kBlock) s += (double)x[base + (long)(e / cpg) * C + (e % cpg)];
    const double mean = block_sum256(s, red) / n;
    double q = 0.0;
    for (int e = threadIdx.x; e < n; e += kBlock) {
        const double d = (double)x[base + (long)(e / cpg) * C + (e % cpg)] - mean;
        q += d * d;
    }
    const double var = block_sum256(q, red) / n;
    const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)eps));
    for (int e = threadIdx.x; e < n; e += kBlock) {
        const int c = g * cpg + (e % cpg);
        const long idx = base + (long)(e / cpg) * C + (e % cpg);
        float v = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(x[idx], m), r), gamma[c]), beta[c]);
        if (ss) v = __fadd_rn(__fmul_rn(v, __fadd_rn(ss[(long)b * ss_ld + c], 1.0f)), ss[(long)b * ss_ld + C + c]);
        v = v / (1.0f + expf(-v));                           // SiLU = v * sigmoid(v)
        y[idx] = res ? __fadd_rn(v, res[idx]) : v;
    }
}

// ----------------------------------------------------------------------------- channel LayerNorm, gain only (:205-209)
// one wave per pixel; optional residual add (Residual(PreNorm(...)) :160-166 folds in at the caller's choice)
__global__ void __launch_bounds__(kBlock) chan_layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               long npix, int C, const float* __restrict__ gain, float eps,
                                                               const float* __restrict__ residual) {
    const int lane = threadIdx.x & 63;
    if (C % 4 == 0 && C <= 256 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)gain & 15) == 0 &&
        (residual == nullptr || ((uintptr_t)residual & 15) == 0)) {
        // A lane owns 4 consecutive channels (one 16-byte access): a C = 64 row takes 16 lanes, so a wave normalises 4 pixels at once
        // with 4-step reductions (one lane per channel left 48 of 64 lanes of every access idle at C = 64 and walked 6 shuffle steps of
        // doubles twice per pixel: 6.3 us per launch at 16 K pixels against a 2.2 us floor; now 3.4 us).  LPP = lanes per pixel.
        const int LPP = C <= 64 ? 16 : (C <= 128 ? 32 : 64), PPW = 64 / LPP;
        const int sub = lane / LPP, l = lane % LPP;
        const bool on = 4 * l < C;
        const float4 g4 = on ? *reinterpret_cast<const float4*>(gain + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
        const long stride = (long)gridDim.x * 4 * PPW;
        for (long p0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PPW; p0 < npix; p0 += stride) {
            const long p = p0 + sub;
            const bool live = on && p < npix;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f), rv = v;
            if (live) {
                v = *reinterpret_cast<const float4*>(x + p * C + 4 * l);
                if (residual) rv = *reinterpret_cast<const float4*>(residual + p * C + 4 * l);
            }
            double s1 = live ? (double)v.x + (double)v.y + (double)v.z + (double)v.w : 0.0;
            for (int o = LPP / 2; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
            const double mean = s1 / C;
            double s2 = 0.0;
            if (live) {
                const double d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
                s2 = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
            for (int o = LPP / 2; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
            const float m = (float)mean, r = rsqrtf(__fadd_rn((float)(s2 / C), eps));
            if (live) {
                float4 ov;
                ov.x = __fmul_rn(__fmul_rn(__fsub_rn(v.x, m), r), g4.x);
                ov.y = __fmul_rn(__fmul_rn(__fsub_rn(v.y, m), r), g4.y);
                ov.z = __fmul_rn(__fmul_rn(__fsub_rn(v.z, m), r), g4.z);
                ov.w = __fmul_rn(__fmul_rn(__fsub_rn(v.w, m), r), g4.w);
                if (residual) { ov.x = __fadd_rn(ov.x, rv.x); ov.y = __fadd_rn(ov.y, rv.y); ov.z = __fadd_rn(ov.z, rv.z); ov.w = __fadd_rn(ov.w, rv.w); }
                *reinterpret_cast<float4*>(y + p * C + 4 * l) = ov;
            }
        }
        return;
    }
    if (C <= 256) {
        // the row lives in registers (one global read) and the gain / residual are requested with it, before the two reductions
        float gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) gv[k] = lane + 64 * k < C ? gain[lane + 64 * k] : 0.f;
        for (long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6); p < npix; p += (long)gridDim.x * 4) {
            const float* xr = x + p * C;
            float xv[4], rv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = lane + 64 * k;
                xv[k] = c < C ? xr[c] : 0.f;
                rv[k] = (residual && c < C) ? residual[p * C + c] : 0.f;
            }
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (lane + 64 * k < C) s += (double)xv[k];
            const double mean = wave_sum_d(s) / C;
            double q = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (lane + 64 * k < C) {
                    const double d = (double)xv[k] - mean;
                    q += d * d;
                }
            const double var = wave_sum_d(q) / C;
            const float m = (float)mean, r = rsqrtf(__fadd_rn((float)var, eps));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = lane + 64 * k;
                if (c < C) {
                    float v = __fmul_rn(__fmul_rn(__fsub_rn(xv[k], m), r), gv[k]);
                    if (residual) v = __fadd_rn(v, rv[k]);
                    y[p * C + c] = v;
                }
            }
        }
        return;
    }
    for (long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6); p < npix; p += (long)gridDim.x * 4) {
        const float* xr = x + p * C;
        double s = 0.0;
        for (int c = lane; c < C; c += 64) s += (double)xr[c];
        const double mean = wave_sum_d(s) / C;
        double q = 0.0;
        for (int c = lane; c < C; c += 64) {
            const double d = (double)xr[c] - mean;
            q += d * d;
        }
        const double var = wave_sum_d(q) / C;
        const float m = (float)mean, r = rsqrtf(__fadd_rn((float)var, eps));
        for (int c = lane; c < C; c += 64) {
            float v = __fmul_rn(__fmul_rn(__fsub_rn(xr[c], m), r), gain[c]);
            if (residual) v = __fadd_rn(v, residual[p * C + c]);
            y[p * C + c] = v;
        }
    }
}

// ----------------------------------------------------------------------------- small linear with fused activations
// out[b][o] = post( bias[o] + sum_i pre(in[b][i]) * W[o][i] );  pre: 0 none, 1 SiLU;  post: 0 none, 1 GELU(erf), 2 SiLU
// One workgroup = 16 output features x up to 64 batch rows: pre(in) [64][I] and W [16][I] are staged in LDS with
// coalesced loads (rows padded by one word: conflict-free), each thread then owns 4 (b, o) pairs.
constexpr int LIN_TO = 16, LIN_TB = 64;
__global__ void __launch_bounds__(kBlock) linear_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int B, int I,
                                                       int O, int out_ld, int pre, int post) {
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    const int IP = I + 1;
    float* xs = lsm;                   // [LIN_TB][IP]
    float* ws = lsm + LIN_TB * IP;     // [LIN_TO][IP]
    const int o0 = blockIdx.x * LIN_TO, b0 = blockIdx.y * LIN_TB;
    if ((I & 3) == 0 && (((uintptr_t)in | (uintptr_t)w) & 15) == 0) {
        // 16-byte loads, eight in flight per thread before LDS is touched (an element-wise staging loop waits out one load
        // latency per element: 64 dependent round trips for I = 256 made this 37 us for a [64 x 256] x [256 x 256] product)
        const int I4 = I >> 2;
        auto stage = [&](const float* __restrict__ src, int row0, int rows, int nrows, float* dst, bool act) {
            const int total = nrows * I4;
            for (int base = threadIdx.x; base < total; base += kBlock * 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = base + u * kBlock;
                    const int r = e / I4, c = e - r * I4;
                    v[u] = (e < total && row0 + r < rows) ? *reinterpret_cast<const float4*>(src + (long)(row0 + r) * I + 4 * c)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = base + u * kBlock;
                    if (e >= total) continue;
                    const int r = e / I4, c = e - r * I4;
                    float t[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    if (act) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[j] = t[j] / (1.0f + expf(-t[j]));
                    }
                    float* d = dst + r * IP + 4 * c;
                    d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3];
                }
            }
        };
        stage(in, b0, B, LIN_TB, xs, pre == 1);
        stage(w, o0, O, LIN_TO, ws, false);
    } else {
        for (int e = threadIdx.x; e < LIN_TB * I; e += kBlock) {
            const int r = e / I, i = e % I;
            float v = (b0 + r < B) ? in[(long)(b0 + r) * I + i] : 0.f;
            if (pre == 1) v = v / (1.0f + expf(-v));
            xs[r * IP + i] = v;
        }
        for (int e = threadIdx.x; e < LIN_TO * I; e += kBlock) {
            const int r = e / I, i = e % I;
            ws[r * IP + i] = (o0 + r < O) ? w[(long)(o0 + r) * I + i] : 0.f;
        }
    }
    __syncthreads();
    const int bl = threadIdx.x % LIN_TB, og = threadIdx.x / LIN_TB;      // og in 0..3 -> outputs 4og..4og+3
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xr = xs + bl * IP;
    for (int i = 0; i < I; ++i) {
        const float xv = xr[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv, ws[(4 * og + j) * IP + i], acc[j]);
    }
    if (b0 + bl < B) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = o0 + 4 * og + j;
            if (o < O) {
                float v = acc[j];
                if (bias) v = __fadd_rn(v, bias[o]);
                if (post == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                else if (post == 2) v = v / (1.0f + expf(-v));
                out[(long)(b0 + bl) * out_ld + o] = v;
            }
        }
    }
}

// sinusoidal position embedding (:231-237): [sin(t f_j) | cos(t f_j)], f_j = exp(-j log(10000)/(half-1))
__global__ void sinusoidal_kernel(const long* __restrict__ t, float* __restrict__ out, int B, int dim) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    const int half = dim / 2;
    if (idx >= B * half) return;
    const int b = idx / half, j = idx % half;
    const float step = (float)(-(9.210340371976184 / (double)(half - 1)));       // -log(10000)/(half-1)
    const float f = expf(__fmul_rn((float)j, step));
    const float a = __fmul_rn((float)t[b], f);
    out[(long)b * dim + j] = sinf(a);
    out[(long)b * dim + half + j] = cosf(a);
}

// random / learned sinusoidal embedding (RandomOrLearnedSinusoidalPosEmb, :240-255): out[b] = [t | sin(a_j) | cos(a_j)] with
// a_j = ((t w_j) 2) pi in fp32 (the reference's `x * weights * 2 * math.pi`, left to right), row length 2*half + 1
__global__ void learned_sinusoidal_kernel(const long* __restrict__ t, const float* __restrict__ w, float* __restrict__ out, int B, int half) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= B * half) return;
    const int b = idx / half, j = idx % half;
    const float tv = (float)t[b];
    const float a = __fmul_rn(__fmul_rn(__fmul_rn(tv, w[j]), 2.0f), 3.14159265358979323846f);
    float* row = out + (long)b * (2 * half + 1);
    if (j == 0) row[0] = tv;
    row[1 + j] = sinf(a);
    row[1 + half + j] = cosf(a);
}
// its weight gradient: dw[j] (+)= sum_b 2 pi t_b (g_sin[b][j] cos a - g_cos[b][j] sin a); one thread per frequency, samples in order
__global__ void learned_sinusoidal_bwd_kernel(const long* __restrict__ t, const float* __restrict__ w, const float* __restrict__ dout,
                                              float* __restrict__ dw, int B, int half, int accumulate) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= half) return;
    double sum = 0.0;
    for (int b = 0; b < B; ++b) {
        const float tv = (float)t[b];
        const float a = __fmul_rn(__fmul_rn(__fmul_rn(tv, w[j]), 2.0f), 3.14159265358979323846f);
        const float* g = dout + (long)b * (2 * half + 1);
        sum += (double)(g[1 + j] * cosf(a) - g[1 + half + j] * sinf(a)) * (double)tv * 6.283185307179586;
    }
    dw[j] = accumulate ? dw[j] + (float)sum : (float)sum;
}

// ----------------------------------------------------------------------------- linear attention core (:320-332)
// qkv: [B][n][3*H*32] (q | k | v, head-major inside each third); out: [B][n][H*32].  One workgroup per (b, head):
//   q <- softmax_d(q) * scale,  k <- softmax_n(k),  v <- v/n,  ctx[d][e] = sum_n k[d][n] v[e][n],
//   out[e][n] = sum_d ctx[d][e] q[d][n]
constexpr int DH = 32;
__global__ void __launch_bounds__(kBlock) linear_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 int n, int heads, float scale) {
    __shared__ float kmax[DH], ksum[DH], ctx[DH][DH + 1];
    __shared__ float part[8][DH];
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int HD = heads * DH;
    int ld = 3 * HD;
    const int qld = 3 * HD;
    const float* base = qkv + (long)b * n * ld;
    const float* qp = base + hd * DH;
    const float* kp = base + HD + hd * DH;
    const float* vp = base + 2 * HD + hd * DH;
    const int d = threadIdx.x % DH, grp = threadIdx.x / DH;          // 8 groups stride over n
    // n <= 256: stage this head's k and v rows (128 B each, coalesced) in LDS once; every pass below reads LDS
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    const bool staged = n <= kBlock;
    if (staged) {
        float* ksm = lsm;
        float* vsm = lsm + n * (DH + 1);
        for (int e = threadIdx.x; e < n * DH; e += kBlock) {
            const int i = e / DH, dd = e % DH;
            ksm[i * (DH + 1) + dd] = kp[(long)i * ld + dd];
            vsm[i * (DH + 1) + dd] = vp[(long)i * ld + dd];
        }
        __syncthreads();
        kp = ksm; vp = vsm; ld = DH + 1;                              // same indexing, now in LDS
    }
    // 1. column max of k over n
    float m = -INFINITY;
    for (int i = grp; i < n; i += 8) m = fmaxf(m, kp[(long)i * ld + d]);
    part[grp][d] = m;
    __syncthreads();
    if (threadIdx.x < DH) {
        float mm = part[0][d];
        for (int g = 1; g < 8; ++g) mm = fmaxf(mm, part[g][d]);
        kmax[d] = mm;
    }
    __syncthreads();
    // 2. column sum of exp(k - max)
    float s = 0.f;
    for (int i = grp; i < n; i += 8) {
        const float ev = expf(kp[(long)i * ld + d] - kmax[d]);
        if (staged) lsm[i * (DH + 1) + d] = ev;                     // keep exp(k - max) in place of k: step 3 needs it for every e
        s += ev;
    }
    part[grp][d] = s;
    __syncthreads();
    if (threadIdx.x < DH) {
        float ss = 0.f;
        for (int g = 0; g < 8; ++g) ss += part[g][d];
        ksum[d] = ss;
    }
    __syncthreads();
    // 3. ctx[d][e], thread owns d and e in [4*grp, 4*grp+4)
    {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float inv_n = 1.0f / (float)n;
        const float km = kmax[d], ks = ksum[d];
        for (int i = 0; i < n; ++i) {
            const float p = (staged ? kp[(long)i * ld + d] : expf(kp[(long)i * ld + d] - km)) / ks;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(p, vp[(long)i * ld + 4 * grp + j] * inv_n, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ctx[d][4 * grp + j] = acc[j];
    }
    __syncthreads();
    // 4. per pixel: softmax over d of q, times scale, then out[e] = sum_d ctx[d][e] q[d]
    if (staged) {
        // k and v are done with: their LDS holds the q rows (coalesced load) and then the output rows (coalesced store); a thread
        // walking its own 128-byte row of q / out in global memory touched 32 different lines per wave instruction
        float* qs = lsm;
        float* os = lsm + n * (DH + 1);
        for (int e = threadIdx.x; e < n * DH; e += kBlock) qs[(e / DH) * (DH + 1) + e % DH] = qp[(long)(e / DH) * qld + e % DH];
        __syncthreads();
        const int i = threadIdx.x;
        if (i < n) {
            float q[DH];
            float mq = -INFINITY;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) { q[dd] = qs[i * (DH + 1) + dd]; mq = fmaxf(mq, q[dd]); }
            float sq = 0.f;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) { q[dd] = expf(q[dd] - mq); sq += q[dd]; }
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) q[dd] = __fmul_rn(q[dd] / sq, scale);
            for (int e = 0; e < DH; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int dd = 0; dd < DH; ++dd) acc = fmaf(ctx[dd][e], q[dd], acc);
                os[i * (DH + 1) + e] = acc;
            }
        }
        __syncthreads();
        float* obase = out + (long)b * n * HD + hd * DH;
        for (int e = threadIdx.x; e < n * DH; e += kBlock) obase[(long)(e / DH) * HD + e % DH] = os[(e / DH) * (DH + 1) + e % DH];
        return;
    }
    for (int i = threadIdx.x; i < n; i += kBlock) {
        float q[DH];
        float mq = -INFINITY;
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) { q[dd] = qp[(long)i * qld + dd]; mq = fmaxf(mq, q[dd]); }
        float sq = 0.f;
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) { q[dd] = expf(q[dd] - mq); sq += q[dd]; }
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) q[dd] = __fmul_rn(q[dd] / sq, scale);
        float* orow = out + ((long)b * n + i) * HD + hd * DH;
        for (int e = 0; e < DH; ++e) {
            float acc = 0.f;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) acc = fmaf(ctx[dd][e], q[dd], acc);
            orow[e] = acc;
        }
    }
}

// The same core for n <= 256 positions with every global access batched: a thread issues all of its (up to 8) 16-byte loads of a
// staging pass before it touches LDS, where the loops above wait out one load latency per element (32 dependent round trips per
// pass at n = 256: ~46 us per launch at 16x16, a tenth of the reverse step of a 2x16x16 part); the softmax denominators are applied
// once to the 32x32 context instead of to every term, and the context is read as broadcast float4.
__global__ void __launch_bounds__(kBlock) linear_attention_staged_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                        int n, int heads, float scale) {
    __shared__ float kmax[DH], ksum[DH];
    __shared__ __attribute__((aligned(16))) float ctx4[DH][DH];
    __shared__ float part[8][DH];
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    constexpr int LP = DH + 1;                                       // LDS row pitch (conflict-free column walks)
    float* ksm = lsm;
    float* vsm = lsm + n * LP;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int HD = heads * DH, ld = 3 * HD;
    const float* base = qkv + (long)b * n * ld;
    const float* qp = base + hd * DH;
    const float* kp = base + HD + hd * DH;
    const float* vp = base + 2 * HD + hd * DH;
    const int tid = threadIdx.x;
    const int r = tid >> 3, c4 = (tid & 7) * 4;                       // staging role: rows r, r + 32, ..; floats c4 .. c4+3
    const int d = tid % DH, grp = tid / DH;
    {
        float4 kv[8], vv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = r + 32 * it;
            if (i < n) {
                kv[it] = *reinterpret_cast<const float4*>(kp + (long)i * ld + c4);
                vv[it] = *reinterpret_cast<const float4*>(vp + (long)i * ld + c4);
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = r + 32 * it;
            if (i < n) {
                float* kd = ksm + i * LP + c4;
                float* vd = vsm + i * LP + c4;
                kd[0] = kv[it].x; kd[1] = kv[it].y; kd[2] = kv[it].z; kd[3] = kv[it].w;
                vd[0] = vv[it].x; vd[1] = vv[it].y; vd[2] = vv[it].z; vd[3] = vv[it].w;
            }
        }
    }
    // q rows: requested now, used in step 4 (their latency hides under steps 1-3)
    float4 qv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int i = r + 32 * it;
        if (i < n) qv[it] = *reinterpret_cast<const float4*>(qp + (long)i * ld + c4);
    }
    __syncthreads();
    // 1. column max of k over n
    float m = -INFINITY;
    for (int i = grp; i < n; i += 8) m = fmaxf(m, ksm[i * LP + d]);
    part[grp][d] = m;
    __syncthreads();
    if (tid < DH) {
        float mm = part[0][d];
        for (int g = 1; g < 8; ++g) mm = fmaxf(mm, part[g][d]);
        kmax[d] = mm;
    }
    __syncthreads();
    // 2. exp(k - max) in place, column sums
    float s = 0.f;
    {
        const float km = kmax[d];
        for (int i = grp; i < n; i += 8) {
            const float ev = expf(ksm[i * LP + d] - km);
            ksm[i * LP + d] = ev;
            s += ev;
        }
    }
    __syncthreads();                                                  // part[] was read by the threads < DH above
    part[grp][d] = s;
    __syncthreads();
    if (tid < DH) {
        float ss = 0.f;
        for (int g = 0; g < 8; ++g) ss += part[g][d];
        ksum[d] = ss;
    }
    __syncthreads();
    // 3. ctx[d][e] = (sum_i exp(k - max)[i][d] v[i][e]) / (ksum[d] n); thread owns d and e in [4 grp, 4 grp + 4)
    {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i) {
            const float p = ksm[i * LP + d];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(p, vsm[i * LP + 4 * grp + j], acc[j]);
        }
        const float f = 1.0f / (ksum[d] * (float)n);
#pragma unroll
        for (int j = 0; j < 4; ++j) ctx4[d][4 * grp + j] = acc[j] * f;
    }
    __syncthreads();                                                  // k, v are done with: their LDS takes the q rows, then the output rows
    float* qs = ksm;
    float* os = vsm;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int i = r + 32 * it;
        if (i < n) {
            float* qd = qs + i * LP + c4;
            qd[0] = qv[it].x; qd[1] = qv[it].y; qd[2] = qv[it].z; qd[3] = qv[it].w;
        }
    }
    __syncthreads();
    // 4. per position: softmax over d of q, times scale, then out[e] = sum_d ctx[d][e] q[d]
    if (tid < n) {
        const int i = tid;
        float q[DH];
        float mq = -INFINITY;
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) { q[dd] = qs[i * LP + dd]; mq = fmaxf(mq, q[dd]); }
        float sq = 0.f;
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) { q[dd] = expf(q[dd] - mq); sq += q[dd]; }
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) q[dd] = __fmul_rn(q[dd] / sq, scale);
        float acc[DH];
#pragma unroll
        for (int e = 0; e < DH; ++e) acc[e] = 0.f;
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) {
#pragma unroll
            for (int e4 = 0; e4 < DH / 4; ++e4) {
                const float4 c = *reinterpret_cast<const float4*>(&ctx4[dd][4 * e4]);      // same address in every lane: broadcast
                acc[4 * e4 + 0] = fmaf(c.x, q[dd], acc[4 * e4 + 0]);
                acc[4 * e4 + 1] = fmaf(c.y, q[dd], acc[4 * e4 + 1]);
                acc[4 * e4 + 2] = fmaf(c.z, q[dd], acc[4 * e4 + 2]);
                acc[4 * e4 + 3] = fmaf(c.w, q[dd], acc[4 * e4 + 3]);
            }
        }
#pragma unroll
        for (int e = 0; e < DH; ++e) os[i * LP + e] = acc[e];
    }
    __syncthreads();
    float* obase = out + (long)b * n * HD + hd * DH;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int i = r + 32 * it;
        if (i < n) {
            const float* od = os + i * LP + c4;
            *reinterpret_cast<float4*>(obase + (long)i * HD + c4) = make_float4(od[0], od[1], od[2], od[3]);
        }
    }
}

// ---- n > 256 positions (the 64x64 / 32x32 latent parts of the larger models): the positions are cut into chunks of 256 and
// spread over blockIdx.y.  Pass 1 (one workgroup per (b, head, chunk), the chunk's k and v staged in LDS as above) produces
// the chunk's column max m_c[d], sum_c[d] = sum_i exp(k - m_c) and the unnormalised context sum_i exp(k - m_c) v[e]; pass 2
// recombines the chunks in chunk order (rescaling by exp(m_c - m): the usual streaming-softmax identity, deterministic) and
// applies the context to its own 256 positions.  With one workgroup per (b, head) walking all n positions serially the 6x64x64
// part of a 128x128 model spent 45 % of its reverse step here.
constexpr int LA_CHUNK = 256;
constexpr int LA_REC = 2 * DH + DH * DH;                        // floats per (b, head, chunk) record: m | s | ctx

__global__ void __launch_bounds__(kBlock) linear_attention_partial_kernel(const float* __restrict__ qkv, float* __restrict__ ws, int n,
                                                                         int heads) {
    __shared__ float kmax[DH];
    __shared__ float part[8][DH];
    __shared__ float ksm[LA_CHUNK][DH + 1], vsm[LA_CHUNK][DH + 1];
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads, c = blockIdx.y;
    const int HD = heads * DH, ld = 3 * HD;
    const int i0 = c * LA_CHUNK;
    const int nc = (n - i0) < LA_CHUNK ? (n - i0) : LA_CHUNK;
    const float* base = qkv + ((long)b * n + i0) * ld;
    const float* kp = base + HD + hd * DH;
    const float* vp = base + 2 * HD + hd * DH;
    const int d = threadIdx.x % DH, grp = threadIdx.x / DH;
    for (int e = threadIdx.x; e < nc * DH; e += kBlock) {
        const int i = e / DH, dd = e % DH;
        ksm[i][dd] = kp[(long)i * ld + dd];
        vsm[i][dd] = vp[(long)i * ld + dd];
    }
    __syncthreads();
    float m = -INFINITY;
    for (int i = grp; i < nc; i += 8) m = fmaxf(m, ksm[i][d]);
    part[grp][d] = m;
    __syncthreads();
    if (threadIdx.x < DH) {
        float mm = part[0][d];
        for (int g = 1; g < 8; ++g) mm = fmaxf(mm, part[g][d]);
        kmax[d] = mm;
    }
    __syncthreads();
    const float km = kmax[d];
    float s = 0.f;
    for (int i = grp; i < nc; i += 8) s += expf(ksm[i][d] - km);
    __syncthreads();
    part[grp][d] = s;
    __syncthreads();
    float* rec = ws + ((size_t)blockIdx.x * gridDim.y + c) * LA_REC;
    if (threadIdx.x < DH) {
        float ss = 0.f;
        for (int g = 0; g < 8; ++g) ss += part[g][d];
        rec[d] = km;
        rec[DH + d] = ss;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                            // thread owns d and e in [4 grp, 4 grp + 4)
    for (int i = 0; i < nc; ++i) {
        const float pexp = expf(ksm[i][d] - km);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(pexp, vsm[i][4 * grp + j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) rec[2 * DH + d * DH + 4 * grp + j] = acc[j];
}

__global__ void __launch_bounds__(kBlock) linear_attention_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ ws,
                                                                       float* __restrict__ out, int n, int heads, float scale) {
    __shared__ float gm[DH], gs[DH], ctx[DH][DH + 1];
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads, c = blockIdx.y, nch = gridDim.y;
    const int HD = heads * DH, qld = 3 * HD;
    const float* recs = ws + (size_t)blockIdx.x * nch * LA_REC;
    const int d = threadIdx.x % DH, grp = threadIdx.x / DH;
    if (threadIdx.x < DH) {
        float m = -INFINITY;
        for (int q = 0; q < nch; ++q) m = fmaxf(m, recs[q * LA_REC + d]);
        float ssum = 0.f;
        for (int q = 0; q < nch; ++q) ssum += recs[q * LA_REC + DH + d] * expf(recs[q * LA_REC + d] - m);
        gm[d] = m;
        gs[d] = ssum;
    }
    __syncthreads();
    {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < nch; ++q) {
            const float r = expf(recs[q * LA_REC + d] - gm[d]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(recs[q * LA_REC + 2 * DH + d * DH + 4 * grp + j], r, acc[j]);
        }
        const float norm = 1.0f / (gs[d] * (float)n);               // softmax denominator and v / n
#pragma unroll
        for (int j = 0; j < 4; ++j) ctx[d][4 * grp + j] = acc[j] * norm;
    }
    __syncthreads();
    // this chunk's q rows in, output rows out: both through LDS so the global accesses are coalesced
    __shared__ float qs[LA_CHUNK][DH + 1], os[LA_CHUNK][DH + 1];
    const int i0 = c * LA_CHUNK;
    const int nc = (n - i0) < LA_CHUNK ? (n - i0) : LA_CHUNK;
    const float* qbase = qkv + ((long)b * n + i0) * qld + hd * DH;
    for (int e = threadIdx.x; e < nc * DH; e += kBlock) qs[e / DH][e % DH] = qbase[(long)(e / DH) * qld + e % DH];
    __syncthreads();
    if (threadIdx.x < nc) {
        const int i = threadIdx.x;
        float q[DH];
        float mq = -INFINITY;
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) { q[dd] = qs[i][dd]; mq = fmaxf(mq, q[dd]); }
        float sq = 0.f;
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) { q[dd] = expf(q[dd] - mq); sq += q[dd]; }
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) q[dd] = __fmul_rn(q[dd] / sq, scale);
        for (int e = 0; e < DH; ++e) {
            float acc = 0.f;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) acc = fmaf(ctx[dd][e], q[dd], acc);
            os[i][e] = acc;
        }
    }
    __syncthreads();
    float* obase = out + ((long)b * n + i0) * HD + hd * DH;
    for (int e = threadIdx.x; e < nc * DH; e += kBlock) obase[(long)(e / DH) * HD + e % DH] = os[e / DH][e % DH];
}

// ----------------------------------------------------------------------------- full attention core (:351-357), n <= 256
// one workgroup per (b, head); K and V staged in LDS (16-byte loads, all of a thread's requests in flight before it touches LDS).  A query
// is shared by NP = 256 / npow2(n) threads, thread (i, p) taking the keys j = p, p + NP, ...: with one thread per query the 8x8 mid block
// of a B = 64 step (n = 64) ran 64 of the 256 threads through 2 x 64 keys x 32 dependent LDS reads - 41.6 us per launch, the longest kernel
// of a reverse step.  The row maximum is exact (a max over the parts' maxima); the parts' exp sums and value sums meet in LDS and are
// added in part order by the thread that writes 32 / NP... of the query's outputs (NP = 1, n > 128: the one-thread arithmetic).
__global__ void __launch_bounds__(kBlock) attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int n,
                                                          int heads, float scale, int np_log2) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // K[n][32], V[n][32] (rows read as broadcast float4: the lanes of a wave share the key), partials [NP][nq][34] (max | den | o[32])
    constexpr int LP = DH, PP = DH + 2;
    float* Ks = sm;
    float* Vs = sm + n * LP;
    float* Ps = sm + 2 * n * LP;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int HD = heads * DH, ld = 3 * HD;
    const float* base = qkv + (long)b * n * ld;
    const int tid = threadIdx.x;
    {
        const int r = tid >> 3, c4 = (tid & 7) * 4;                  // staging role: rows r, r + 32, ..; floats c4 .. c4 + 3
        float4 kv[8], vv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = r + 32 * it;
            if (i < n) {
                kv[it] = *reinterpret_cast<const float4*>(base + (long)i * ld + HD + hd * DH + c4);
                vv[it] = *reinterpret_cast<const float4*>(base + (long)i * ld + 2 * HD + hd * DH + c4);
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = r + 32 * it;
            if (i < n) {
                *reinterpret_cast<float4*>(Ks + i * LP + c4) = kv[it];
                *reinterpret_cast<float4*>(Vs + i * LP + c4) = vv[it];
            }
        }
    }
    const int NP = 1 << np_log2, nq = kBlock >> np_log2;             // threads per query; query slots of this workgroup
    const int si = tid & (nq - 1), p = tid >> (8 - np_log2);
    const int i = blockIdx.y * nq + si;                              // (gridDim.y workgroups share a (sample, head): nq queries each, all keys)
    const bool active = i < n;
    const int iq = active ? i : 0;
    float q[DH];
#pragma unroll
    for (int dd = 0; dd < DH; dd += 4) {
        const float4 v = *reinterpret_cast<const float4*>(base + (long)iq * ld + hd * DH + dd);
        q[dd] = __fmul_rn(v.x, scale); q[dd + 1] = __fmul_rn(v.y, scale); q[dd + 2] = __fmul_rn(v.z, scale); q[dd + 3] = __fmul_rn(v.w, scale);
    }
    __syncthreads();
    float mx = -INFINITY;
    auto dot = [&](int j) __attribute__((always_inline)) {
        const float4* kr = reinterpret_cast<const float4*>(Ks + j * LP);
        float sdot = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < DH / 4; ++d4) {
            const float4 kv = kr[d4];
            sdot = fmaf(q[4 * d4], kv.x, sdot); sdot = fmaf(q[4 * d4 + 1], kv.y, sdot);
            sdot = fmaf(q[4 * d4 + 2], kv.z, sdot); sdot = fmaf(q[4 * d4 + 3], kv.w, sdot);
        }
        return sdot;
    };
    for (int j = p; j < n; j += NP) mx = fmaxf(mx, dot(j));
    if (NP > 1) {                                                    // (uniform) the row maximum over the parts
        Ps[(p * nq + si) * PP] = mx;
        __syncthreads();
        for (int pp = 0; pp < NP; ++pp) mx = fmaxf(mx, Ps[(pp * nq + si) * PP]);
        __syncthreads();
    }
    float o[DH];
#pragma unroll
    for (int dd = 0; dd < DH; ++dd) o[dd] = 0.f;
    float den = 0.f;
    for (int j = p; j < n; j += NP) {
        const float pr = expf(dot(j) - mx);
        den += pr;
        const float4* vr = reinterpret_cast<const float4*>(Vs + j * LP);
#pragma unroll
        for (int d4 = 0; d4 < DH / 4; ++d4) {
            const float4 vv = vr[d4];
            o[4 * d4] = fmaf(pr, vv.x, o[4 * d4]); o[4 * d4 + 1] = fmaf(pr, vv.y, o[4 * d4 + 1]);
            o[4 * d4 + 2] = fmaf(pr, vv.z, o[4 * d4 + 2]); o[4 * d4 + 3] = fmaf(pr, vv.w, o[4 * d4 + 3]);
        }
    }
    if (NP == 1) {
        if (!active) return;
        float* orow = out + ((long)b * n + i) * HD + hd * DH;
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) orow[dd] = o[dd] / den;
        return;
    }
    float* mine = Ps + (p * nq + si) * PP;
    mine[1] = den;
#pragma unroll
    for (int dd = 0; dd < DH; ++dd) mine[2 + dd] = o[dd];
    __syncthreads();
    // thread (i, p) finishes the outputs dd = p, p + NP, ... of query i: parts added in order
    if (!active) return;
    float dsum = 0.f;
    for (int pp = 0; pp < NP; ++pp) dsum += Ps[(pp * nq + si) * PP + 1];
    float* orow = out + ((long)b * n + i) * HD + hd * DH;
    for (int dd = p; dd < DH; dd += NP) {
        float v = 0.f;
        for (int pp = 0; pp < NP; ++pp) v += Ps[(pp * nq + si) * PP + 2 + dd];
        orow[dd] = v / dsum;
    }
}

// n > 256: queries in chunks of 256 over blockIdx.y, keys / values streamed through LDS in chunks of 256, the same two passes
// (row max, then exp(s - max) sums) so the arithmetic is that of the one-chunk kernel.
__global__ void __launch_bounds__(kBlock) attention_long_kernel(const float* __restrict__ qkv, float* __restrict__ out, int n,
                                                               int heads, float scale) {
    __shared__ float Ks[kBlock][DH + 1], Vs[kBlock][DH + 1];
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int HD = heads * DH, ld = 3 * HD;
    const float* base = qkv + (long)b * n * ld;
    const int i = blockIdx.y * kBlock + threadIdx.x;
    const bool active = i < n;
    float q[DH];
#pragma unroll
    for (int dd = 0; dd < DH; ++dd) q[dd] = active ? __fmul_rn(base[(long)i * ld + hd * DH + dd], scale) : 0.f;
    float mx = -INFINITY;
    for (int j0 = 0; j0 < n; j0 += kBlock) {
        const int nc = (n - j0) < kBlock ? (n - j0) : kBlock;
        __syncthreads();
        for (int e = threadIdx.x; e < nc * DH; e += kBlock) Ks[e / DH][e % DH] = base[(long)(j0 + e / DH) * ld + HD + hd * DH + e % DH];
        __syncthreads();
        for (int j = 0; j < nc; ++j) {
            float sdot = 0.f;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) sdot = fmaf(q[dd], Ks[j][dd], sdot);
            mx = fmaxf(mx, sdot);
        }
    }
    float o[DH];
#pragma unroll
    for (int dd = 0; dd < DH; ++dd) o[dd] = 0.f;
    float den = 0.f;
    for (int j0 = 0; j0 < n; j0 += kBlock) {
        const int nc = (n - j0) < kBlock ? (n - j0) : kBlock;
        __syncthreads();
        for (int e = threadIdx.x; e < nc * DH; e += kBlock) {
            Ks[e / DH][e % DH] = base[(long)(j0 + e / DH) * ld + HD + hd * DH + e % DH];
            Vs[e / DH][e % DH] = base[(long)(j0 + e / DH) * ld + 2 * HD + hd * DH + e % DH];
        }
        __syncthreads();
        for (int j = 0; j < nc; ++j) {
            float sdot = 0.f;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) sdot = fmaf(q[dd], Ks[j][dd], sdot);
            const float p = expf(sdot - mx);
            den += p;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) o[dd] = fmaf(p, Vs[j][dd], o[dd]);
        }
    }
    if (!active) return;
    float* orow = out + ((long)b * n + i) * HD + hd * DH;
#pragma unroll
    for (int dd = 0; dd < DH; ++dd) orow[dd] = o[dd] / den;
}

__global__ void upsample2x_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int C, int H, int W) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int c = (int)(i % C);
        long r = i / C;
        const int ox = (int)(r % (2 * W));
        r /= 2 * W;
        const int oy = (int)(r % (2 * H));
        const long b = r / (2 * H);
        y[i] = x[((b * H + (oy >> 1)) * W + (ox >> 1)) * C + c];
    }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) out[i] = __fadd_rn(a[i], b[i]);
}

}  // namespace

extern "C" int nfdp_ws_standardize_f32(const float* w, float* out, int Cout, int fan_in, float eps, nfdp_stream_t stream) {
    NFDP_REQUIRE(w && out, NFDP_EINVAL, "ws_standardize: null pointer");
    NFDP_REQUIRE(Cout > 0 && fan_in > 0, NFDP_ESHAPE, "ws_standardize: bad shape");
    hipLaunchKernelGGL(ws_standardize_kernel, dim3(Cout), dim3(kBlock), 0, as_stream(stream), w, out, fan_in, eps);
    NFDP_CHECK_LAUNCH("ws_standardize");
    return NFDP_OK;
}

extern "C" int nfdp_groupnorm_silu_res_f32(const float* x, float* y, int B, int HW, int C, int groups, const float* gamma,
                                           const float* beta, const float* scale_shift, int ss_ld, float eps,
                                           const float* residual, nfdp_stream_t stream);

// ----------------------------------------------------------------------------- GroupNorm + SiLU with statistics that came with the tensor
// Block 2 of a ResnetBlock where a conv tile is a part of an image: the conv's epilogue left the fp64 (sum, sum of squares) of every
// (pixel tile, group) (nfdp_conv2d_stats_nhwc_f32 / nfdp_conv2d_gnin_nhwc_f32), so the norm needs no reduction of its own and no
// (sample, group) workgroups: a workgroup takes 4096 / C consecutive pixels with ALL their channels (whole 4 C-byte rows: the grouped
// kernel reads 16 bytes of every row per workgroup when a group is 4 channels wide), adds the image's tile sums up (LP lanes per group,
// a fixed xor tree, as conv3x3_ldsx_kernel<.., GNIN>), and streams.  Same expressions in the same order as groupnorm_silu_kernel.
__global__ void __launch_bounds__(kBlock) groupnorm_silu_sums_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    const double* __restrict__ sums, int parts, int HW, int C, int groups,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    const float* __restrict__ ss, int ss_ld, float eps,
                                                                    const float* __restrict__ res) {
    __shared__ float2 mr[64];                           // (mean, rstd) per group
    const int q4 = C >> 2, pxw = 4096 / C;              // float4 per pixel row, pixels per workgroup (1024 units, 4 per thread)
    const long p0 = (long)blockIdx.x * pxw;
    const int b = (int)(p0 / HW);
    const int tid = threadIdx.x;
    float4 v[4], rv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                       // the rows first: they fly while the statistics are formed
        const int u = tid + kBlock * k;
        v[k] = *reinterpret_cast<const float4*>(x + (p0 + u / q4) * C + (u % q4) * 4);
    }
    if (res) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = tid + kBlock * k;
            rv[k] = *reinterpret_cast<const float4*>(res + (p0 + u / q4) * C + (u % q4) * 4);
        }
    }
    const int cpg = C / groups;
    const int LP = parts < 16 ? parts : 16, lsh = __builtin_ctz((unsigned)LP);
    for (int g0 = 0; g0 < groups; g0 += kBlock >> lsh) {                 // (uniform)
        const int g = g0 + (tid >> lsh), j = tid & (LP - 1), gc = g < groups ? g : groups - 1;
        const double* sp = sums + ((size_t)b * parts * groups + gc) * 2;
        double g1 = 0.0, g2 = 0.0;
        for (int i = j; i < parts; i += LP) {
            const double2 t = *reinterpret_cast<const double2*>(sp + (size_t)i * groups * 2);
            g1 += t.x;
            g2 += t.y;
        }
        for (int o = 1; o < LP; o <<= 1) {
            g1 += __shfl_xor(g1, o);
            g2 += __shfl_xor(g2, o);
        }
        if (g < groups && j == 0) {
            const double rn = 1.0 / ((double)HW * cpg);
            const double mean = g1 * rn;
            double var = g2 * rn - mean * mean;
            var = var > 0.0 ? var : 0.0;
            mr[g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
        }
    }
    __syncthreads();
    const float* sr = ss ? ss + (long)b * ss_ld : nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int u = tid + kBlock * k;
        const int c0 = (u % q4) * 4;
        const long off = (p0 + u / q4) * C + c0;
        const float2 st = mr[c0 / cpg];
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c0), be = *reinterpret_cast<const float4*>(beta + c0);
        float o[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        const float gq[4] = {ga.x, ga.y, ga.z, ga.w}, bq[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(o[r], st.x), st.y), gq[r]), bq[r]);
            if (sr) t = __fadd_rn(__fmul_rn(t, __fadd_rn(sr[c0 + r], 1.0f)), sr[C + c0 + r]);
            o[r] = t / (1.0f + expf(-t));
        }
        if (res) {
            o[0] = __fadd_rn(o[0], rv[k].x); o[1] = __fadd_rn(o[1], rv[k].y); o[2] = __fadd_rn(o[2], rv[k].z); o[3] = __fadd_rn(o[3], rv[k].w);
        }
        *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int nfdp_groupnorm_silu_sums_supported(int B, int HW, int C, int groups, int tile) {
    if (B <= 0 || HW <= 0 || C < 4 || groups <= 0 || groups > 64 || tile <= 0 || C % groups != 0 || HW % tile != 0) return 0;
    const int cpg = C / groups, parts = HW / tile;
    if (4096 % C != 0 || cpg % 4 != 0 || (parts & (parts - 1)) != 0) return 0;
    const int pxw = 4096 / C;
    return HW % pxw == 0 && (((long)HW * cpg) & ((long)HW * cpg - 1)) == 0;          // whole workgroups inside one image; 1 / n exact
}

// y = SiLU(GroupNorm(x) * (scale + 1) + shift) + residual as nfdp_groupnorm_silu_res_f32, statistics from `sums`
// ([B * HW / tile][groups][2] fp64 (sum, sum of squares): nfdp_conv2d_stats_nhwc_f32 / nfdp_conv2d_gnin_nhwc_f32's sums_out).
extern "C" int nfdp_groupnorm_silu_sums_f32(const float* x, float* y, const double* sums, int tile, int B, int HW, int C, int groups,
                                            const float* gamma, const float* beta, const float* scale_shift, int ss_ld, float eps,
                                            const float* residual, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y && sums && gamma && beta, NFDP_EINVAL, "groupnorm_silu_sums: null pointer");
    NFDP_REQUIRE(nfdp_groupnorm_silu_sums_supported(B, HW, C, groups, tile), NFDP_EUNSUPPORTED,
                 "groupnorm_silu_sums: unsupported shape B=%d HW=%d C=%d groups=%d tile=%d", B, HW, C, groups, tile);
    NFDP_REQUIRE(!scale_shift || ss_ld >= 2 * C, NFDP_ESHAPE, "groupnorm_silu_sums: ss_ld < 2C");
    NFDP_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)sums | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)residual) % 16) == 0, NFDP_ESHAPE,
                 "groupnorm_silu_sums: tensors and vectors must be 16-byte aligned");
    const long nwg = (long)B * HW / (4096 / C);
    hipLaunchKernelGGL(groupnorm_silu_sums_kernel, dim3((unsigned)nwg), dim3(kBlock), 0, as_stream(stream), x, y, sums, HW / tile, HW, C, groups,
                       gamma, beta, scale_shift, ss_ld, eps, residual);
    NFDP_CHECK_LAUNCH("groupnorm_silu_sums");
    return NFDP_OK;
}

extern "C" int nfdp_groupnorm_silu_f32(const float* x, float* y, int B, int HW, int C, int groups, const float* gamma,
                                       const float* beta, const float* scale_shift, int ss_ld, float eps,
                                       nfdp_stream_t stream) {
    return nfdp_groupnorm_silu_res_f32(x, y, B, HW, C, groups, gamma, beta, scale_shift, ss_ld, eps, nullptr, stream);
}

extern "C" int nfdp_groupnorm_silu_res_f32(const float* x, float* y, int B, int HW, int C, int groups, const float* gamma,
                                           const float* beta, const float* scale_shift, int ss_ld, float eps,
                                           const float* residual, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y && gamma && beta, NFDP_EINVAL, "groupnorm_silu: null pointer");
    NFDP_REQUIRE(B >= 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, NFDP_ESHAPE, "groupnorm_silu: bad shape");
    if (B == 0) return NFDP_OK;
    NFDP_REQUIRE(!scale_shift || ss_ld >= 2 * C, NFDP_ESHAPE, "groupnorm_silu: ss_ld < 2C");
    const long n = (long)HW * (C / groups);
    constexpr long kGnLdsMax = 36 * 1024;                        // floats: 144 KiB of the 160 KiB LDS
    const int lds_cached = n > 256 * 16 && n <= kGnLdsMax && (C / groups) % 4 == 0 && C % 4 == 0 && ((uintptr_t)x & 15) == 0 &&
                           ((uintptr_t)y & 15) == 0 && ((uintptr_t)residual & 15) == 0;
    static NfdpPerDevice gn_attr;
    if (lds_cached && !gn_attr) {
        nfdp_func_attr(reinterpret_cast<const void*>(&groupnorm_silu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(kGnLdsMax * sizeof(float)));
        gn_attr = true;
    }
    hipLaunchKernelGGL(groupnorm_silu_kernel, dim3(B * groups), dim3(kBlock), lds_cached ? (size_t)n * sizeof(float) : 0,
                       as_stream(stream), x, y, HW, C, groups, gamma, beta, scale_shift, ss_ld, eps, lds_cached, residual);
    NFDP_CHECK_LAUNCH("groupnorm_silu");
    return NFDP_OK;
}

extern "C" int nfdp_chan_layernorm_f32(const float* x, float* y, long npix, int C, const float* gain, float eps,
                                       const float* residual, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y && gain, NFDP_EINVAL, "chan_layernorm: null pointer");
    NFDP_REQUIRE(npix >= 0 && C > 0, NFDP_ESHAPE, "chan_layernorm: bad shape");
    if (npix == 0) return NFDP_OK;
    // (4 waves per workgroup; a wave takes 64 / LPP pixels per round in the 16-byte form - see the kernel - else one)
    const int ppw = (C % 4 == 0 && C <= 256) ? 64 / (C <= 64 ? 16 : (C <= 128 ? 32 : 64)) : 1;
    const long g = cdiv(npix, 4L * ppw);
    hipLaunchKernelGGL(chan_layernorm_kernel, dim3((int)(g < 2048 ? g : 2048)), dim3(kBlock), 0, as_stream(stream), x, y, npix, C,
                       gain, eps, residual);
    NFDP_CHECK_LAUNCH("chan_layernorm");
    return NFDP_OK;
}

extern "C" int nfdp_linear_f32(const float* in, const float* w, const float* bias, float* out, int B, int I, int O, int pre_act,
                               int post_act, nfdp_stream_t stream) {
    NFDP_REQUIRE(in && w && out, NFDP_EINVAL, "linear: null pointer");
    NFDP_REQUIRE(B >= 0 && I > 0 && O > 0, NFDP_ESHAPE, "linear: bad shape");
    NFDP_REQUIRE(I <= 480, NFDP_EUNSUPPORTED, "linear: I=%d (<= 480: LDS-staged tiles)", I);
    NFDP_REQUIRE(pre_act >= 0 && pre_act <= 1 && post_act >= 0 && post_act <= 2, NFDP_EINVAL, "linear: bad activation enum");
    if (B == 0) return NFDP_OK;
    const size_t lds = (size_t)(LIN_TB + LIN_TO) * (I + 1) * sizeof(float);
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&linear_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (LIN_TB + LIN_TO) * 481 * (int)sizeof(float));      // 153,920 B <= 160 KiB
        attr_set = true;
    }
    hipLaunchKernelGGL(linear_kernel, dim3((int)cdiv(O, LIN_TO), (int)cdiv(B, LIN_TB)), dim3(kBlock), lds, as_stream(stream), in, w,
                       bias, out, B, I, O, O, pre_act, post_act);
    NFDP_CHECK_LAUNCH("linear");
    return NFDP_OK;
}

extern "C" int nfdp_learned_sinusoidal_emb_f32(const long* t, const float* w, float* out, int B, int half, nfdp_stream_t stream) {
    NFDP_REQUIRE(t && w && out, NFDP_EINVAL, "learned_sinusoidal_emb: null pointer");
    NFDP_REQUIRE(B >= 0 && half >= 1, NFDP_ESHAPE, "learned_sinusoidal_emb: bad shape");
    if (B == 0) return NFDP_OK;
    hipLaunchKernelGGL(learned_sinusoidal_kernel, dim3((int)cdiv((long)B * half, kBlock)), dim3(kBlock), 0, as_stream(stream), t, w, out, B,
                       half);
    NFDP_CHECK_LAUNCH("learned_sinusoidal_emb");
    return NFDP_OK;
}

extern "C" int nfdp_learned_sinusoidal_bwd_f32(const long* t, const float* w, const float* dout, float* dw, int B, int half,
                                               int accumulate, nfdp_stream_t stream) {
    NFDP_REQUIRE(t && w && dout && dw, NFDP_EINVAL, "learned_sinusoidal_bwd: null pointer");
    NFDP_REQUIRE(B >= 0 && half >= 1, NFDP_ESHAPE, "learned_sinusoidal_bwd: bad shape");
    hipLaunchKernelGGL(learned_sinusoidal_bwd_kernel, dim3((int)cdiv((long)half, kBlock)), dim3(kBlock), 0, as_stream(stream), t, w, dout, dw,
                       B, half, accumulate);
    NFDP_CHECK_LAUNCH("learned_sinusoidal_bwd");
    return NFDP_OK;
}

extern "C" int nfdp_sinusoidal_emb_f32(const long* t, float* out, int B, int dim, nfdp_stream_t stream) {
    NFDP_REQUIRE(t && out, NFDP_EINVAL, "sinusoidal_emb: null pointer");
    NFDP_REQUIRE(B >= 0 && dim >= 4 && dim % 2 == 0, NFDP_ESHAPE, "sinusoidal_emb: bad shape");
    if (B == 0) return NFDP_OK;
    hipLaunchKernelGGL(sinusoidal_kernel, dim3((int)cdiv((long)B * (dim / 2), kBlock)), dim3(kBlock), 0, as_stream(stream), t, out,
                       B, dim);
    NFDP_CHECK_LAUNCH("sinusoidal_emb");
    return NFDP_OK;
}

extern "C" int nfdp_linear_attention_f32(const float* qkv, float* out, int B, int n, int heads, int dim_head, float scale,
                                         nfdp_stream_t stream) {
    NFDP_REQUIRE(qkv && out, NFDP_EINVAL, "linear_attention: null pointer");
    NFDP_REQUIRE(B >= 0 && n > 0 && heads > 0, NFDP_ESHAPE, "linear_attention: bad shape");
    NFDP_REQUIRE(dim_head == DH, NFDP_EUNSUPPORTED, "linear_attention: dim_head=%d (32 only)", dim_head);
    if (B == 0) return NFDP_OK;
    const size_t la_lds = n <= kBlock ? (size_t)2 * n * (DH + 1) * sizeof(float) : 0;
    static NfdpPerDevice la_attr;
    if (!la_attr) {
        nfdp_func_attr(reinterpret_cast<const void*>(&linear_attention_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBlock * (DH + 1) * (int)sizeof(float));
        la_attr = true;
    }
    if (n <= kBlock && (((uintptr_t)qkv | (uintptr_t)out) & 15) == 0) {
        static NfdpPerDevice las_attr;
        if (!las_attr) {
            nfdp_func_attr(reinterpret_cast<const void*>(&linear_attention_staged_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBlock * (DH + 1) * (int)sizeof(float));
            las_attr = true;
        }
        hipLaunchKernelGGL(linear_attention_staged_kernel, dim3(B * heads), dim3(kBlock), la_lds, as_stream(stream), qkv, out, n, heads,
                           scale);
    } else {
        hipLaunchKernelGGL(linear_attention_kernel, dim3(B * heads), dim3(kBlock), la_lds, as_stream(stream), qkv, out, n, heads, scale);
    }
    NFDP_CHECK_LAUNCH("linear_attention");
    return NFDP_OK;
}

extern "C" size_t nfdp_linear_attention_ws_bytes(int B, int n, int heads) {
    if (B <= 0 || n <= kBlock || heads <= 0) return 0;
    return (size_t)B * heads * cdiv(n, LA_CHUNK) * LA_REC * sizeof(float);
}

extern "C" int nfdp_linear_attention_chunked_f32(const float* qkv, float* out, int B, int n, int heads, int dim_head, float scale,
                                                 void* ws, nfdp_stream_t stream) {
    NFDP_REQUIRE(qkv && out && ws, NFDP_EINVAL, "linear_attention_chunked: null pointer");
    NFDP_REQUIRE(B >= 0 && n > 0 && heads > 0, NFDP_ESHAPE, "linear_attention_chunked: bad shape");
    NFDP_REQUIRE(dim_head == DH, NFDP_EUNSUPPORTED, "linear_attention_chunked: dim_head=%d (32 only)", dim_head);
    if (B == 0) return NFDP_OK;
    const int nch = (int)cdiv(n, LA_CHUNK);
    NFDP_REQUIRE(nch <= 65535, NFDP_ESHAPE, "linear_attention_chunked: n=%d too long", n);
    const dim3 grid(B * heads, nch);
    hipLaunchKernelGGL(linear_attention_partial_kernel, grid, dim3(kBlock), 0, as_stream(stream), qkv, reinterpret_cast<float*>(ws), n, heads);
    hipLaunchKernelGGL(linear_attention_apply_kernel, grid, dim3(kBlock), 0, as_stream(stream), qkv, reinterpret_cast<const float*>(ws), out,
                       n, heads, scale);
    NFDP_CHECK_LAUNCH("linear_attention_chunked");
    return NFDP_OK;
}

extern "C" int nfdp_attention_f32(const float* qkv, float* out, int B, int n, int heads, int dim_head, float scale,
                                  nfdp_stream_t stream) {
    NFDP_REQUIRE(qkv && out, NFDP_EINVAL, "attention: null pointer");
    NFDP_REQUIRE(B >= 0 && n > 0 && heads > 0, NFDP_ESHAPE, "attention: bad shape");
    NFDP_REQUIRE(dim_head == DH, NFDP_EUNSUPPORTED, "attention: dim_head=%d (32 only)", dim_head);
    if (B == 0) return NFDP_OK;
    if (n > kBlock) {
        NFDP_REQUIRE(cdiv(n, kBlock) <= 65535, NFDP_ESHAPE, "attention: n=%d too long", n);
        hipLaunchKernelGGL(attention_long_kernel, dim3(B * heads, (int)cdiv(n, kBlock)), dim3(kBlock), 0, as_stream(stream), qkv, out, n,
                           heads, scale);
        NFDP_CHECK_LAUNCH("attention (long)");
        return NFDP_OK;
    }
    // threads per query: 256 / npow2(queries per workgroup), at most 32 (the finishing loop deals the 32 outputs of a query to its
    // threads).  More than 64 queries are dealt to several workgroups of 64 (each stages all keys): with one thread per query a
    // 256-query head walked 2 x 256 keys x 32 dependent LDS reads per thread - 159 us for the 16x16 mid block of a 6 x 64 x 64 latent part
    const int qpw = n > 64 ? 64 : n, qc = (int)cdiv(n, qpw);
    int np_log2 = 0;
    while ((qpw << (np_log2 + 1)) <= kBlock && np_log2 < 5) ++np_log2;
    const size_t lds = ((size_t)2 * n * DH + (np_log2 ? (size_t)kBlock * (DH + 2) : 0)) * sizeof(float);
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (2 * kBlock * (DH + 1) + kBlock * (DH + 2)) * (int)sizeof(float));
        attr_set = true;
    }
    hipLaunchKernelGGL(attention_kernel, dim3(B * heads, qc), dim3(kBlock), lds, as_stream(stream), qkv, out, n, heads, scale, np_log2);
    NFDP_CHECK_LAUNCH("attention");
    return NFDP_OK;
}

extern "C" int nfdp_upsample_nearest2x_f32(const float* x, float* y, int B, int C, int H, int W, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y, NFDP_EINVAL, "upsample: null pointer");
    NFDP_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, NFDP_ESHAPE, "upsample: bad shape");
    const long total = (long)B * C * H * W * 4;
    if (total == 0) return NFDP_OK;
    hipLaunchKernelGGL(upsample2x_kernel, dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream), x, y, total, C, H, W);
    NFDP_CHECK_LAUNCH("upsample");
    return NFDP_OK;
}

// ----------------------------------------------------------------------------- Unet.init_conv: 7x7 over 1-8 input channels (:383)
// K = 49 * Cin <= 196 with Cin not a multiple of 4: the implicit-GEMM kernel gathers it element by element (13.6 us for the 16x16 / B = 64
// latent part - 0.2 GFLOP).  A direct conv on the vector ALU: a workgroup takes 64 consecutive pixels x 64 output channels, a thread 4
// consecutive pixels of one image row (W % 4 == 0) x 4 outputs - lane = (output quad, pixel group), so that a pixel's 64 outputs leave
// the wave as one 256-byte row.  The pixels and their halo (the flat range [m0 - 3(W+1), m0 + 64 + 3(W+1)) holds every 7x7 neighbour
// whatever the geometry) and the 64 outputs' weights ([49][Cin][Cout] copy of the parameter, nfdp_conv_k7_pack_f32: a straight 16-byte
// copy) sit in LDS; per row of taps and channel the thread reads its 10-pixel window once (a column outside the image is zeroed there:
// validity belongs to the SOURCE column, whichever of the 4 pixels and 7 taps reads it) and one float4 of weights per tap feeds 16 FMAs
// (fp32 chains, tap-major, channel-minor).  Tried on the way: workgroups of 4 or 16 outputs (16 CUs write 16-byte pieces of every output
// row: 10.5 us whatever the inner loop), weights through the scalar cache (s_load_dwordx16 per (tap, channel): ~700 cycles each, 28 us).
__global__ void conv_k7_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin) {
    const int n = Cout * Cin * 49;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int o = i % Cout, r = i / Cout, ci = r % Cin, tap = r / Cin;
        out[i] = w[((size_t)o * Cin + ci) * 49 + tap];
    }
}

template <int CIN>
__global__ void __launch_bounds__(256) conv_k7_narrow_kernel(const float* __restrict__ x, int x_ld, const float* __restrict__ wk,
                                                            const float* __restrict__ bias, float* __restrict__ y, int y_ld, int npix,
                                                            int H, int W, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float k7sm[];
    const int HALO = 3 * (W + 1), NP = 64 + 2 * HALO;
    float* ws = k7sm;                                   // [49 * CIN][64]
    float* xs = k7sm + 49 * CIN * 64;                   // [NP][CIN]
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64, tid = threadIdx.x;
    {   // weights: 49 * CIN rows of 16 float4, all requested before the first is stored
        constexpr int NV = 49 * CIN * 16, NE = (NV + 255) / 256;
        float4 v[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = e * 256 + tid, ic = i < NV ? i : NV - 1;
            const int row = ic >> 4, q4 = (ic & 15) * 4;
            const int n = n0 + q4 < Cout ? n0 + q4 : 0;                 // (Cout % 4 == 0; beyond Cout: a valid address, never used)
            v[e] = *reinterpret_cast<const float4*>(wk + (size_t)row * Cout + n);
        }
        for (int i0 = 0; i0 < NP * CIN; i0 += 256 * 4) {               // four loads in flight (wide images: NP * CIN runs to 10 x 256)
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + e * 256 + tid, ic = i < NP * CIN ? i : NP * CIN - 1;
                const int g = m0 - HALO + ic / CIN, c = ic % CIN;
                const int gc = g < 0 ? 0 : (g < npix ? g : npix - 1);
                const float tv = x[(size_t)gc * x_ld + c];
                t[e] = (g >= 0 && g < npix) ? tv : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = i0 + e * 256 + tid;
                if (i < NP * CIN) xs[i] = t[e];
            }
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = e * 256 + tid;
            if (i < NV) *reinterpret_cast<float4*>(ws + 4 * i) = v[e];
        }
    }
    __syncthreads();
    const int oq = tid & 15, pg = tid >> 4;             // output quad, pixel group
    const int p0 = m0 + 4 * pg;                         // 4 pixels of one row (npix % 4 == 0: all four or none)
    if (p0 >= npix || n0 + 4 * oq >= Cout) return;
    const int r = p0 % (H * W), py = r / W, px0 = r % W;
    float acc[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[k][o] = 0.f;
#pragma unroll 1
    for (int dy = -3; dy <= 3; ++dy) {
        const bool rowok = (unsigned)(py + dy) < (unsigned)H;
        const float* xrow = xs + (4 * pg + HALO + dy * W - 3) * CIN;
        float xw[CIN][10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const bool ok = rowok && (unsigned)(px0 + j - 3) < (unsigned)W;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float v = xrow[j * CIN + ci];
                xw[ci][j] = ok ? v : 0.f;
            }
        }
        const float* wrow = ws + (dy + 3) * 7 * CIN * 64 + 4 * oq;
#pragma unroll
        for (int dx = 0; dx < 7; ++dx)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float4 wv = *reinterpret_cast<const float4*>(wrow + (dx * CIN + ci) * 64);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xv = xw[ci][k + dx];
                    acc[k][0] = __fmaf_rn(xv, wv.x, acc[k][0]); acc[k][1] = __fmaf_rn(xv, wv.y, acc[k][1]);
                    acc[k][2] = __fmaf_rn(xv, wv.z, acc[k][2]); acc[k][3] = __fmaf_rn(xv, wv.w, acc[k][3]);
                }
            }
    }
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b4 = *reinterpret_cast<const float4*>(bias + n0 + 4 * oq);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        *reinterpret_cast<float4*>(y + (size_t)(p0 + k) * y_ld + n0 + 4 * oq) =
            make_float4(__fadd_rn(acc[k][0], b4.x), __fadd_rn(acc[k][1], b4.y), __fadd_rn(acc[k][2], b4.z), __fadd_rn(acc[k][3], b4.w));
}

static size_t k7_lds_bytes(int W, int Cin) { return (size_t)(49 * Cin * 64 + (64 + 6 * (W + 1)) * Cin) * 4; }

extern "C" int nfdp_conv_k7_narrow_supported(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || W % 4 != 0 || Cin < 1 || Cin > 8 || Cout < 4 || Cout % 4 != 0) return 0;
    // (the flat halo of a 7x7 window is six image rows: at W = 64 a 64-pixel tile stages 454 pixels, and with 6 channels the launch takes
    // 140 us against the implicit-GEMM kernel's 101 - wide images with more than 4 channels stay there)
    if (Cin > 4 && W > 32) return 0;
    return (long)B * H * W < (1L << 31) - 4096 && k7_lds_bytes(W, Cin) <= 144 * 1024;
}

// w_k7 [49][Cin][Cout] fp32 from the [Cout][Cin][7][7] parameter
extern "C" int nfdp_conv_k7_pack_f32(const float* w_oihw, float* w_k7, int Cout, int Cin, nfdp_stream_t stream) {
    NFDP_REQUIRE(w_oihw && w_k7, NFDP_EINVAL, "conv_k7_pack: null pointer");
    NFDP_REQUIRE(Cin >= 1 && Cout >= 1, NFDP_ESHAPE, "conv_k7_pack: bad shape");
    hipLaunchKernelGGL(conv_k7_pack_kernel, dim3(stream_grid((long)Cout * Cin * 49, kBlock)), dim3(kBlock), 0, as_stream(stream), w_oihw, w_k7, Cout, Cin);
    NFDP_CHECK_LAUNCH("conv_k7_pack");
    return NFDP_OK;
}

// y [npix][y_ld] = conv7x7(x [B][H][W][Cin] (rows x_ld), padding 3) + bias; w_k7 from nfdp_conv_k7_pack_f32.
extern "C" int nfdp_conv_k7_narrow_f32(const float* x, int x_ld, const float* w_k7, const float* bias, float* y, int y_ld, int B, int H,
                                       int W, int Cin, int Cout, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && w_k7 && y, NFDP_EINVAL, "conv_k7_narrow: null pointer");
    NFDP_REQUIRE(nfdp_conv_k7_narrow_supported(B, H, W, Cin, Cout), NFDP_EUNSUPPORTED,
                 "conv_k7_narrow: unsupported geometry B=%d H=%d W=%d (%% 4) Cin=%d (1..8) Cout=%d (%% 4)", B, H, W, Cin, Cout);
    NFDP_REQUIRE(x_ld >= Cin && y_ld >= Cout && y_ld % 4 == 0 && (((uintptr_t)y | (uintptr_t)bias | (uintptr_t)w_k7) % 16) == 0, NFDP_ESHAPE,
                 "conv_k7_narrow: output rows, weights and bias must be 16-byte aligned");
    const int npix = B * H * W;
    const dim3 grid((npix + 63) / 64, (Cout + 63) / 64);
    const size_t lds = k7_lds_bytes(W, Cin);
    hipStream_t s = as_stream(stream);
    static NfdpPerDevice attr_set;
    if (!attr_set) {                                    // (5-8 channels: the 64 outputs' weights alone are 61-98 KB)
        nfdp_func_attr(reinterpret_cast<const void*>(&conv_k7_narrow_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        nfdp_func_attr(reinterpret_cast<const void*>(&conv_k7_narrow_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        nfdp_func_attr(reinterpret_cast<const void*>(&conv_k7_narrow_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        nfdp_func_attr(reinterpret_cast<const void*>(&conv_k7_narrow_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        attr_set = true;
    }
#define NFDP_K7_CASE(CI) case CI: hipLaunchKernelGGL(conv_k7_narrow_kernel<CI>, grid, dim3(256), lds, s, x, x_ld, w_k7, bias, y, y_ld, npix, H, W, Cout); break
    switch (Cin) {
        NFDP_K7_CASE(1); NFDP_K7_CASE(2); NFDP_K7_CASE(3); NFDP_K7_CASE(4); NFDP_K7_CASE(5); NFDP_K7_CASE(6); NFDP_K7_CASE(7);
        default: hipLaunchKernelGGL(conv_k7_narrow_kernel<8>, grid, dim3(256), lds, s, x, x_ld, w_k7, bias, y, y_ld, npix, H, W, Cout); break;
    }
#undef NFDP_K7_CASE
    NFDP_CHECK_LAUNCH("conv_k7_narrow");
    return NFDP_OK;
}

extern "C" int nfdp_add_f32(const float* a, const float* b, float* out, long n, nfdp_stream_t stream) {
    NFDP_REQUIRE(a && b && out, NFDP_EINVAL, "add: null pointer");
    NFDP_REQUIRE(n >= 0, NFDP_ESHAPE, "add: bad size");
    if (n == 0) return NFDP_OK;
    hipLaunchKernelGGL(add_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, as_stream(stream), a, b, out, n);
    NFDP_CHECK_LAUNCH("add");
    return NFDP_OK;
}
