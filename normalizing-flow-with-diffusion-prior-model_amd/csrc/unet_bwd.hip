// Backward kernels of the diffusion prior's U-Net (training the prior on frozen-flow latents: loss.backward() of
// diffusion_prior/trainer.py:134-150 through GaussianDiffusion.p_losses, gaussian_diffusion.py:823-859), exact fp32.
// The convolutions' gradients are the flow's kernels (csrc/backward.hip: weight gradient over pixels, data gradient as
// the forward conv on flipped weights); here are the derivatives of the glue ops of csrc/unet.hip:
//   Block: GroupNorm -> x*(scale+1)+shift -> SiLU        gaussian_diffusion.py:267-276
//   channel LayerNorm (gain only)                        :205-209
//   LinearAttention core                                 :320-332          Attention core :351-357
//   nearest-neighbour x2 upsample                        :169-173          GELU / SiLU of the time MLP :282-285, :407-412
//   p_losses reduction                                   :855-859
// Like the forward kernels these are latency/HBM-bound, one launch each, statistics in fp64, fixed summation orders.
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int DH = 32;

__device__ __forceinline__ double block_sum256(double v, double* red) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ----------------------------------------------------------------------------- GroupNorm + scale/shift + SiLU backward
// one workgroup per (sample, group).  x: input of the norm, dy: gradient of the SiLU output.
//   xh = (x-mu) rstd, n = xh g + b, m = n (1+sc) + sh, y = m sigmoid(m)
//   dm = dy sig (1 + m (1-sig));  dsh[b,c] = sum_px dm;  dsc[b,c] = sum_px dm n;  dn = dm (1+sc)
//   dbeta_b[b,c] = sum_px dn;  dgamma_b[b,c] = sum_px dn xh;  dxh = dn g
//   dx = rstd (dxh - mean_grp(dxh) - xh mean_grp(dxh xh))
// Threads: channel = tid % cpg, pixel lane = tid / cpg (RT = 256 / cpg lanes), so per-channel sums stay in one thread.
__global__ void __launch_bounds__(kBlock) groupnorm_silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   int HW, int C, int groups, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const float* __restrict__ ss,
                                                                   int ss_ld, float eps, float* __restrict__ dx,
                                                                   float* __restrict__ dgamma_b, float* __restrict__ dbeta_b,
                                                                   float* __restrict__ dss) {
    __shared__ double red[4];
    extern __shared__ __attribute__((aligned(16))) double part[];        // [4][RT][cpg]
    const int b = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cpg = C / groups, n = HW * cpg;
    const int RT = kBlock / cpg;
    const int cc = threadIdx.x % cpg, pl = threadIdx.x / cpg;
    const bool active = pl < RT;
    const int c = g * cpg + cc;
    const long base = (long)b * HW * C + (long)g * cpg;
    double s = 0.0;
    if (active)
        for (int p = pl; p < HW; p += RT) s += (double)x[base + (long)p * C + cc];
    const double mean = block_sum256(s, red) / n;
    double q = 0.0;
    if (active)
        for (int p = pl; p < HW; p += RT) {
            const double d = (double)x[base + (long)p * C + cc] - mean;
            q += d * d;
        }
    const double var = block_sum256(q, red) / n;
    const float mu = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float ga = gamma[c], be = beta[c];
    const float sc1 = ss ? __fadd_rn(ss[(long)b * ss_ld + c], 1.0f) : 1.0f;
    const float sh = ss ? ss[(long)b * ss_ld + C + c] : 0.0f;
    double a_dm = 0.0, a_dmn = 0.0, a_dn = 0.0, a_dnx = 0.0;
    if (active)
        for (int p = pl; p < HW; p += RT) {
            const long idx = base + (long)p * C + cc;
            const float xh = __fmul_rn(__fsub_rn(x[idx], mu), rstd);
            const float nn = __fadd_rn(__fmul_rn(xh, ga), be);
            const float m = __fadd_rn(__fmul_rn(nn, sc1), sh);
            const float sig = 1.0f / (1.0f + expf(-m));
            const float dm = dy[idx] * sig * (1.0f + m * (1.0f - sig));
            const float dn = dm * sc1;
            a_dm += (double)dm; a_dmn += (double)(dm * nn); a_dn += (double)dn; a_dnx += (double)(dn * xh);
        }
    // per-channel sums across the pixel lanes (fixed order), then the two group sums
    double* P0 = part;
    double* P1 = part + RT * cpg;
    double* P2 = part + 2 * RT * cpg;
    double* P3 = part + 3 * RT * cpg;
    if (active) {
        P0[pl * cpg + cc] = a_dm; P1[pl * cpg + cc] = a_dmn; P2[pl * cpg + cc] = a_dn; P3[pl * cpg + cc] = a_dnx;
    }
    __syncthreads();
    double c_dn = 0.0, c_dnx = 0.0;
    if (threadIdx.x < cpg) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        for (int r = 0; r < RT; ++r) {
            t0 += P0[r * cpg + threadIdx.x]; t1 += P1[r * cpg + threadIdx.x];
            t2 += P2[r * cpg + threadIdx.x]; t3 += P3[r * cpg + threadIdx.x];
        }
        const int ch = g * cpg + threadIdx.x;
        if (dss) { dss[(long)b * 2 * C + ch] = (float)t1; dss[(long)b * 2 * C + C + ch] = (float)t0; }
        dbeta_b[(long)b * C + ch] = (float)t2;
        dgamma_b[(long)b * C + ch] = (float)t3;
        c_dn = t2 * (double)gamma[ch];          // sum dxh      of this channel
        c_dnx = t3 * (double)gamma[ch];         // sum dxh * xh of this channel
    }
    const double S1 = block_sum256(c_dn, red) / n;
    const double S2 = block_sum256(c_dnx, red) / n;
    const float s1 = (float)S1, s2 = (float)S2;
    if (active)
        for (int p = pl; p < HW; p += RT) {
            const long idx = base + (long)p * C + cc;
            const float xh = __fmul_rn(__fsub_rn(x[idx], mu), rstd);
            const float nn = __fadd_rn(__fmul_rn(xh, ga), be);
            const float m = __fadd_rn(__fmul_rn(nn, sc1), sh);
            const float sig = 1.0f / (1.0f + expf(-m));
            const float dm = dy[idx] * sig * (1.0f + m * (1.0f - sig));
            const float dxh = dm * sc1 * ga;
            dx[idx] = rstd * (dxh - s1 - xh * s2);
        }
}

// ----------------------------------------------------------------------------- channel LayerNorm (gain only) backward
// one wave per pixel: xh = (x-mu) rstd, y = xh g;  dxh = dy g;  dx = rstd (dxh - mean(dxh) - xh mean(dxh xh));
// dgx[p][c] = dy xh  (the caller sums it over pixels for dg)
__global__ void __launch_bounds__(kBlock) chan_layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   long npix, int C, const float* __restrict__ gain, float eps,
                                                                   float* __restrict__ dx, float* __restrict__ dgx) {
    const int lane = threadIdx.x & 63;
    for (long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6); p < npix; p += (long)gridDim.x * 4) {
        const float* xr = x + p * C;
        const float* dr = dy + p * C;
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
        double a = 0.0, bsum = 0.0;
        for (int c = lane; c < C; c += 64) {
            const float xh = __fmul_rn(__fsub_rn(xr[c], m), r);
            const float dxh = dr[c] * gain[c];
            a += (double)dxh;
            bsum += (double)(dxh * xh);
        }
        const float s1 = (float)(wave_sum_d(a) / C), s2 = (float)(wave_sum_d(bsum) / C);
        for (int c = lane; c < C; c += 64) {
            const float xh = __fmul_rn(__fsub_rn(xr[c], m), r);
            const float dxh = dr[c] * gain[c];
            dx[p * C + c] = r * (dxh - s1 - xh * s2);
            dgx[p * C + c] = dr[c] * xh;
        }
    }
}

// ----------------------------------------------------------------------------- linear attention core backward
// one workgroup per (b, head), n <= 256.  LDS rows [n][33]: qsm (softmax_d q), ksm (softmax_n k), vs (v/n), dO.
//   ctx[d][e] = sum_i ksm[i][d] vs[i][e]           out[i][e] = sum_d ctx[d][e] qsm[i][d] scale
//   dctx[d][e] = sum_i dO[i][e] qsm[i][d] scale     dqs[i][d] = sum_e ctx[d][e] dO[i][e]
//   dq = qsm (scale dqs - sum_d qsm scale dqs)      dvs[i][e] = sum_d dctx[d][e] ksm[i][d],  dv = dvs / n
//   dks[i][d] = sum_e dctx[d][e] vs[i][e]           dk = ksm (dks - sum_i dks ksm)
__global__ void __launch_bounds__(kBlock) linear_attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                     float* __restrict__ dqkv, int n, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int LD = DH + 1;
    float* Q = sm;                       // [n][33]
    float* Kk = Q + n * LD;
    float* V = Kk + n * LD;
    float* G = V + n * LD;               // dO, later dks
    float* ctx = G + n * LD;             // [32][33]
    float* dctx = ctx + DH * LD;
    float* part = dctx + DH * LD;        // [8][32]
    float* colv = part + 8 * DH;         // [32]
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int HD = heads * DH, ld = 3 * HD;
    const float* base = qkv + (long)b * n * ld;
    float* dbase = dqkv + (long)b * n * ld;
    const float inv_n = 1.0f / (float)n;
    for (int e = threadIdx.x; e < n * DH; e += kBlock) {
        const int i = e / DH, dd = e % DH;
        Q[i * LD + dd] = base[(long)i * ld + hd * DH + dd];
        Kk[i * LD + dd] = base[(long)i * ld + HD + hd * DH + dd];
        V[i * LD + dd] = base[(long)i * ld + 2 * HD + hd * DH + dd] * inv_n;
        G[i * LD + dd] = dout[((long)b * n + i) * HD + hd * DH + dd];
    }
    __syncthreads();
    // q: softmax over d per pixel (thread per pixel)
    for (int i = threadIdx.x; i < n; i += kBlock) {
        float mq = -INFINITY;
        for (int dd = 0; dd < DH; ++dd) mq = fmaxf(mq, Q[i * LD + dd]);
        float sq = 0.f;
        for (int dd = 0; dd < DH; ++dd) { const float ev = expf(Q[i * LD + dd] - mq); Q[i * LD + dd] = ev; sq += ev; }
        for (int dd = 0; dd < DH; ++dd) Q[i * LD + dd] = Q[i * LD + dd] / sq;
    }
    // k: softmax over n per d (column max / sum, 8 pixel groups per column)
    const int d = threadIdx.x % DH, grp = threadIdx.x / DH;
    float m = -INFINITY;
    for (int i = grp; i < n; i += 8) m = fmaxf(m, Kk[i * LD + d]);
    part[grp * DH + d] = m;
    __syncthreads();
    if (threadIdx.x < DH) {
        float mm = part[d];
        for (int g2 = 1; g2 < 8; ++g2) mm = fmaxf(mm, part[g2 * DH + d]);
        colv[d] = mm;
    }
    __syncthreads();
    float s = 0.f;
    for (int i = grp; i < n; i += 8) { const float ev = expf(Kk[i * LD + d] - colv[d]); Kk[i * LD + d] = ev; s += ev; }
    __syncthreads();
    part[grp * DH + d] = s;
    __syncthreads();
    if (threadIdx.x < DH) {
        float tot = 0.f;
        for (int g2 = 0; g2 < 8; ++g2) tot += part[g2 * DH + d];
        colv[d] = tot;
    }
    __syncthreads();
    for (int i = grp; i < n; i += 8) Kk[i * LD + d] = Kk[i * LD + d] / colv[d];
    __syncthreads();
    // ctx and dctx: thread owns d and e in [4 grp, 4 grp + 4)
    {
        float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i) {
            const float kv = Kk[i * LD + d], qv = Q[i * LD + d] * scale;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0[j] = fmaf(kv, V[i * LD + 4 * grp + j], a0[j]);
                a1[j] = fmaf(qv, G[i * LD + 4 * grp + j], a1[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { ctx[d * LD + 4 * grp + j] = a0[j]; dctx[d * LD + 4 * grp + j] = a1[j]; }
    }
    __syncthreads();
    // per pixel: dq, dv, dks (dks overwrites this pixel's dO row, which only this thread reads)
    for (int i = threadIdx.x; i < n; i += kBlock) {
        float go[DH];
#pragma unroll
        for (int e = 0; e < DH; ++e) go[e] = G[i * LD + e];
        float dot = 0.f;
        float dqs[DH];
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) {
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < DH; ++e) acc = fmaf(ctx[dd * LD + e], go[e], acc);
            dqs[dd] = acc * scale;
            dot = fmaf(Q[i * LD + dd], dqs[dd], dot);
        }
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) dbase[(long)i * ld + hd * DH + dd] = Q[i * LD + dd] * (dqs[dd] - dot);
#pragma unroll
        for (int e = 0; e < DH; ++e) {
            float acc = 0.f;
            for (int dd = 0; dd < DH; ++dd) acc = fmaf(dctx[dd * LD + e], Kk[i * LD + dd], acc);
            dbase[(long)i * ld + 2 * HD + hd * DH + e] = acc * inv_n;
        }
        float vrow[DH];
#pragma unroll
        for (int e = 0; e < DH; ++e) vrow[e] = V[i * LD + e];
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) {
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < DH; ++e) acc = fmaf(dctx[dd * LD + e], vrow[e], acc);
            G[i * LD + dd] = acc;                                   // dks[i][dd]
        }
    }
    __syncthreads();
    // k softmax backward over n: column dot, then dk
    float cd = 0.f;
    for (int i = grp; i < n; i += 8) cd = fmaf(G[i * LD + d], Kk[i * LD + d], cd);
    part[grp * DH + d] = cd;
    __syncthreads();
    if (threadIdx.x < DH) {
        float tot = 0.f;
        for (int g2 = 0; g2 < 8; ++g2) tot += part[g2 * DH + d];
        colv[d] = tot;
    }
    __syncthreads();
    for (int i = grp; i < n; i += 8) dbase[(long)i * ld + HD + hd * DH + d] = Kk[i * LD + d] * (G[i * LD + d] - colv[d]);
}

// ----------------------------------------------------------------------------- full attention core backward, n <= 256
// one workgroup per (b, head).  Phase A, thread i = query: row max / denominator / rowdot = sum_j p_ij dattn_ij, and dq_i;
// phase B, thread j = key: dk_j = sum_i dsim_ij qs_i, dv_j = sum_i p_ij dO_i (rows recomputed, nothing is atomically added).
__global__ void __launch_bounds__(kBlock) attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                              float* __restrict__ dqkv, int n, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int LD = DH + 1;
    float* Q = sm;                       // scaled q
    float* Kk = Q + n * LD;
    float* V = Kk + n * LD;
    float* G = V + n * LD;               // dO
    float* rmax = G + n * LD;            // [n]
    float* rden = rmax + n;
    float* rdot = rden + n;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int HD = heads * DH, ld = 3 * HD;
    const float* base = qkv + (long)b * n * ld;
    float* dbase = dqkv + (long)b * n * ld;
    for (int e = threadIdx.x; e < n * DH; e += kBlock) {
        const int i = e / DH, dd = e % DH;
        Q[i * LD + dd] = __fmul_rn(base[(long)i * ld + hd * DH + dd], scale);
        Kk[i * LD + dd] = base[(long)i * ld + HD + hd * DH + dd];
        V[i * LD + dd] = base[(long)i * ld + 2 * HD + hd * DH + dd];
        G[i * LD + dd] = dout[((long)b * n + i) * HD + hd * DH + dd];
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < n) {
        float q[DH], go[DH];
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) { q[dd] = Q[t * LD + dd]; go[dd] = G[t * LD + dd]; }
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) {
            float sdot = 0.f;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) sdot = fmaf(q[dd], Kk[j * LD + dd], sdot);
            mx = fmaxf(mx, sdot);
        }
        float den = 0.f, rd = 0.f;
        for (int j = 0; j < n; ++j) {
            float sdot = 0.f, da = 0.f;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) { sdot = fmaf(q[dd], Kk[j * LD + dd], sdot); da = fmaf(go[dd], V[j * LD + dd], da); }
            const float p = expf(sdot - mx);
            den += p;
            rd = fmaf(p, da, rd);
        }
        rd /= den;
        rmax[t] = mx; rden[t] = den; rdot[t] = rd;
        float dq[DH];
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) dq[dd] = 0.f;
        for (int j = 0; j < n; ++j) {
            float sdot = 0.f, da = 0.f;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) { sdot = fmaf(q[dd], Kk[j * LD + dd], sdot); da = fmaf(go[dd], V[j * LD + dd], da); }
            const float dsim = (expf(sdot - mx) / den) * (da - rd);
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) dq[dd] = fmaf(dsim, Kk[j * LD + dd], dq[dd]);
        }
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) dbase[(long)t * ld + hd * DH + dd] = dq[dd] * scale;
    }
    __syncthreads();
    if (t < n) {
        float kj[DH], vj[DH], dk[DH], dv[DH];
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) { kj[dd] = Kk[t * LD + dd]; vj[dd] = V[t * LD + dd]; dk[dd] = 0.f; dv[dd] = 0.f; }
        for (int i = 0; i < n; ++i) {
            float sdot = 0.f, da = 0.f;
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) { sdot = fmaf(Q[i * LD + dd], kj[dd], sdot); da = fmaf(G[i * LD + dd], vj[dd], da); }
            const float p = expf(sdot - rmax[i]) / rden[i];
            const float dsim = p * (da - rdot[i]);
#pragma unroll
            for (int dd = 0; dd < DH; ++dd) {
                dk[dd] = fmaf(dsim, Q[i * LD + dd], dk[dd]);
                dv[dd] = fmaf(p, G[i * LD + dd], dv[dd]);
            }
        }
#pragma unroll
        for (int dd = 0; dd < DH; ++dd) {
            dbase[(long)t * ld + HD + hd * DH + dd] = dk[dd];
            dbase[(long)t * ld + 2 * HD + hd * DH + dd] = dv[dd];
        }
    }
}

// dx[b,y,x,c] = sum of the 2x2 block of dy (backward of nearest-neighbour x2)
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long total, int C, int H, int W) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int c = (int)(i % C);
        long r = i / C;
        const int xx = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const long b = r / H;
        const long o = ((b * 2 * H + 2 * y) * (2 * W) + 2 * xx) * C + c;
        const long row = (long)2 * W * C;
        dx[i] = (dy[o] + dy[o + C]) + (dy[o + row] + dy[o + row + C]);
    }
}

// act: 1 GELU (erf form, nn.GELU default), 2 SiLU
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const float v = x[i];
        y[i] = act == 1 ? 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)) : v / (1.0f + expf(-v));
    }
}
__global__ void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n, int act) {
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const float v = x[i];
        float g;
        if (act == 1) {
            const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
            g = cdf + v * 0.3989422804014327f * expf(-0.5f * v * v);
        } else {
            const float sg = 1.0f / (1.0f + expf(-v));
            g = sg * (1.0f + v * (1.0f - sg));
        }
        dx[i] = dy[i] * g;
    }
}

// d pred of  mean_b( mean_chw(|pred-target|^p) * w[t_b] ):  gout * w[t_b] / (B * per_sample) * (2 (pred-target) | sign)
__global__ void diffusion_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                          const long* __restrict__ t, const float* __restrict__ w,
                                          const float* __restrict__ gout, long total, int per_sample, int B, int l1,
                                          float* __restrict__ dpred) {
    const float g0 = gout[0] / ((float)B * (float)per_sample);
    for (long i = blockIdx.x * (long)kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const float d = __fsub_rn(pred[i], target[i]);
        const float c = g0 * w[t[i / per_sample]];
        dpred[i] = l1 ? c * (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f)) : 2.0f * c * d;
    }
}

}  // namespace

extern "C" int nfdp_groupnorm_silu_bwd_f32(const float* x, const float* dy, int B, int HW, int C, int groups, const float* gamma,
                                           const float* beta, const float* scale_shift, int ss_ld, float eps, float* dx,
                                           float* dgamma_b, float* dbeta_b, float* dscale_shift, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && dy && gamma && beta && dx && dgamma_b && dbeta_b, NFDP_EINVAL, "groupnorm_silu_bwd: null pointer");
    NFDP_REQUIRE(B >= 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && C / groups <= kBlock, NFDP_ESHAPE,
                 "groupnorm_silu_bwd: bad shape B=%d HW=%d C=%d groups=%d", B, HW, C, groups);
    NFDP_REQUIRE((scale_shift == nullptr) == (dscale_shift == nullptr), NFDP_EINVAL,
                 "groupnorm_silu_bwd: scale_shift and its gradient go together");
    if (B == 0) return NFDP_OK;
    const int cpg = C / groups, RT = kBlock / cpg;
    const size_t lds = (size_t)4 * RT * cpg * sizeof(double);
    hipLaunchKernelGGL(groupnorm_silu_bwd_kernel, dim3(B * groups), dim3(kBlock), lds, as_stream(stream), x, dy, HW, C, groups, gamma,
                       beta, scale_shift, ss_ld, eps, dx, dgamma_b, dbeta_b, dscale_shift);
    NFDP_CHECK_LAUNCH("groupnorm_silu_bwd");
    return NFDP_OK;
}

extern "C" int nfdp_chan_layernorm_bwd_f32(const float* x, const float* dy, long npix, int C, const float* gain, float eps,
                                           float* dx, float* dgx, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && dy && gain && dx && dgx, NFDP_EINVAL, "chan_layernorm_bwd: null pointer");
    NFDP_REQUIRE(npix >= 0 && C > 0, NFDP_ESHAPE, "chan_layernorm_bwd: bad shape");
    if (npix == 0) return NFDP_OK;
    long g = cdiv(npix, 4);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(chan_layernorm_bwd_kernel, dim3((int)g), dim3(kBlock), 0, as_stream(stream), x, dy, npix, C, gain, eps, dx, dgx);
    NFDP_CHECK_LAUNCH("chan_layernorm_bwd");
    return NFDP_OK;
}

static size_t attn_bwd_lds(int n, bool linear) {
    const size_t rows = (size_t)4 * n * (DH + 1);
    return (rows + (linear ? (size_t)2 * DH * (DH + 1) + 8 * DH + DH : (size_t)3 * n)) * sizeof(float);
}

extern "C" int nfdp_linear_attention_bwd_f32(const float* qkv, const float* dout, float* dqkv, int B, int n, int heads,
                                             int dim_head, float scale, nfdp_stream_t stream) {
    NFDP_REQUIRE(qkv && dout && dqkv, NFDP_EINVAL, "linear_attention_bwd: null pointer");
    NFDP_REQUIRE(dim_head == DH, NFDP_EUNSUPPORTED, "linear_attention_bwd: dim_head=%d (32 only)", dim_head);
    NFDP_REQUIRE(B >= 0 && n > 0 && n <= 256 && heads > 0, NFDP_ESHAPE, "linear_attention_bwd: n=%d outside [1,256]", n);
    if (B == 0) return NFDP_OK;
    const size_t lds = attn_bwd_lds(n, true);
    static NfdpPerDevice attr;
    if (!attr) {
        nfdp_func_attr(reinterpret_cast<const void*>(linear_attention_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)attn_bwd_lds(256, true));
        attr = true;
    }
    hipLaunchKernelGGL(linear_attention_bwd_kernel, dim3(B * heads), dim3(kBlock), lds, as_stream(stream), qkv, dout, dqkv, n, heads,
                       scale);
    NFDP_CHECK_LAUNCH("linear_attention_bwd");
    return NFDP_OK;
}

extern "C" int nfdp_attention_bwd_f32(const float* qkv, const float* dout, float* dqkv, int B, int n, int heads, int dim_head,
                                      float scale, nfdp_stream_t stream) {
    NFDP_REQUIRE(qkv && dout && dqkv, NFDP_EINVAL, "attention_bwd: null pointer");
    NFDP_REQUIRE(dim_head == DH, NFDP_EUNSUPPORTED, "attention_bwd: dim_head=%d (32 only)", dim_head);
    NFDP_REQUIRE(B >= 0 && n > 0 && n <= 256 && heads > 0, NFDP_ESHAPE, "attention_bwd: n=%d outside [1,256]", n);
    if (B == 0) return NFDP_OK;
    const size_t lds = attn_bwd_lds(n, false);
    static NfdpPerDevice attr;
    if (!attr) {
        nfdp_func_attr(reinterpret_cast<const void*>(attention_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)attn_bwd_lds(256, false));
        attr = true;
    }
    hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * heads), dim3(kBlock), lds, as_stream(stream), qkv, dout, dqkv, n, heads, scale);
    NFDP_CHECK_LAUNCH("attention_bwd");
    return NFDP_OK;
}

extern "C" int nfdp_upsample_nearest2x_bwd_f32(const float* dy, float* dx, int B, int C, int H, int W, nfdp_stream_t stream) {
    NFDP_REQUIRE(dy && dx, NFDP_EINVAL, "upsample_nearest2x_bwd: null pointer");
    NFDP_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, NFDP_ESHAPE, "upsample_nearest2x_bwd: bad shape");
    const long total = (long)B * H * W * C;
    if (total == 0) return NFDP_OK;
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream), dy, dx, total, C, H, W);
    NFDP_CHECK_LAUNCH("upsample_nearest2x_bwd");
    return NFDP_OK;
}

extern "C" int nfdp_act_fwd_f32(const float* x, float* y, long n, int act, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && y, NFDP_EINVAL, "act_fwd: null pointer");
    NFDP_REQUIRE(n >= 0 && (act == 1 || act == 2), NFDP_EINVAL, "act_fwd: act=%d (1 GELU, 2 SiLU)", act);
    if (n == 0) return NFDP_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, as_stream(stream), x, y, n, act);
    NFDP_CHECK_LAUNCH("act_fwd");
    return NFDP_OK;
}

extern "C" int nfdp_act_bwd_f32(const float* x, const float* dy, float* dx, long n, int act, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && dy && dx, NFDP_EINVAL, "act_bwd: null pointer");
    NFDP_REQUIRE(n >= 0 && (act == 1 || act == 2), NFDP_EINVAL, "act_bwd: act=%d (1 GELU, 2 SiLU)", act);
    if (n == 0) return NFDP_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, as_stream(stream), x, dy, dx, n, act);
    NFDP_CHECK_LAUNCH("act_bwd");
    return NFDP_OK;
}

extern "C" int nfdp_diffusion_loss_bwd_f32(const float* pred, const float* target, const long* t, const float* p2_weight,
                                           const float* grad_out, int B, int per_sample, int l1, float* dpred,
                                           nfdp_stream_t stream) {
    NFDP_REQUIRE(pred && target && t && p2_weight && grad_out && dpred, NFDP_EINVAL, "diffusion_loss_bwd: null pointer");
    NFDP_REQUIRE(B > 0 && per_sample > 0, NFDP_ESHAPE, "diffusion_loss_bwd: bad shape");
    const long total = (long)B * per_sample;
    hipLaunchKernelGGL(diffusion_loss_bwd_kernel, dim3(stream_grid(total, kBlock)), dim3(kBlock), 0, as_stream(stream), pred, target,
                       t, p2_weight, grad_out, total, per_sample, B, l1, dpred);
    NFDP_CHECK_LAUNCH("diffusion_loss_bwd");
    return NFDP_OK;
}
