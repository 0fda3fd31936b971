// The boundary between two Glow levels, and the end of the forward pass, as single launches.
//
// Reference (normalizing_flow/): after the K flows of a level, GlowBlock.transform (glow.py:107-113) runs Split.transform
// (transforms.py:270-289): chunk -> ZeroConv2d 3x3 on the kept half (utils.py:43-44: (conv + bias) * exp(3 logs)) -> the Gaussian
// log-density of the dropped half (prior.py:36-37) into logp; the next block then squeezes the kept half (transforms.py:226).
// Op by op that is a narrow conv launch, the log-density launch, a copy of the dropped half and the squeeze - four launches of 5-20 us
// on 0.4-3 MB of data, each at its launch floor, twice per forward pass of a 3-level model: ~100 of the 197 us that the headline step
// spends outside its three level kernels (profiles/r05_a_fwd_per_step.txt).
//
// split_prior_squeeze_kernel: one workgroup per image.  The kept half of the image sits in LDS (<= 24 KB); a thread owns a pixel and
// CPT of its dropped channels and accumulates their 2 * CPT conv outputs (mean, log-sd) as fp32 FMA chains in k = (tap, ci) order
// over zero-padded 3x3 neighbourhoods (weights broadcast from LDS), then the log-density terms in the reference's op order, summed
// per image in fp64 (every term is an fp32 value: the fp64 sum of a few thousand of them is exact, so its order is immaterial).  The
// same workgroup writes the dropped half out contiguously (the latent part) and the kept half squeezed into the next level's buffer.
// HBM-side work: one read of the level buffer, one write of each output: 2 * C * 4 bytes per pixel.
//
// ldj_add_levels_kernel: the weight-only log-det terms of ALL levels (H*W * sum_k log|det W_k| per level, transforms.py:131) added
// to the per-sample accumulator by one launch instead of one per level; per level the same fp64 sum -> fp32 -> hw * term rounding as
// nfdp_ldj_add_scaled_f32, added in level order: bit-identical.
//
// gauss_logp_prior_kernel: GaussianPrior.compute_log_prob (prior.py:70-83) with the ZeroConv2d-on-zeros output h = bias * exp(3 logs)
// formed in the kernel (it was a launch of its own in front of the log-density launch).
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

// -0.5 * (Log2PI + 2*logs + d^2 * exp(-2*logs)), fp32 op by op as prior.py:36 (the expression of gauss_logp_kernel, coupling.hip)
__device__ __forceinline__ float gauss_term(float zv, float mean, float logs) {
    const float d = __fsub_rn(zv, mean);
    const float t = __fmul_rn(__fmul_rn(d, d), expf(__fmul_rn(-2.0f, logs)));
    return __fmul_rn(-0.5f, __fadd_rn(__fadd_rn(kLog2Pi, __fmul_rn(2.0f, logs)), t));
}

// CZ: channels of the kept half (= of the dropped half); CPT: dropped channels per thread; TPP = CZ / CPT threads share a pixel.
template <int CZ, int CPT>
__global__ void __launch_bounds__(kBlock) split_prior_squeeze_kernel(const float* __restrict__ x, int H, int W,
                                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                                    const float* __restrict__ logs, float* __restrict__ z,
                                                                    float* __restrict__ ynext, double* __restrict__ logp) {
    constexpr int C = 2 * CZ, TPP = CZ / CPT, K = 9 * CZ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Y = smem;                                   // [HW][CZ] kept half
    const int HW = H * W;
    float* Wl = smem + ((HW * CZ + 3) & ~3);           // [K][C]: k = tap * CZ + ci, then the output channel
    __shared__ double red[kBlock / 64];
    const int tid = threadIdx.x;
    const float* xb = x + (long)blockIdx.x * HW * C;
    float* zb = z + (long)blockIdx.x * HW * CZ;
    // the image: kept half to LDS, dropped half straight out (the latent part)
    for (int e = tid; e < HW * C; e += kBlock) {
        const int p = e / C, c = e - p * C;
        const float v = xb[e];
        if (c < CZ) Y[p * CZ + c] = v;
        else zb[p * CZ + (c - CZ)] = v;
    }
    const bool prior = logp != nullptr;                // (uniform)
    if (prior && w != nullptr)
        for (int e = tid; e < K * C; e += kBlock) {    // OIHW [C][CZ][3][3] -> [k][o]
            const int k = e / C, o = e - k * C;
            const int tap = k / CZ, ci = k - tap * CZ;
            Wl[e] = w[(o * CZ + ci) * 9 + tap];
        }
    __syncthreads();
    // the squeezed kept half: ynext[h][w][4c + 2h1 + w1] = Y[2h + h1][2w + w1][c]   (transforms.py:226)
    {
        const int h2 = H / 2, w2 = W / 2;
        float4* yo = reinterpret_cast<float4*>(ynext + (long)blockIdx.x * HW * CZ);
        for (int e = tid; e < h2 * w2 * CZ; e += kBlock) {
            const int c = e % CZ, q = e / CZ;
            const int wq = q % w2, hq = q / w2;
            const float* p = Y + ((2 * hq) * W + 2 * wq) * CZ + c;
            yo[e] = make_float4(p[0], p[CZ], p[W * CZ], p[W * CZ + CZ]);
        }
    }
    if (!prior) return;
    double acc = 0.0;
    for (int item = tid; item < HW * TPP; item += kBlock) {
        const int p = item / TPP, c0 = (item - p * TPP) * CPT;
        const int py = p / W, px = p - py * W;
        float am[CPT], al[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) { am[j] = 0.f; al[j] = 0.f; }
        if (w != nullptr) {
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {       // zero padding: the term is +0
                    const float* yr = Y + (yy * W + xx) * CZ;
                    const float* wr = Wl + tap * CZ * C + c0;
#pragma unroll
                    for (int ci = 0; ci < CZ; ++ci) {
                        const float yv = yr[ci];
#pragma unroll
                        for (int j = 0; j < CPT; ++j) {
                            am[j] = fmaf(yv, wr[ci * C + j], am[j]);
                            al[j] = fmaf(yv, wr[ci * C + CZ + j], al[j]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = c0 + j;
            float mean = 0.f, lsd = 0.f;               // learn_prior_mean_logs = False: h = 0 (transforms.py:266)
            if (w != nullptr) {
                mean = __fmul_rn(__fadd_rn(am[j], bias[c]), expf(__fmul_rn(logs[c], 3.0f)));            // utils.py:44
                lsd = __fmul_rn(__fadd_rn(al[j], bias[CZ + c]), expf(__fmul_rn(logs[CZ + c], 3.0f)));
            }
            acc += (double)gauss_term(xb[p * C + CZ + c], mean, lsd);
        }
    }
    const double s = block_sum_d(acc, red);
    if (tid == 0) logp[blockIdx.x] += s;
}

constexpr int kMaxLevels = 8;
struct LevelTerms {
    const float* v[kMaxLevels];
    int n[kMaxLevels];
    float hw[kMaxLevels];
    int levels;
};
__global__ void __launch_bounds__(kBlock) ldj_add_levels_kernel(double* __restrict__ ldj, int B, const LevelTerms t) {
    __shared__ double red[kBlock / 64];
    __shared__ float term[kMaxLevels];
    for (int l = 0; l < t.levels; ++l) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < t.n[l]; i += kBlock) acc += (double)t.v[l][i];
        const double s = block_sum_d(acc, red);
        if (threadIdx.x == 0) term[l] = __fmul_rn(t.hw[l], (float)s);
        __syncthreads();
    }
    for (int b = threadIdx.x; b < B; b += kBlock) {
        double v = ldj[b];
        for (int l = 0; l < t.levels; ++l) v += (double)term[l];
        ldj[b] = v;
    }
}

__global__ void __launch_bounds__(kBlock) gauss_logp_prior_kernel(const float* __restrict__ z, int z_ld, const float* __restrict__ bias,
                                                                 const float* __restrict__ logs, int HW, int Cz,
                                                                 double* __restrict__ logp) {
    __shared__ double red[kBlock / 64];
    extern __shared__ __attribute__((aligned(16))) float hsm[];              // [2 * Cz]: (mean | log-sd), the same at every pixel
    for (int i = threadIdx.x; i < 2 * Cz; i += kBlock)
        hsm[i] = bias ? __fmul_rn(bias[i], expf(__fmul_rn(logs[i], 3.0f))) : 0.f;       // (0 + bias) * exp(logs * 3), utils.py:44
    __syncthreads();
    const long zb = (long)blockIdx.x * HW * z_ld;
    const int n = HW * Cz;
    double acc = 0.0;
    for (int e = threadIdx.x; e < n; e += kBlock) {
        const int p = e / Cz, c = e - p * Cz;
        acc += (double)gauss_term(z[zb + (long)p * z_ld + c], hsm[c], hsm[Cz + c]);
    }
    const double s = block_sum_d(acc, red);
    if (threadIdx.x == 0) logp[blockIdx.x] += s;
}

template <int CZ, int CPT>
int launch_boundary(const float* x, int B, int H, int W, const float* w, const float* bias, const float* logs, float* z, float* ynext,
                    double* logp, hipStream_t s) {
    const size_t lds = (size_t)(((H * W * CZ + 3) & ~3) + 9 * CZ * 2 * CZ) * sizeof(float);
    auto k = split_prior_squeeze_kernel<CZ, CPT>;
    static NfdpPerDevice attr_set;
    if (lds > 48 * 1024 && !attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(k, dim3(B), dim3(kBlock), lds, s, x, H, W, w, bias, logs, z, ynext, logp);
    return NFDP_OK;
}

template <int CZ>
int launch_boundary_cz(const float* x, int B, int H, int W, const float* w, const float* bias, const float* logs, float* z,
                       float* ynext, double* logp, hipStream_t s) {
    const int HW = H * W;
    int tpp = 1;
    for (int d = 1; d <= CZ; ++d)
        if (CZ % d == 0 && d * HW <= kBlock) tpp = d;
    const int cpt = CZ / tpp;
    // (CPT values that occur: CZ itself for images of >= 256 pixels, CZ / 2, CZ / 4 ... for smaller ones)
    if (cpt == CZ) return launch_boundary<CZ, CZ>(x, B, H, W, w, bias, logs, z, ynext, logp, s);
    if constexpr (CZ % 2 == 0) if (cpt == CZ / 2) return launch_boundary<CZ, CZ / 2>(x, B, H, W, w, bias, logs, z, ynext, logp, s);
    if constexpr (CZ % 3 == 0) if (cpt == CZ / 3) return launch_boundary<CZ, CZ / 3>(x, B, H, W, w, bias, logs, z, ynext, logp, s);
    if constexpr (CZ % 4 == 0) if (cpt == CZ / 4) return launch_boundary<CZ, CZ / 4>(x, B, H, W, w, bias, logs, z, ynext, logp, s);
    if constexpr (CZ % 6 == 0) if (cpt == CZ / 6) return launch_boundary<CZ, CZ / 6>(x, B, H, W, w, bias, logs, z, ynext, logp, s);
    if constexpr (CZ % 12 == 0) if (cpt == CZ / 12) return launch_boundary<CZ, CZ / 12>(x, B, H, W, w, bias, logs, z, ynext, logp, s);
    if constexpr (CZ % 8 == 0) if (cpt == CZ / 8) return launch_boundary<CZ, CZ / 8>(x, B, H, W, w, bias, logs, z, ynext, logp, s);
    return launch_boundary<CZ, 1>(x, B, H, W, w, bias, logs, z, ynext, logp, s);
}

}  // namespace

extern "C" int nfdp_split_prior_squeeze_supported(int C, int H, int W) {
    if (C <= 0 || C % 2 || H <= 0 || W <= 0 || H % 2 || W % 2) return 0;
    const int cz = C / 2;
    if (cz != 2 && cz != 4 && cz != 6 && cz != 8 && cz != 12 && cz != 24) return 0;
    const long lds = ((long)H * W * cz + 3 + 9L * cz * 2 * cz) * 4;
    return lds <= 96 * 1024;
}

extern "C" int nfdp_split_prior_squeeze_f32(const float* x, int B, int H, int W, int C, const float* w, const float* bias,
                                            const float* logs, float* z, float* ynext, double* logp, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && z && ynext, NFDP_EINVAL, "split_prior_squeeze: null pointer");
    NFDP_REQUIRE(B >= 0 && nfdp_split_prior_squeeze_supported(C, H, W), NFDP_EUNSUPPORTED,
                 "split_prior_squeeze: unsupported geometry B=%d H=%d W=%d C=%d", B, H, W, C);
    NFDP_REQUIRE(w == nullptr || (bias && logs), NFDP_EINVAL, "split_prior_squeeze: conv weight without bias / logs");
    NFDP_REQUIRE(((uintptr_t)ynext & 15) == 0, NFDP_ESHAPE, "split_prior_squeeze: ynext must be 16-byte aligned");
    if (B == 0) return NFDP_OK;
    hipStream_t s = as_stream(stream);
    switch (C / 2) {
        case 2: launch_boundary_cz<2>(x, B, H, W, w, bias, logs, z, ynext, logp, s); break;
        case 4: launch_boundary_cz<4>(x, B, H, W, w, bias, logs, z, ynext, logp, s); break;
        case 6: launch_boundary_cz<6>(x, B, H, W, w, bias, logs, z, ynext, logp, s); break;
        case 8: launch_boundary_cz<8>(x, B, H, W, w, bias, logs, z, ynext, logp, s); break;
        case 12: launch_boundary_cz<12>(x, B, H, W, w, bias, logs, z, ynext, logp, s); break;
        default: launch_boundary_cz<24>(x, B, H, W, w, bias, logs, z, ynext, logp, s); break;
    }
    NFDP_CHECK_LAUNCH("split_prior_squeeze");
    return NFDP_OK;
}

extern "C" int nfdp_ldj_add_levels_f32(double* ldj, int B, const float* const* v, const int* n_sum, const float* hw, int levels,
                                       nfdp_stream_t stream) {
    NFDP_REQUIRE(ldj && v && n_sum && hw, NFDP_EINVAL, "ldj_add_levels: null pointer");
    NFDP_REQUIRE(B >= 0 && levels >= 1 && levels <= kMaxLevels, NFDP_ESHAPE, "ldj_add_levels: 1..%d levels (%d)", kMaxLevels, levels);
    LevelTerms t;
    t.levels = levels;
    for (int l = 0; l < kMaxLevels; ++l) {
        const bool on = l < levels;
        NFDP_REQUIRE(!on || (v[l] && n_sum[l] >= 0), NFDP_EINVAL, "ldj_add_levels: level %d", l);
        t.v[l] = on ? v[l] : nullptr; t.n[l] = on ? n_sum[l] : 0; t.hw[l] = on ? hw[l] : 0.f;
    }
    if (B == 0) return NFDP_OK;
    hipLaunchKernelGGL(ldj_add_levels_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), ldj, B, t);
    NFDP_CHECK_LAUNCH("ldj_add_levels");
    return NFDP_OK;
}

extern "C" int nfdp_gauss_logp_prior_f32(const float* z, int z_ld, const float* bias, const float* logs, int B, int HW, int Cz,
                                         double* logp, nfdp_stream_t stream) {
    NFDP_REQUIRE(z && logp, NFDP_EINVAL, "gauss_logp_prior: null pointer");
    NFDP_REQUIRE(bias == nullptr || logs != nullptr, NFDP_EINVAL, "gauss_logp_prior: bias without logs");
    NFDP_REQUIRE(B >= 0 && HW > 0 && Cz > 0 && Cz <= 4096 && z_ld >= Cz, NFDP_ESHAPE, "gauss_logp_prior: bad shape");
    if (B == 0) return NFDP_OK;
    hipLaunchKernelGGL(gauss_logp_prior_kernel, dim3(B), dim3(kBlock), 2 * Cz * sizeof(float), as_stream(stream), z, z_ld, bias, logs, HW,
                       Cz, logp);
    NFDP_CHECK_LAUNCH("gauss_logp_prior");
    return NFDP_OK;
}
