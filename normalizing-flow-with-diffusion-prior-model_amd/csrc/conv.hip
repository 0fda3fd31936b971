// Stride-1 "same" convolution on NHWC tensors as an MFMA implicit GEMM for gfx950.
//
// Replaces the F.conv2d calls of the coupling network (normalizing_flow/utils.py:43-44, 67-69: 3x3 C/2->512,
// 1x1 512->512, 3x3 512->C with the ActNorm / exp(3*logs) scale folded into the epilogue) and of the U-Net
// (diffusion_prior/gaussian_diffusion.py: 7x7, 3x3, 1x1).
//
// GEMM view: D[p][n] = sum_k A[p][k] * Wp[n][k],  p = output pixel (M), n = output channel (N),
//            k = (ky*KW+kx)*Cin + ci (K), A gathered on the fly (implicit im2col, zero padding).
// MFMA:      compute=F32  -> v_mfma_f32_16x16x4_f32   (exact fp32 FMA chain; the 1e-4 parity mode)
//            compute=BF16 -> v_mfma_f32_16x16x32_bf16 (fp32 accumulate; the speed mode)
// Both use the same 32-deep K tile: lane (r = lane&15, q = lane>>4) owns k = 8q..8q+7 of the tile for its row,
// read from LDS as one (bf16) or two (fp32) 16-byte chunks.  For fp32 the eight 16x16x4 steps take k = 8q+kk,
// kk = 0..7 — a permutation of the K order that A and B share, so the sum is unchanged.
// LDS rows are 32 k wide (64 B bf16 / 128 B fp32) with the 16-byte chunk index XORed by (row>>1) so the
// ds_read_b128 of 16 consecutive rows hits 16 distinct bank groups.
// Pipeline: register-staged prefetch (global loads for tile t+1 issued before the MFMAs of tile t, written to
// the other LDS buffer after them), one barrier per K tile.
// Block->tile map is XCD-aware: the N tiles of one M tile run back to back on the same XCD (blockIdx%8 labels
// the XCD group), so the gathered A rows are served from that XCD's L2 instead of HBM for N tiles 2..Nt.
#include <type_traits>
#include "nfdp_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int BK = 32;

struct ConvParams {
    const void* x;
    const void* w;
    void* y;
    const float* add;
    const float* mul;
    long npix;
    int x_ld, y_ld, y_bf16;
    int H, W, Cin, Cout, KH, KW, pad;
    int K, Kpad, Mt, Nt, relu;
    float mul_exp;
    const void* mask_y = nullptr;     // pw_gemm masked epilogue (fused scale/ReLU backward): the saved layer output, bf16
    int mask_ld = 0;
    double* part = nullptr;           // [Mt][2][Cout] partial sums of  g*y  and  dc
    // split-K conv with the Block epilogue (GroupNorm -> x*(scale+1)+shift -> SiLU (+ residual)): conv_splitk_f32_kernel<.., GN = true>
    const float* gn_gamma = nullptr;
    const float* gn_beta = nullptr;
    const float* gn_ss = nullptr;     // [B][2*Cout] rows (scale | shift), row stride gn_ss_ld; null: no scale/shift
    const float* gn_res = nullptr;    // residual [npix][gn_res_ld], added after the SiLU; null: none
    int gn_ss_ld = 0, gn_res_ld = 0, gn_groups = 0;
    float gn_eps = 0.f;
    // two-source input (torch.cat((x, skip), dim=1) without the concatenated tensor, reference gaussian_diffusion.py:479-486): input
    // channels [0, Cin1) come from x, [Cin1, Cin) from x2 (fp32, row stride x2_ld); Cin1 % 8 == 0.  Default: one source.
    const void* x2 = nullptr;
    int x2_ld = 0, Cin1 = 1 << 30;
    // host-side only: compute = NFDP_F32X6 - fp32 storage, every product as six bf16 MFMAs on three-way split operands (the EMU kernels)
    int emu = 0;
    size_t wsplit_off = 0;            // byte offset of the split image behind the fp32 matrix (conv3x3_ldsx_kernel)
    // GroupNorm of this conv's OUTPUT applied by the conv that reads it (ResnetBlock: block1's norm inside block2's conv): the plain
    // split-K epilogue also writes, per (pixel tile, group), the fp64 sums of the outputs and of their squares:
    // gs_out[(tile * gs_groups + group) * 2 + {0, 1}]; a tile is a 1 / 2^k part of one image (nfdp_conv2d_stats_tile)
    double* gs_out = nullptr;
    int gs_groups = 0;
    // ... and the reading side (conv3x3_ldsx_kernel<.., GNIN = true>): x is the RAW output of such a conv; what enters the products is
    // SiLU(GN(x) * (scale + 1) + shift) (Block.forward, reference gaussian_diffusion.py:267-276), formed while the tile is staged
    // ... and ResnetBlock's res_conv (a 1x1 conv of the SAME input, reference :288, :301) in the launch of block1's 3x3 conv
    // (conv3x3_ldsx_kernel<.., RES = true>): a second accumulator set over the centre tap of the staged tile, weights from w2's split image
    const void* w2 = nullptr;         // packed fp32 1x1 weight with its split image at w2split_off
    size_t w2split_off = 0;
    const float* add2 = nullptr;      // its bias (nullable)
    float* y2 = nullptr;              // its output [npix][y2_ld]
    int y2_ld = 0;
    int sq = 0;                       // conv_splitk_f32_kernel: KH = KW = 2, stride 2, no padding on x [B][2H][2W][Cin] (Downsample, :176-180)
    int tw = 0, th = 0;               // conv3x3_ldsx_kernel: tw > 0: a workgroup's pixels are a th x tw block of one image (wide images) instead of a flat range
    int up2 = 0;                      // conv3x3_ldsx_kernel<.., UP>: x is [B][H/2][W/2][Cin], read as its nearest-neighbour x2 upsampling (:169-173)
    const double* gi_stats = nullptr; // the producer's sums, gi_parts tiles per image
    const float* gi_gamma = nullptr;
    const float* gi_beta = nullptr;
    const float* gi_ss = nullptr;     // [B][2*Cin] rows (scale | shift), row stride gi_ss_ld; null: none
    int gi_ss_ld = 0, gi_groups = 0, gi_parts = 0;
    float gi_eps = 0.f;
};

__device__ __forceinline__ uint2 pack_bf16x4(float4 v) {
    bf16x4 h;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    return __builtin_bit_cast(uint2, h);
}

// EMU: the fp32 products on the bf16 matrix cores.  v = h + m + l with three bf16 pieces (below)
// represents an fp32 value exactly (3 x 8 significand bits), and a*b = ah*bh + (ah*bm + am*bh) + (am*bm + ah*bl + al*bh) + terms
// below 2^-24 |a*b|: six v_mfma_f32_16x16x32_bf16 (16 cycles each, fp32 accumulation) replace eight v_mfma_f32_16x16x4_f32
// (32 cycles each) per 32 k: 2.7x fewer matrix-pipe cycles at the accuracy of the fp32 FMA chain (measured against fp64 on
// K = 576..4608 sums: 2e-7 relative, the chain itself 5e-7).  The splits are VALU work on the fragments already in registers.
// The pieces are ROUNDED to nearest (h = bf16(v), m = bf16(v - h), l = bf16(v - h - m)): |m| <= 2^-9 |v|, |l| <= 2^-18 |v|, so the three
// dropped products (m l, l m, l l) stay below 2^-27.  Pieces cut by truncation (two ANDs instead of the conversions) are cheaper but only
// reach |m| < 2^-7, |l| < 2^-14 - and all carry v's sign: the dropped terms then add up coherently at 2^-21 per product, and the sampling
// goldens fail (tried in round 5).
__device__ __forceinline__ void split3(const float (&v)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hh = (__bf16)v[e];
        // inf / NaN operands (and finite ones above the bf16 range, whose high piece rounds to inf) travel in h alone: v - h would be
        // inf - inf = NaN and turn an inf result into NaN; with m = l = 0 they propagate as in the fp32 chain
        const float r1 = __builtin_isfinite((float)hh) ? v[e] - (float)hh : 0.f;
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;
        h[e] = hh; m[e] = mm; l[e] = (__bf16)r2;
    }
}

template <bool BF16, bool XBF16, bool VEC, int BM, int BN, int WAVES_M, int WAVES_N, bool EMU = false>
__global__ void __launch_bounds__(kThreads) conv_kernel(const ConvParams P) {
    static_assert(!EMU || !BF16, "the split applies to fp32 operands");
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    static_assert(!XBF16 || BF16, "bf16 storage needs bf16 compute");
    constexpr int ROWB = BF16 ? 64 : 128;              // LDS bytes per 32-k row
    constexpr int NCH = ROWB / 16;                     // 16-byte chunks per row
    constexpr int SWZ = NCH - 1;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int AR = BM / 32;                        // A rows staged per thread
    constexpr int BCH = (BN * NCH + kThreads - 1) / kThreads;   // B chunks staged per thread
    constexpr bool BFULL = (BN * NCH) % kThreads == 0;          // every thread stages BCH chunks: no bounds test
    using AReg = typename std::conditional<XBF16, uint2, float4>::type;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // buffer b: A at smem + b*(A_BYTES+B_BYTES), B right behind it (computed, not table-indexed: stays in registers)

    // ---- XCD-aware block -> (M tile, N tile)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int nt = slot % P.Nt;
    const int mt = (slot / P.Nt) * 8 + xcd;
    if (mt >= P.Mt) return;                            // whole block leaves together
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WTM, wn0 = (wave % WAVES_N) * WTN;
    const int r16 = lane & 15, q = lane >> 4;

    // ---- per-thread A staging rows
    const int kq = tid & 7, rbase = tid >> 3;
    int rp[AR], roy[AR], rox[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const long p = m0 + rbase + 32 * i;
        if (p < P.npix) {
            rp[i] = (int)p;
            rox[i] = (int)(p % P.W);
            roy[i] = (int)((p / P.W) % P.H);
        } else {
            rp[i] = -1; rox[i] = 0; roy[i] = 0;
        }
    }
    AReg areg[AR];
    uint4 breg[BCH];

    auto load_tile = [&](int kt) {
        const int k = kt * BK + kq * 4;
        if constexpr (VEC) {
            const int tap = k / P.Cin, ci = k - tap * P.Cin;
            const int ky = tap / P.KW, kx = tap - ky * P.KW;
            const int dy = ky - P.pad, dx = kx - P.pad;
            const bool kok = k < P.K;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int iy = roy[i] + dy, ix = rox[i] + dx;
                const bool ok = kok && rp[i] >= 0 && (unsigned)iy < (unsigned)P.H && (unsigned)ix < (unsigned)P.W;
                const long off = ((long)rp[i] + dy * P.W + dx) * P.x_ld + ci;
                if constexpr (XBF16) {
                    uint2 v = make_uint2(0u, 0u);
                    if (ok) v = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(P.x) + off);
                    areg[i] = __builtin_bit_cast(AReg, v);
                } else {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    const bool second = ci >= P.Cin1;             // (two-source input: the 4 channels of a load never straddle Cin1)
                    const float* src = second ? reinterpret_cast<const float*>(P.x2) + ((long)rp[i] + dy * P.W + dx) * P.x2_ld + (ci - P.Cin1)
                                              : reinterpret_cast<const float*>(P.x) + off;
                    if (ok) v = *reinterpret_cast<const float4*>(src);
                    areg[i] = __builtin_bit_cast(AReg, v);
                }
            }
        } else {
            int tapj[4], cij[4];
            bool kokj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = k + j;
                tapj[j] = kk / P.Cin;
                cij[j] = kk - tapj[j] * P.Cin;
                kokj[j] = kk < P.K;
            }
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ky = tapj[j] / P.KW, kx = tapj[j] - ky * P.KW;
                    const int dy = ky - P.pad, dx = kx - P.pad;
                    const int iy = roy[i] + dy, ix = rox[i] + dx;
                    const bool ok = kokj[j] && rp[i] >= 0 && (unsigned)iy < (unsigned)P.H && (unsigned)ix < (unsigned)P.W;
                    const long off = ((long)rp[i] + dy * P.W + dx) * P.x_ld + cij[j];
                    float v = 0.f;
                    if (ok) {
                        if (XBF16) v = (float)reinterpret_cast<const __bf16*>(P.x)[off];
                        else if (cij[j] >= P.Cin1) v = reinterpret_cast<const float*>(P.x2)[((long)rp[i] + dy * P.W + dx) * P.x2_ld + (cij[j] - P.Cin1)];
                        else v = reinterpret_cast<const float*>(P.x)[off];
                    }
                    e[j] = v;
                }
                const float4 f = make_float4(e[0], e[1], e[2], e[3]);
                if constexpr (XBF16) areg[i] = __builtin_bit_cast(AReg, pack_bf16x4(f));
                else areg[i] = __builtin_bit_cast(AReg, f);
            }
        }
        // weights: packed [Cout_pad][Kpad], 16-byte chunks of the current k tile
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int idx = tid + i * kThreads;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (BFULL || idx < BN * NCH) {
                const int n = idx / NCH, c = idx % NCH;
                const char* src = reinterpret_cast<const char*>(P.w) +
                                  ((long)(n0 + n) * P.Kpad + (long)kt * BK) * (BF16 ? 2 : 4) + c * 16;
                v = *reinterpret_cast<const uint4*>(src);
            }
            breg[i] = v;
        }
    };

    auto store_tile = [&](int buf) {
        char* A = smem + buf * (A_BYTES + B_BYTES);
        char* Bs = A + A_BYTES;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int r = rbase + 32 * i;
            if constexpr (BF16) {
                uint2 v;
                if constexpr (XBF16) v = __builtin_bit_cast(uint2, areg[i]);
                else v = pack_bf16x4(__builtin_bit_cast(float4, areg[i]));
                *reinterpret_cast<uint2*>(A + r * ROWB + (((kq >> 1) ^ ((r >> 1) & SWZ)) << 4) + ((kq & 1) << 3)) = v;
            } else {
                *reinterpret_cast<float4*>(A + r * ROWB + ((kq ^ ((r >> 1) & SWZ)) << 4)) = __builtin_bit_cast(float4, areg[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int idx = tid + i * kThreads;
            if (BFULL || idx < BN * NCH) {
                const int n = idx / NCH, c = idx % NCH;
                *reinterpret_cast<uint4*>(Bs + n * ROWB + ((c ^ ((n >> 1) & SWZ)) << 4)) = breg[i];
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute_tile = [&](int buf) {
        const char* A = smem + buf * (A_BYTES + B_BYTES);
        const char* Bs = A + A_BYTES;
        if constexpr (BF16) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm0 + i * 16 + r16;
                a[i] = *reinterpret_cast<const bf16x8*>(A + r * ROWB + ((q ^ ((r >> 1) & SWZ)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = wn0 + j * 16 + r16;
                b[j] = *reinterpret_cast<const bf16x8*>(Bs + n * ROWB + ((q ^ ((n >> 1) & SWZ)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
            float a[TM][8], b[TN][8];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm0 + i * 16 + r16;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 v = *reinterpret_cast<const float4*>(A + r * ROWB + (((2 * q + h) ^ ((r >> 1) & SWZ)) << 4));
                    a[i][4 * h] = v.x; a[i][4 * h + 1] = v.y; a[i][4 * h + 2] = v.z; a[i][4 * h + 3] = v.w;
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = wn0 + j * 16 + r16;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 v = *reinterpret_cast<const float4*>(Bs + n * ROWB + (((2 * q + h) ^ ((n >> 1) & SWZ)) << 4));
                    b[j][4 * h] = v.x; b[j][4 * h + 1] = v.y; b[j][4 * h + 2] = v.z; b[j][4 * h + 3] = v.w;
                }
            }
            if constexpr (EMU) {                                   // six bf16 MFMAs per 32 k instead of eight fp32 ones (split3)
                bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) split3(a[i], ah[i], am[i], al[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) split3(b[j], bh[j], bm[j], bl[j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        auto c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bm[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm[j], c, 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], c, 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
            }
        }
    };

    const int nkt = P.Kpad / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);
        compute_tile(cur);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: v = (acc + add[n]) * mul[n]; relu; store fp32 or bf16
    //      C/D map of the 16x16 MFMA: col = lane&15 (n), row = 4*(lane>>4) + reg (pixel)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 16 + r16;
        const bool nok = n < P.Cout;
        const float addv = (nok && P.add) ? P.add[n] : 0.f;
        float mulv = (nok && P.mul) ? P.mul[n] : 1.f;
        if (P.mul && P.mul_exp != 0.f) mulv = expf(__fmul_rn(mulv, P.mul_exp));   // exp(scale) / exp(3*logs)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const long p = m0 + wm0 + i * 16 + 4 * q + rg;
                if (nok && p < P.npix) {
                    float v = acc[i][j][rg];
                    if (P.add) v = __fadd_rn(v, addv);
                    if (P.mul) v = __fmul_rn(v, mulv);
                    if (P.relu) v = fmaxf(v, 0.f);
                    if (P.y_bf16) reinterpret_cast<__bf16*>(P.y)[p * P.y_ld + n] = (__bf16)v;
                    else reinterpret_cast<float*>(P.y)[p * P.y_ld + n] = v;
                }
            }
        }
    }
}

// the same pieces for 4 values (one 16-byte load), each piece as 4 packed bf16: element for element the arithmetic of split3
__device__ __forceinline__ void split3x4(const float4 v4, uint2& h, uint2& m, uint2& l) {
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    bf16x4 hq, mq, lq;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 hh = (__bf16)v[e];
        const float r1 = __builtin_isfinite((float)hh) ? v[e] - (float)hh : 0.f;
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;
        hq[e] = hh; mq[e] = mm; lq[e] = (__bf16)r2;
    }
    h = __builtin_bit_cast(uint2, hq); m = __builtin_bit_cast(uint2, mq); l = __builtin_bit_cast(uint2, lq);
}

// ----------------------------------------------------------------------------- small fp32 convs: K split over the waves
// The U-Net layers of the reverse-diffusion step (B*H*W of a few thousand pixels, K = 9*Cin up to 1728) fill the chip with one
// 64x64 tile per CU, and conv_kernel then spends its time on the load -> LDS -> barrier -> MFMA round trip of each of its
// K/32 tiles (24 us for 7.7 us of fp32 MFMA work).  Here every wave computes the WHOLE 64x64 tile over a quarter of the K
// tiles: operands go global -> VGPR already in MFMA fragment shape (a lane's 8 consecutive k of one pixel / one output
// channel = 32 contiguous bytes; the four q-lanes of a row cover a 128-byte line), double-buffered in registers, no LDS and
// no barrier in the loop; the four partial tiles meet once in LDS and are summed in wave order (deterministic), then the
// epilogue writes float4 rows.  Roles as in pw_gemm (A := weights) so a lane owns 4 consecutive output channels.
// Needs Cin % 8 == 0 (8 consecutive k stay inside one tap), fp32 in and out.
constexpr int SK_LD = 64 + 4;                                  // partial-tile row stride in floats
#ifndef NFDP_SK_NBUF
#define NFDP_SK_NBUF 3
#endif

// PT: 16-pixel tiles per workgroup (4 = the 64 x 64 tile; 2 / 1 when 64-pixel tiles would leave CUs without a workgroup: the 8x8 and
// 4x4 levels of a B = 64 reverse step have 4096 / 1024 pixels - every wave's MFMA chain, the critical path of the launch, shrinks
// with PT while the weight fragments are re-read from L2)
// GN: the Block epilogue (reference gaussian_diffusion.py:267-276: GroupNorm -> x*(scale+1)+shift -> SiLU, and ResnetBlock's residual add
// :301) inside the conv's launch.  Possible when the workgroup's MT pixels are whole images (H*W divides MT: the 8x8, 4x4 and 2x2 stages
// of the U-Nets) and its 64 output channels are whole groups: the summed tile is already in LDS, the statistics of its (image, group)
// sets are taken there in fp64 (two passes, like groupnorm_silu_kernel), and the launch of that kernel - 9 us at its latency floor, 22
// per reverse step - and one round trip of the activation through memory disappear.

// The end of the split-K convs (conv_splitk_f32_kernel, conv3x3_ldsx_kernel): the waves' partial tiles meet in LDS and are summed in wave
// order (deterministic); then the plain epilogue (bias, scale, ReLU) or the Block epilogue (GN, see conv_splitk_f32_kernel).  `part` may
// alias memory the main loop read: the caller synchronises the workgroup first.
template <int SK_WAVES, int PT, bool GN, bool SECOND = false, bool T2D = false>    // SECOND: the res_conv tile of conv3x3_ldsx_kernel<.., RES> (y2, add2; plain); T2D: block tiles
__device__ __forceinline__ void splitk_epilogue(const ConvParams& P, float* part, const f32x4 (&acc)[4][PT], long m0, int n0, int tid,
                                                int wave, int r16, int q, long tile) {
    static_assert(!(GN && SECOND), "the second output is a plain conv");
    // pixel m of the tile: m0 + m, or (P.tw: a th x tw block whose first pixel is m0) row m / tw, column m % tw of the block
    auto pix = [&](int m) -> long {
        if constexpr (T2D) return m0 + (long)(m / P.tw) * P.W + (m % P.tw);
        else return m0 + m;
    };
    constexpr int SK_THREADS = SK_WAVES * 64, MT = 16 * PT;
    const int HW = P.H * P.W;
    // ---- the partial tiles meet in LDS: D[row = n = 4q + r][col = pixel = r16]
    float* mine = part + wave * MT * SK_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
            *reinterpret_cast<float4*>(mine + (j * 16 + r16) * SK_LD + i * 16 + 4 * q) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    __syncthreads();
    constexpr int NF4 = MT * 16;                                   // float4 of the MT x 64 tile
    if constexpr (GN) {
        // ---- A. summed tile (+ bias) back into wave 0's region: all sums first (registers), ONE barrier, then the stores - a barrier
        // inside the loop would be skipped by the threads without an element (MT = 16 with 512 threads) and pair up with a later one
        constexpr int NE = (NF4 + SK_THREADS - 1) / SK_THREADS;
        float4 sums[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = tid + SK_THREADS * e;
            sums[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NF4 % SK_THREADS != 0 && idx >= NF4) continue;
            const int m = idx >> 4, n4 = (idx & 15) * 4;
            float4 v = *reinterpret_cast<const float4*>(part + m * SK_LD + n4);
#pragma unroll
            for (int w = 1; w < SK_WAVES; ++w) {
                const float4 t = *reinterpret_cast<const float4*>(part + (w * MT + m) * SK_LD + n4);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            if (P.add && n0 + n4 < P.Cout) {
                const float4 a4 = *reinterpret_cast<const float4*>(P.add + n0 + n4);
                v.x = __fadd_rn(v.x, a4.x); v.y = __fadd_rn(v.y, a4.y); v.z = __fadd_rn(v.z, a4.z); v.w = __fadd_rn(v.w, a4.w);
            }
            sums[e] = v;
        }
        __syncthreads();                                           // every thread has read the partial tiles
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = tid + SK_THREADS * e;
            if (NF4 % SK_THREADS != 0 && idx >= NF4) continue;
            *reinterpret_cast<float4*>(part + (idx >> 4) * SK_LD + (idx & 15) * 4) = sums[e];
        }
        __syncthreads();
        // ---- B. statistics of the tile's (image, group) sets: 32 lanes per set, fp64, mean first then the centred squares
        const int cpg = P.Cout / P.gn_groups, gpt = 64 / cpg, ipt = MT / HW, nset = ipt * gpt, nval = HW * cpg;
        double* stat = reinterpret_cast<double*>(part + MT * SK_LD);           // wave 1's region: free after pass A
        const int l32 = tid & 31;
        for (int st = tid >> 5; st < nset; st += SK_THREADS / 32) {
            const int img = st / gpt, grp = st - img * gpt;
            const float* src = part + (img * HW) * SK_LD + grp * cpg;
            double s1 = 0.0;
            for (int i = l32; i < nval; i += 32) s1 += (double)src[(i / cpg) * SK_LD + (i % cpg)];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 32);
            const double mean = s1 / nval;
            double s2 = 0.0;
            for (int i = l32; i < nval; i += 32) {
                const double d = (double)src[(i / cpg) * SK_LD + (i % cpg)] - mean;
                s2 += d * d;
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 32);
            if (l32 == 0) {
                stat[2 * st] = mean;
                stat[2 * st + 1] = 1.0 / sqrt(s2 / nval + (double)P.gn_eps);
            }
        }
        __syncthreads();
        // ---- C. normalise, scale/shift, SiLU, residual; float4 rows out
#pragma unroll
        for (int e = 0; e < (NF4 + SK_THREADS - 1) / SK_THREADS; ++e) {
            const int idx = tid + SK_THREADS * e;
            if (NF4 % SK_THREADS != 0 && idx >= NF4) continue;
            const int m = idx >> 4, n4 = (idx & 15) * 4;
            const long p = m0 + m;
            const int n = n0 + n4;
            if (p >= P.npix || n >= P.Cout) continue;
            const float4 v = *reinterpret_cast<const float4*>(part + m * SK_LD + n4);
            const int st = (m / HW) * gpt + n4 / cpg;
            const float mu = (float)stat[2 * st], rs = (float)stat[2 * st + 1];
            const float4 ga = *reinterpret_cast<const float4*>(P.gn_gamma + n), be = *reinterpret_cast<const float4*>(P.gn_beta + n);
            float o[4] = {v.x, v.y, v.z, v.w};
            const float gq[4] = {ga.x, ga.y, ga.z, ga.w}, bq[4] = {be.x, be.y, be.z, be.w};
            const float* sr = P.gn_ss ? P.gn_ss + (p / HW) * (long)P.gn_ss_ld : nullptr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(o[r], mu), rs), gq[r]), bq[r]);
                if (sr) t = __fadd_rn(__fmul_rn(t, __fadd_rn(sr[n + r], 1.0f)), sr[P.Cout + n + r]);
                o[r] = t / (1.0f + expf(-t));
            }
            if (P.gn_res) {
                const float4 rv = *reinterpret_cast<const float4*>(P.gn_res + p * P.gn_res_ld + n);
                o[0] = __fadd_rn(o[0], rv.x); o[1] = __fadd_rn(o[1], rv.y); o[2] = __fadd_rn(o[2], rv.z); o[3] = __fadd_rn(o[3], rv.w);
            }
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(P.y) + p * P.y_ld + n) = make_float4(o[0], o[1], o[2], o[3]);
        }
        return;
    }
    double gs1 = 0.0, gs2 = 0.0;                                   // (gs_out) this thread's 4 channels: always the same group
#pragma unroll
    for (int e = 0; e < (NF4 + SK_THREADS - 1) / SK_THREADS; ++e) {
        const int idx = tid + SK_THREADS * e;
        if (NF4 % SK_THREADS != 0 && idx >= NF4) continue;
        const int m = idx >> 4, n4 = (idx & 15) * 4;
        const long p = pix(m);
        const int n = n0 + n4;
        if (p >= P.npix || n >= P.Cout) continue;
        float4 v = *reinterpret_cast<const float4*>(part + m * SK_LD + n4);
#pragma unroll
        for (int w = 1; w < SK_WAVES; ++w) {
            const float4 t = *reinterpret_cast<const float4*>(part + (w * MT + m) * SK_LD + n4);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float o[4] = {v.x, v.y, v.z, v.w};
        if constexpr (SECOND) {
            if (P.add2) {
                const float4 a4 = *reinterpret_cast<const float4*>(P.add2 + n);
                o[0] = __fadd_rn(o[0], a4.x); o[1] = __fadd_rn(o[1], a4.y); o[2] = __fadd_rn(o[2], a4.z); o[3] = __fadd_rn(o[3], a4.w);
            }
            *reinterpret_cast<float4*>(P.y2 + p * P.y2_ld + n) = make_float4(o[0], o[1], o[2], o[3]);
            continue;
        }
        if (P.add) {
            const float4 a4 = *reinterpret_cast<const float4*>(P.add + n);
            o[0] = __fadd_rn(o[0], a4.x); o[1] = __fadd_rn(o[1], a4.y); o[2] = __fadd_rn(o[2], a4.z); o[3] = __fadd_rn(o[3], a4.w);
        }
        if (P.gs_out) {                                            // (uniform; host: no mul / relu, whole tiles, Cout % 64 == 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double d = (double)o[r];
                gs1 += d;
                gs2 += d * d;
            }
        }
        if (P.mul) {
            const float4 m4 = *reinterpret_cast<const float4*>(P.mul + n);
            float mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (P.mul_exp != 0.f) mv[r] = expf(__fmul_rn(mv[r], P.mul_exp));
                o[r] = __fmul_rn(o[r], mv[r]);
            }
        }
        if (P.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
        }
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(P.y) + p * P.y_ld + n) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (!SECOND && P.gs_out) {
        // the tile's sums per group: lanes that hold the same group meet by xor-shuffles (lane bits 4, 5: other pixels; the low bits below
        // cpg / 4: the group's other channel quads), the waves in LDS, summed in wave order - fp64 throughout, a fixed association
        __shared__ double gsw[SK_WAVES][16][2];
        const int cpg = P.Cout / P.gs_groups, lpg = cpg >> 2;      // lanes (of the low 16) per group
        gs1 += __shfl_xor(gs1, 16); gs2 += __shfl_xor(gs2, 16);
        gs1 += __shfl_xor(gs1, 32); gs2 += __shfl_xor(gs2, 32);
        for (int o = 1; o < lpg; o <<= 1) {
            gs1 += __shfl_xor(gs1, o);
            gs2 += __shfl_xor(gs2, o);
        }
        const int lane = tid & 63;
        if (lane < 16 && (lane & (lpg - 1)) == 0) {
            gsw[wave][lane / lpg][0] = gs1;
            gsw[wave][lane / lpg][1] = gs2;
        }
        __syncthreads();
        const int gpt = 64 / cpg;                                  // groups of this 64-channel tile
        if (tid < 2 * gpt) {
            const int g = tid >> 1, k = tid & 1;
            double t = gsw[0][g][k];
#pragma unroll
            for (int w = 1; w < SK_WAVES; ++w) t += gsw[w][g][k];
            if constexpr (T2D && PT == 8) {                        // the consumers count 64-pixel tiles: this one's sums, then a zero record
                P.gs_out[((2 * tile) * P.gs_groups + n0 / cpg + g) * 2 + k] = t;
                P.gs_out[((2 * tile + 1) * P.gs_groups + n0 / cpg + g) * 2 + k] = 0.0;
            } else {
                P.gs_out[(tile * P.gs_groups + n0 / cpg + g) * 2 + k] = t;
            }
        }
    }
}

template <int SK_WAVES, int SK_KL, bool EMU, int PT = 4, bool GN = false, int SK_NBUF = (EMU ? NFDP_SK_NBUF : 2)>   // waves per workgroup, consecutive k per lane per stage (a stage is 4*SK_KL deep), register sets of the K pipeline
__global__ void __launch_bounds__(SK_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, SK_WAVES / 4)))
conv_splitk_f32_kernel(const ConvParams P) {
    constexpr int SK_BK = 4 * SK_KL, SK_THREADS = SK_WAVES * 64;
    static_assert(!EMU || SK_KL == 8, "the bf16 MFMA takes 8 consecutive k per lane");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = 16 * PT;                                // pixels per workgroup
    float* part = reinterpret_cast<float*>(smem);              // [SK_WAVES][MT pixels][SK_LD]
    const int nt = blockIdx.x % P.Nt, mt = blockIdx.x / P.Nt;
    const long m0 = (long)mt * MT;
    const int n0 = nt * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: the stage index is scalar)
    const int r16 = lane & 15, q = lane >> 4;
    const float* __restrict__ xg = reinterpret_cast<const float*>(P.x);
    const float* __restrict__ x2g = reinterpret_cast<const float*>(P.x2);
    const float* __restrict__ wg = reinterpret_cast<const float*>(P.w);
    const int HW = P.H * P.W;

    // this lane's PT pixels (one per pixel tile) and 4 weight rows.  pcen / pcen2: the lane's CENTRE pixels in the two sources (always valid
    // addresses: absent pixels point at pixel 0) - a stage then adds one wave-uniform tap offset and the lane's channel offset, where the
    // first version recomputed (batch, y, x) -> address with 64-bit multiplies per pixel and stage (the launch was bound by that
    // arithmetic and the split3 VALU work, ~20 us whatever the matrix shape: profiles/r05_u_small_conv.txt)
    int py[PT], px[PT];
    bool pok[PT];
    const float* pcen[PT];
    const float* pcen2[PT];
#pragma unroll
    for (int j = 0; j < PT; ++j) {
        const long p = m0 + j * 16 + r16;
        pok[j] = p < P.npix;
        const int pc = pok[j] ? (int)p : 0;              // (npix < 2^31: 32-bit divisions - the 64-bit ones were ~1 us of every launch)
        const int b = pc / HW;
        const int r = pc - b * HW;
        py[j] = r / P.W; px[j] = r - py[j] * P.W;
        // (sq) output pixel (b, y, x) reads the 2 x 2 block at (2y, 2x) of the 2H x 2W source: the "centre" is the block's first pixel
        pcen[j] = xg + (P.sq ? (((long)b * 2 * P.H + 2 * py[j]) * 2 * P.W + 2 * px[j]) : (long)pc) * P.x_ld;
        pcen2[j] = x2g ? x2g + (long)pc * P.x2_ld : xg;
    }
    const int wsrc = P.sq ? 2 * P.W : P.W;               // pixels per source row
    // a stage (SK_BK consecutive k) lies inside ONE tap when Cin is a multiple of the stage depth (every 3x3 conv of the U-Nets: 64, 128,
    // 192 channels): tap and channel base are then scalar
    const bool one_tap = P.Cin % SK_BK == 0 && P.x2 == nullptr;
    const float* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wrow[i] = wg + (size_t)(n0 + i * 16 + r16) * P.Kpad + SK_KL * q;

    f32x4 acc[4][PT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // (x as loaded + the lane's validity mask: the mask is applied in compute(), i.e. when the stage is consumed - applied right behind the
    // load it made the wave wait for the data at once, and the prefetch hid nothing)
    struct Frags { float4 w[4][SK_KL / 4]; uint4 x[PT][SK_KL / 4]; unsigned msk[PT]; };
    auto load = [&](Frags& f, int kt_req) {
        const bool live = kt_req < P.Kpad / SK_BK;       // (uniform) a stage beyond the end: valid addresses of the last stage, x masked
        const int kt = live ? kt_req : P.Kpad / SK_BK - 1;
        const int kb = kt * SK_BK;                       // wave-uniform
        int tap, ci;
        if (one_tap) {
            tap = kb / P.Cin;                            // (scalar)
            ci = kb - tap * P.Cin + SK_KL * q;
        } else {
            const int k0 = kb + SK_KL * q;
            tap = k0 / P.Cin;
            ci = k0 - tap * P.Cin;
        }
        const int dy = tap / P.KW - P.pad, dx = tap - (tap / P.KW) * P.KW - P.pad;
        const bool kok = live && kb + SK_KL * q < P.K;
        const bool second = kok && ci >= P.Cin1;        // two-source input: this lane's SK_KL channels lie wholly in x2 (Cin1 % 8 == 0)
        const int shift = dy * wsrc + dx;                // pixel offset of the tap
        const int off1 = shift * P.x_ld + ci, off2 = shift * P.x2_ld + ci - P.Cin1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < SK_KL / 4; ++h) f.w[i][h] = *reinterpret_cast<const float4*>(wrow[i] + kb + 4 * h);
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int y = py[j] + dy, xx = px[j] + dx;
            const bool ok = kok && pok[j] && (P.sq || ((unsigned)y < (unsigned)P.H && (unsigned)xx < (unsigned)P.W));
            // always load (from a valid address: the centre pixel's first channels) and mask the bits: a "load or zero" select makes
            // hipcc branch around every dword and wait for each one
            const float* src = second ? pcen2[j] + (ok ? off2 : 0) : pcen[j] + (ok ? off1 : 0);
            f.msk[j] = ok ? 0xffffffffu : 0u;
#pragma unroll
            for (int h = 0; h < SK_KL / 4; ++h) f.x[j][h] = *reinterpret_cast<const uint4*>(src + 4 * h);
        }
    };
    auto compute = [&](const Frags& f) {
        if constexpr (EMU) {
            bf16x8 wh[4], wm[4], wl[4], xh[PT], xm[PT], xl[PT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float wv[8] = {f.w[i][0].x, f.w[i][0].y, f.w[i][0].z, f.w[i][0].w, f.w[i][1].x, f.w[i][1].y, f.w[i][1].z, f.w[i][1].w};
#if defined(NFDP_CONV_ABLATE_SPLIT)      // timing-only build (garbage results): the three pieces without the split arithmetic
                wh[i] = __builtin_bit_cast(bf16x8, f.w[i][0]); wm[i] = __builtin_bit_cast(bf16x8, f.w[i][1]); wl[i] = wh[i];
#else
                split3(wv, wh[i], wm[i], wl[i]);
#endif
            }
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const unsigned k = f.msk[i];
                const float xv[8] = {__builtin_bit_cast(float, f.x[i][0].x & k), __builtin_bit_cast(float, f.x[i][0].y & k),
                                     __builtin_bit_cast(float, f.x[i][0].z & k), __builtin_bit_cast(float, f.x[i][0].w & k),
                                     __builtin_bit_cast(float, f.x[i][1].x & k), __builtin_bit_cast(float, f.x[i][1].y & k),
                                     __builtin_bit_cast(float, f.x[i][1].z & k), __builtin_bit_cast(float, f.x[i][1].w & k)};
#if defined(NFDP_CONV_ABLATE_SPLIT) && NFDP_CONV_ABLATE_SPLIT == 2
                xh[i] = __builtin_bit_cast(bf16x8, f.x[i][0]); xm[i] = __builtin_bit_cast(bf16x8, f.x[i][1]); xl[i] = xh[i];
#else
                split3(xv, xh[i], xm[i], xl[i]);
#endif
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j) {                     // smallest terms first
                    f32x4 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], xh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xl[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[i], xm[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm[i], xh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xm[j], c, 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xh[j], c, 0, 0, 0);
                }
            return;
        }
        float xm_[PT][SK_KL];
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int kk = 0; kk < SK_KL; ++kk)
                xm_[j][kk] = __builtin_bit_cast(float, reinterpret_cast<const unsigned*>(&f.x[j][0])[kk] & f.msk[j]);
#pragma unroll
        for (int kk = 0; kk < SK_KL; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = reinterpret_cast<const float*>(&f.w[i][0])[kk];
#pragma unroll
                for (int j = 0; j < PT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xm_[j][kk], acc[i][j], 0, 0, 0);
            }
        }
    };
    // K stages wave, wave+SK_WAVES, ...: NB register sets, the loads of the next NB - 1 stages fly under the MFMAs of the current one.
    // Every wave runs the SAME number of stages (ceil(nk / SK_WAVES): a stage beyond the end re-reads the last one with its x bits masked
    // to zero) and every load is issued unconditionally: with `if (stage exists) load(...)` hipcc put the loads under exec-masked branches
    // and drained the whole queue (s_waitcnt vmcnt(0)) in front of every stage's MFMAs - no prefetch at all, one L2 round trip per stage,
    // ~20 us per launch whatever the arithmetic (exact chain 20.2, split products 19.6 us at 64 x 16x16 x 64 -> 64).  The stage order
    // per wave is unchanged, a masked stage adds exact zeros: the same sums, bit for bit.
    const int nk = P.Kpad / SK_BK;
    constexpr int NB = SK_NBUF;
    Frags f[NB];
    const int S = (nk + SK_WAVES - 1) / SK_WAVES;
#pragma unroll
    for (int b = 0; b < NB - 1; ++b) load(f[b], wave + b * SK_WAVES);
    for (int s0 = 0; s0 < S; s0 += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (s0 + b < S) {                                       // (uniform)
                load(f[(b + NB - 1) % NB], wave + (s0 + b + NB - 1) * SK_WAVES);
                __builtin_amdgcn_sched_barrier(0);                  // (the prefetch is ISSUED here: hipcc otherwise sinks it to its use)
                compute(f[b]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    splitk_epilogue<SK_WAVES, PT, GN>(P, part, acc, m0, n0, tid, wave, r16, q, mt);
}

#include "conv_ldsx.inc"      // the LDS-staged split-product 3x3 conv
#include "conv_attn.inc"      // 1x1 convs with LayerNorms, attention cores with to_qkv inside

// ----------------------------------------------------------------------------- pointwise conv on bf16 activations
// The 512->512 1x1 convs of the training path (forward and data gradient) and the 512 -> 9*C tap GEMMs read a bf16
// activation tensor whose rows ARE the GEMM's K-contiguous operand rows: no gather, no conversion.  Both operands go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 rows x 128 B per wave instruction, no VGPRs), double
// buffered: the DMA of K tile t+1 is issued before the MFMAs of tile t, one barrier per K tile.  The LDS image is
// lane-linear, so the bank swizzle (16-byte chunk index ^ (row>>1)&7: the 16 rows of a ds_read_b128 lane group land on 16
// distinct slots of the 256-byte bank row) is applied to the per-lane SOURCE address and again on the fragment read.
// Roles are swapped against conv_kernel (A := weights, B := pixels) so a lane ends up with 4 consecutive output
// channels of one pixel: 8- or 16-byte stores and float4 epilogue parameters.
constexpr int PW_BM = 128, PW_BN = 128, PW_BK = 64;
constexpr int PW_TILE_BYTES = PW_BM * PW_BK * 2;           // 16 KiB per operand per stage
constexpr int PW_LDS_BYTES = PW_BM * (PW_BN * 4 + 16);      // the staged fp32 output tile (66 KiB) >= the 4 operand buffers (64 KiB)

// MASK: the epilogue is the scale/ReLU backward of the layer whose OUTPUT gradient this GEMM produces (see below).
template <bool MASK>
__global__ void __launch_bounds__(kThreads, 2) pw_gemm_bf16_kernel(const ConvParams P) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [2 stages][pixels 16 KiB | weights 16 KiB]
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int nt = slot % P.Nt;
    const int mt = (slot / P.Nt) * 8 + xcd;
    if (mt >= P.Mt) return;
    const long m0 = (long)mt * PW_BM;
    const int n0 = nt * PW_BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int r16 = lane & 15, q = lane >> 4;
    const __bf16* __restrict__ xg = reinterpret_cast<const __bf16*>(P.x);
    const __bf16* __restrict__ wg = reinterpret_cast<const __bf16*>(P.w);

    // this lane's source rows for its wave's 4 pieces of each operand tile (piece = 8 rows x 128 B)
    const __bf16* src_x[4];
    const __bf16* src_w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);              // logical 16-byte chunk stored at this lane's slot
        long px = m0 + row;
        if (px >= P.npix) px = P.npix - 1;                         // tail rows: any valid address, results are not stored
        src_x[i] = xg + px * P.x_ld + c * 8;
        src_w[i] = wg + (size_t)(n0 + row) * P.Kpad + c * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * 2 * PW_TILE_BYTES + wave * 4 * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_x[i] + kt * PW_BK),
                                             (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_w[i] + kt * PW_BK),
                                             (__attribute__((address_space(3))) void*)(base + PW_TILE_BYTES + i * 1024), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];                                               // [n tile][pixel tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes inside an operand tile) for k-step 0; k-step 1 flips chunk bit 2
    int off_w[4], off_x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rw = wn * 64 + i * 16 + r16, rx = wm * 64 + i * 16 + r16;
        off_w[i] = rw * 128 + ((q ^ ((rw >> 1) & 7)) << 4);
        off_x[i] = rx * 128 + ((q ^ ((rx >> 1) & 7)) << 4);
    }
    const int nk = P.Kpad / PW_BK;
    // MASK: the saved y this thread's epilogue rows need (8 x 16 bytes) is fetched under the last two K tiles, not after them
    bf16x8 ypre[MASK ? 8 : 1];
    const int e_ch = tid & 15, e_rg = tid >> 4;                   // epilogue role: 8 channels n0 + 8 e_ch .., rows e_rg, e_rg + 16, ...
    // per-channel epilogue parameters, requested before the first tile so their latency is not paid after the last one (a K = 64
    // GEMM is one tile long: its workgroups live a few microseconds)
    float4 addq[MASK ? 1 : 4], mulq[MASK ? 2 : 4];
    if constexpr (MASK) {
        const int n = n0 + e_ch * 8;
        const bool ok = n < P.Cout && P.mul;
        mulq[0] = ok ? *reinterpret_cast<const float4*>(P.mul + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        mulq[1] = ok ? *reinterpret_cast<const float4*>(P.mul + n + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + 4 * q;
            const bool nok = n < P.Cout;
            addq[i] = (P.add && nok) ? *reinterpret_cast<const float4*>(P.add + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            mulq[i] = (P.mul && nok) ? *reinterpret_cast<const float4*>(P.mul + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        if constexpr (MASK) {
            if (kt == (nk >= 2 ? nk - 2 : 0)) {
                const __bf16* yb = reinterpret_cast<const __bf16*>(P.mask_y);
                const int n = n0 + e_ch * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    long px = m0 + e_rg + 16 * e;
                    if (px >= P.npix) px = P.npix - 1;             // clamped: loaded, never used
                    ypre[e] = *reinterpret_cast<const bf16x8*>(yb + px * P.mask_ld + (n < P.Cout ? n : 0));
                }
            }
        }
        const char* xs = smem + cur * 2 * PW_TILE_BYTES;
        const char* ws = xs + PW_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(ws + (off_w[i] ^ (ks << 6)));
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(xs + (off_x[j] ^ (ks << 6)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                           // drains the DMA of tile kt+1 (vmcnt(0)) and frees buffer cur
    }

    // ---- epilogue: D[row = n = 4q + r][col = pixel = r16];  v = (acc + add[n]) * exp(f mul[n]); relu.
    // The finished tile goes through LDS (the stage buffers are free after the loop's last barrier) so that the global stores
    // are 16 bytes per lane along full output rows: stored straight from the accumulators a lane owns 8 bytes of a row, a wave
    // store touches 16 rows, and at K = 64 the kernel was bound by those stores (1.6 TB/s).
    if constexpr (MASK) {
        // Fused scale/ReLU backward (what nfdp_scale_bwd does in a pass of its own): the accumulators are dy, the gradient w.r.t. the
        // OUTPUT y = relu((u + b) e^{f s}) of the previous layer; with the saved y (bf16) this epilogue writes
        // dc = [y > 0] dy e^{f s[n]} (bf16) - the operand of that layer's data- and weight-gradient GEMMs - and the tile's column sums
        // of [y > 0] dy y and of dc (-> ds, db), so dy never goes to memory and y is read once.
        constexpr int pitchF = PW_BN * 4 + 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(smem + (wm * 64 + j * 16 + r16) * pitchF + (wn * 64 + i * 16 + 4 * q) * 4) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        __syncthreads();
        const int ch = tid & 15, rg = tid >> 4;                    // 8 channels n0 + 8 ch .., rows rg, rg + 16, ...
        const int n = n0 + ch * 8;
        const bool nok = n < P.Cout;                               // Cout % 8 == 0 (checked by the host)
        float fac[8], s1[8], s2[8];
        const float mraw[8] = {mulq[0].x, mulq[0].y, mulq[0].z, mulq[0].w, mulq[1].x, mulq[1].y, mulq[1].z, mulq[1].w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            fac[j] = (nok && P.mul) ? expf(__fmul_rn(mraw[j], P.mul_exp)) : 1.0f;
            s1[j] = 0.f; s2[j] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pl = rg + 16 * e;
            const long px = m0 + pl;
            if (px < P.npix && nok) {
                const float4 lo = *reinterpret_cast<const float4*>(smem + pl * pitchF + ch * 32);
                const float4 hi = *reinterpret_cast<const float4*>(smem + pl * pitchF + ch * 32 + 16);
                const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const bf16x8 yv = ypre[e];
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float yj = (float)yv[j];
                    const float g = yj > 0.f ? v[j] : 0.f;
                    s1[j] = fmaf(g, yj, s1[j]);
                    const float d = __fmul_rn(g, fac[j]);
                    s2[j] += d;
                    o[j] = (__bf16)d;
                }
                *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(P.y) + px * P.y_ld + n) = o;
            }
        }
        __syncthreads();                                           // everyone is done with the staged tile: reuse it for the column sums
        float* red = reinterpret_cast<float*>(smem);               // [16 row groups][128 columns][2]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[(rg * PW_BN + ch * 8 + j) * 2] = s1[j];
            red[(rg * PW_BN + ch * 8 + j) * 2 + 1] = s2[j];
        }
        __syncthreads();
        {
            const int col = tid & (PW_BN - 1), which = tid >> 7;   // 256 threads = 128 columns x 2 sums
            if (n0 + col < P.Cout) {
                double acc2 = 0.0;
                for (int g = 0; g < 16; ++g) acc2 += (double)red[(g * PW_BN + col) * 2 + which];      // fixed order
                P.part[((size_t)mt * 2 + which) * P.Cout + n0 + col] = acc2;
            }
        }
        return;
    }
    const int esz = P.y_bf16 ? 2 : 4;
    const int pitch = PW_BN * esz + 16;                            // bytes per staged pixel row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nl = wn * 64 + i * 16 + 4 * q;
        const int n = n0 + nl;
        const bool nok = n < P.Cout;                               // Cout % 4 == 0: the 4 channels are in or out together
        float addv[4] = {0.f, 0.f, 0.f, 0.f}, mulv[4] = {1.f, 1.f, 1.f, 1.f};
        if (P.add && nok) { const float4 t = addq[MASK ? 0 : i]; addv[0] = t.x; addv[1] = t.y; addv[2] = t.z; addv[3] = t.w; }
        if (P.mul && nok) {
            const float4 t = mulq[MASK ? 0 : i];
            mulv[0] = t.x; mulv[1] = t.y; mulv[2] = t.z; mulv[3] = t.w;
            if (P.mul_exp != 0.f) {
#pragma unroll
                for (int r = 0; r < 4; ++r) mulv[r] = expf(__fmul_rn(mulv[r], P.mul_exp));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pl = wm * 64 + j * 16 + r16;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r];
                if (P.add) v[r] = __fadd_rn(v[r], addv[r]);
                if (P.mul) v[r] = __fmul_rn(v[r], mulv[r]);
                if (P.relu) v[r] = fmaxf(v[r], 0.f);
            }
            char* dst = smem + pl * pitch + nl * esz;
            if (P.y_bf16) *reinterpret_cast<uint2*>(dst) = pack_bf16x4(make_float4(v[0], v[1], v[2], v[3]));
            else *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    __syncthreads();
    const int cpr = PW_BN * esz / 16;                              // 16-byte chunks per row: 16 (bf16) or 32 (fp32)
    const int n_el = 16 / esz;                                     // channels per chunk
    for (int idx = tid; idx < PW_BM * cpr; idx += kThreads) {
        const int pl = idx / cpr, ch = idx - pl * cpr;
        const long px = m0 + pl;
        const int n = n0 + ch * n_el;
        if (px >= P.npix || n >= P.Cout) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(smem + pl * pitch + ch * 16);
        char* dst = reinterpret_cast<char*>(P.y) + (px * P.y_ld + n) * esz;
        if (n + n_el <= P.Cout) {
            *reinterpret_cast<uint4*>(dst) = v;
        } else {                                                   // last chunk of a Cout that is a multiple of 4 but not of 8
            *reinterpret_cast<uint2*>(dst) = make_uint2(v.x, v.y);
        }
    }
}

// element (n, k) of the packed matrix of a weight w[Cout][Cin][KK] (see pack_weights_ex_kernel for flip / taps)
__device__ __forceinline__ float packed_value(const float* __restrict__ w, int n, int k, int Cout, int Cin, int KK, int flip, int taps,
                                              int rows, int K) {
    if (n >= rows || k >= K) return 0.f;
    const int Co = flip ? Cin : Cout, Ci = flip ? Cout : Cin;
    int co, ci, tap;
    if (taps) { tap = n / Co; co = n - tap * Co; ci = k; }
    else { tap = k / Ci; ci = k - tap * Ci; co = n; }
    return flip ? w[((long)ci * Cin + co) * KK + (KK - 1 - tap)] : w[((long)co * Cin + ci) * KK + tap];
}
// The split image behind a packed fp32 3x3 weight (conv3x3_ldsx_kernel): blocks (64-row tile nt, 32-k stage kt) of 12 KiB, in a block the
// A fragments (row tile i, piece h / m / l) as register images: lane = r16 + 16 q holds row nt*64 + i*16 + r16, k = kt*32 + 8q .. +7.
// One thread per (block, i, lane): 8 values -> 3 x 16 bytes, with split3 - the arithmetic the gathering kernel does per launch.
__device__ __forceinline__ void split_image_write(const float* __restrict__ w, char* __restrict__ img, long t, int Cout, int Cin, int KK,
                                                  int flip, int taps, int rows, int K, int Kpad) {
    const int lane = (int)(t & 63), i = (int)((t >> 6) & 3);
    const long blk = t >> 8;
    const int nkk = Kpad / 32, kt = (int)(blk % nkk), nt = (int)(blk / nkk);
    const int n = nt * 64 + i * 16 + (lane & 15), k0 = kt * 32 + 8 * (lane >> 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = packed_value(w, n, k0 + e, Cout, Cin, KK, flip, taps, rows, K);
    bf16x8 h, m, l;
    split3(v, h, m, l);
    char* d = img + blk * LX_BLOCK + (3 * i) * 1024 + lane * 16;
    *reinterpret_cast<bf16x8*>(d) = h;
    *reinterpret_cast<bf16x8*>(d + 1024) = m;
    *reinterpret_cast<bf16x8*>(d + 2048) = l;
}
__global__ void split_image_kernel(const float* __restrict__ w, char* __restrict__ img, int Cout, int Cin, int KK, int flip, int taps,
                                   int rows, int K, int Kpad, long total) {
    for (long t = blockIdx.x * (long)kThreads + threadIdx.x; t < total; t += (long)gridDim.x * kThreads)
        split_image_write(w, img, t, Cout, Cin, KK, flip, taps, rows, K, Kpad);
}

// OIHW fp32 -> [Cout_pad][Kpad] in the compute dtype, k = (ky*KW+kx)*Cin+ci, zero padded
template <bool BF16>
__global__ void pack_weights_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout, int Cin, int KH, int KW,
                                    int K, int Kpad, long total) {
    for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int k = (int)(i % Kpad);
        const int n = (int)(i / Kpad);
        float v = 0.f;
        if (n < Cout && k < K) {
            const int tap = k / Cin, ci = k - tap * Cin;
            const int ky = tap / KW, kx = tap - ky * KW;
            v = w[(((long)n * Cin + ci) * KH + ky) * KW + kx];
        }
        if (BF16) reinterpret_cast<__bf16*>(out)[i] = (__bf16)v;
        else reinterpret_cast<float*>(out)[i] = v;
    }
}

// The same packing for the weights the training path derives from a parameter, straight from the parameter (one launch instead of
// flip / permute copies + pack):  flip: the data-gradient weight W'[o'][i'][ky][kx] = w[i'][o'][K-1-ky][K-1-kx] (Cout' = Cin,
// Cin' = Cout);  taps: the pointwise form of a 3x3 conv with few outputs, rows n = tap*Cout' + co', columns ci' (conv3x3_narrow).
template <bool BF16>
__global__ void pack_weights_ex_kernel(const float* __restrict__ w, void* __restrict__ out, int Cout, int Cin, int KK, int flip,
                                       int taps, int rows, int K, int Kpad, long total) {
    const int Co = flip ? Cin : Cout, Ci = flip ? Cout : Cin;     // logical (primed) channel counts
    for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int k = (int)(i % Kpad);
        const int n = (int)(i / Kpad);
        float v = 0.f;
        if (n < rows && k < K) {
            int co, ci, tap;
            if (taps) { tap = n / Co; co = n - tap * Co; ci = k; }
            else { tap = k / Ci; ci = k - tap * Ci; co = n; }
            v = flip ? w[((long)ci * Cin + co) * KK + (KK - 1 - tap)] : w[((long)co * Cin + ci) * KK + tap];
        }
        if (BF16) reinterpret_cast<__bf16*>(out)[i] = (__bf16)v;
        else reinterpret_cast<float*>(out)[i] = v;
    }
}

inline int tile_n_for(int Cout) {
    if (Cout <= 16) return 16;
    if (Cout <= 32) return 32;
    if (Cout <= 48) return 48;
    if (Cout <= 64) return 64;
    return 128;
}
inline int cout_pad_for(int Cout) {
    const int bn = tile_n_for(Cout);
    return (int)(cdiv(Cout, bn) * bn);
}
// K is padded to 64: whole 64-deep stages for pw_gemm_bf16_kernel, and the packed 3x3 weight [Cout_pad][Kpad] (k = tap*Cin + ci)
// is then also the 1x1 weight that acts on nfdp_im2col3x3_bf16's rows
inline int kpad_for(int Cin, int KH, int KW) { return (int)(cdiv((long)Cin * KH * KW, 2 * BK) * 2 * BK); }

template <bool BF16, bool XBF16, bool VEC, int BM, int BN, int WAVES_M, int WAVES_N>
void launch_cfg(ConvParams& P, hipStream_t s) {
    P.Mt = (int)cdiv(P.npix, BM);
    P.Nt = cout_pad_for(P.Cout) / BN;
    const int grid = (int)(cdiv(P.Mt, 8) * 8) * P.Nt;
    const size_t lds = 2 * (size_t)(BM + BN) * (BF16 ? 64 : 128);
    static NfdpPerDevice attr_set;   // > 64 KB of dynamic LDS must be opted into once per kernel
    if (lds > 65536 && !attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&conv_kernel<BF16, XBF16, VEC, BM, BN, WAVES_M, WAVES_N>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if constexpr (!BF16) {
        if (P.emu) {                                     // fp32 products as six bf16 MFMAs (compute = NFDP_F32X6)
            static NfdpPerDevice attr_set_e;
            if (lds > 65536 && !attr_set_e) {
                nfdp_func_attr(reinterpret_cast<const void*>(&conv_kernel<BF16, XBF16, VEC, BM, BN, WAVES_M, WAVES_N, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_set_e = true;
            }
            hipLaunchKernelGGL((conv_kernel<BF16, XBF16, VEC, BM, BN, WAVES_M, WAVES_N, true>), dim3(grid), dim3(kThreads), lds, s, P);
            return;
        }
    }
    hipLaunchKernelGGL((conv_kernel<BF16, XBF16, VEC, BM, BN, WAVES_M, WAVES_N>), dim3(grid), dim3(kThreads), lds, s, P);
}

template <bool BF16, bool XBF16, bool VEC>
void launch_cfg_small(ConvParams& P, hipStream_t s) {
    constexpr int BM = 64, BN = 64;
    P.Mt = (int)cdiv(P.npix, BM);
    P.Nt = P.Cout / BN;                                   // Cout % 64 == 0: the packed rows (padded to 64/128) cover it
    const int grid = (int)(cdiv(P.Mt, 8) * 8) * P.Nt;
    const size_t lds = 2 * (size_t)(BM + BN) * (BF16 ? 64 : 128);
    if constexpr (!BF16) {
        if (P.emu) {
            hipLaunchKernelGGL((conv_kernel<BF16, XBF16, VEC, BM, BN, 2, 2, true>), dim3(grid), dim3(kThreads), lds, s, P);
            return;
        }
    }
    hipLaunchKernelGGL((conv_kernel<BF16, XBF16, VEC, BM, BN, 2, 2>), dim3(grid), dim3(kThreads), lds, s, P);
}

template <bool BF16, bool XBF16, bool VEC>
void launch_by_cout(ConvParams& P, hipStream_t s) {
    // Small problems (U-Net layers at 8x8 / 4x4, B <= 64) would fill only a few dozen CUs with 128-pixel tiles:
    // use 64x64 tiles (4x the workgroups) whenever the big-tile grid is under one wave of the chip.
    if (P.Cout >= 64 && P.Cout % 64 == 0 && cdiv(P.npix, 128) * cdiv(P.Cout, tile_n_for(P.Cout)) < 256) {
        launch_cfg_small<BF16, XBF16, VEC>(P, s);
        return;
    }
    if constexpr (!BF16) {
        // a narrow fp32 output at sampling sizes (the U-Nets' final 1x1 conv, 64 -> 2..16 over 16 k pixels): 256-pixel tiles are 64
        // workgroups on 256 CUs; 64-pixel tiles fill the chip (10.8 -> us per call at B = 64, 16x16)
        if (tile_n_for(P.Cout) == 16 && cdiv(P.npix, 256) < 256) {
            launch_cfg<BF16, XBF16, VEC, 64, 16, 4, 1>(P, s);
            return;
        }
    }
    switch (tile_n_for(P.Cout)) {
        case 16: launch_cfg<BF16, XBF16, VEC, 256, 16, 4, 1>(P, s); break;
        case 32: launch_cfg<BF16, XBF16, VEC, 256, 32, 4, 1>(P, s); break;
        case 48: launch_cfg<BF16, XBF16, VEC, 256, 48, 4, 1>(P, s); break;
        case 64: launch_cfg<BF16, XBF16, VEC, 128, 64, 4, 1>(P, s); break;
        default: launch_cfg<BF16, XBF16, VEC, 128, 128, 2, 2>(P, s); break;
    }
}

}  // namespace


extern "C" size_t nfdp_conv_packed_bytes(int Cout, int Cin, int KH, int KW, int compute) {
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 0;
    const size_t matrix = (size_t)cout_pad_for(Cout) * kpad_for(Cin, KH, KW) * (compute == NFDP_BF16 ? 2 : 4);
    // fp32 3x3 weights with Cin % 32 == 0 also carry their three-way bf16 split as MFMA fragment images (conv3x3_ldsx_kernel)
    return matrix + (compute != NFDP_BF16 && split_image_shape(Cout, Cin, KH, KW) ? split_image_bytes(Cout, kpad_for(Cin, KH, KW)) : 0);
}

// the split image behind the fp32 matrix of a qualifying weight (rows x Ci x 3 x 3 after flip)
static void launch_split_image(const float* w_oihw, void* w_packed, int Cout, int Cin, int KK, int flip, int rows, int Ci, hipStream_t s) {
    const int K = Ci * KK, Kpad = KK == 9 ? kpad_for(Ci, 3, 3) : kpad_for(Ci, 1, 1);
    char* img = reinterpret_cast<char*>(w_packed) + (size_t)cout_pad_for(rows) * Kpad * 4;
    const long total = (long)cdiv(rows, 64) * (Kpad / 32) * 256;
    hipLaunchKernelGGL(split_image_kernel, dim3(stream_grid(total, kThreads)), dim3(kThreads), 0, s, w_oihw, img, Cout, Cin, KK, flip, 0,
                       rows, K, Kpad, total);
}

extern "C" int nfdp_conv_pack_weights(const float* w_oihw, void* w_packed, int Cout, int Cin, int KH, int KW, int compute,
                                      nfdp_stream_t stream) {
    NFDP_REQUIRE(w_oihw && w_packed, NFDP_EINVAL, "conv_pack_weights: null pointer");
    NFDP_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0, NFDP_ESHAPE, "conv_pack_weights: bad shape");
    NFDP_REQUIRE(compute == NFDP_F32 || compute == NFDP_BF16, NFDP_EINVAL, "conv_pack_weights: compute=%d", compute);
    const int K = Cin * KH * KW, Kpad = kpad_for(Cin, KH, KW);
    const long total = (long)cout_pad_for(Cout) * Kpad;
    const int grid = stream_grid(total, kThreads);
    if (compute == NFDP_BF16)
        hipLaunchKernelGGL((pack_weights_kernel<true>), dim3(grid), dim3(kThreads), 0, as_stream(stream), w_oihw, w_packed,
                           Cout, Cin, KH, KW, K, Kpad, total);
    else
        hipLaunchKernelGGL((pack_weights_kernel<false>), dim3(grid), dim3(kThreads), 0, as_stream(stream), w_oihw, w_packed,
                           Cout, Cin, KH, KW, K, Kpad, total);
    if (compute == NFDP_F32 && split_image_shape(Cout, Cin, KH, KW))
        launch_split_image(w_oihw, w_packed, Cout, Cin, KH * KW, 0, Cout, Cin, as_stream(stream));
    NFDP_CHECK_LAUNCH("conv_pack_weights");
    return NFDP_OK;
}

extern "C" size_t nfdp_conv_packed_bytes_ex(int Cout, int Cin, int KH, int KW, int compute, int flip, int taps) {
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 0;
    const int Co = flip ? Cin : Cout, Ci = flip ? Cout : Cin;
    return taps ? nfdp_conv_packed_bytes(KH * KW * Co, Ci, 1, 1, compute) : nfdp_conv_packed_bytes(Co, Ci, KH, KW, compute);
}

extern "C" int nfdp_conv_pack_weights_ex(const float* w_oihw, void* w_packed, int Cout, int Cin, int KH, int KW, int compute, int flip,
                                         int taps, nfdp_stream_t stream) {
    NFDP_REQUIRE(w_oihw && w_packed, NFDP_EINVAL, "conv_pack_weights_ex: null pointer");
    NFDP_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW == KH, NFDP_ESHAPE, "conv_pack_weights_ex: bad shape");
    const int Co = flip ? Cin : Cout, Ci = flip ? Cout : Cin, KK = KH * KW;
    const int rows = taps ? KK * Co : Co;
    const int K = taps ? Ci : Ci * KK;
    const int Kpad = taps ? kpad_for(Ci, 1, 1) : kpad_for(Ci, KH, KW);
    const long total = (long)cout_pad_for(rows) * Kpad;
    const int grid = stream_grid(total, kThreads);
    if (compute == NFDP_BF16)
        hipLaunchKernelGGL((pack_weights_ex_kernel<true>), dim3(grid), dim3(kThreads), 0, as_stream(stream), w_oihw, w_packed, Cout, Cin,
                           KK, flip, taps, rows, K, Kpad, total);
    else
        hipLaunchKernelGGL((pack_weights_ex_kernel<false>), dim3(grid), dim3(kThreads), 0, as_stream(stream), w_oihw, w_packed, Cout, Cin,
                           KK, flip, taps, rows, K, Kpad, total);
    if (compute == NFDP_F32 && !taps && split_image_shape(Co, Ci, KH, KW))
        launch_split_image(w_oihw, w_packed, Cout, Cin, KK, flip, Co, Ci, as_stream(stream));
    NFDP_CHECK_LAUNCH("conv_pack_weights_ex");
    return NFDP_OK;
}

// ---- batched packing: every derived weight layout of a training step (3 forward + 3 backward forms per coupling network, ~290
// launches of ~4.7 us for Glow L3 K16) in ONE launch.  A job is NFDP_PACK_JOB_WORDS int64 words in device memory:
// {w, w_packed, Cout, Cin, KK, flip, taps, rows, K, Kpad, total, 0}, filled on the host by nfdp_conv_pack_job.
constexpr int kPackJobWords = 12;

template <bool BF16>
__global__ void pack_weights_batch_kernel(const long* __restrict__ jobs) {
    const long* j = jobs + (long)blockIdx.y * kPackJobWords;
    const float* __restrict__ w = reinterpret_cast<const float*>(j[0]);
    void* out = reinterpret_cast<void*>(j[1]);
    const int Cout = (int)j[2], Cin = (int)j[3], KK = (int)j[4], flip = (int)j[5], taps = (int)j[6], rows = (int)j[7], K = (int)j[8],
              Kpad = (int)j[9];
    const long total = j[10];
    const int Co = flip ? Cin : Cout, Ci = flip ? Cout : Cin;
    for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        const int k = (int)(i % Kpad);
        const int n = (int)(i / Kpad);
        float v = 0.f;
        if (n < rows && k < K) {
            int co, ci, tap;
            if (taps) { tap = n / Co; co = n - tap * Co; ci = k; }
            else { tap = k / Ci; ci = k - tap * Ci; co = n; }
            v = flip ? w[((long)ci * Cin + co) * KK + (KK - 1 - tap)] : w[((long)co * Cin + ci) * KK + tap];
        }
        if (BF16) reinterpret_cast<__bf16*>(out)[i] = (__bf16)v;
        else reinterpret_cast<float*>(out)[i] = v;
    }
    if (!BF16 && j[11]) {                                           // a qualifying 3x3 weight: its split image behind the matrix
        char* img = reinterpret_cast<char*>(out) + total * 4;
        const long nthreads = (long)((rows + 63) / 64) * (Kpad / 32) * 256;
        for (long t = blockIdx.x * (long)kThreads + threadIdx.x; t < nthreads; t += (long)gridDim.x * kThreads)
            split_image_write(w, img, t, Cout, Cin, KK, flip, taps, rows, K, Kpad);
    }
}

extern "C" int nfdp_conv_pack_job(const float* w_oihw, void* w_packed, int Cout, int Cin, int KH, int KW, int flip, int taps, long* job) {
    NFDP_REQUIRE(w_oihw && w_packed && job, NFDP_EINVAL, "conv_pack_job: null pointer");
    NFDP_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW == KH, NFDP_ESHAPE, "conv_pack_job: bad shape");
    const int Co = flip ? Cin : Cout, Ci = flip ? Cout : Cin, KK = KH * KW;
    const int rows = taps ? KK * Co : Co;
    const int K = taps ? Ci : Ci * KK;
    const int Kpad = taps ? kpad_for(Ci, 1, 1) : kpad_for(Ci, KH, KW);
    job[0] = (long)(uintptr_t)w_oihw; job[1] = (long)(uintptr_t)w_packed;
    job[2] = Cout; job[3] = Cin; job[4] = KK; job[5] = flip ? 1 : 0; job[6] = taps ? 1 : 0; job[7] = rows; job[8] = K; job[9] = Kpad;
    job[10] = (long)cout_pad_for(rows) * Kpad;
    job[11] = (!taps && split_image_shape(Co, Ci, KH, KW)) ? 1 : 0;      // (read by the fp32 batch kernel only)
    return NFDP_OK;
}

extern "C" int nfdp_conv_pack_weights_batch(const long* jobs_dev, int n, long max_total, int compute, nfdp_stream_t stream) {
    NFDP_REQUIRE(jobs_dev && n > 0 && max_total > 0, NFDP_EINVAL, "conv_pack_weights_batch: no jobs");
    NFDP_REQUIRE(compute == NFDP_F32 || compute == NFDP_BF16, NFDP_EINVAL, "conv_pack_weights_batch: compute=%d", compute);
    long gx = cdiv(max_total, (long)kThreads * 8);                 // ~8 elements per thread of the largest job
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    if (compute == NFDP_BF16)
        hipLaunchKernelGGL((pack_weights_batch_kernel<true>), dim3((int)gx, n), dim3(kThreads), 0, as_stream(stream), jobs_dev);
    else
        hipLaunchKernelGGL((pack_weights_batch_kernel<false>), dim3((int)gx, n), dim3(kThreads), 0, as_stream(stream), jobs_dev);
    NFDP_CHECK_LAUNCH("conv_pack_weights_batch");
    return NFDP_OK;
}

void nfdp_internal_scale_bwd_finish(const double* part, int nblk, int C, float f, float* dmul, float* dadd, int acc, hipStream_t s);   // backward.hip

extern "C" int nfdp_pointwise_dgrad_masked_supported(long npix, int Cin, int Cout) {
    return Cin % PW_BK == 0 && Cout > 64 && Cout % 8 == 0 && npix >= 4 * PW_BM;
}

extern "C" size_t nfdp_pointwise_dgrad_masked_ws_bytes(long npix, int Cout) {
    return (size_t)cdiv(npix, PW_BM) * 2 * Cout * sizeof(double);
}

extern "C" int nfdp_pointwise_dgrad_masked_bf16_ex(const void* x, int x_ld, const void* w_packed, const void* y_saved, int y_ld,
                                                   const float* mul, float f, long npix, int Cin, int Cout, void* dc, int dc_ld,
                                                   float* dmul, float* dadd, void* ws, int accumulate, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && w_packed && y_saved && dc && dmul && dadd && ws, NFDP_EINVAL, "pointwise_dgrad_masked: null pointer");
    NFDP_REQUIRE(nfdp_pointwise_dgrad_masked_supported(npix, Cin, Cout), NFDP_EUNSUPPORTED,
                 "pointwise_dgrad_masked: npix=%ld Cin=%d Cout=%d (Cin %% 64, Cout %% 8, npix >= 512)", npix, Cin, Cout);
    NFDP_REQUIRE(x_ld >= Cin && x_ld % 8 == 0 && y_ld >= Cout && y_ld % 8 == 0 && dc_ld >= Cout && dc_ld % 8 == 0 &&
                     (((uintptr_t)x | (uintptr_t)y_saved | (uintptr_t)dc) & 15) == 0, NFDP_ESHAPE,
                 "pointwise_dgrad_masked: bf16 rows must be 16-byte aligned with strides that are multiples of 8");
    ConvParams P;
    P.x = x; P.w = w_packed; P.y = dc; P.add = nullptr; P.mul = mul; P.npix = npix;
    P.x_ld = x_ld; P.y_ld = dc_ld; P.y_bf16 = 1;
    P.H = 1; P.W = 1; P.Cin = Cin; P.Cout = Cout; P.KH = 1; P.KW = 1; P.pad = 0;
    P.K = Cin; P.Kpad = kpad_for(Cin, 1, 1); P.relu = 1; P.mul_exp = f;
    P.mask_y = y_saved; P.mask_ld = y_ld; P.part = reinterpret_cast<double*>(ws);
    P.Mt = (int)cdiv(npix, PW_BM);
    P.Nt = cout_pad_for(Cout) / PW_BN;
    const int grid = (int)(cdiv(P.Mt, 8) * 8) * P.Nt;
    hipStream_t s = as_stream(stream);
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&pw_gemm_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  PW_LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL(pw_gemm_bf16_kernel<true>, dim3(grid), dim3(kThreads), PW_LDS_BYTES, s, P);
    nfdp_internal_scale_bwd_finish(P.part, P.Mt, Cout, f, dmul, dadd, accumulate, s);
    NFDP_CHECK_LAUNCH("pointwise_dgrad_masked");
    return NFDP_OK;
}

extern "C" int nfdp_pointwise_dgrad_masked_bf16(const void* x, int x_ld, const void* w_packed, const void* y_saved, int y_ld,
                                                const float* mul, float f, long npix, int Cin, int Cout, void* dc, int dc_ld, float* dmul,
                                                float* dadd, void* ws, nfdp_stream_t stream) {
    return nfdp_pointwise_dgrad_masked_bf16_ex(x, x_ld, w_packed, y_saved, y_ld, mul, f, npix, Cin, Cout, dc, dc_ld, dmul, dadd, ws, 0,
                                               stream);
}

// ---- the LDS-staged form of the split-product 3x3 convs (conv3x3_ldsx_kernel): taken where its geometry holds
static thread_local int g_conv_ldsx = 1;                          // test hook (thread-local): 0 = always the gathering kernel, 2 = flat tiles only, 3 = no 128-pixel tiles
extern "C" void nfdp_debug_set_conv_ldsx(int v) { g_conv_ldsx = v; }
constexpr size_t kLdsxMax = 160 * 1024;
#ifdef NFDP_LX_STAMPS
static thread_local void* g_lx_stamps = nullptr;
extern "C" void nfdp_debug_set_conv_stamps(void* p) { g_lx_stamps = p; }
#endif
constexpr size_t kLdsxStatic = 1024;                              // the epilogue's group-sum scratch (splitk_epilogue: gsw)
static bool ldsx_ok(const ConvParams& P, int pt) {
    return g_conv_ldsx && P.KH == 3 && split_image_shape(P.Cout, P.Cin, P.KH, P.KW) && (P.x2 == nullptr || P.Cin1 % 4 == 0) &&
           ldsx_np(P.W, pt) <= 1024 &&                 // (images up to ~500 pixels wide: beyond, the halo dwarfs the tile)
           ldsx_bytes(P.W, P.Cin, pt, P.H * P.W, P.gi_stats ? P.gi_groups : 0) + kLdsxStatic <= kLdsxMax;
}
template <bool GN, bool GNIN, bool RES, bool UP = false>
static void launch_ldsx_t(const ConvParams& P, int pt, dim3 grid, hipStream_t s);
template <bool GN>
static void launch_ldsx(const ConvParams& P, int pt, dim3 grid, hipStream_t s) {
    if (P.w2) launch_ldsx_t<GN, false, true>(P, pt, grid, s);         // (host: never together with gi_stats / up2)
    else if (P.gi_stats) launch_ldsx_t<GN, true, false>(P, pt, grid, s);
    else if (!GN && P.up2) launch_ldsx_t<false, false, false, true>(P, pt, grid, s);
    else launch_ldsx_t<GN, false, false>(P, pt, grid, s);
}
template <bool GN, bool GNIN, bool RES, bool UP>
static void launch_ldsx_t(const ConvParams& P, int pt, dim3 grid, hipStream_t s) {
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&conv3x3_ldsx_kernel<4, GN, GNIN, RES, UP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsxMax - kLdsxStatic));
        nfdp_func_attr(reinterpret_cast<const void*>(&conv3x3_ldsx_kernel<2, GN, GNIN, RES, UP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsxMax - kLdsxStatic));
        nfdp_func_attr(reinterpret_cast<const void*>(&conv3x3_ldsx_kernel<1, GN, GNIN, RES, UP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsxMax - kLdsxStatic));
        attr_set = true;
    }
    size_t lds = ldsx_bytes(P.W, P.Cin, pt, P.H * P.W, GNIN ? P.gi_groups : 0);
    ConvParams Q = P;
    Q.wsplit_off = (size_t)cout_pad_for(P.Cout) * P.Kpad * 4;
    if (RES) Q.w2split_off = (size_t)cout_pad_for(P.Cout) * kpad_for(P.Cin, 1, 1) * 4;
    // wide images: 8 x 8 pixel blocks (100 staged pixels) instead of flat 64-pixel ranges (64 + 2 (W + 1))
    if constexpr (!GN && !GNIN && !RES) {
        // ... and 16 x 8 blocks (128 pixels per workgroup: a stage's twelve weight fragments meet twice the MFMAs; the accumulators live in
        // AGPRs, no spills) where that still leaves two chip-wide waves of workgroups: 79 -> 71 us for the 64 -> 64 conv at 32 x 64 x 64
        if (pt == 4 && g_conv_ldsx == 1 && P.W >= 32 && P.W % 8 == 0 && P.H % 16 == 0 && (P.npix / 128) * P.Nt >= 512) {
            Q.tw = 8; Q.th = 16; Q.Mt = (int)(P.npix / 128);
            const size_t planes = (size_t)3 * (18 * 10 + 1) * (P.Cin + 8) * 2;
            const size_t part = (size_t)LX_WAVES * 128 * SK_LD * 4;
            lds = planes > part ? planes : part;
            if (lds + kLdsxStatic <= kLdsxMax) {
                static NfdpPerDevice attr8;
                if (!attr8) {
                    nfdp_func_attr(reinterpret_cast<const void*>(&conv3x3_ldsx_kernel<8, false, false, false, UP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsxMax - kLdsxStatic));
                    attr8 = true;
                }
#ifdef NFDP_LX_STAMPS
                Q.part = reinterpret_cast<double*>(g_lx_stamps);
#endif
                hipLaunchKernelGGL((conv3x3_ldsx_kernel<8, false, false, false, UP, true>), dim3(Q.Mt * Q.Nt), dim3(LX_WAVES * 64), lds, s, Q);
                return;
            }
            Q.Mt = P.Mt;
            lds = ldsx_bytes(P.W, P.Cin, pt, P.H * P.W, 0);
        }
    }
    if constexpr (!GN) {
        if (pt == 4 && g_conv_ldsx != 2 && P.W >= 32 && P.W % 8 == 0 && P.H % 8 == 0) {
            Q.tw = 8; Q.th = 8;
            const size_t planes = (size_t)3 * (10 * 10 + 1) * (P.Cin + 8) * 2 + (GNIN ? ldsx_gi_bytes(P.W, P.H * P.W, P.Cin, P.gi_groups, pt) : 0);
            const size_t part = (size_t)LX_WAVES * 64 * SK_LD * 4;
            lds = planes > part ? planes : part;
            static NfdpPerDevice attr2d;
            if (!attr2d) {
                nfdp_func_attr(reinterpret_cast<const void*>(&conv3x3_ldsx_kernel<4, false, GNIN, RES, UP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kLdsxMax - kLdsxStatic));
                attr2d = true;
            }
#ifdef NFDP_LX_STAMPS
            Q.part = reinterpret_cast<double*>(g_lx_stamps);
#endif
            hipLaunchKernelGGL((conv3x3_ldsx_kernel<4, false, GNIN, RES, UP, true>), grid, dim3(LX_WAVES * 64), lds, s, Q);
            return;
        }
    }
#ifdef NFDP_LX_STAMPS
    Q.part = reinterpret_cast<double*>(g_lx_stamps);
#endif
    if (pt == 4) hipLaunchKernelGGL((conv3x3_ldsx_kernel<4, GN, GNIN, RES, UP>), grid, dim3(LX_WAVES * 64), lds, s, Q);
    else if (pt == 2) hipLaunchKernelGGL((conv3x3_ldsx_kernel<2, GN, GNIN, RES, UP>), grid, dim3(LX_WAVES * 64), lds, s, Q);
    else hipLaunchKernelGGL((conv3x3_ldsx_kernel<1, GN, GNIN, RES, UP>), grid, dim3(LX_WAVES * 64), lds, s, Q);
}

// ---- a ResnetBlock's first GroupNorm carried from block1's conv to block2's conv (ConvParams::gs_* / gi_*)
struct ConvExtra {
    double* gs_out = nullptr;         // producer: per (tile, group) sums of the outputs and their squares
    int gs_groups = 0;
    const double* gi_stats = nullptr; // consumer: the producer's sums ...
    const float* gi_gamma = nullptr;  // ... and the norm's parameters
    const float* gi_beta = nullptr;
    const float* gi_ss = nullptr;
    int gi_ss_ld = 0, gi_groups = 0, gi_parts = 0;
    float gi_eps = 0.f;
    const void* w2 = nullptr;         // res_conv in the same launch (ConvParams::w2 ...)
    const float* add2 = nullptr;
    float* y2 = nullptr;
    int y2_ld = 0, up2 = 0;
};
static void apply_extra(ConvParams& P, const ConvExtra* ex) {
    if (!ex) return;
    P.w2 = ex->w2; P.add2 = ex->add2; P.y2 = ex->y2; P.y2_ld = ex->y2_ld; P.up2 = ex->up2;
    P.gs_out = ex->gs_out; P.gs_groups = ex->gs_groups;
    P.gi_stats = ex->gi_stats; P.gi_gamma = ex->gi_gamma; P.gi_beta = ex->gi_beta; P.gi_ss = ex->gi_ss;
    P.gi_ss_ld = ex->gi_ss_ld; P.gi_groups = ex->gi_groups; P.gi_parts = ex->gi_parts; P.gi_eps = ex->gi_eps;
}
static inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
// Beyond 1024 tiles of 64 x 64 the gathering split-K kernel loses to the implicit-GEMM kernel - but the LDS-staged 3x3 kernel does not (the
// 64 -> 64 convs of a 32 x 6 x 64 x 64 latent part: 114.5 us on conv_kernel, 9.7 GFLOP at 20 % of the bf16 peak; the same FLOPs at 32 x 32
// take 42.7 us staged): large problems stay in the split-K family when that kernel takes them (W, Cin: its 64-pixel tile with halo in LDS).
static inline bool big_ldsx_shape(int W, int Cin, int Cout, int K) {
    return K == 3 && g_conv_ldsx && split_image_shape(Cout, Cin, 3, 3) && ldsx_np(W, 4) <= 1024 && ldsx_bytes(W, Cin, 4) + kLdsxStatic <= kLdsxMax;
}
// pixels per workgroup of the plain split-K conv for this problem (conv2d_impl's rule), 0 when it takes another kernel
static int splitk_tile_pixels(long npix, int Cin, int Cout, int K, int W) {
    if (Cin % 8 != 0 || Cout < 64 || Cout % 4 != 0 || kpad_for(Cin, K, K) < 8 * 32) return 0;
    if (cdiv(npix, 64) * cdiv(Cout, 64) > 1024 && !big_ldsx_shape(W, Cin, Cout, K)) return 0;
    const long t64 = cdiv(npix, 64) * cdiv(Cout, 64);
    return 16 * (t64 >= 256 ? 4 : (2 * t64 >= 256 ? 2 : 1));
}

// ---- split-K conv with the Block epilogue (GroupNorm -> scale/shift -> SiLU -> + residual) in the same launch
static int gn_pixel_tiles(long npix, int Cout, int HW) {
    const long t64 = cdiv(npix, 64) * cdiv(Cout, 64);
    const int pt = t64 >= 256 ? 4 : (2 * t64 >= 256 ? 2 : 1);  // the tile the plain conv would take for this grid
    // only when that tile already holds whole images: a larger tile would leave CUs without a workgroup and lengthen every wave's MFMA
    // chain - measured: forcing 64-pixel tiles on the 8x8 stages of a B = 64 step cost more than the GroupNorm launches it saved
    return (16 * pt) % HW == 0 ? pt : 0;
}

extern "C" int nfdp_conv2d_gn_silu_supported(int B, int H, int W, int Cin, int Cout, int K, int groups) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || groups <= 0 || (K % 2) != 1 || K > 7) return 0;
    if (Cin % 8 != 0 || Cout % 64 != 0 || Cout % groups != 0) return 0;
    const int cpg = Cout / groups, HW = H * W;
    if (cpg % 4 != 0 || 64 % cpg != 0 || HW > 64 || 64 % HW != 0) return 0;
    const long npix = (long)B * HW;
    if (kpad_for(Cin, K, K) < 8 * 32 || cdiv(npix, 64) * cdiv(Cout, 64) > 1024) return 0;       // the split-K kernel's own range
    return gn_pixel_tiles(npix, Cout, HW) != 0;
}

static int conv2d_gn_silu_impl(const float* x, int x_ld, const void* w_packed, float* y, int y_ld, int B, int H, int W, int Cin,
                               int Cout, int K, const float* bias, int groups, const float* gamma, const float* beta,
                               const float* scale_shift, int ss_ld, float eps, const float* residual, int res_ld, int compute,
                               nfdp_stream_t stream, const float* x2, int x2_ld, int cin1, const ConvExtra* ex = nullptr);

extern "C" int nfdp_conv2d_gn_silu_nhwc_f32(const float* x, int x_ld, const void* w_packed, float* y, int y_ld, int B, int H, int W, int Cin,
                                            int Cout, int K, const float* bias, int groups, const float* gamma, const float* beta,
                                            const float* scale_shift, int ss_ld, float eps, const float* residual, int res_ld,
                                            int compute, nfdp_stream_t stream) {
    return conv2d_gn_silu_impl(x, x_ld, w_packed, y, y_ld, B, H, W, Cin, Cout, K, bias, groups, gamma, beta, scale_shift, ss_ld, eps,
                               residual, res_ld, compute, stream, nullptr, 0, 0);
}

// the same on torch.cat((x1, x2), dim=1) without the concatenated tensor (see nfdp_conv2d_cat_nhwc_f32)
extern "C" int nfdp_conv2d_gn_silu_cat_nhwc_f32(const float* x1, int x1_ld, int Cin1, const float* x2, int x2_ld, int Cin2,
                                                const void* w_packed, float* y, int y_ld, int B, int H, int W, int Cout, int K,
                                                const float* bias, int groups, const float* gamma, const float* beta,
                                                const float* scale_shift, int ss_ld, float eps, const float* residual, int res_ld,
                                                int compute, nfdp_stream_t stream) {
    NFDP_REQUIRE(x1 && x2, NFDP_EINVAL, "conv2d_gn_silu_cat: null pointer");
    NFDP_REQUIRE(Cin1 > 0 && Cin2 > 0 && Cin1 % 8 == 0 && Cin2 % 8 == 0, NFDP_EUNSUPPORTED,
                 "conv2d_gn_silu_cat: channel counts %d + %d (multiples of 8)", Cin1, Cin2);
    NFDP_REQUIRE(x2_ld >= Cin2 && x2_ld % 4 == 0 && ((uintptr_t)x2 % 16) == 0, NFDP_ESHAPE, "conv2d_gn_silu_cat: rows must be 16-byte aligned");
    return conv2d_gn_silu_impl(x1, x1_ld, w_packed, y, y_ld, B, H, W, Cin1 + Cin2, Cout, K, bias, groups, gamma, beta, scale_shift, ss_ld,
                               eps, residual, res_ld, compute, stream, x2, x2_ld, Cin1);
}

static int conv2d_gn_silu_impl(const float* x, int x_ld, const void* w_packed, float* y, int y_ld, int B, int H, int W, int Cin,
                               int Cout, int K, const float* bias, int groups, const float* gamma, const float* beta,
                               const float* scale_shift, int ss_ld, float eps, const float* residual, int res_ld, int compute,
                               nfdp_stream_t stream, const float* x2, int x2_ld, int cin1, const ConvExtra* ex) {
    NFDP_REQUIRE(x && w_packed && y && gamma && beta, NFDP_EINVAL, "conv2d_gn_silu: null pointer");
    NFDP_REQUIRE(compute == NFDP_F32 || compute == NFDP_F32X6, NFDP_EINVAL, "conv2d_gn_silu: compute=%d (fp32 or the six-product split form)", compute);
    NFDP_REQUIRE(nfdp_conv2d_gn_silu_supported(B, H, W, Cin, Cout, K, groups), NFDP_EUNSUPPORTED,
                 "conv2d_gn_silu: unsupported geometry B=%d H=%d W=%d Cin=%d Cout=%d K=%d groups=%d", B, H, W, Cin, Cout, K, groups);
    NFDP_REQUIRE(x_ld >= (x2 ? cin1 : Cin) && x_ld % 4 == 0 && y_ld >= Cout && y_ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) % 16) == 0,
                 NFDP_ESHAPE, "conv2d_gn_silu: rows must be 16-byte aligned");
    NFDP_REQUIRE((bias == nullptr || (uintptr_t)bias % 16 == 0) && ((uintptr_t)gamma % 16) == 0 && ((uintptr_t)beta % 16) == 0, NFDP_ESHAPE,
                 "conv2d_gn_silu: parameter vectors must be 16-byte aligned");
    NFDP_REQUIRE(scale_shift == nullptr || ss_ld >= 2 * Cout, NFDP_ESHAPE, "conv2d_gn_silu: scale_shift rows shorter than 2*Cout");
    NFDP_REQUIRE(residual == nullptr || (res_ld >= Cout && res_ld % 4 == 0 && ((uintptr_t)residual % 16) == 0), NFDP_ESHAPE,
                 "conv2d_gn_silu: residual rows must be 16-byte aligned");
    const long npix = (long)B * H * W;
    ConvParams P;
    P.x = x; P.w = w_packed; P.y = y; P.add = bias; P.mul = nullptr; P.npix = npix;
    P.x_ld = x_ld; P.y_ld = y_ld; P.y_bf16 = 0;
    P.H = H; P.W = W; P.Cin = Cin; P.Cout = Cout; P.KH = K; P.KW = K; P.pad = (K - 1) / 2;
    P.K = Cin * K * K; P.Kpad = kpad_for(Cin, K, K); P.relu = 0; P.mul_exp = 0.f;
    P.gn_gamma = gamma; P.gn_beta = beta; P.gn_ss = scale_shift; P.gn_ss_ld = ss_ld; P.gn_res = residual; P.gn_res_ld = res_ld;
    P.gn_groups = groups; P.gn_eps = eps;
    if (x2) { P.x2 = x2; P.x2_ld = x2_ld; P.Cin1 = cin1; }
    apply_extra(P, ex);
    P.Nt = (int)cdiv(Cout, 64);
    const int pt = gn_pixel_tiles(npix, Cout, H * W);
    NFDP_REQUIRE((!P.gi_stats && !P.w2) || (compute == NFDP_F32X6 && ldsx_ok(P, pt)), NFDP_EUNSUPPORTED,
                 "conv2d_gnin / conv2d_res: available in the LDS-staged 3x3 conv only (nfdp_conv2d_gnin_supported / nfdp_conv2d_res_supported)");
    P.Mt = (int)cdiv(npix, 16 * pt);
    const dim3 grid(P.Mt * P.Nt);
    hipStream_t s = as_stream(stream);
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&conv_splitk_f32_kernel<8, 4, false, 4, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 64 * SK_LD * 4);
        nfdp_func_attr(reinterpret_cast<const void*>(&conv_splitk_f32_kernel<4, 8, true, 4, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * SK_LD * 4);
        attr_set = true;
    }
    if (compute == NFDP_F32X6 && ldsx_ok(P, pt)) {
        launch_ldsx<true>(P, pt, grid, s);
    } else if (compute == NFDP_F32X6) {
        if (pt == 4) hipLaunchKernelGGL((conv_splitk_f32_kernel<4, 8, true, 4, true>), grid, dim3(256), 4 * 64 * SK_LD * 4, s, P);
        else if (pt == 2) hipLaunchKernelGGL((conv_splitk_f32_kernel<4, 8, true, 2, true>), grid, dim3(256), 4 * 32 * SK_LD * 4, s, P);
        else hipLaunchKernelGGL((conv_splitk_f32_kernel<4, 8, true, 1, true>), grid, dim3(256), 4 * 16 * SK_LD * 4, s, P);
    } else {
        if (pt == 4) hipLaunchKernelGGL((conv_splitk_f32_kernel<8, 4, false, 4, true>), grid, dim3(512), 8 * 64 * SK_LD * 4, s, P);
        else if (pt == 2) hipLaunchKernelGGL((conv_splitk_f32_kernel<8, 4, false, 2, true>), grid, dim3(512), 8 * 32 * SK_LD * 4, s, P);
        else hipLaunchKernelGGL((conv_splitk_f32_kernel<8, 4, false, 1, true>), grid, dim3(512), 8 * 16 * SK_LD * 4, s, P);
    }
    NFDP_CHECK_LAUNCH("conv2d_gn_silu");
    return NFDP_OK;
}

static int conv2d_impl(const void* x, int x_dtype, int x_ld, const void* w_packed, void* y, int y_dtype, int y_ld,
                       int B, int H, int W, int Cin, int Cout, int KH, int KW, const float* add, const float* mul,
                       float mul_exp, int relu, int compute, nfdp_stream_t stream, const void* x2, int x2_ld, int cin1,
                       const ConvExtra* ex = nullptr);

extern "C" int nfdp_conv2d_nhwc(const void* x, int x_dtype, int x_ld, const void* w_packed, void* y, int y_dtype, int y_ld,
                                int B, int H, int W, int Cin, int Cout, int KH, int KW, const float* add, const float* mul,
                                float mul_exp, int relu, int compute, nfdp_stream_t stream) {
    return conv2d_impl(x, x_dtype, x_ld, w_packed, y, y_dtype, y_ld, B, H, W, Cin, Cout, KH, KW, add, mul, mul_exp, relu, compute, stream,
                       nullptr, 0, 0);
}

// conv2d(torch.cat((x1, x2), dim=1)) without the concatenated tensor (the U-Net's skip connections, reference
// gaussian_diffusion.py:479-486): exact fp32, both inputs fp32 pixel-strided, Cin1 % 8 == 0.
extern "C" int nfdp_conv2d_cat_nhwc_f32(const float* x1, int x1_ld, int Cin1, const float* x2, int x2_ld, int Cin2, const void* w_packed,
                                        float* y, int y_ld, int B, int H, int W, int Cout, int K, const float* add, int compute,
                                        nfdp_stream_t stream) {
    NFDP_REQUIRE(x1 && x2, NFDP_EINVAL, "conv2d_cat: null pointer");
    NFDP_REQUIRE(compute == NFDP_F32 || compute == NFDP_F32X6, NFDP_EINVAL, "conv2d_cat: compute=%d (fp32 or the six-product split form)", compute);
    NFDP_REQUIRE(Cin1 > 0 && Cin2 > 0 && Cin1 % 8 == 0 && Cin2 % 4 == 0, NFDP_EUNSUPPORTED,
                 "conv2d_cat: channel counts %d + %d (first a multiple of 8, second of 4)", Cin1, Cin2);
    NFDP_REQUIRE(x1_ld >= Cin1 && x2_ld >= Cin2 && x1_ld % 4 == 0 && x2_ld % 4 == 0 && (((uintptr_t)x1 | (uintptr_t)x2) % 16) == 0,
                 NFDP_ESHAPE, "conv2d_cat: input rows must be 16-byte aligned");
    return conv2d_impl(x1, NFDP_F32, x1_ld, w_packed, y, NFDP_F32, y_ld, B, H, W, Cin1 + Cin2, Cout, K, K, add, nullptr, 0.f, 0, compute,
                       stream, x2, x2_ld, Cin1);
}

// ---- ResnetBlock (reference gaussian_diffusion.py:279-302): block1's GroupNorm -> scale/shift -> SiLU inside block2's conv.  A GroupNorm
// needs a whole (sample, group) before its first output; where a conv's workgroup tile is PART of an image (the 16x16 / 8x8 stages at
// sampling batch sizes) the norm was a launch of its own between the two convs.  Now block1's conv also leaves the fp64 sums of every
// (tile, group) (nfdp_conv2d_stats_nhwc_f32), and block2's conv - which stages its input tile in LDS anyway - adds them up per sample and
// normalises while it stages (nfdp_conv2d_gnin_nhwc_f32): the normalised tensor is never written, one launch of ~6 us per ResnetBlock gone.
extern "C" int nfdp_conv2d_stats_tile(int B, int H, int W, int Cin, int Cout, int K, int groups) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || groups <= 0 || (K % 2) != 1 || K > 7) return 0;
    if (Cout % 64 != 0 || Cout % groups != 0) return 0;
    const int cpg = Cout / groups, HW = H * W;
    if (!pow2(cpg) || cpg < 4 || cpg > 64) return 0;
    const int mt = splitk_tile_pixels((long)B * HW, Cin, Cout, K, W);
    return (mt > 0 && HW % mt == 0) ? mt : 0;          // whole tiles, each inside one image
}

extern "C" int nfdp_conv2d_stats_nhwc_f32(const float* x1, int x1_ld, int Cin1, const float* x2, int x2_ld, int Cin2, const void* w_packed,
                                          float* y, int y_ld, int B, int H, int W, int Cout, int K, const float* bias, int groups,
                                          double* sums, int compute, nfdp_stream_t stream) {
    NFDP_REQUIRE(x1 && sums, NFDP_EINVAL, "conv2d_stats: null pointer");
    NFDP_REQUIRE(compute == NFDP_F32 || compute == NFDP_F32X6, NFDP_EINVAL, "conv2d_stats: compute=%d (fp32 or the six-product split form)", compute);
    NFDP_REQUIRE((x2 == nullptr) == (Cin2 == 0) && Cin1 > 0 && Cin2 >= 0, NFDP_EINVAL, "conv2d_stats: second source and Cin2 go together");
    NFDP_REQUIRE(nfdp_conv2d_stats_tile(B, H, W, Cin1 + Cin2, Cout, K, groups) != 0, NFDP_EUNSUPPORTED,
                 "conv2d_stats: unsupported geometry B=%d H=%d W=%d Cin=%d Cout=%d K=%d groups=%d", B, H, W, Cin1 + Cin2, Cout, K, groups);
    NFDP_REQUIRE(x1_ld % 4 == 0 && y_ld % 4 == 0 && (((uintptr_t)x1 | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)sums) % 16) == 0, NFDP_ESHAPE,
                 "conv2d_stats: rows and vectors must be 16-byte aligned");
    if (x2) {
        NFDP_REQUIRE(Cin1 % 8 == 0 && Cin2 % 4 == 0 && x2_ld >= Cin2 && x2_ld % 4 == 0 && ((uintptr_t)x2 % 16) == 0, NFDP_ESHAPE,
                     "conv2d_stats: two sources: %d + %d channels (first a multiple of 8, second of 4), 16-byte aligned rows", Cin1, Cin2);
    }
    ConvExtra ex;
    ex.gs_out = sums; ex.gs_groups = groups;
    return conv2d_impl(x1, NFDP_F32, x1_ld, w_packed, y, NFDP_F32, y_ld, B, H, W, Cin1 + Cin2, Cout, K, K, bias, nullptr, 0.f, 0, compute, stream,
                       x2, x2_ld, x2 ? Cin1 : 0, &ex);
}

// ResnetBlock's res_conv (1x1, the same input) in block1's launch: y = conv3x3(x) (+ the group sums when `sums` is given, see above),
// y_res = conv1x1(x).  Both weights packed fp32 with their split images; the LDS-staged kernel only.
extern "C" int nfdp_conv2d_res_supported(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !g_conv_ldsx) return 0;
    const int pt = splitk_tile_pixels((long)B * H * W, Cin, Cout, 3, W) / 16;
    if (pt == 0 || !split_image_shape(Cout, Cin, 3, 3) || !split_image_shape(Cout, Cin, 1, 1) || ldsx_np(W, pt) > 1024) return 0;
    return ldsx_bytes(W, Cin, pt) + kLdsxStatic <= kLdsxMax;
}

extern "C" int nfdp_conv2d_res_nhwc_f32(const float* x1, int x1_ld, int Cin1, const float* x2, int x2_ld, int Cin2, const void* w_packed,
                                        float* y, int y_ld, int B, int H, int W, int Cout, const float* bias, int groups, double* sums,
                                        const void* w_res_packed, const float* bias_res, float* y_res, int y_res_ld, nfdp_stream_t stream) {
    NFDP_REQUIRE(x1 && w_res_packed && y_res, NFDP_EINVAL, "conv2d_res: null pointer");
    NFDP_REQUIRE((x2 == nullptr) == (Cin2 == 0) && Cin1 > 0 && Cin2 >= 0, NFDP_EINVAL, "conv2d_res: second source and Cin2 go together");
    NFDP_REQUIRE(nfdp_conv2d_res_supported(B, H, W, Cin1 + Cin2, Cout), NFDP_EUNSUPPORTED,
                 "conv2d_res: unsupported geometry B=%d H=%d W=%d Cin=%d Cout=%d", B, H, W, Cin1 + Cin2, Cout);
    NFDP_REQUIRE(!sums || nfdp_conv2d_stats_tile(B, H, W, Cin1 + Cin2, Cout, 3, groups) != 0, NFDP_EUNSUPPORTED,
                 "conv2d_res: no group sums for B=%d H=%d W=%d Cin=%d Cout=%d groups=%d (nfdp_conv2d_stats_tile)", B, H, W, Cin1 + Cin2, Cout, groups);
    NFDP_REQUIRE(x1_ld % 4 == 0 && y_ld % 4 == 0 && y_res_ld >= Cout && y_res_ld % 4 == 0 &&
                 (((uintptr_t)x1 | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)sums | (uintptr_t)bias_res | (uintptr_t)y_res) % 16) == 0, NFDP_ESHAPE,
                 "conv2d_res: rows and vectors must be 16-byte aligned");
    if (x2) {
        NFDP_REQUIRE(Cin1 % 8 == 0 && Cin2 % 4 == 0 && x2_ld >= Cin2 && x2_ld % 4 == 0 && ((uintptr_t)x2 % 16) == 0, NFDP_ESHAPE,
                     "conv2d_res: two sources: %d + %d channels (first a multiple of 8, second of 4), 16-byte aligned rows", Cin1, Cin2);
    }
    ConvExtra ex;
    ex.gs_out = sums; ex.gs_groups = groups;
    ex.w2 = w_res_packed; ex.add2 = bias_res; ex.y2 = y_res; ex.y2_ld = y_res_ld;
    return conv2d_impl(x1, NFDP_F32, x1_ld, w_packed, y, NFDP_F32, y_ld, B, H, W, Cin1 + Cin2, Cout, 3, 3, bias, nullptr, 0.f, 0, NFDP_F32X6, stream,
                       x2, x2_ld, x2 ? Cin1 : 0, &ex);
}

// Downsample (reference :176-180: Rearrange("b c (h p1) (w p2) -> b (c p1 p2) h w") -> Conv2d 1x1) as one launch: the rearranged channel
// index is c * 4 + p1 * 2 + p2, so the 1x1 weight [Cout][4C] viewed as [Cout][C][2][2] is the weight of a 2x2 conv with stride 2 on the
// un-rearranged tensor.  x [B][2H][2W][C], w_packed = nfdp_conv_pack_weights of that view (KH = KW = 2), y [B][H][W][Cout]; the split-K conv.
extern "C" int nfdp_conv2d_squeeze_supported(int B, int H, int W, int C, int Cout) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0 || C % 32 != 0) return 0;
    if (Cout < 64 || Cout % 4 != 0 || kpad_for(C, 2, 2) < 8 * 32 || cdiv((long)B * H * W, 64) * cdiv(Cout, 64) > 1024) return 0;
    return (long)B * H * W * 4 < (1L << 31) - 4096;
}

extern "C" int nfdp_conv2d_squeeze_nhwc_f32(const float* x, int x_ld, const void* w_packed, float* y, int y_ld, int B, int H, int W, int C,
                                            int Cout, const float* bias, int compute, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && w_packed && y, NFDP_EINVAL, "conv2d_squeeze: null pointer");
    NFDP_REQUIRE(compute == NFDP_F32 || compute == NFDP_F32X6, NFDP_EINVAL, "conv2d_squeeze: compute=%d (fp32 or the six-product split form)", compute);
    NFDP_REQUIRE(nfdp_conv2d_squeeze_supported(B, H, W, C, Cout), NFDP_EUNSUPPORTED,
                 "conv2d_squeeze: unsupported geometry B=%d H=%d W=%d (output) C=%d Cout=%d", B, H, W, C, Cout);
    NFDP_REQUIRE(x_ld >= C && x_ld % 4 == 0 && y_ld >= Cout && y_ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)bias) % 16) == 0, NFDP_ESHAPE,
                 "conv2d_squeeze: rows and vectors must be 16-byte aligned");
    const long npix = (long)B * H * W;
    ConvParams P;
    P.x = x; P.w = w_packed; P.y = y; P.add = bias; P.mul = nullptr; P.npix = npix;
    P.x_ld = x_ld; P.y_ld = y_ld; P.y_bf16 = 0;
    P.H = H; P.W = W; P.Cin = C; P.Cout = Cout; P.KH = 2; P.KW = 2; P.pad = 0;
    P.K = 4 * C; P.Kpad = kpad_for(C, 2, 2); P.relu = 0; P.mul_exp = 0.f;
    P.sq = 1;
    P.Nt = (int)cdiv(Cout, 64);
    const int pt = splitk_tile_pixels(npix, C, Cout, 2, W) / 16;
    P.Mt = (int)cdiv(npix, 16 * pt);
    const dim3 grid(P.Mt * P.Nt);
    hipStream_t s = as_stream(stream);
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&conv_splitk_f32_kernel<8, 4, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 64 * SK_LD * 4);
        nfdp_func_attr(reinterpret_cast<const void*>(&conv_splitk_f32_kernel<4, 8, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * SK_LD * 4);
        attr_set = true;
    }
    if (compute == NFDP_F32X6) {
        if (pt == 4) hipLaunchKernelGGL((conv_splitk_f32_kernel<4, 8, true, 4>), grid, dim3(256), 4 * 64 * SK_LD * 4, s, P);
        else if (pt == 2) hipLaunchKernelGGL((conv_splitk_f32_kernel<4, 8, true, 2>), grid, dim3(256), 4 * 32 * SK_LD * 4, s, P);
        else hipLaunchKernelGGL((conv_splitk_f32_kernel<4, 8, true, 1>), grid, dim3(256), 4 * 16 * SK_LD * 4, s, P);
    } else {
        if (pt == 4) hipLaunchKernelGGL((conv_splitk_f32_kernel<8, 4, false, 4>), grid, dim3(512), 8 * 64 * SK_LD * 4, s, P);
        else if (pt == 2) hipLaunchKernelGGL((conv_splitk_f32_kernel<8, 4, false, 2>), grid, dim3(512), 8 * 32 * SK_LD * 4, s, P);
        else hipLaunchKernelGGL((conv_splitk_f32_kernel<8, 4, false, 1>), grid, dim3(512), 8 * 16 * SK_LD * 4, s, P);
    }
    NFDP_CHECK_LAUNCH("conv2d_squeeze");
    return NFDP_OK;
}

// Upsample (reference :169-173: nn.Upsample(scale_factor = 2, mode = "nearest") -> Conv2d 3x3) as one launch: x [B][H/2][W/2][Cin] is read
// through the index map of its x2 nearest-neighbour upsampling while the input tile is staged; H, W: the OUTPUT sides (powers of two).
extern "C" int nfdp_conv2d_up2_supported(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || H < 2 || W < 2 || Cin <= 0 || Cout <= 0 || !g_conv_ldsx || !pow2(H) || !pow2(W)) return 0;
    const int pt = splitk_tile_pixels((long)B * H * W, Cin, Cout, 3, W) / 16;
    if (pt == 0 || !split_image_shape(Cout, Cin, 3, 3) || ldsx_np(W, pt) > 1024) return 0;
    return ldsx_bytes(W, Cin, pt) + kLdsxStatic <= kLdsxMax;
}

extern "C" int nfdp_conv2d_up2_nhwc_f32(const float* x, int x_ld, const void* w_packed, float* y, int y_ld, int B, int H, int W, int Cin,
                                        int Cout, const float* bias, nfdp_stream_t stream) {
    NFDP_REQUIRE(x, NFDP_EINVAL, "conv2d_up2: null pointer");
    NFDP_REQUIRE(nfdp_conv2d_up2_supported(B, H, W, Cin, Cout), NFDP_EUNSUPPORTED,
                 "conv2d_up2: unsupported geometry B=%d H=%d W=%d (output) Cin=%d Cout=%d", B, H, W, Cin, Cout);
    NFDP_REQUIRE(x_ld % 4 == 0 && y_ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)bias) % 16) == 0, NFDP_ESHAPE,
                 "conv2d_up2: rows and vectors must be 16-byte aligned");
    ConvExtra ex;
    ex.up2 = 1;
    return conv2d_impl(x, NFDP_F32, x_ld, w_packed, y, NFDP_F32, y_ld, B, H, W, Cin, Cout, 3, 3, bias, nullptr, 0.f, 0, NFDP_F32X6, stream,
                       nullptr, 0, 0, &ex);
}

// ... and the same beside the one-launch Block (tiles that hold whole images: the 4x4 / 2x2 stages): y = SiLU(GN(conv3x3(x) + bias) *
// (scale + 1) + shift) as nfdp_conv2d_gn_silu_(cat_)nhwc_f32, y_res = conv1x1(x) + bias_res.
extern "C" int nfdp_conv2d_gn_silu_res_supported(int B, int H, int W, int Cin, int Cout, int groups) {
    if (!nfdp_conv2d_gn_silu_supported(B, H, W, Cin, Cout, 3, groups) || !g_conv_ldsx) return 0;
    const int pt = gn_pixel_tiles((long)B * H * W, Cout, H * W);
    if (pt == 0 || !split_image_shape(Cout, Cin, 3, 3) || !split_image_shape(Cout, Cin, 1, 1) || ldsx_np(W, pt) > 1024) return 0;
    return ldsx_bytes(W, Cin, pt) + kLdsxStatic <= kLdsxMax;
}

extern "C" int nfdp_conv2d_gn_silu_res_nhwc_f32(const float* x1, int x1_ld, int Cin1, const float* x2, int x2_ld, int Cin2,
                                                const void* w_packed, float* y, int y_ld, int B, int H, int W, int Cout, const float* bias,
                                                int groups, const float* gamma, const float* beta, const float* scale_shift, int ss_ld,
                                                float eps, const void* w_res_packed, const float* bias_res, float* y_res, int y_res_ld,
                                                nfdp_stream_t stream) {
    NFDP_REQUIRE(x1 && w_res_packed && y_res, NFDP_EINVAL, "conv2d_gn_silu_res: null pointer");
    NFDP_REQUIRE((x2 == nullptr) == (Cin2 == 0) && Cin1 > 0 && Cin2 >= 0, NFDP_EINVAL, "conv2d_gn_silu_res: second source and Cin2 go together");
    NFDP_REQUIRE(nfdp_conv2d_gn_silu_res_supported(B, H, W, Cin1 + Cin2, Cout, groups), NFDP_EUNSUPPORTED,
                 "conv2d_gn_silu_res: unsupported geometry B=%d H=%d W=%d Cin=%d Cout=%d groups=%d", B, H, W, Cin1 + Cin2, Cout, groups);
    NFDP_REQUIRE(y_res_ld >= Cout && y_res_ld % 4 == 0 && (((uintptr_t)bias_res | (uintptr_t)y_res) % 16) == 0, NFDP_ESHAPE,
                 "conv2d_gn_silu_res: the second output's rows and bias must be 16-byte aligned");
    if (x2) {
        NFDP_REQUIRE(Cin1 % 8 == 0 && Cin2 % 8 == 0 && x2_ld >= Cin2 && x2_ld % 4 == 0 && ((uintptr_t)x2 % 16) == 0, NFDP_ESHAPE,
                     "conv2d_gn_silu_res: two sources: %d + %d channels (multiples of 8), 16-byte aligned rows", Cin1, Cin2);
    }
    ConvExtra ex;
    ex.w2 = w_res_packed; ex.add2 = bias_res; ex.y2 = y_res; ex.y2_ld = y_res_ld;
    return conv2d_gn_silu_impl(x1, x1_ld, w_packed, y, y_ld, B, H, W, Cin1 + Cin2, Cout, 3, bias, groups, gamma, beta, scale_shift, ss_ld, eps,
                               nullptr, 0, NFDP_F32X6, stream, x2, x2_ld, x2 ? Cin1 : 0, &ex);
}

extern "C" int nfdp_conv2d_gnin_supported(int B, int H, int W, int Cin, int Cout, int in_groups, int out_groups) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || in_groups <= 0 || in_groups % 2 != 0 || Cin % in_groups != 0) return 0;
    const int cpg = Cin / in_groups, HW = H * W;
    if (!pow2(cpg) || cpg < 4 || !pow2(HW) || !g_conv_ldsx) return 0;
    const long npix = (long)B * HW;
    int pt;
    if (out_groups > 0) {
        if (!nfdp_conv2d_gn_silu_supported(B, H, W, Cin, Cout, 3, out_groups)) return 0;
        pt = gn_pixel_tiles(npix, Cout, HW);
    } else {
        pt = splitk_tile_pixels(npix, Cin, Cout, 3, W) / 16;
    }
    if (pt == 0 || !split_image_shape(Cout, Cin, 3, 3) || ldsx_np(W, pt) > 1024) return 0;
    return ldsx_bytes(W, Cin, pt, HW, in_groups) + kLdsxStatic <= kLdsxMax;
}

extern "C" int nfdp_conv2d_gnin_nhwc_f32(const float* x, int x_ld, const double* in_sums, int in_tile, int in_groups, const float* in_gamma,
                                         const float* in_beta, const float* in_scale_shift, int in_ss_ld, float in_eps, const void* w_packed,
                                         float* y, int y_ld, int B, int H, int W, int Cin, int Cout, const float* bias, int out_groups,
                                         const float* out_gamma, const float* out_beta, const float* out_scale_shift, int out_ss_ld,
                                         float out_eps, const float* residual, int res_ld, int sums_groups, double* sums_out,
                                         nfdp_stream_t stream) {
    NFDP_REQUIRE(x && in_sums && in_gamma && in_beta && w_packed && y, NFDP_EINVAL, "conv2d_gnin: null pointer");
    NFDP_REQUIRE(nfdp_conv2d_gnin_supported(B, H, W, Cin, Cout, in_groups, out_groups), NFDP_EUNSUPPORTED,
                 "conv2d_gnin: unsupported geometry B=%d H=%d W=%d Cin=%d Cout=%d groups %d -> %d", B, H, W, Cin, Cout, in_groups, out_groups);
    NFDP_REQUIRE(in_tile > 0 && (H * W) % in_tile == 0, NFDP_ESHAPE, "conv2d_gnin: in_tile=%d does not divide the image (nfdp_conv2d_stats_tile)", in_tile);
    NFDP_REQUIRE(in_scale_shift == nullptr || in_ss_ld >= 2 * Cin, NFDP_ESHAPE, "conv2d_gnin: scale_shift rows shorter than 2*Cin");
    NFDP_REQUIRE(x_ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)in_sums | (uintptr_t)in_gamma | (uintptr_t)in_beta | (uintptr_t)in_scale_shift) % 16) == 0 &&
                 in_ss_ld % 4 == 0, NFDP_ESHAPE, "conv2d_gnin: rows and vectors must be 16-byte aligned");
    NFDP_REQUIRE(out_groups > 0 || residual == nullptr, NFDP_EINVAL, "conv2d_gnin: a residual goes with the output Block (out_groups > 0)");
    NFDP_REQUIRE(!sums_out || (out_groups == 0 && nfdp_conv2d_stats_tile(B, H, W, Cin, Cout, 3, sums_groups) != 0 && (uintptr_t)sums_out % 16 == 0),
                 NFDP_EUNSUPPORTED, "conv2d_gnin: no group sums of the output for this geometry (nfdp_conv2d_stats_tile; out_groups must be 0)");
    ConvExtra ex;
    ex.gs_out = sums_out; ex.gs_groups = sums_groups;
    ex.gi_stats = in_sums; ex.gi_gamma = in_gamma; ex.gi_beta = in_beta; ex.gi_ss = in_scale_shift; ex.gi_ss_ld = in_ss_ld;
    ex.gi_groups = in_groups; ex.gi_parts = (H * W) / in_tile; ex.gi_eps = in_eps;
    if (out_groups > 0)
        return conv2d_gn_silu_impl(x, x_ld, w_packed, y, y_ld, B, H, W, Cin, Cout, 3, bias, out_groups, out_gamma, out_beta, out_scale_shift,
                                   out_ss_ld, out_eps, residual, res_ld, NFDP_F32X6, stream, nullptr, 0, 0, &ex);
    return conv2d_impl(x, NFDP_F32, x_ld, w_packed, y, NFDP_F32, y_ld, B, H, W, Cin, Cout, 3, 3, bias, nullptr, 0.f, 0, NFDP_F32X6, stream,
                       nullptr, 0, 0, &ex);
}

// PreNorm -> LinearAttention (to_qkv + core) in one launch (linattn_qkv_kernel): x [B][n][C] (rows x_ld), w_packed = the to_qkv weight
// [3 * heads * 32][C][1][1] packed fp32 (with its split image), out [B][n][heads * 32].
extern "C" int nfdp_linear_attention_qkv_supported(int n, int C, int heads, int dim_head) {
    return ((n >= 1 && n <= 64) || (n >= 256 && n % 256 == 0)) && (C == 64 || C == 128) && dim_head == 32 && heads >= 1 && heads <= 8 &&
           split_image_shape(3 * heads * 32, C, 1, 1);
}
// n > 256: the chunked form's workspace (a record of column maxima, exp sums and context per (sample, head, 256-pixel chunk)); else 0
extern "C" size_t nfdp_linear_attention_qkv_ws_bytes(int B, int n, int heads) {
    if (B <= 0 || n <= 256 || heads <= 0) return 0;
    return (size_t)B * heads * (n / 256) * (2 * 32 + 32 * 32) * sizeof(float);
}
extern "C" int nfdp_linear_attention_qkv_f32(const float* x, int x_ld, const float* pre_gain, const void* w_packed, float* out, int B, int n,
                                             int C, int heads, int dim_head, float scale, float eps, void* ws, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && pre_gain && w_packed && out, NFDP_EINVAL, "linear_attention_qkv: null pointer");
    NFDP_REQUIRE(n <= 256 || (ws && (uintptr_t)ws % 16 == 0), NFDP_EINVAL, "linear_attention_qkv: n > 256 needs the workspace (nfdp_linear_attention_qkv_ws_bytes)");
    NFDP_REQUIRE(B >= 0, NFDP_ESHAPE, "linear_attention_qkv: bad shape");
    NFDP_REQUIRE(nfdp_linear_attention_qkv_supported(n, C, heads, dim_head), NFDP_EUNSUPPORTED,
                 "linear_attention_qkv: n=%d (<= 64 or a multiple of 256) C=%d (64, 128) heads=%d dim_head=%d (32)", n, C, heads, dim_head);
    NFDP_REQUIRE(x_ld >= C && x_ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)pre_gain | (uintptr_t)out) % 16) == 0, NFDP_ESHAPE,
                 "linear_attention_qkv: rows and vectors must be 16-byte aligned");
    if (B == 0) return NFDP_OK;
    const int rows = 3 * heads * 32;
    const char* wimg = reinterpret_cast<const char*>(w_packed) + (size_t)cout_pad_for(rows) * kpad_for(C, 1, 1) * 4;
    const size_t lds = (size_t)2 * n * 33 * sizeof(float);
    hipStream_t s = as_stream(stream);
    static NfdpPerDevice attr_set;
    if (!attr_set) {
        nfdp_func_attr(reinterpret_cast<const void*>(&linattn_qkv_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 33 * 4);
        nfdp_func_attr(reinterpret_cast<const void*>(&linattn_qkv_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 33 * 4);
        attr_set = true;
    }
    if (n > 256) {
        const dim3 grid2(B * heads, n / 256);
        float* wsf = reinterpret_cast<float*>(ws);
        if (C == 64) {
            hipLaunchKernelGGL((linattn_partial_qkv_kernel<2>), grid2, dim3(256), 0, s, x, x_ld, pre_gain, wimg, wsf, n, heads, eps);
            hipLaunchKernelGGL((linattn_apply_qkv_kernel<2>), grid2, dim3(256), 0, s, x, x_ld, pre_gain, wimg, wsf, out, n, heads, scale, eps);
        } else {
            hipLaunchKernelGGL((linattn_partial_qkv_kernel<4>), grid2, dim3(256), 0, s, x, x_ld, pre_gain, wimg, wsf, n, heads, eps);
            hipLaunchKernelGGL((linattn_apply_qkv_kernel<4>), grid2, dim3(256), 0, s, x, x_ld, pre_gain, wimg, wsf, out, n, heads, scale, eps);
        }
        NFDP_CHECK_LAUNCH("linear_attention_qkv (chunked)");
        return NFDP_OK;
    }
    const dim3 grid(B * heads), blk(256);
    if (n < 64) {                                        // ragged: the 4x4 / 2x2 stages, odd image sizes
        const size_t lds1 = (size_t)2 * 64 * 33 * sizeof(float);
        if (C == 64) hipLaunchKernelGGL((linattn_qkv_kernel<2, 1, true>), grid, blk, lds1, s, x, x_ld, pre_gain, wimg, out, heads, scale, eps, n);
        else hipLaunchKernelGGL((linattn_qkv_kernel<4, 1, true>), grid, blk, lds1, s, x, x_ld, pre_gain, wimg, out, heads, scale, eps, n);
    } else if (C == 64 && n == 256) hipLaunchKernelGGL((linattn_qkv_kernel<2, 4>), grid, blk, lds, s, x, x_ld, pre_gain, wimg, out, heads, scale, eps, n);
    else if (C == 64) hipLaunchKernelGGL((linattn_qkv_kernel<2, 1>), grid, blk, lds, s, x, x_ld, pre_gain, wimg, out, heads, scale, eps, n);
    else if (n == 256) hipLaunchKernelGGL((linattn_qkv_kernel<4, 4>), grid, blk, lds, s, x, x_ld, pre_gain, wimg, out, heads, scale, eps, n);
    else hipLaunchKernelGGL((linattn_qkv_kernel<4, 1>), grid, blk, lds, s, x, x_ld, pre_gain, wimg, out, heads, scale, eps, n);
    NFDP_CHECK_LAUNCH("linear_attention_qkv");
    return NFDP_OK;
}

// PreNorm -> Attention (to_qkv + softmax core) in one launch (attn_qkv_kernel): n = 64 keys, C in {64, 128}.
extern "C" int nfdp_attention_qkv_supported(int n, int C, int heads, int dim_head) {
    return n >= 1 && n <= 64 && (C == 64 || C == 128) && dim_head == 32 && heads >= 1 && heads <= 8 && split_image_shape(3 * heads * 32, C, 1, 1);
}
extern "C" int nfdp_attention_qkv_f32(const float* x, int x_ld, const float* pre_gain, const void* w_packed, float* out, int B, int n, int C,
                                      int heads, int dim_head, float scale, float eps, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && pre_gain && w_packed && out, NFDP_EINVAL, "attention_qkv: null pointer");
    NFDP_REQUIRE(B >= 0, NFDP_ESHAPE, "attention_qkv: bad shape");
    NFDP_REQUIRE(nfdp_attention_qkv_supported(n, C, heads, dim_head), NFDP_EUNSUPPORTED,
                 "attention_qkv: n=%d (<= 64) C=%d (64, 128) heads=%d dim_head=%d (32)", n, C, heads, dim_head);
    NFDP_REQUIRE(x_ld >= C && x_ld % 4 == 0 && (((uintptr_t)x | (uintptr_t)pre_gain | (uintptr_t)out) % 16) == 0, NFDP_ESHAPE,
                 "attention_qkv: rows and vectors must be 16-byte aligned");
    if (B == 0) return NFDP_OK;
    const char* wimg = reinterpret_cast<const char*>(w_packed) + (size_t)cout_pad_for(3 * heads * 32) * kpad_for(C, 1, 1) * 4;
    hipStream_t s = as_stream(stream);
    if (C == 64) hipLaunchKernelGGL((attn_qkv_kernel<2>), dim3(B * heads), dim3(256), 0, s, x, x_ld, pre_gain, wimg, out, heads, scale, eps, n);
    else hipLaunchKernelGGL((attn_qkv_kernel<4>), dim3(B * heads), dim3(256), 0, s, x, x_ld, pre_gain, wimg, out, heads, scale, eps, n);
    NFDP_CHECK_LAUNCH("attention_qkv");
    return NFDP_OK;
}

// 1x1 conv with the channel LayerNorm of its input (pre_gain) and / or of its output (post_gain, + residual) in the same launch: the
// to_qkv / to_out convs of the U-Nets' attention blocks (conv1x1_ln_kernel).  fp32 tensors, split-product arithmetic (compute = NFDP_F32X6).
extern "C" int nfdp_conv1x1_ln_supported(long npix, int Cin, int Cout, int pre, int post) {
    if (npix <= 0 || !split_image_shape(Cout, Cin, 1, 1) || Cin > 128 || Cout % 4 != 0) return 0;
    if (post && Cout > 128) return 0;                    // the output LayerNorm needs all outputs of a pixel in one workgroup
    return 1;                                            // (neither LayerNorm: the plain conv, for its residual epilogue)
}
extern "C" int nfdp_conv1x1_ln_f32(const float* x, int x_ld, const void* w_packed, float* y, int y_ld, long npix, int Cin, int Cout,
                                   const float* bias, const float* pre_gain, const float* post_gain, const float* residual, int res_ld,
                                   float eps, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && w_packed && y, NFDP_EINVAL, "conv1x1_ln: null pointer");
    NFDP_REQUIRE(nfdp_conv1x1_ln_supported(npix, Cin, Cout, pre_gain != nullptr, post_gain != nullptr), NFDP_EUNSUPPORTED,
                 "conv1x1_ln: unsupported shape npix=%ld Cin=%d Cout=%d (Cin %% 32, Cin <= 128; output LayerNorm: Cout <= 128)", npix, Cin, Cout);
    NFDP_REQUIRE(x_ld >= Cin && x_ld % 4 == 0 && y_ld >= Cout && y_ld % 4 == 0 && (!residual || (res_ld >= Cout && res_ld % 4 == 0)) &&
                 (((uintptr_t)x | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)pre_gain | (uintptr_t)post_gain | (uintptr_t)residual) % 16) == 0,
                 NFDP_ESHAPE, "conv1x1_ln: rows and vectors must be 16-byte aligned");
    ConvParams P;
    P.x = x; P.w = w_packed; P.y = y; P.add = bias; P.mul = nullptr; P.npix = npix;
    P.x_ld = x_ld; P.y_ld = y_ld; P.y_bf16 = 0;
    P.H = 1; P.W = 1; P.Cin = Cin; P.Cout = Cout; P.KH = 1; P.KW = 1; P.pad = 0;
    P.K = Cin; P.Kpad = kpad_for(Cin, 1, 1); P.relu = 0; P.mul_exp = 0.f;
    P.Nt = (int)cdiv(Cout, 64); P.Mt = (int)cdiv(npix, 64);
    P.wsplit_off = (size_t)cout_pad_for(Cout) * P.Kpad * 4;
    LnConv ln{pre_gain, post_gain, residual, res_ld, eps};
    hipStream_t s = as_stream(stream);
    const bool ns2 = P.Kpad / 32 <= 2;
    const dim3 blk(256);
    if (post_gain) {
        const dim3 grid(P.Mt);
        const bool two = P.Nt == 2;
        if (pre_gain) {
            if (ns2) { if (two) hipLaunchKernelGGL((conv1x1_ln_kernel<2, 2, true, true>), grid, blk, 0, s, P, ln); else hipLaunchKernelGGL((conv1x1_ln_kernel<2, 1, true, true>), grid, blk, 0, s, P, ln); }
            else { if (two) hipLaunchKernelGGL((conv1x1_ln_kernel<4, 2, true, true>), grid, blk, 0, s, P, ln); else hipLaunchKernelGGL((conv1x1_ln_kernel<4, 1, true, true>), grid, blk, 0, s, P, ln); }
        } else {
            if (ns2) { if (two) hipLaunchKernelGGL((conv1x1_ln_kernel<2, 2, false, true>), grid, blk, 0, s, P, ln); else hipLaunchKernelGGL((conv1x1_ln_kernel<2, 1, false, true>), grid, blk, 0, s, P, ln); }
            else { if (two) hipLaunchKernelGGL((conv1x1_ln_kernel<4, 2, false, true>), grid, blk, 0, s, P, ln); else hipLaunchKernelGGL((conv1x1_ln_kernel<4, 1, false, true>), grid, blk, 0, s, P, ln); }
        }
    } else {
        const dim3 grid(P.Mt * P.Nt);
        if (pre_gain) {
            if (ns2) hipLaunchKernelGGL((conv1x1_ln_kernel<2, 1, true, false>), grid, blk, 0, s, P, ln);
            else hipLaunchKernelGGL((conv1x1_ln_kernel<4, 1, true, false>), grid, blk, 0, s, P, ln);
        } else {
            if (ns2) hipLaunchKernelGGL((conv1x1_ln_kernel<2, 1, false, false>), grid, blk, 0, s, P, ln);
            else hipLaunchKernelGGL((conv1x1_ln_kernel<4, 1, false, false>), grid, blk, 0, s, P, ln);
        }
    }
    NFDP_CHECK_LAUNCH("conv1x1_ln");
    return NFDP_OK;
}

static int conv2d_impl(const void* x, int x_dtype, int x_ld, const void* w_packed, void* y, int y_dtype, int y_ld,
                       int B, int H, int W, int Cin, int Cout, int KH, int KW, const float* add, const float* mul,
                       float mul_exp, int relu, int compute, nfdp_stream_t stream, const void* x2, int x2_ld, int cin1,
                       const ConvExtra* ex) {
    NFDP_REQUIRE(x && w_packed && y, NFDP_EINVAL, "conv2d: null pointer");
    NFDP_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, NFDP_ESHAPE, "conv2d: bad shape");
    NFDP_REQUIRE(KH == KW && (KH % 2) == 1 && KH <= 7, NFDP_EUNSUPPORTED, "conv2d: kernel %dx%d (odd square <= 7 only)", KH, KW);
    NFDP_REQUIRE(x_ld >= (x2 ? cin1 : Cin) && y_ld >= Cout, NFDP_ESHAPE, "conv2d: ld smaller than channel count");
    NFDP_REQUIRE(compute == NFDP_F32 || compute == NFDP_BF16 || compute == NFDP_F32X6, NFDP_EINVAL, "conv2d: compute=%d", compute);
    const int emu = compute == NFDP_F32X6;       // fp32 storage and packed weights; products from three-way bf16 splits
    if (emu) compute = NFDP_F32;
    NFDP_REQUIRE((x_dtype == NFDP_F32 || x_dtype == NFDP_BF16) && (y_dtype == NFDP_F32 || y_dtype == NFDP_BF16), NFDP_EINVAL,
                 "conv2d: bad dtype");
    NFDP_REQUIRE(compute == NFDP_BF16 || (x_dtype == NFDP_F32 && y_dtype == NFDP_F32), NFDP_EUNSUPPORTED,
                 "conv2d: bf16 storage requires bf16 compute");
    const long npix = (long)B * H * W;
    NFDP_REQUIRE(npix < (1L << 31) - 4096, NFDP_ESHAPE, "conv2d: too many pixels (%ld)", npix);
    if (npix == 0) return NFDP_OK;
    ConvParams P;
    P.x = x; P.w = w_packed; P.y = y; P.add = add; P.mul = mul; P.npix = npix;
    P.x_ld = x_ld; P.y_ld = y_ld; P.y_bf16 = (y_dtype == NFDP_BF16);
    P.H = H; P.W = W; P.Cin = Cin; P.Cout = Cout; P.KH = KH; P.KW = KW; P.pad = (KH - 1) / 2;
    P.K = Cin * KH * KW; P.Kpad = kpad_for(Cin, KH, KW); P.relu = relu; P.Mt = 0; P.Nt = 0; P.mul_exp = mul_exp;
    P.emu = emu;
    if (x2) { P.x2 = x2; P.x2_ld = x2_ld; P.Cin1 = cin1; }
    apply_extra(P, ex);
    const int esz = (x_dtype == NFDP_BF16) ? 2 : 4;
    // vector path: 4 consecutive k stay inside one tap and the 4-element load is naturally aligned
    const bool vec = (Cin % 4 == 0) && (x_ld % 4 == 0) && (((uintptr_t)x) % (4 * esz) == 0);
    hipStream_t s = as_stream(stream);
    // pointwise conv on bf16 activations with a wide output: the LDS-DMA GEMM
    const int y_esz = (y_dtype == NFDP_BF16) ? 2 : 4;
    if (compute == NFDP_BF16 && x_dtype == NFDP_BF16 && KH == 1 && Cin % PW_BK == 0 && Cout > 64 && Cout % 4 == 0 && x_ld % 8 == 0 &&
        ((uintptr_t)x % 16) == 0 && y_ld % (16 / y_esz) == 0 && ((uintptr_t)y % 16) == 0 && (add == nullptr || (uintptr_t)add % 16 == 0) &&
        (mul == nullptr || (uintptr_t)mul % 16 == 0) && npix >= 4 * PW_BM) {
        P.Mt = (int)cdiv(npix, PW_BM);
        P.Nt = cout_pad_for(Cout) / PW_BN;
        const int grid = (int)(cdiv(P.Mt, 8) * 8) * P.Nt;
        static NfdpPerDevice attr_set;
        if (!attr_set) {
            nfdp_func_attr(reinterpret_cast<const void*>(&pw_gemm_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      PW_LDS_BYTES);
            attr_set = true;
        }
        hipLaunchKernelGGL(pw_gemm_bf16_kernel<false>, dim3(grid), dim3(kThreads), PW_LDS_BYTES, s, P);
        NFDP_CHECK_LAUNCH("conv2d (pointwise bf16 GEMM)");
        return NFDP_OK;
    }
    // small fp32 problems with a long K: every wave takes a quarter of the K tiles of a 64 x 64 output tile
    if (compute == NFDP_F32 && Cin % 8 == 0 && x_ld % 4 == 0 && ((uintptr_t)x % 16) == 0 && Cout >= 64 && Cout % 4 == 0 &&
        y_ld % 4 == 0 && ((uintptr_t)y % 16) == 0 && P.Kpad >= 8 * 32 &&
        (cdiv(npix, 64) * cdiv(Cout, 64) <= 1024 || (emu && (x2 == nullptr || cin1 % 4 == 0) && big_ldsx_shape(W, Cin, Cout, KH))) &&
        (add == nullptr || (uintptr_t)add % 16 == 0) && (mul == nullptr || (uintptr_t)mul % 16 == 0)) {
        P.Nt = (int)cdiv(Cout, 64);
        // pixel tiles per workgroup: 64-pixel tiles unless they leave CUs without a workgroup (then 32 or 16 pixels)
        const long t64 = cdiv(npix, 64) * P.Nt;
        const int pt = t64 >= 256 ? 4 : (2 * t64 >= 256 ? 2 : 1);
        P.Mt = (int)cdiv(npix, 16 * pt);
        const dim3 grid(P.Mt * P.Nt);
        static NfdpPerDevice sk_attr_set;
        if (!sk_attr_set) {
            nfdp_func_attr(reinterpret_cast<const void*>(&conv_splitk_f32_kernel<8, 4, false, 4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 64 * SK_LD * 4);
            nfdp_func_attr(reinterpret_cast<const void*>(&conv_splitk_f32_kernel<4, 8, true, 4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * SK_LD * 4);
            sk_attr_set = true;
        }
        NFDP_REQUIRE((!P.gi_stats && !P.w2 && !P.up2) || (emu && ldsx_ok(P, pt)), NFDP_EUNSUPPORTED,
                     "conv2d_gnin / conv2d_res / conv2d_up2: available in the LDS-staged 3x3 conv only (nfdp_conv2d_*_supported)");
        if (emu && ldsx_ok(P, pt)) {  // fp32 products as six bf16 MFMAs (NFDP_F32X6), 3x3: the input tile staged and split once in LDS
            launch_ldsx<false>(P, pt, grid, s);
        } else if (emu) {             // ... operands gathered per tap
            if (pt == 4) hipLaunchKernelGGL((conv_splitk_f32_kernel<4, 8, true, 4>), grid, dim3(256), 4 * 64 * SK_LD * 4, s, P);
            else if (pt == 2) hipLaunchKernelGGL((conv_splitk_f32_kernel<4, 8, true, 2>), grid, dim3(256), 4 * 32 * SK_LD * 4, s, P);
            else hipLaunchKernelGGL((conv_splitk_f32_kernel<4, 8, true, 1>), grid, dim3(256), 4 * 16 * SK_LD * 4, s, P);
        } else {
            if (pt == 4) hipLaunchKernelGGL((conv_splitk_f32_kernel<8, 4, false, 4>), grid, dim3(512), 8 * 64 * SK_LD * 4, s, P);
            else if (pt == 2) hipLaunchKernelGGL((conv_splitk_f32_kernel<8, 4, false, 2>), grid, dim3(512), 8 * 32 * SK_LD * 4, s, P);
            else hipLaunchKernelGGL((conv_splitk_f32_kernel<8, 4, false, 1>), grid, dim3(512), 8 * 16 * SK_LD * 4, s, P);
        }
        NFDP_CHECK_LAUNCH("conv2d (split-K fp32)");
        return NFDP_OK;
    }
    NFDP_REQUIRE(!P.gs_out && !P.gi_stats && !P.w2 && !P.up2, NFDP_EUNSUPPORTED,
                 "conv2d_stats / conv2d_gnin / conv2d_res / conv2d_up2: this problem does not take the split-K kernels");
    if (compute == NFDP_F32) {
        if (vec) launch_by_cout<false, false, true>(P, s);
        else launch_by_cout<false, false, false>(P, s);
    } else if (x_dtype == NFDP_BF16) {
        if (vec) launch_by_cout<true, true, true>(P, s);
        else launch_by_cout<true, true, false>(P, s);
    } else {
        if (vec) launch_by_cout<true, false, true>(P, s);
        else launch_by_cout<true, false, false>(P, s);
    }
    NFDP_CHECK_LAUNCH("conv2d");
    return NFDP_OK;
}
