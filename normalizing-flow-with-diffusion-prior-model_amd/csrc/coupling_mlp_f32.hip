// The coupling network in EXACT fp32 with the hidden activations on chip: the fp32-accurate fast path.
//
// Reference arithmetic: coupling_network  normalizing_flow/utils.py:83-89  (Conv2dActNorm :67-69, ZeroConv2d :43-44), fp32 end to end.
//
//   x_a --gather (3x3 taps, zero padding)--> GEMM1 (9*C/2 -> 512) -> (.+b1)*e^s1 -> ReLU -> GEMM2 (512 -> 512) -> (.+b2)*e^s2 -> ReLU
//       -> GEMM3 (512 -> 9 taps x C, pointwise)  ==> T[pixel][tap*C + co]   (nfdp_tap_shift_sum_f32 adds the taps up, + bias, * e^{3 logs})
//
// Why this kernel exists.  The bf16 fused flow-step kernels meet BASELINE.json's bits/dim gate but not its 1e-4 PIXEL gate (8-bit
// mantissa operands: 1e-2 on latents and sampled pixels); the per-op fp32 path meets it at 34 ms per 512-image step, 16 ms of which
// are the three narrow 512 -> C 3x3 convs (10-43 TFLOP/s as implicit GEMMs) and the rest moves two 268 MB hidden tensors per flow
// through HBM.  gfx950 has no TF32-like mode; v_mfma_f32_32x32x2_f32 multiplies and accumulates in plain fp32 at 1/16 of the bf16
// rate, so at that rate everything except the matrix pipe is cheap and the design is the simple one:
//  * a workgroup = ONE wave = 32 consecutive pixels (any image size: the spatial parts of the two 3x3 convs are a gather in front and
//    the tap shift-sum behind) with the whole 512-register budget of its SIMD; waves share nothing, so there is no LDS and no barrier,
//    four of them run per CU, and the 4x4 level of a 512-image batch (8192 pixels) still gives every CU a wave;
//  * all three GEMMs TRANSPOSED (rows = output channel, lanes = pixel): a 32x32 accumulator tile of one GEMM, after its epilogue, IS
//    the B operand of the next - register r of lane (h, p) holds channel drow(r, h) of pixel p, and the instruction's two k values are
//    the two lane halves, so "k-pair r" of a hidden block is accumulator register r as it stands.  The 512 hidden channels of 32
//    pixels are 256 registers; H2 is produced one 32-channel block at a time and consumed at once by GEMM3 (tap accumulators: up to 7
//    row tiles = 112 registers; C = 48 has 14: two waves (blockIdx.y) take seven each and both compute H1 and H2);
//  * weights are read straight from global memory (L2) in MFMA-fragment order, one global_load_dwordx4 per lane = 4 k-pairs of a
//    32-row tile = 1 KiB per wave-instruction per 4 MFMAs (256 cycles): 16 B/clk per CU, far under the per-CU L2 rate, so no LDS
//    staging and no barriers at all; an 8-deep register ring keeps ~1 us of weights in flight per wave;
//  * products and sums are fp32 FMAs in a fixed order (k ascending within the permuted order baked into the images): results match
//    the per-op fp32 kernels and the reference to summation order (tests: goldens at 1e-4, element-wise).
// Bound: fp32 matrix pipe, 157.3 TFLOP/s; algorithmic work 2*(9*C/2*512 + 512^2 + 9*C*512) FLOP per pixel.
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "nfdp_common.h"

namespace {

constexpr int kThreads = 256;                         // packer
constexpr int kWave = 64;                              // the network kernel: one wave per workgroup
constexpr int HID = 512;
constexpr int NHB = HID / 32;
constexpr int DMAX = 16;                               // weight fragments in flight per wave (1 KiB each): 16, or 8 where registers are short
constexpr int FRAG = 256;                              // floats per fragment: 64 lanes x 4 k-pairs

typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr int drow(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

template <int C>
struct GeoF {
    static constexpr int Ch = C / 2;
    static constexpr int K1 = 9 * Ch;
    static constexpr int NQ1 = (K1 + 7) / 8;           // fragments (4 k-pairs = 8 k) per hidden block of GEMM1
    static constexpr int KP1 = 4 * NQ1;                // k-pairs of GEMM1, padded
    static constexpr int D = C >= 48 ? 8 : (C >= 16 ? 12 : DMAX);   // ring depth, by what the registers leave (no scratch in the loops)
    static constexpr int NROW = 9 * C;
    static constexpr int NRT = (NROW + 31) / 32;
    static constexpr int NG = (NRT + 6) / 7;           // GEMM3 row groups (tap accumulators: at most 7 tiles = 112 registers)
    static constexpr int NRTG = (NRT + NG - 1) / NG;
    static constexpr int NB = 4 * NHB + 4 * NRTG;      // fragments one hidden block of GEMM2 + its GEMM3 slice consume
    static constexpr int NBP = (NB + D - 1) / D * D;   // ... padded so the register ring's phase is the same in every iteration
    static constexpr int NW1 = NHB * NQ1;
    static constexpr long S1 = 0;                                      // fragment index of the W1 stream
    static constexpr long S2 = NW1;                                    // ... of the GEMM2/3 stream
    static constexpr long NFRAG = NW1 + (long)NG * NHB * NBP + D;      // + D zero fragments the prefetch runs into
    static constexpr long PARAMS = NFRAG * FRAG;                       // float offset of b1 | e^s1 | b2 | e^s2 (natural channel order)
    // Row-split streams (coupling_mlp_f32_rs_kernel: 4 waves share a 32-pixel tile, wave w owns hidden blocks w, w+4, w+8, w+12 of both
    // hidden GEMMs and tap-row tiles w, w+4, ...): per wave one linear stream [4 x NQ1 W1 | 4 x 64 W2 | 16 x NTW x 4 W3] fragments.
    static constexpr int DR = 16;                                      // ring depth of the row-split kernel (registers are plentiful there)
    static constexpr int NTW = (NRT + 3) / 4;                          // tap-row tiles per wave
    static constexpr int NRS = 4 * NQ1 + 4 * 4 * NHB + NHB * NTW * 4;  // fragments per wave
    static constexpr long RS_OFF = PARAMS + 4 * HID;                   // float offset of wave 0's stream
    static constexpr long TOTAL_FLOATS = RS_OFF + 4L * NRS * FRAG;
    static_assert(C % 4 == 0, "rows of T are written as float4");
};

// ----------------------------------------------------------------------------- packer
template <int C>
__global__ void pack_mlp_f32_kernel(const float* __restrict__ w1, const float* __restrict__ scale1, const float* __restrict__ bias1,
                                    const float* __restrict__ w2, const float* __restrict__ scale2, const float* __restrict__ bias2,
                                    const float* __restrict__ w3, float* __restrict__ out) {
    using G = GeoF<C>;
    for (long e = blockIdx.x * (long)kThreads + threadIdx.x; e < G::TOTAL_FLOATS; e += (long)gridDim.x * kThreads) {
        float v = 0.f;
        if (e >= G::RS_OFF) {                                           // ---- the four row-split streams
            const long r = e - G::RS_OFF;
            const int w = (int)(r / ((long)G::NRS * FRAG));
            const int n = (int)(r / FRAG) % G::NRS;
            const int l = (int)(r % FRAG) / 4, i = (int)(r % 4);
            const int m = l & 31, h = l >> 5;
            if (n < 4 * G::NQ1) {                                       // W1 of hidden block w + 4 (n / NQ1)
                const int hb = w + 4 * (n / G::NQ1), q = n % G::NQ1;
                const int k = 2 * (4 * q + i) + h;
                if (k < G::K1) v = w1[((long)(32 * hb + m) * G::Ch + k % G::Ch) * 9 + k / G::Ch];
            } else if (n < 4 * G::NQ1 + 16 * NHB) {                     // W2 rows of hidden block w + 4 i2, input block hb, quarter q4
                const int j = n - 4 * G::NQ1;
                const int hb2 = w + 4 * (j / (4 * NHB)), hb = (j % (4 * NHB)) / 4, q4 = j % 4;
                v = w2[(long)(32 * hb2 + m) * HID + 32 * hb + 8 * q4 + 4 * h + i];
            } else {                                                    // W3: input block hb2, this wave's tile t (rows of tile w + 4 t), quarter q4
                const int j = n - 4 * G::NQ1 - 16 * NHB;
                const int hb2 = j / (4 * G::NTW), t = (j / 4) % G::NTW, q4 = j % 4;
                const int nrow = 32 * (w + 4 * t) + m, in = 32 * hb2 + 8 * q4 + 4 * h + i;
                if (nrow < G::NROW) v = w3[((long)(nrow % C) * HID + in) * 9 + nrow / C];
            }
        } else if (e >= G::PARAMS) {
            const int q = (int)(e - G::PARAMS), which = q / HID, ch = q % HID;
            v = which == 0 ? bias1[ch] : which == 1 ? expf(scale1[ch]) : which == 2 ? bias2[ch] : expf(scale2[ch]);
        } else {
            const long f = e / FRAG;
            const int l = (int)(e % FRAG) / 4, i = (int)(e % 4);
            const int m = l & 31, h = l >> 5;
            if (f < G::S2) {                                            // W1[hb][q]: k = 2*(4q+i) + h, columns (tap, ci)
                const int hb = (int)(f / G::NQ1), q = (int)(f % G::NQ1);
                const int k = 2 * (4 * q + i) + h;
                if (k < G::K1) v = w1[((long)(32 * hb + m) * G::Ch + k % G::Ch) * 9 + k / G::Ch];
            } else if (f < G::NFRAG - G::D) {
                const long r = f - G::S2;
                const int g = (int)(r / (NHB * G::NBP)), hb2 = (int)(r / G::NBP) % NHB, j = (int)(r % G::NBP);
                if (j < 4 * NHB) {                                      // W2[hb2][hb][q4]: input channel 32 hb + 8 q4 + 4 h + i
                    const int hb = j / 4, q4 = j % 4;
                    v = w2[(long)(32 * hb2 + m) * HID + 32 * hb + 8 * q4 + 4 * h + i];
                } else if (j < G::NB) {                                 // W3 slice [rtl][q4]: row n = tap*C + co, input channel of block hb2
                    const int rtl = (j - 4 * NHB) / 4, q4 = (j - 4 * NHB) % 4;
                    const int n = 32 * (g * G::NRTG + rtl) + m, in = 32 * hb2 + 8 * q4 + 4 * h + i;
                    if (n < G::NROW) v = w3[((long)(n % C) * HID + in) * 9 + n / C];
                }
            }
        }
        out[e] = v;
    }
}

// ----------------------------------------------------------------------------- helpers
template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ f32x4 ld_frag(const float* __restrict__ p) { return *reinterpret_cast<const f32x4*>(p); }

// relu((acc + b) * e^s) per row: lane half h holds rows drow(r, h) = 8 (r >> 2) + 4 h + (r & 3), i.e. four float4s of each parameter
// vector.  The parameters are REQUESTED at the top of a hidden block (they are older than every weight fragment of the block, so they
// are back long before its 256 MFMAs are done) - loaded in the epilogue itself they exposed a full L2 latency per block.
struct EpiRegs { f32x4 b[4], e[4]; };
__device__ __forceinline__ EpiRegs epi_load(const float* __restrict__ bvec, const float* __restrict__ evec, int ch0) {
    EpiRegs r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        r.b[q] = *reinterpret_cast<const f32x4*>(bvec + ch0 + 8 * q);
        r.e[q] = *reinterpret_cast<const f32x4*>(evec + ch0 + 8 * q);
    }
    return r;
}
__device__ __forceinline__ void epilogue(f32x16& acc, const EpiRegs& p) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * q + i] = fmaxf(__fmul_rn(__fadd_rn(acc[4 * q + i], p.b[q][i]), p.e[q][i]), 0.f);
}

// ----------------------------------------------------------------------------- the kernel
template <int C>
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(1, 1)))
coupling_mlp_f32_kernel(const float* __restrict__ x, int x_ld, long npix, int H, int W, const float* __restrict__ img,
                        float* __restrict__ T) {
    using G = GeoF<C>;
    constexpr int Ch = G::Ch, D = G::D;
    const int lane = threadIdx.x;
    const int h = lane >> 5, p = lane & 31;
    const long pix = (long)blockIdx.x * 32 + p;
    const bool live = pix < npix;
    const long pc = live ? pix : npix - 1;                               // clamped: absent pixels compute on a copy, store nothing
    const int xx = (int)(pc % W), yy = (int)((pc / W) % H);
    const float* __restrict__ params = img + G::PARAMS;

    // ---- the wave's B operands of GEMM1: k-pair jj of lane (h, p) is column k = 2 jj + h of the pixel's im2col row, k = tap*Ch + ci
    unsigned vmask = 0;
    int poff[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const bool ok = (unsigned)(yy + dy) < (unsigned)H && (unsigned)(xx + dx) < (unsigned)W;
        vmask |= ok ? 1u << tap : 0u;
        poff[tap] = ok ? (dy * W + dx) * x_ld : 0;
    }
    const float* __restrict__ xrow = x + pc * x_ld;
    float bx[G::KP1];
    sfor<G::KP1>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        constexpr int k0 = 2 * jj, k1 = 2 * jj + 1;
        constexpr int t0 = k0 < G::K1 ? k0 / Ch : 9, t1 = k1 < G::K1 ? k1 / Ch : 9;      // tap 9: beyond K1, never valid
        constexpr int c0 = k0 < G::K1 ? k0 % Ch : 0, c1 = k1 < G::K1 ? k1 % Ch : 0;
        const int tap = h ? t1 : t0;
        const bool ok = tap < 9 && ((vmask >> tap) & 1u);
        int off = 0;
        if constexpr (t0 < 9) { if (!h) off = poff[t0] + c0; }
        if constexpr (t1 < 9) { if (h) off = poff[t1] + c1; }
        const float v = xrow[ok ? off : 0];
        bx[jj] = ok ? v : 0.f;
    });

    // ---- GEMM1: 16 hidden blocks, A fragments streamed in order through the register ring
    const float* __restrict__ wp = img + (long)lane * 4;                  // this lane's 16 bytes of fragment 0
    f32x4 ring[D];
    sfor<D>([&](auto i) { ring[decltype(i)::value] = ld_frag(wp + (long)decltype(i)::value * FRAG); });
    f32x16 h1[NHB];
    sfor<NHB>([&](auto hbc) {
        constexpr int hb = decltype(hbc)::value;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const EpiRegs ep = epi_load(params, params + HID, 32 * hb + 4 * h);
        __builtin_amdgcn_sched_barrier(0);
        sfor<G::NQ1>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int f = hb * G::NQ1 + q;                            // fragment index in the whole stream
            const f32x4 a = ring[f % D];
            ring[f % D] = ld_frag(wp + (long)(f + D) * FRAG);             // the stream is one array: GEMM1 runs on into GEMM2/3's
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bx[4 * q + i], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);      // keep the refill where it is: under register pressure hipcc sinks it to its use
        });
        epilogue(acc, ep);
        h1[hb] = acc;
    });

    // ---- GEMM2 + GEMM3, one hidden block of H2 at a time; GEMM3's row tiles in groups of at most 7, one group per wave (blockIdx.y)
    constexpr int PH = 0;                                                 // ring phase at the start of the GEMM2/3 walk
    static_assert(4 * NHB + D <= G::NBP, "the refills of GEMM2's fragments stay inside the block");
    const int g = G::NG > 1 ? (int)blockIdx.y : 0;
    // Every wave of the launch walks the same weight image; in lockstep they would all ask the same L2 channel for the same KiB at the
    // same time (measured: 69 % of the fp32 matrix peak at the 16x16 level).  H2's blocks are independent, so wave t starts at block
    // t mod 16 and wraps: at any moment the chip's reads are spread over the whole image.  (GEMM3 then adds the blocks' contributions in
    // a rotated order: fp32 sums that differ between tiles by rounding only, fixed for a given pixel index.)
    const float* __restrict__ gbase = wp + ((long)G::S2 + (long)g * NHB * G::NBP) * FRAG;   // block b of this group: gbase + b * NBP * FRAG
    const int rot = (int)(blockIdx.x & (NHB - 1));
    const float* __restrict__ sp = gbase + (long)rot * G::NBP * FRAG;     // fragment j of the current block: sp + j * FRAG
    {                                                                     // the ring restarts at the first block of this wave's walk
#pragma unroll
        for (int i = 0; i < D; ++i) ring[(PH + i) % D] = ld_frag(sp + (long)i * FRAG);
    }
    {
        f32x16 tap[G::NRTG];
#pragma unroll
        for (int t = 0; t < G::NRTG; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) tap[t][r] = 0.f;
#pragma unroll 1
        for (int it = 0; it < NHB; ++it) {
            const int hb2 = (rot + it) & (NHB - 1);
            const float* __restrict__ spn = gbase + (long)((hb2 + 1) & (NHB - 1)) * G::NBP * FRAG;    // the next block of the walk
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const EpiRegs ep = epi_load(params + 2 * HID, params + 3 * HID, 32 * hb2 + 4 * h);
            __builtin_amdgcn_sched_barrier(0);
            sfor<4 * NHB>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr int hb = j / 4, q4 = j % 4;
                const f32x4 a = ring[(PH + j) % D];
                ring[(PH + j) % D] = ld_frag(sp + (long)(j + D) * FRAG);              // (j + D < NBP here: 64 + D <= NBP)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], h1[hb][4 * q4 + i], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            epilogue(acc, ep);
            sfor<G::NBP - 4 * NHB>([&](auto jc) {
                constexpr int j = 4 * NHB + decltype(jc)::value;
                const f32x4 a = ring[(PH + j) % D];
                if constexpr (j + D < G::NBP) ring[(PH + j) % D] = ld_frag(sp + (long)(j + D) * FRAG);
                else ring[(PH + j) % D] = ld_frag(spn + (long)(j + D - G::NBP) * FRAG);   // the ring runs on into the next block
                if constexpr (j < G::NB) {
                    constexpr int rtl = (j - 4 * NHB) / 4, q4 = (j - 4 * NHB) % 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) tap[rtl] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], acc[4 * q4 + i], tap[rtl], 0, 0, 0);
                } else {
                    asm volatile("" ::"v"(a));                            // padding fragment: loaded (keeps the ring phase), not used
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            sp = spn;
        }
        // ---- this group's tap contributions: lane (h, p) holds rows 32 rt + 8 q + 4 h + 0..3 of pixel p as one float4
        if (live) {
            float* __restrict__ trow = T + pix * (long)G::NROW;
#pragma unroll
            for (int t = 0; t < G::NRTG; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n0 = 32 * (g * G::NRTG + t) + 8 * q + 4 * h;
                    if (n0 < G::NROW) {
                        f32x4 v = {tap[t][4 * q], tap[t][4 * q + 1], tap[t][4 * q + 2], tap[t][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(trow + n0) = v;
                    }
                }
        }
    }
    // drain the ring (its last D loads hit the zero fragments behind the stream)
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const f32x4 r = ring[i];
        asm volatile("" ::"v"(r));
    }
}


// ----------------------------------------------------------------------------- row-split variant for small pixel counts
// With few pixels (the 4x4 level of a 512-image batch: 8192 pixels = 256 tiles) one wave per tile leaves three SIMDs of every CU idle,
// and C = 48 additionally computes H1 and H2 twice (two row groups).  Here FOUR waves share a 32-pixel tile and split the OUTPUT ROWS of
// every GEMM: wave w computes hidden blocks w, w+4, w+8, w+12 of GEMM1 and of GEMM2 and tap-row tiles w, w+4, ... of GEMM3 - no work is
// duplicated.  The hidden blocks meet in LDS exactly as they leave the accumulators ([block][lane][16 floats]: the reading lane is the
// writing lane, so no transposition), 64 KB (H2 takes H1's place after a barrier), three barriers per launch; a wave's B operands of a block are four
// ds_read_b128 per 16 MFMAs (1024 cycles), requested one block ahead.  Weights: each wave walks its own linear stream of the image
// (GeoF::RS_OFF) through a 16-deep register ring.  Per wave and flow at C = 48: 432 + 1024 + 1024 MFMAs (one-wave-per-tile: 7616).
template <int C>
__global__ void __launch_bounds__(256) coupling_mlp_f32_rs_kernel(const float* __restrict__ x, int x_ld, long npix, int H, int W,
                                                                  const float* __restrict__ img, float* __restrict__ T) {
    using G = GeoF<C>;
    constexpr int Ch = G::Ch, D = G::DR, NRS = G::NRS, NQ1 = G::NQ1, NTW = G::NTW;
    extern __shared__ __attribute__((aligned(16))) float rs_lds[];
    float* H1s = rs_lds;                                   // [16 blocks][64 lanes][16]
    float* H2s = rs_lds;                                   // ... H2 overwrites H1 once every wave has finished GEMM2
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, p = lane & 31;
    const long pix = (long)blockIdx.x * 32 + p;
    const bool live = pix < npix;
    const long pc = live ? pix : npix - 1;
    const int xx = (int)(pc % W), yy = (int)((pc / W) % H);
    const float* __restrict__ params = img + G::PARAMS;

    unsigned vmask = 0;
    int poff[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const bool ok = (unsigned)(yy + dy) < (unsigned)H && (unsigned)(xx + dx) < (unsigned)W;
        vmask |= ok ? 1u << tap : 0u;
        poff[tap] = ok ? (dy * W + dx) * x_ld : 0;
    }
    const float* __restrict__ xrow = x + pc * x_ld;
    float bx[G::KP1];
    sfor<G::KP1>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        constexpr int k0 = 2 * jj, k1 = 2 * jj + 1;
        constexpr int t0 = k0 < G::K1 ? k0 / Ch : 9, t1 = k1 < G::K1 ? k1 / Ch : 9;
        constexpr int c0 = k0 < G::K1 ? k0 % Ch : 0, c1 = k1 < G::K1 ? k1 % Ch : 0;
        const int tap = h ? t1 : t0;
        const bool ok = tap < 9 && ((vmask >> tap) & 1u);
        int off = 0;
        if constexpr (t0 < 9) { if (!h) off = poff[t0] + c0; }
        if constexpr (t1 < 9) { if (h) off = poff[t1] + c1; }
        const float v = xrow[ok ? off : 0];
        bx[jj] = ok ? v : 0.f;
    });

    const float* __restrict__ sp = img + G::RS_OFF + (long)w * NRS * FRAG + (long)lane * 4;     // fragment n of this wave: sp + n * FRAG
    f32x4 ring[D];
    sfor<D>([&](auto i) { ring[decltype(i)::value] = ld_frag(sp + (long)decltype(i)::value * FRAG); });
    auto next = [&](auto nc) {                              // consume fragment n, refill its slot with fragment n + D
        constexpr int n = decltype(nc)::value;
        const f32x4 a = ring[n % D];
        if constexpr (n + D < NRS) ring[n % D] = ld_frag(sp + (long)(n + D) * FRAG);
        return a;
    };
    float* mine = H1s + lane * 16;                          // this lane's 16 floats of block 0 (blocks are 1024 floats apart)

    // ---- GEMM1: hidden blocks w, w+4, w+8, w+12
    sfor<4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int hb = w + 4 * i;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const EpiRegs ep = epi_load(params, params + HID, 32 * hb + 4 * h);
        __builtin_amdgcn_sched_barrier(0);
        sfor<NQ1>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const f32x4 a = next(std::integral_constant<int, i * NQ1 + q>{});
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ii], bx[4 * q + ii], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        epilogue(acc, ep);
        float* dst = mine + hb * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *reinterpret_cast<f32x4*>(dst + 4 * q) = v;
        }
    });
    __syncthreads();

    // ---- GEMM2: hidden blocks w, w+4, ... of H2; B operands = H1 blocks from LDS, one block ahead
    constexpr int N1 = 4 * NQ1;
    auto load_blk = [&](const float* base, int blk, f32x4 (&bv)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = *reinterpret_cast<const f32x4*>(base + blk * 1024 + lane * 16 + 4 * q);
    };
    f32x16 acc2[4];                                        // this wave's four H2 blocks stay in registers until every wave has read H1
    sfor<4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int hb2 = w + 4 * i;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const EpiRegs ep = epi_load(params + 2 * HID, params + 3 * HID, 32 * hb2 + 4 * h);
        f32x4 bv[2][4];
        load_blk(H1s, 0, bv[0]);
        sfor<NHB>([&](auto hc) {
            constexpr int hb = decltype(hc)::value;
            if constexpr (hb + 1 < NHB) load_blk(H1s, hb + 1, bv[(hb + 1) & 1]);
            sfor<4>([&](auto qc) {
                constexpr int q4 = decltype(qc)::value;
                const f32x4 a = next(std::integral_constant<int, N1 + i * 4 * NHB + hb * 4 + q4>{});
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ii], bv[hb & 1][q4][ii], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        epilogue(acc, ep);
        acc2[i] = acc;
    });
    __syncthreads();                                       // H1 is dead: H2 takes its place (64 KB of LDS per workgroup, two workgroups per CU)
    sfor<4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        float* dst = H2s + lane * 16 + (w + 4 * i) * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc2[i][4 * q], acc2[i][4 * q + 1], acc2[i][4 * q + 2], acc2[i][4 * q + 3]};
            *reinterpret_cast<f32x4*>(dst + 4 * q) = v;
        }
    });
    __syncthreads();

    // ---- GEMM3: tap-row tiles w, w+4, ... over all 16 blocks of H2
    constexpr int N2 = N1 + 16 * NHB;
    f32x16 tap[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) tap[t][r] = 0.f;
    {
        f32x4 bv[2][4];
        load_blk(H2s, 0, bv[0]);
        sfor<NHB>([&](auto hc) {
            constexpr int hb2 = decltype(hc)::value;
            if constexpr (hb2 + 1 < NHB) load_blk(H2s, hb2 + 1, bv[(hb2 + 1) & 1]);
            sfor<NTW>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                sfor<4>([&](auto qc) {
                    constexpr int q4 = decltype(qc)::value;
                    const f32x4 a = next(std::integral_constant<int, N2 + (hb2 * NTW + t) * 4 + q4>{});
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) tap[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ii], bv[hb2 & 1][q4][ii], tap[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
    }
    if (live) {
        float* __restrict__ trow = T + pix * (long)G::NROW;
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = 32 * (w + 4 * t) + 8 * q + 4 * h;
                if (n0 < G::NROW) {
                    const f32x4 v = {tap[t][4 * q], tap[t][4 * q + 1], tap[t][4 * q + 2], tap[t][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(trow + n0) = v;
                }
            }
    }
}

// variant 0: by the pixel count (row split while one wave per tile would leave the chip under ~3 waves per CU), 1: one wave per tile,
// 2: row split
template <int C>
int launch_mlp(const float* x, int x_ld, long npix, int H, int W, const float* img, float* T, int variant, hipStream_t s) {
    const long nwg = cdiv(npix, 32);
    static const int forced = [] { const char* e = NFDP_DIAG_ENV("NFDP_F32_VARIANT"); return e ? atoi(e) : 0; }();     // A/B measurements
    if (variant == 0 && (forced == 1 || forced == 2)) variant = forced;
    const bool rs = variant == 2 || (variant == 0 && nwg * GeoF<C>::NG < 768);
    if (rs) {
        constexpr int kLds = NHB * 64 * 16 * (int)sizeof(float);                 // H1, then H2 in its place: 64 KB
        static NfdpPerDevice attr_set;
        if (!attr_set) {
            nfdp_func_attr(reinterpret_cast<const void*>(&coupling_mlp_f32_rs_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
            attr_set = true;
        }
        hipLaunchKernelGGL((coupling_mlp_f32_rs_kernel<C>), dim3((unsigned)nwg), dim3(256), kLds, s, x, x_ld, npix, H, W, img, T);
    } else {
        hipLaunchKernelGGL((coupling_mlp_f32_kernel<C>), dim3((unsigned)nwg, GeoF<C>::NG), dim3(kWave), 0, s, x, x_ld, npix, H, W, img, T);
    }
    return NFDP_OK;
}

template <int C>
int launch_pack(const float* w1, const float* s1, const float* b1, const float* w2, const float* s2, const float* b2, const float* w3,
                float* out, hipStream_t s) {
    hipLaunchKernelGGL((pack_mlp_f32_kernel<C>), dim3(stream_grid(GeoF<C>::TOTAL_FLOATS, kThreads)), dim3(kThreads), 0, s, w1, s1, b1, w2,
                       s2, b2, w3, out);
    return NFDP_OK;
}

}  // namespace

#define NFDP_MLP_F32_DISPATCH(C, CALL)                 \
    switch (C) {                                       \
        case 4: CALL(4); break;                        \
        case 8: CALL(8); break;                        \
        case 12: CALL(12); break;                      \
        case 16: CALL(16); break;                      \
        case 24: CALL(24); break;                      \
        case 48: CALL(48); break;                      \
        default: break;                                \
    }

extern "C" size_t nfdp_coupling_mlp_f32_packed_bytes(int C) {
    size_t n = 0;
#define CALL(CC) n = (size_t)GeoF<CC>::TOTAL_FLOATS * 4
    NFDP_MLP_F32_DISPATCH(C, CALL)
#undef CALL
    return n;
}

extern "C" int nfdp_coupling_mlp_f32_pack(const float* w1, const float* scale1, const float* bias1, const float* w2, const float* scale2,
                                          const float* bias2, const float* w3, void* packed, int C, nfdp_stream_t stream) {
    NFDP_REQUIRE(w1 && scale1 && bias1 && w2 && scale2 && bias2 && w3 && packed, NFDP_EINVAL, "coupling_mlp_f32_pack: null pointer");
    NFDP_REQUIRE(nfdp_coupling_mlp_f32_packed_bytes(C) != 0, NFDP_EUNSUPPORTED, "coupling_mlp_f32_pack: C=%d (supported: 4,8,12,16,24,48)", C);
    NFDP_REQUIRE(((uintptr_t)packed & 15) == 0, NFDP_ESHAPE, "coupling_mlp_f32_pack: packed must be 16-byte aligned");
    int rc = NFDP_OK;
#define CALL(CC) rc = launch_pack<CC>(w1, scale1, bias1, w2, scale2, bias2, w3, reinterpret_cast<float*>(packed), as_stream(stream))
    NFDP_MLP_F32_DISPATCH(C, CALL)
#undef CALL
    NFDP_CHECK_LAUNCH("coupling_mlp_f32_pack");
    return rc;
}

extern "C" int nfdp_coupling_mlp_f32_ex(const float* x, int x_ld, int B, int H, int W, const void* packed, float* T, int C, int variant,
                                        nfdp_stream_t stream);
extern "C" int nfdp_coupling_mlp_f32(const float* x, int x_ld, int B, int H, int W, const void* packed, float* T, int C,
                                     nfdp_stream_t stream) {
    return nfdp_coupling_mlp_f32_ex(x, x_ld, B, H, W, packed, T, C, 0, stream);
}

extern "C" int nfdp_coupling_mlp_f32_ex(const float* x, int x_ld, int B, int H, int W, const void* packed, float* T, int C, int variant,
                                        nfdp_stream_t stream) {
    NFDP_REQUIRE(x && packed && T, NFDP_EINVAL, "coupling_mlp_f32: null pointer");
    NFDP_REQUIRE(variant >= 0 && variant <= 2, NFDP_EINVAL, "coupling_mlp_f32: variant=%d (0 auto, 1 one wave per tile, 2 row split)", variant);
    NFDP_REQUIRE(nfdp_coupling_mlp_f32_packed_bytes(C) != 0, NFDP_EUNSUPPORTED, "coupling_mlp_f32: C=%d (supported: 4,8,12,16,24,48)", C);
    NFDP_REQUIRE(B >= 0 && H > 0 && W > 0 && x_ld >= C / 2, NFDP_ESHAPE, "coupling_mlp_f32: bad shape B=%d H=%d W=%d x_ld=%d", B, H, W, x_ld);
    NFDP_REQUIRE(((uintptr_t)packed & 15) == 0 && ((uintptr_t)T & 15) == 0, NFDP_ESHAPE, "coupling_mlp_f32: packed / T must be 16-byte aligned");
    const long npix = (long)B * H * W;
    if (npix == 0) return NFDP_OK;
    NFDP_REQUIRE(cdiv(npix, 32) < (1L << 31), NFDP_ESHAPE, "coupling_mlp_f32: too many pixels");
    int rc = NFDP_OK;
#define CALL(CC) rc = launch_mlp<CC>(x, x_ld, npix, H, W, reinterpret_cast<const float*>(packed), T, variant, as_stream(stream))
    NFDP_MLP_F32_DISPATCH(C, CALL)
#undef CALL
    NFDP_CHECK_LAUNCH("coupling_mlp_f32");
    return rc;
}
