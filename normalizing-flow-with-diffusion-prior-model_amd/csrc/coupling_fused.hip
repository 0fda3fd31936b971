// Fused affine-coupling layer for gfx950: the whole coupling network + coupling epilogue in ONE launch.
//
// Reference arithmetic: AffineCoupling.transform/invert  normalizing_flow/transforms.py:179-185, :196-200
//                       coupling_network               normalizing_flow/utils.py:83-89
//                       Conv2dActNorm / ZeroConv2d     normalizing_flow/utils.py:43-44, :67-69
//
//   x_a --im2col--> GEMM1 (3x3, C/2->512) -> ActNorm+ReLU -> GEMM2 (1x1, 512->512) -> ActNorm+ReLU
//       -> GEMM3 (512 -> 9 taps x C, pointwise) -> shift-sum over the 3x3 taps (+bias)*exp(3*logs) -> (log_s, t)
//       -> s = sigmoid(log_s+2);  x_b <- (x_b+t)*s  [or y_b/(s+1e-6) - t];  ldj[b] += sum log(s+1e-6)
//
// Why this shape (MI355X-first, not a translation of the eager op list):
//  * The unfused path moves the 512-channel hidden tensors through HBM (2 x 134 MB per flow at B=512) and runs the
//    last conv as a K=4608, N=12..48 GEMM that is bound by im2col staging (20-70 TFLOP/s measured).  Here a
//    workgroup owns WHOLE images (256 or 128 pixels), every wave owns 64 or 32 pixels, and the hidden
//    activations never leave the wave's registers:
//      - all three GEMMs are computed TRANSPOSED (rows = output channel, cols = pixel on the lanes) with
//        v_mfma_f32_32x32x16_bf16, so a 32x32 accumulator tile of one GEMM is, after ActNorm+ReLU and a pairwise
//        bf16 convert, directly the B operand of the next GEMM (no LDS round trip, no transpose); the price is a
//        fixed permutation of the K order inside each 16-step, which is baked into the packed weights;
//      - the 3x3 output conv is a pointwise GEMM to 9*C "tap contributions" per pixel; the spatial shift-sum
//        happens once per layer through LDS (fp32), inside the workgroup because it owns whole images.
//  * Weights (A operands) are pre-packed in global memory as the exact MFMA register image
//    [k-step][lane][8 x bf16], so streaming them is a linear copy: global_load_lds_dwordx4 (direct to LDS, no
//    VGPRs) into a 2-deep ring, one 1-KiB piece per wave-instruction, and every A fetch is one conflict-free
//    linear ds_read_b128.  One fetch feeds NCT MFMAs.  The ring is shared by the 4 waves (one per SIMD; the
//    kernel uses the full 512-register budget: 256 for the hidden B fragments, up to 224 for the tap accumulators).
//  * bf16 MFMA inputs, fp32 accumulation, fp32 flow state: the same rounding points as the unfused bf16 path
//    (x_a, h1, h2, weights -> bf16), so the two agree to summation order.
//
// Supported geometry: hidden width 512, square images with H*W in {16, 64, 256} (every level of a 32x32 Glow),
// C in {4,8,16} (1-channel data) and {12,24,48} (3-channel data).  Other shapes use the unfused kernels.
#include <type_traits>
#include <utility>
#include "nfdp_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int PF = 4;            // A-fragment prefetch distance (k-steps in flight ahead of the MFMAs)
constexpr int HID = 512;
constexpr int NHB = HID / 32;

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int C>
struct Geo {
    static constexpr int Ch = C / 2;
    static constexpr int K1 = 9 * Ch;                      // GEMM1 depth (taps x in-channels)
    static constexpr int KS1 = (K1 + 15) / 16;             // 16-deep k-steps of GEMM1
    static constexpr int NROW = 9 * C;                     // GEMM3 rows: n = tap*C + co
    static constexpr int NRT = (NROW + 31) / 32;           // 32-row tiles of GEMM3
    static constexpr int BLOB_A = (KS1 + 1) * 1024;        // W1 image + 1 KiB params block
    static constexpr int BLOB_B = (32 + 2 * NRT + 1) * 1024;   // W2 image + W3 slice image + params block
    static constexpr int EPI = 1024;                       // bias3[C], exp(3*logs3)[C]
    static constexpr size_t TOTAL = (size_t)NHB * BLOB_A + (size_t)NHB * BLOB_B + EPI;
};

// D-tile row held by (reg, lane-half h) of a 32x32 MFMA accumulator
__host__ __device__ constexpr int drow(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
// hidden channel (within a 32-block) sitting at B-fragment position (k-step s, lane-half h, element j) when the
// fragment is an accumulator tile converted in place (regs 8s..8s+7)
__host__ __device__ constexpr int kperm(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

// ----------------------------------------------------------------------------- weight packer
template <int C>
__global__ void pack_fused_kernel(const float* __restrict__ w1, const float* __restrict__ scale1,
                                  const float* __restrict__ bias1, const float* __restrict__ w2,
                                  const float* __restrict__ scale2, const float* __restrict__ bias2,
                                  const float* __restrict__ w3, const float* __restrict__ bias3,
                                  const float* __restrict__ logs3, char* __restrict__ out) {
    using G = Geo<C>;
    constexpr long nA = (long)NHB * G::KS1 * 512;           // bf16 elements of the W1 images
    constexpr long nB2 = (long)NHB * 32 * 512;              // W2 images
    constexpr long nB3 = (long)NHB * 2 * G::NRT * 512;      // W3 images
    constexpr long nP = (long)NHB * 64;                     // params floats per GEMM (b[2][16], es[2][16])
    constexpr long total = nA + nB2 + nB3 + 2 * nP + 2 * C;
    char* blobA = out;
    char* blobB = out + (size_t)NHB * G::BLOB_A;
    char* epi = blobB + (size_t)NHB * G::BLOB_B;
    for (long i = blockIdx.x * (long)kThreads + threadIdx.x; i < total; i += (long)gridDim.x * kThreads) {
        long e = i;
        if (e < nA) {                                       // ---- W1: natural k order
            const int hb = (int)(e / (G::KS1 * 512));
            const int r0 = (int)(e % (G::KS1 * 512));
            const int ks = r0 / 512, l = (r0 % 512) / 8, j = r0 % 8;
            const int r = l & 31, h = l >> 5;
            const int o = 32 * hb + r, k = 16 * ks + 8 * h + j;
            float v = 0.f;
            if (k < G::K1) {
                const int tap = k / G::Ch, ci = k % G::Ch;
                v = w1[((long)o * G::Ch + ci) * 9 + tap];
            }
            reinterpret_cast<__bf16*>(blobA + (size_t)hb * G::BLOB_A)[r0] = (__bf16)v;
            continue;
        }
        e -= nA;
        if (e < nB2) {                                      // ---- W2: k order permuted to the accumulator image
            const int hb = (int)(e / (32 * 512));
            const int r0 = (int)(e % (32 * 512));
            const int ks = r0 / 512, l = (r0 % 512) / 8, j = r0 % 8;
            const int r = l & 31, h = l >> 5;
            const int o = 32 * hb + r, in = 32 * (ks >> 1) + kperm(ks & 1, h, j);
            reinterpret_cast<__bf16*>(blobB + (size_t)hb * G::BLOB_B)[r0] = (__bf16)w2[(long)o * HID + in];
            continue;
        }
        e -= nB2;
        if (e < nB3) {                                      // ---- W3 slice for hidden block hb: [s][rt][lane][8]
            const int hb = (int)(e / (2 * G::NRT * 512));
            const int r0 = (int)(e % (2 * G::NRT * 512));
            const int s = r0 / (G::NRT * 512), rt = (r0 / 512) % G::NRT, l = (r0 % 512) / 8, j = r0 % 8;
            const int r = l & 31, h = l >> 5;
            const int n = 32 * rt + r, in = 32 * hb + kperm(s, h, j);
            float v = 0.f;
            if (n < G::NROW) {
                const int tap = n / C, co = n % C;
                v = w3[((long)co * HID + in) * 9 + tap];
            }
            reinterpret_cast<__bf16*>(blobB + (size_t)hb * G::BLOB_B + 32 * 1024)[r0] = (__bf16)v;
            continue;
        }
        e -= nB3;
        if (e < 2 * nP) {                                   // ---- ActNorm params in accumulator-row order
            const bool second = e >= nP;
            if (second) e -= nP;
            const int hb = (int)(e / 64), q = (int)(e % 64);
            const int is_es = q / 32, h = (q % 32) / 16, reg = q % 16;
            const int ch = 32 * hb + drow(reg, h);
            const float* sc = second ? scale2 : scale1;
            const float* bi = second ? bias2 : bias1;
            char* base = second ? blobB + (size_t)hb * G::BLOB_B + (32 + 2 * G::NRT) * 1024
                                : blobA + (size_t)hb * G::BLOB_A + G::KS1 * 1024;
            reinterpret_cast<float*>(base)[q] = is_es ? expf(sc[ch]) : bi[ch];
            continue;
        }
        e -= 2 * nP;
        {                                                   // ---- epilogue params
            const int co = (int)(e % C);
            reinterpret_cast<float*>(epi)[e] = e < C ? bias3[co] : expf(__fmul_rn(logs3[co], 3.0f));
        }
    }
}

// ----------------------------------------------------------------------------- helpers
template <int NBYTES>
__device__ __forceinline__ void stream_blob(const char* __restrict__ g, char* lds, int wave, int lane) {
    // linear copy, one 1-KiB piece per wave-instruction, direct to LDS (no VGPR destination)
    constexpr int NP = NBYTES / 1024;
#pragma unroll
    for (int i = 0; i < (NP + 3) / 4; ++i) {
        const int piece = wave + 4 * i;
        if (piece < NP)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(lds + piece * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[8 * s + j];
    return r;
}

// (acc + b) * es, ReLU — ActNorm (transforms.py:80) + nn.ReLU on an accumulator tile; params in drow() order
__device__ __forceinline__ void actnorm_relu(f32x16& v, const char* params, int h) {
    const float4* pb = reinterpret_cast<const float4*>(params + h * 64);
    const float4* pe = reinterpret_cast<const float4*>(params + 128 + h * 64);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = pb[g], e = pe[g];
        v[4 * g + 0] = fmaxf(__fmul_rn(__fadd_rn(v[4 * g + 0], b.x), e.x), 0.f);
        v[4 * g + 1] = fmaxf(__fmul_rn(__fadd_rn(v[4 * g + 1], b.y), e.y), 0.f);
        v[4 * g + 2] = fmaxf(__fmul_rn(__fadd_rn(v[4 * g + 2], b.z), e.z), 0.f);
        v[4 * g + 3] = fmaxf(__fmul_rn(__fadd_rn(v[4 * g + 3], b.w), e.w), 0.f);
    }
}

constexpr int round16(int v) { return (v + 15) & ~15; }

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
// compile-time unrolled loop whose index is a constant expression (usable as an asm immediate)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// A-fragment stream: hand-issued LDS reads with hand-counted waits (cdna_hip_programming.md §5.7 form ii).
// hipcc's own waits for this pattern are always lgkmcnt(0), which with one wave per SIMD exposes a full LDS
// latency per group; LDS returns in order, so "all but the N youngest" is exact.
template <int OFF>
__device__ __forceinline__ void lds_read_frag(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait_frag(bf16x8& frag) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "i"(N));
    __builtin_amdgcn_sched_barrier(0);          // keep the consuming MFMAs below the wait (rule 18)
}

// Pipelined A stream of NA fragments starting at LDS byte address `base` (1 KiB apart): `use(i, frag)` is called
// for i = 0..NA-1 with fragment i ready; up to PF reads are in flight.
template <int NA, class Use>
__device__ __forceinline__ void a_stream(unsigned base, Use&& use) {
    bf16x8 ab[PF];
    static_for<(PF < NA ? PF : NA)>([&](auto i) { lds_read_frag<decltype(i)::value * 1024>(ab[decltype(i)::value], base); });
    static_for<NA>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int after = (NA - 1 - i) < (PF - 1) ? (NA - 1 - i) : (PF - 1);     // reads issued after fragment i
        lds_wait_frag<after>(ab[i % PF]);
        use(ic, ab[i % PF]);
        if constexpr (i + PF < NA) lds_read_frag<(i + PF) * 1024>(ab[i % PF], base);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Where an H1 fragment lives: the accumulator half of the register file (AGPR) holds the MFMA accumulators first;
// H1 fragments fill what is left, the rest stay in arch VGPRs.  Pinning the class at creation stops hipcc from
// parking them in scratch / shuffling them through v_mov before every MFMA.
template <int C, int NCT>
struct RegPlan {
    static constexpr int NACC = 16 * NCT * (1 + Geo<C>::NRT);                   // acc + accT registers
    static constexpr int AFREE = 256 - NACC - 8 > 0 ? 256 - NACC - 8 : 0;
    static constexpr int FRAGS_IN_AGPR = AFREE / 4 < 32 * NCT ? AFREE / 4 : 32 * NCT;
};
template <bool IN_AGPR>
__device__ __forceinline__ void pin_frag(bf16x8& f) {
    if constexpr (IN_AGPR) asm volatile("" : "+a"(f));
    else asm volatile("" : "+v"(f));
}

template <int C, int WD, int NCT>
struct Lds {
    using G = Geo<C>;
    static constexpr int P = WD * WD;
    static constexpr int M = 128 * NCT;
    static constexpr int XS = C + 1;
    static constexpr int XBYTES = round16(M * XS * 4);
    static constexpr int RED = 256;                                   // 16 doubles + slack
    static constexpr int RING = 2 * G::BLOB_B;
    static constexpr int T1 = M * (G::NRT * 32 + 1) * 4;              // one-pass tap buffer
    static constexpr int BUDGET = 160 * 1024 - XBYTES - RED;
    static constexpr int NPASS = T1 <= BUDGET ? 1 : 2;
    static constexpr int RT_PER_PASS = (G::NRT + NPASS - 1) / NPASS;
    static constexpr int TS = RT_PER_PASS * 32 + 1;                   // tap-buffer row stride (floats, odd)
    static constexpr int TBYTES = M * TS * 4;
    static constexpr int REGION = RING > TBYTES ? RING : TBYTES;
    static constexpr int TOTAL = XBYTES + RED + REGION;
    static_assert(TOTAL <= 160 * 1024, "LDS budget");
    static_assert(M % P == 0, "a workgroup owns whole images");
};

// ----------------------------------------------------------------------------- the kernel
template <int C, int WD, int NCT, bool INV>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1)))
coupling_fused_kernel(float* __restrict__ x, const char* __restrict__ packed, int B, double* __restrict__ ldj,
                      const float* __restrict__ ldj_const) {
    using G = Geo<C>;
    using L = Lds<C, WD, NCT>;
    constexpr int P = L::P, M = L::M, XS = L::XS, Ch = G::Ch;
    constexpr int IPW = M / P;                 // images per workgroup
    constexpr int NPW = 32 * NCT;              // pixels per wave

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* X = reinterpret_cast<float*>(smem);                        // [M][XS] fp32 flow state of this tile
    double* red = reinterpret_cast<double*>(smem + L::XBYTES);
    char* ring = smem + L::XBYTES + L::RED;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5;
    const long img0 = (long)blockIdx.x * IPW;
    const int n_img = (int)((B - img0) < IPW ? (B - img0) : IPW);
    const int npx = n_img * P;
    float* xg = x + img0 * P * C;
    const char* blobA = packed;
    const char* blobB = packed + (size_t)NHB * G::BLOB_A;
    const float* epi = reinterpret_cast<const float*>(blobB + (size_t)NHB * G::BLOB_B);

    // ---- 0. start the weight stream, stage the flow-state tile (coalesced float4 loads)
    stream_blob<G::BLOB_A>(blobA, ring, wave, lane);
    for (int i = tid; i < M * C / 4; i += kThreads) {
        const int e = i * 4;
        const int r = e / C, c = e % C;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < npx) v = reinterpret_cast<const float4*>(xg)[i];
        float* d = X + r * XS + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();

    // ---- 1. im2col B fragments of x_a for this wave's pixels (natural k = tap*Ch + ci, zero padding)
    bf16x8 bim[NCT][G::KS1];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int pl = wave * NPW + ct * 32 + (lane & 31);
        const int py = (pl % P) / WD, pxx = pl % WD;
#pragma unroll
        for (int ks = 0; ks < G::KS1; ++ks) {
            bf16x8 f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * ks + 8 * h + j;
                float v = 0.f;
                if (k < G::K1) {
                    const int tap = k / Ch, ci = k % Ch;
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    if ((unsigned)(py + dy) < (unsigned)WD && (unsigned)(pxx + dx) < (unsigned)WD)
                        v = X[(pl + dy * WD + dx) * XS + ci];
                }
                f[j] = (__bf16)v;
            }
            bim[ct][ks] = f;
        }
    }

    // ---- 2. GEMM1 + ActNorm + ReLU -> H1 kept as B fragments (k-steps 2*hb, 2*hb+1 of GEMM2)
    bf16x8 H1[NCT][2 * NHB];
    using RP = RegPlan<C, NCT>;
    static_for<NHB>([&](auto hbc) {
        constexpr int hb = decltype(hbc)::value;
        char* cur = ring + (hb & 1) * G::BLOB_B;
        char* nxt = ring + ((hb + 1) & 1) * G::BLOB_B;
        if (hb + 1 < NHB) stream_blob<G::BLOB_A>(blobA + (size_t)(hb + 1) * G::BLOB_A, nxt, wave, lane);
        else stream_blob<G::BLOB_B>(blobB, nxt, wave, lane);
        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const unsigned abase1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(cur) + lane * 16;
        a_stream<G::KS1>(abase1, [&](auto ks, const bf16x8& a) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bim[ct][decltype(ks)::value], acc[ct], 0, 0, 0);
        });
        static_for<NCT>([&](auto ctc) {
            constexpr int ct = decltype(ctc)::value;
            actnorm_relu(acc[ct], cur + G::KS1 * 1024, h);
            H1[ct][2 * hb] = pack8(acc[ct], 0);
            H1[ct][2 * hb + 1] = pack8(acc[ct], 1);
            // fragment order for the plan: f = ks * NCT + ct  (the last FRAGS_IN_AGPR fragments go to AGPRs)
            pin_frag<((2 * hb) * NCT + ct >= 32 * NCT - RP::FRAGS_IN_AGPR)>(H1[ct][2 * hb]);
            pin_frag<((2 * hb + 1) * NCT + ct >= 32 * NCT - RP::FRAGS_IN_AGPR)>(H1[ct][2 * hb + 1]);
        });
        __syncthreads();
    });

    // ---- 3. GEMM2 (+ActNorm+ReLU) chained into GEMM3 (tap contributions), hidden block by hidden block
    f32x16 accT[G::NRT][NCT];
#pragma unroll
    for (int rt = 0; rt < G::NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) accT[rt][ct][r] = 0.f;

    for (int hb = 0; hb < NHB; ++hb) {
        const char* cur = ring + (hb & 1) * G::BLOB_B;
        if (hb + 1 < NHB) stream_blob<G::BLOB_B>(blobB + (size_t)(hb + 1) * G::BLOB_B, ring + ((hb + 1) & 1) * G::BLOB_B, wave, lane);
        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        // A stream of one stage (contiguous in the blob): 32 W2 k-steps, then 2*NRT W3 fragments; the GEMM2
        // epilogue (ActNorm+ReLU, bf16 pack) runs at the seam while the first W3 fragments are already in flight.
        constexpr int NA = 32 + 2 * G::NRT;
        const unsigned abase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(cur) + lane * 16;
        bf16x8 H2[NCT][2];
        a_stream<NA>(abase, [&](auto ic, const bf16x8& a) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < 32) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, H1[ct][i], acc[ct], 0, 0, 0);
            } else {
                if constexpr (i == 32) {
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        actnorm_relu(acc[ct], cur + (32 + 2 * G::NRT) * 1024, h);
                        H2[ct][0] = pack8(acc[ct], 0);
                        H2[ct][1] = pack8(acc[ct], 1);
                    }
                }
                constexpr int sidx = (i - 32) / G::NRT, rt = (i - 32) % G::NRT;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    accT[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, H2[ct][sidx], accT[rt][ct], 0, 0, 0);
            }
        });
        __syncthreads();
    }

    // ---- 4. shift-sum of the tap contributions through LDS (the ring is free now), then the coupling epilogue
    float* T = reinterpret_cast<float*>(ring);                         // [M][TS]
    constexpr int NG = kThreads / M;                                   // channel groups per pixel (1 or 2)
    constexpr int CPT = Ch / NG;                                       // (log_s, t) pairs per thread
    static_assert(Ch % NG == 0, "channel split");
    const int p = tid % M, cg = tid / M;
    const int py = (p % P) / WD, pxx = p % WD;
    float sls[CPT], st[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) { sls[i] = 0.f; st[i] = 0.f; }

#pragma unroll
    for (int pass = 0; pass < L::NPASS; ++pass) {
        constexpr int RPP = L::RT_PER_PASS;
        const int n0 = pass * RPP * 32;
#pragma unroll
        for (int rt = 0; rt < G::NRT; ++rt) {
            if (rt / RPP != pass) continue;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                float* row = T + (wave * NPW + ct * 32 + (lane & 31)) * L::TS + (rt * 32 - n0);
#pragma unroll
                for (int r = 0; r < 16; ++r) row[drow(r, 0) + 4 * h] = accT[rt][ct][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if ((unsigned)(py + dy) < (unsigned)WD && (unsigned)(pxx + dx) < (unsigned)WD) {
                const float* row = T + (p + dy * WD + dx) * L::TS - n0;
#pragma unroll
                for (int i = 0; i < CPT; ++i) {
                    const int n_ls = tap * C + cg * CPT + i, n_t = n_ls + Ch;
                    if (n_ls >= n0 && n_ls < n0 + RPP * 32) sls[i] += row[n_ls];
                    if (n_t >= n0 && n_t < n0 + RPP * 32) st[i] += row[n_t];
                }
            }
        }
        __syncthreads();
    }

    double lsum = 0.0;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = cg * CPT + i;
        const float log_s = __fmul_rn(__fadd_rn(sls[i], epi[c]), epi[C + c]);             // (conv+bias)*exp(3 logs)
        const float t = __fmul_rn(__fadd_rn(st[i], epi[Ch + c]), epi[C + Ch + c]);
        const float s = 1.0f / (1.0f + expf(-__fadd_rn(log_s, 2.0f)));                   // sigmoid(log_s + 2)
        float* xb = X + p * XS + Ch + c;
        if (!INV) {
            *xb = __fmul_rn(__fadd_rn(*xb, t), s);
            lsum += (double)logf(__fadd_rn(s, 1e-6f));
        } else {
            *xb = __fsub_rn(__fdiv_rn(*xb, __fadd_rn(s, 1e-6f)), t);
        }
    }

    if (!INV && ldj != nullptr) {
        // per-image sum: shuffle tree over the lanes of one image segment, then a fixed-order combine
        constexpr int SEG = P < 64 ? P : 64;
#pragma unroll
        for (int o = SEG / 2; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
        if ((lane % SEG) == 0) red[tid / SEG] = lsum;
        __syncthreads();
        if (tid < n_img) {
            double s = 0.0;
            for (int g = 0; g < kThreads / SEG; ++g)
                if (((g * SEG) % M) / P == tid) s += red[g];
            double v = ldj[img0 + tid];
            if (ldj_const) v = (v + (double)ldj_const[0]) + (double)ldj_const[1];
            ldj[img0 + tid] = v + s;
        }
    }
    __syncthreads();

    // ---- 5. write the tile back (x_a unchanged, x_b updated), coalesced float4
    for (int i = tid; i < npx * C / 4; i += kThreads) {
        const int e = i * 4;
        const float* s = X + (e / C) * XS + (e % C);
        reinterpret_cast<float4*>(xg)[i] = make_float4(s[0], s[1], s[2], s[3]);
    }
}

template <int C, int WD, int NCT>
int launch_fused(float* x, const void* packed, int B, double* ldj, const float* ldj_const, int inverse, hipStream_t s) {
    using L = Lds<C, WD, NCT>;
    constexpr int IPW = L::M / L::P;
    const int grid = (B + IPW - 1) / IPW;
    static bool attr_set[2] = {false, false};
    auto kf = coupling_fused_kernel<C, WD, NCT, false>;
    auto ki = coupling_fused_kernel<C, WD, NCT, true>;
    if (!attr_set[inverse ? 1 : 0]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(inverse ? ki : kf), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  L::TOTAL);
        attr_set[inverse ? 1 : 0] = true;
    }
    if (inverse)
        hipLaunchKernelGGL(ki, dim3(grid), dim3(kThreads), L::TOTAL, s, x, reinterpret_cast<const char*>(packed), B, ldj, ldj_const);
    else
        hipLaunchKernelGGL(kf, dim3(grid), dim3(kThreads), L::TOTAL, s, x, reinterpret_cast<const char*>(packed), B, ldj, ldj_const);
    return NFDP_OK;
}

template <int C>
int launch_pack(const float* w1, const float* s1, const float* b1, const float* w2, const float* s2, const float* b2,
                const float* w3, const float* b3, const float* l3, void* packed, hipStream_t s) {
    hipLaunchKernelGGL((pack_fused_kernel<C>), dim3(1024), dim3(kThreads), 0, s, w1, s1, b1, w2, s2, b2, w3, b3, l3,
                       reinterpret_cast<char*>(packed));
    return NFDP_OK;
}

}  // namespace

extern "C" int nfdp_coupling_fused_supported(int H, int W, int C, int n_features) {
    if (n_features != HID || H != W) return 0;
    return (H == 16 && (C == 4 || C == 12)) || (H == 8 && (C == 8 || C == 24)) || (H == 4 && (C == 16 || C == 48));
}

extern "C" size_t nfdp_coupling_fused_packed_bytes(int C) {
    switch (C) {
        case 4: return Geo<4>::TOTAL;
        case 8: return Geo<8>::TOTAL;
        case 16: return Geo<16>::TOTAL;
        case 12: return Geo<12>::TOTAL;
        case 24: return Geo<24>::TOTAL;
        case 48: return Geo<48>::TOTAL;
        default: return 0;
    }
}

extern "C" int nfdp_coupling_fused_pack(const float* w1, const float* scale1, const float* bias1, const float* w2,
                                        const float* scale2, const float* bias2, const float* w3, const float* bias3,
                                        const float* logs3, void* packed, int C, nfdp_stream_t stream) {
    NFDP_REQUIRE(w1 && scale1 && bias1 && w2 && scale2 && bias2 && w3 && bias3 && logs3 && packed, NFDP_EINVAL,
                 "coupling_fused_pack: null pointer");
    hipStream_t s = as_stream(stream);
    switch (C) {
        case 4: launch_pack<4>(w1, scale1, bias1, w2, scale2, bias2, w3, bias3, logs3, packed, s); break;
        case 8: launch_pack<8>(w1, scale1, bias1, w2, scale2, bias2, w3, bias3, logs3, packed, s); break;
        case 16: launch_pack<16>(w1, scale1, bias1, w2, scale2, bias2, w3, bias3, logs3, packed, s); break;
        case 12: launch_pack<12>(w1, scale1, bias1, w2, scale2, bias2, w3, bias3, logs3, packed, s); break;
        case 24: launch_pack<24>(w1, scale1, bias1, w2, scale2, bias2, w3, bias3, logs3, packed, s); break;
        case 48: launch_pack<48>(w1, scale1, bias1, w2, scale2, bias2, w3, bias3, logs3, packed, s); break;
        default: nfdp_set_error("coupling_fused_pack: unsupported C=%d", C); return NFDP_EUNSUPPORTED;
    }
    NFDP_CHECK_LAUNCH("coupling_fused_pack");
    return NFDP_OK;
}

extern "C" int nfdp_coupling_fused_bf16(float* x, const void* packed, int B, int H, int W, int C, double* ldj,
                                        const float* ldj_const, int inverse, nfdp_stream_t stream) {
    NFDP_REQUIRE(x && packed, NFDP_EINVAL, "coupling_fused: null pointer");
    NFDP_REQUIRE(B >= 0, NFDP_ESHAPE, "coupling_fused: bad batch");
    NFDP_REQUIRE(nfdp_coupling_fused_supported(H, W, C, HID), NFDP_EUNSUPPORTED,
                 "coupling_fused: unsupported geometry H=%d W=%d C=%d (use the unfused kernels)", H, W, C);
    if (B == 0) return NFDP_OK;
    hipStream_t s = as_stream(stream);
    switch (C) {
        case 4: launch_fused<4, 16, 2>(x, packed, B, ldj, ldj_const, inverse, s); break;
        case 12: launch_fused<12, 16, 2>(x, packed, B, ldj, ldj_const, inverse, s); break;
        case 8: launch_fused<8, 8, 1>(x, packed, B, ldj, ldj_const, inverse, s); break;
        case 24: launch_fused<24, 8, 1>(x, packed, B, ldj, ldj_const, inverse, s); break;
        case 16: launch_fused<16, 4, 1>(x, packed, B, ldj, ldj_const, inverse, s); break;
        case 48: launch_fused<48, 4, 1>(x, packed, B, ldj, ldj_const, inverse, s); break;
    }
    NFDP_CHECK_LAUNCH("coupling_fused");
    return NFDP_OK;
}
