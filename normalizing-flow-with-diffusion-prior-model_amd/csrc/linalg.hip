// Batched log|det W| and W^-1 for the invertible 1x1 convolutions, in fp64.
// Reference: InvConv2d.transform  transforms.py:131  torch.slogdet(W.double())[1].float()
//            InvConv2d.invert     transforms.py:144  W.inverse()            (fp32 LAPACK there; fp64 here)
// The reference recomputes both on every call (48 slogdet per forward at L3K16); here ONE launch handles every
// flow's matrix of a level: one 256-thread workgroup per matrix runs Gauss-Jordan with partial pivoting on [W | I]
// held in LDS (C <= 64: 64 KB) or in a caller-provided global workspace (C up to 256, L2-resident).
// Deterministic: pivot ties go to the lowest row, every sum has a fixed order.
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxLdsC = 64;

// Gauss-Jordan on [W | I] with IMPLICIT partial pivoting (no row swaps): step k picks, among the rows not yet used as
// a pivot, the one with the largest |A[r][k]| (lowest r on ties), normalises it and eliminates column k from every
// other row.  After C steps physical row rowof[k] carries e_k on the left and row k of W^-1 on the right.
// Three barriers per step: (1) per-wave shuffle argmax -> LDS, (2) pivot row / column factors staged, (3) update.
template <bool IN_LDS>
__global__ void __launch_bounds__(kBlock) gauss_jordan_kernel(const float* __restrict__ w, int C,
                                                              float* __restrict__ logabsdet, float* __restrict__ winv,
                                                              double* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int W2 = 2 * C;
    double* A = IN_LDS ? sm : ws + (size_t)blockIdx.x * C * W2;
    double* rowk = IN_LDS ? sm + (size_t)C * W2 : sm;             // [2C] normalised pivot row
    double* fcol = rowk + W2;                                      // [C]  elimination factors
    double* wval = fcol + C;                                       // [4]  per-wave argmax value
    int* widx = reinterpret_cast<int*>(wval + 4);                  // [4]  per-wave argmax row
    int* rowof = widx + 4;                                         // [C]  pivot row of step k
    int* used = rowof + C;                                         // [C]
    const float* wm = w + (size_t)blockIdx.x * C * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int e = tid; e < C * W2; e += kBlock) {
        const int i = e / W2, j = e % W2;
        A[e] = j < C ? (double)wm[i * C + j] : (j - C == i ? 1.0 : 0.0);
    }
    for (int i = tid; i < C; i += kBlock) used[i] = 0;
    double logdet = 0.0;
    const int i0 = tid / W2, j0 = tid % W2, di = kBlock / W2, dj = kBlock % W2;
    __syncthreads();

    for (int k = 0; k < C; ++k) {
        // 1. pivot search: row r = tid (C <= 256), shuffle argmax inside each wave, lowest row wins ties
        double best = -1.0;
        int bi = 1 << 30;
        if (tid < C && !used[tid]) { best = fabs(A[tid * W2 + k]); bi = tid; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { wval[wave] = best; widx[wave] = bi; }
        __syncthreads();
        int p = widx[0];
        double pb = wval[0];
#pragma unroll
        for (int q = 1; q < kBlock / 64; ++q)
            if (wval[q] > pb || (wval[q] == pb && widx[q] < p)) { pb = wval[q]; p = widx[q]; }
        // 2. stage the normalised pivot row and the column factors (reads only; nothing is modified yet)
        const double pv = A[p * W2 + k];
        for (int j = tid; j < W2; j += kBlock) rowk[j] = A[p * W2 + j] / pv;
        for (int i = tid; i < C; i += kBlock) fcol[i] = (i == p) ? 0.0 : A[i * W2 + k];
        if (tid == 0) { logdet += log(fabs(pv)); rowof[k] = p; used[p] = 1; }
        __syncthreads();
        // 3. update every element; (i, j) advance incrementally (no runtime div/mod in the hot loop)
        {
            int i = i0, j = j0;
            for (int e = tid; e < C * W2; e += kBlock) {
                if (i == p) A[e] = rowk[j];
                else {
                    const double f = fcol[i];
                    if (f != 0.0) A[e] -= f * rowk[j];
                }
                i += di; j += dj;
                if (j >= W2) { j -= W2; ++i; }
            }
        }
        __syncthreads();
    }
    if (tid == 0) logabsdet[blockIdx.x] = (float)logdet;
    if (winv) {
        float* out = winv + (size_t)blockIdx.x * C * C;
        for (int e = tid; e < C * C; e += kBlock) out[e] = (float)A[rowof[e / C] * W2 + C + (e % C)];
    }
}

}  // namespace

extern "C" size_t nfdp_invconv_prepare_ws_bytes(int n, int C) {
    if (n <= 0 || C <= kMaxLdsC) return 16;    // LDS-resident; a token allocation keeps callers uniform
    return (size_t)n * C * 2 * C * sizeof(double);
}

extern "C" int nfdp_invconv_prepare_f32(const float* w, int n, int C, float* logabsdet, float* winv, void* ws,
                                        nfdp_stream_t stream) {
    NFDP_REQUIRE(w && logabsdet, NFDP_EINVAL, "invconv_prepare: null pointer");
    NFDP_REQUIRE(n >= 0 && C > 0 && C <= 256, NFDP_ESHAPE, "invconv_prepare: bad shape n=%d C=%d (C<=256)", n, C);
    if (n == 0) return NFDP_OK;
    const size_t red = (size_t)(3 * C + 4) * sizeof(double) + (size_t)(2 * C + 4) * sizeof(int) + 16;
    if (C <= kMaxLdsC) {
        const size_t lds = (size_t)C * 2 * C * sizeof(double) + red;
        hipLaunchKernelGGL((gauss_jordan_kernel<true>), dim3(n), dim3(kBlock), lds, as_stream(stream), w, C, logabsdet,
                           winv, nullptr);
    } else {
        NFDP_REQUIRE(ws, NFDP_EINVAL, "invconv_prepare: workspace required for C=%d", C);
        hipLaunchKernelGGL((gauss_jordan_kernel<false>), dim3(n), dim3(kBlock), red, as_stream(stream), w, C, logabsdet,
                           winv, reinterpret_cast<double*>(ws));
    }
    NFDP_CHECK_LAUNCH("invconv_prepare");
    return NFDP_OK;
}
