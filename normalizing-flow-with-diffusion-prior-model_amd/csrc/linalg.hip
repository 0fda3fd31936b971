// Batched log|det W| and W^-1 for the invertible 1x1 convolutions, in fp64.
// Reference: InvConv2d.transform  transforms.py:131  torch.slogdet(W.double())[1].float()
//            InvConv2d.invert     transforms.py:144  W.inverse()            (fp32 LAPACK there; fp64 here)
// The reference recomputes both on every call (48 slogdet per forward at L3K16); here ONE launch handles every
// flow's matrix: one 256-thread workgroup per matrix runs Gauss-Jordan with partial pivoting on [W | I] held in
// LDS (C <= 64: 64 KB) or in a caller-provided global workspace (C up to 256, L2-resident).  Deterministic:
// pivot ties go to the lowest row, every sum has a fixed order.
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxLdsC = 64;

template <bool IN_LDS>
__global__ void __launch_bounds__(kBlock) gauss_jordan_kernel(const float* __restrict__ w, int C,
                                                              float* __restrict__ logabsdet, float* __restrict__ winv,
                                                              double* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int W2 = 2 * C;
    double* A = IN_LDS ? sm : ws + (size_t)blockIdx.x * C * W2;
    double* fcol = IN_LDS ? sm + (size_t)C * W2 : sm;            // [C] elimination factors
    double* rval = fcol + C;                                      // [kBlock]
    int* ridx = reinterpret_cast<int*>(rval + kBlock);            // [kBlock]
    const float* wm = w + (size_t)blockIdx.x * C * C;
    const int tid = threadIdx.x;

    for (int e = tid; e < C * W2; e += kBlock) {
        const int i = e / W2, j = e % W2;
        A[e] = j < C ? (double)wm[i * C + j] : (j - C == i ? 1.0 : 0.0);
    }
    double logdet = 0.0;
    __syncthreads();

    for (int k = 0; k < C; ++k) {
        // 1. pivot = argmax_{i>=k} |A[i][k]|, lowest index on ties
        double best = -1.0;
        int bi = C;
        for (int i = k + tid; i < C; i += kBlock) {
            const double v = fabs(A[i * W2 + k]);
            if (v > best) { best = v; bi = i; }
        }
        rval[tid] = best;
        ridx[tid] = bi;
        __syncthreads();
        for (int s = kBlock / 2; s > 0; s >>= 1) {
            if (tid < s) {
                const double v = rval[tid + s];
                const int o = ridx[tid + s];
                if (v > rval[tid] || (v == rval[tid] && o < ridx[tid])) { rval[tid] = v; ridx[tid] = o; }
            }
            __syncthreads();
        }
        const int p = ridx[0];
        // 2. swap rows k <-> p, then normalise row k by the pivot
        const double pv = A[p * W2 + k];
        __syncthreads();                                           // everyone has read pv before row p changes
        if (p != k) {
            for (int j = tid; j < W2; j += kBlock) {
                const double a = A[k * W2 + j], b = A[p * W2 + j];
                A[k * W2 + j] = b;
                A[p * W2 + j] = a;
            }
        }
        if (tid == 0) logdet += log(fabs(pv));
        __syncthreads();
        for (int j = tid; j < W2; j += kBlock) A[k * W2 + j] = A[k * W2 + j] / pv;
        for (int i = tid; i < C; i += kBlock) fcol[i] = (i == k) ? 0.0 : A[i * W2 + k];
        __syncthreads();
        // 3. eliminate column k from every other row
        for (int e = tid; e < C * W2; e += kBlock) {
            const int i = e / W2, j = e % W2;
            const double f = fcol[i];
            if (f != 0.0) A[e] -= f * A[k * W2 + j];
        }
        __syncthreads();
    }
    if (tid == 0) logabsdet[blockIdx.x] = (float)logdet;
    if (winv) {
        float* out = winv + (size_t)blockIdx.x * C * C;
        for (int e = tid; e < C * C; e += kBlock) out[e] = (float)A[(e / C) * W2 + C + (e % C)];
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
    const size_t red = (size_t)C * sizeof(double) + kBlock * (sizeof(double) + sizeof(int));
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
