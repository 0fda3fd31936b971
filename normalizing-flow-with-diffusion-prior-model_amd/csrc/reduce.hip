// ActNorm data-dependent initialisation: per-channel mean and UNBIASED std over all pixels of an NHWC tensor.
// Reference: ActNorm.transform, normalizing_flow/transforms.py:74-78
//     scale = -log(x.std(dim=(0,2,3)) + 1e-6),  bias = -x.mean(dim=(0,2,3))
// torch's CPU std accumulates fp32 data in double (Welford); here: two deterministic fp64 passes
// (sum -> mean, then sum of squared deviations), fixed slab partition and fixed combine order, so the result
// is run-to-run identical and independent of the launch.  Loads are coalesced: consecutive lanes read
// consecutive channels of a pixel row.
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kSlabs = 128;   // number of row slabs == grid size of the partial kernels

// partial[slab][c] = sum over the slab's rows of (x[r][c] - shift[c])^P   (P=1: shift ignored)
template <bool SQUARED>
__global__ void __launch_bounds__(kBlock) chan_partial_kernel(const float* __restrict__ x, int ld, long npix, int C,
                                                              const double* __restrict__ mean,
                                                              double* __restrict__ partial) {
    __shared__ double red[kBlock];
    const int cth = C < kBlock ? C : kBlock;      // channels covered per pass
    const int R = kBlock / cth;                   // rows covered per iteration
    const int ro = threadIdx.x / cth, c0 = threadIdx.x % cth;
    const bool active = ro < R;
    const long rows_per_slab = (npix + kSlabs - 1) / kSlabs;
    const long r_begin = blockIdx.x * rows_per_slab;
    long r_end = r_begin + rows_per_slab;
    if (r_end > npix) r_end = npix;
    for (int cb = 0; cb < C; cb += cth) {
        const int c = cb + c0;
        double acc = 0.0;
        if (active && c < C) {
            const double m = SQUARED ? mean[c] : 0.0;
            for (long r = r_begin + ro; r < r_end; r += R) {
                const double v = (double)x[r * ld + c];
                if (SQUARED) {
                    const double d = v - m;
                    acc += d * d;
                } else {
                    acc += v;
                }
            }
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (active && ro == 0 && c < C) {
            double s = 0.0;
            for (int j = 0; j < R; ++j) s += red[j * cth + c0];
            partial[(long)blockIdx.x * C + c] = s;
        }
        __syncthreads();
    }
}

__global__ void chan_mean_kernel(const double* __restrict__ partial, long npix, int C, double* __restrict__ mean) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < kSlabs; ++b) s += partial[(long)b * C + c];
    mean[c] = s / (double)npix;
}

__global__ void chan_finish_kernel(const double* __restrict__ partial, const double* __restrict__ mean, long npix, int C,
                                   float* __restrict__ scale, float* __restrict__ bias) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < kSlabs; ++b) s += partial[(long)b * C + c];
    const double var = s / (double)(npix - 1);            // unbiased (torch.std default)
    const float std_f = (float)sqrt(var);
    scale[c] = -logf(__fadd_rn(std_f, 1e-6f));
    bias[c] = -(float)mean[c];
}

}  // namespace

extern "C" size_t nfdp_actnorm_init_ws_bytes(int C) {
    return C > 0 ? ((size_t)kSlabs * C + C) * sizeof(double) : 0;
}

extern "C" int nfdp_actnorm_init_f32(const float* x, int ld, long npix, int C, float* scale, float* bias, void* ws,
                                     nfdp_stream_t stream) {
    NFDP_REQUIRE(x && scale && bias && ws, NFDP_EINVAL, "actnorm_init: null pointer");
    NFDP_REQUIRE(npix > 0 && C > 0 && ld >= C, NFDP_ESHAPE, "actnorm_init: bad shape npix=%ld C=%d ld=%d", npix, C, ld);
    hipStream_t s = as_stream(stream);
    double* partial = reinterpret_cast<double*>(ws);
    double* mean = partial + (size_t)kSlabs * C;
    const int gc = (int)cdiv(C, kBlock);
    hipLaunchKernelGGL((chan_partial_kernel<false>), dim3(kSlabs), dim3(kBlock), 0, s, x, ld, npix, C, nullptr, partial);
    hipLaunchKernelGGL(chan_mean_kernel, dim3(gc), dim3(kBlock), 0, s, partial, npix, C, mean);
    hipLaunchKernelGGL((chan_partial_kernel<true>), dim3(kSlabs), dim3(kBlock), 0, s, x, ld, npix, C, mean, partial);
    hipLaunchKernelGGL(chan_finish_kernel, dim3(gc), dim3(kBlock), 0, s, partial, mean, npix, C, scale, bias);
    NFDP_CHECK_LAUNCH("actnorm_init");
    return NFDP_OK;
}
