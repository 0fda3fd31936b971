// Fused optimizer step on flat fp32 buffers: clip_grad_value_ + clip_grad_norm_ + Adam / AdamW in two launches, no
// host synchronisation (the global norm stays on the device).
// Reference: normalizing_flow/trainer.py:165-167 (torch.nn.utils.clip_grad_value_(flow.parameters(), 1),
//            clip_grad_norm_(flow.parameters(), 1), optimizer.step()) with the optimizers of utils.py:120-137
//            (torch.optim.Adam / AdamW, default betas/eps/weight decay).
// HBM-bound: pass 1 reads and rewrites the clipped part of the gradient (8 B/element), pass 2 reads p, g, m, v and
// writes p, m, v (28 B/element); 22 M parameters -> ~0.8 GB per step.
#include "nfdp_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxPartials = 1024;

// g[i] = clamp(scale * g[i], -clip, clip) for i < n_clip (clip <= 0: no clamp); partial[blk] = sum of the clamped squares.
// scale = 1/world turns the all-reduced SUM of the data-parallel ranks into the average in the pass that reads the gradient anyway.
__global__ void __launch_bounds__(kBlock) clip_value_sumsq_kernel(float* __restrict__ g, long n_clip, float clip, float scale,
                                                                 double* __restrict__ partial) {
    __shared__ double red[kBlock / 64];
    double acc = 0.0;
    const bool touch = clip > 0.f || scale != 1.0f;
    auto one = [&](float v) -> float {
        if (scale != 1.0f) v *= scale;
        if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
        acc += (double)v * (double)v;
        return v;
    };
    // 16-byte accesses where the buffer allows it; a thread adds its elements' squares in index order as before, only the assignment
    // of elements to threads differs from the element-by-element loop (the partial sums are fp64)
    const long n4 = (((uintptr_t)g & 15) == 0) ? n_clip / 4 : 0;
    for (long q = blockIdx.x * (long)kBlock + threadIdx.x; q < n4; q += (long)gridDim.x * kBlock) {
        float4 v = reinterpret_cast<const float4*>(g)[q];
        v.x = one(v.x); v.y = one(v.y); v.z = one(v.z); v.w = one(v.w);
        if (touch) reinterpret_cast<float4*>(g)[q] = v;
    }
    for (long i = 4 * n4 + blockIdx.x * (long)kBlock + threadIdx.x; i < n_clip; i += (long)gridDim.x * kBlock) {
        const float v = one(g[i]);
        if (touch) g[i] = v;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// torch.optim.Adam / AdamW (amsgrad=False, maximize=False) on elements [0, n); the first n_clip elements of g are
// first scaled by min(1, max_norm / (||g[:n_clip]|| + 1e-6)) (clip_grad_norm_, max_norm <= 0: off).
__global__ void __launch_bounds__(kBlock) adam_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, long n, long n_clip,
                                                          const double* __restrict__ partial, int n_partial, float max_norm,
                                                          float lr, float beta1, float beta2, float eps, float weight_decay,
                                                          int adamw, float bias_c1, float bias_c2_sqrt,
                                                          float* __restrict__ norm_out, const double* __restrict__ hyper,
                                                          float grad_scale, long scaled_upto) {
    __shared__ float coef_s, hyp_s[3];
    if (threadIdx.x == 0) {
        if (hyper != nullptr) {                  // device-resident learning rate and step count: the launch is replayable in a hipGraph
            const double step = hyper[1];
            hyp_s[0] = (float)hyper[0];
            hyp_s[1] = (float)(1.0 - pow((double)beta1, step));
            hyp_s[2] = (float)sqrt(1.0 - pow((double)beta2, step));
        } else {
            hyp_s[0] = lr; hyp_s[1] = bias_c1; hyp_s[2] = bias_c2_sqrt;
        }
        float coef = 1.0f;
        if (max_norm > 0.f && n_clip > 0) {
            double s = 0.0;
            for (int q = 0; q < n_partial; ++q) s += partial[q];          // fixed order: the same value in every block
            const float total = (float)sqrt(s);
            coef = fminf(max_norm / (total + 1e-6f), 1.0f);
            if (blockIdx.x == 0 && norm_out) norm_out[0] = total;
        }
        coef_s = coef;
    }
    __syncthreads();
    const float coef = coef_s;
    lr = hyp_s[0];
    bias_c2_sqrt = hyp_s[2];
    const float step_size = lr / hyp_s[1];
    // one element: the in-place gradient scalings, then torch's Adam expressions; returns whether g changed
    auto elem = [&](long i, float& gi, float& pi, float& mi, float& vi) -> bool {
        bool gch = false;
        if (i >= scaled_upto && grad_scale != 1.0f) { gi *= grad_scale; gch = true; }     // elements the clip pass did not visit: 1/world here
        if (i < n_clip && coef != 1.0f) { gi *= coef; gch = true; }                         // clip_grad_norm_ scales the gradients in place
        if (weight_decay != 0.f) {
            if (adamw) pi *= 1.0f - lr * weight_decay;
            else gi = fmaf(weight_decay, pi, gi);
        }
        mi = fmaf(beta1, mi, (1.0f - beta1) * gi);
        vi = fmaf(beta2, vi, (1.0f - beta2) * gi * gi);
        const float denom = sqrtf(vi) / bias_c2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        return gch;
    };
    // 16-byte accesses where the four buffers allow it (flat optimizer buffers do), the n % 4 tail element by element
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    const long n4 = vec ? n / 4 : 0;
    for (long q = blockIdx.x * (long)kBlock + threadIdx.x; q < n4; q += (long)gridDim.x * kBlock) {
        float4 g4 = reinterpret_cast<const float4*>(g)[q], p4 = reinterpret_cast<const float4*>(p)[q];
        float4 m4 = reinterpret_cast<const float4*>(m)[q], v4 = reinterpret_cast<const float4*>(v)[q];
        float g4a[4] = {g4.x, g4.y, g4.z, g4.w};                            // gradient as scaled in place (before weight decay is folded in)
        float gi[4] = {g4.x, g4.y, g4.z, g4.w}, pi[4] = {p4.x, p4.y, p4.z, p4.w}, mi[4] = {m4.x, m4.y, m4.z, m4.w}, vi[4] = {v4.x, v4.y, v4.z, v4.w};
        bool gch = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long i = 4 * q + e;
            if (i >= scaled_upto && grad_scale != 1.0f) { g4a[e] *= grad_scale; gch = true; }
            if (i < n_clip && coef != 1.0f) { g4a[e] *= coef; gch = true; }
            elem(i, gi[e], pi[e], mi[e], vi[e]);
        }
        if (gch) reinterpret_cast<float4*>(g)[q] = make_float4(g4a[0], g4a[1], g4a[2], g4a[3]);
        reinterpret_cast<float4*>(p)[q] = make_float4(pi[0], pi[1], pi[2], pi[3]);
        reinterpret_cast<float4*>(m)[q] = make_float4(mi[0], mi[1], mi[2], mi[3]);
        reinterpret_cast<float4*>(v)[q] = make_float4(vi[0], vi[1], vi[2], vi[3]);
    }
    for (long i = 4 * n4 + blockIdx.x * (long)kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        float gi = g[i], pi = p[i], mi = m[i], vi = v[i];
        float gs = gi;
        bool gch = false;
        if (i >= scaled_upto && grad_scale != 1.0f) { gs *= grad_scale; gch = true; }
        if (i < n_clip && coef != 1.0f) { gs *= coef; gch = true; }
        elem(i, gi, pi, mi, vi);
        if (gch) g[i] = gs;
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ void adam_tick_kernel(double* __restrict__ hyper) {
    if (threadIdx.x == 0 && blockIdx.x == 0) hyper[1] += 1.0;
}

}  // namespace

extern "C" size_t nfdp_fused_adam_ws_bytes(void) { return kMaxPartials * sizeof(double); }

static int fused_adam_launch(float* p, float* g, float* m, float* v, long n, long n_clip, float clip_value, float max_norm, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int adamw, long step, double* hyper, void* ws,
                             float* norm_out, nfdp_stream_t stream, float grad_scale = 1.0f);

extern "C" int nfdp_fused_adam_step_f32(float* p, float* g, float* m, float* v, long n, long n_clip, float clip_value,
                                        float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                                        int adamw, long step, void* ws, float* norm_out, nfdp_stream_t stream) {
    NFDP_REQUIRE(step >= 1, NFDP_ESHAPE, "fused_adam_step: step=%ld", step);
    return fused_adam_launch(p, g, m, v, n, n_clip, clip_value, max_norm, lr, beta1, beta2, eps, weight_decay, adamw, step, nullptr, ws,
                             norm_out, stream);
}

extern "C" int nfdp_fused_adam_step_dev_f32(float* p, float* g, float* m, float* v, long n, long n_clip, float clip_value,
                                            float max_norm, double* hyper, float beta1, float beta2, float eps, float weight_decay,
                                            int adamw, void* ws, float* norm_out, nfdp_stream_t stream) {
    NFDP_REQUIRE(hyper, NFDP_EINVAL, "fused_adam_step_dev: null hyper");
    return fused_adam_launch(p, g, m, v, n, n_clip, clip_value, max_norm, 0.f, beta1, beta2, eps, weight_decay, adamw, 1, hyper, ws,
                             norm_out, stream);
}

extern "C" int nfdp_fused_adam_step_dev_scaled_f32(float* p, float* g, float* m, float* v, long n, long n_clip, float clip_value,
                                                   float max_norm, double* hyper, float beta1, float beta2, float eps,
                                                   float weight_decay, int adamw, float grad_scale, void* ws, float* norm_out,
                                                   nfdp_stream_t stream) {
    NFDP_REQUIRE(hyper, NFDP_EINVAL, "fused_adam_step_dev_scaled: null hyper");
    NFDP_REQUIRE(grad_scale > 0.f && grad_scale <= 1.f, NFDP_EINVAL, "fused_adam_step_dev_scaled: grad_scale=%g (expected 1/world)",
                 (double)grad_scale);
    return fused_adam_launch(p, g, m, v, n, n_clip, clip_value, max_norm, 0.f, beta1, beta2, eps, weight_decay, adamw, 1, hyper, ws,
                             norm_out, stream, grad_scale);
}

static int fused_adam_launch(float* p, float* g, float* m, float* v, long n, long n_clip, float clip_value, float max_norm, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int adamw, long step, double* hyper, void* ws,
                             float* norm_out, nfdp_stream_t stream, float grad_scale) {
    NFDP_REQUIRE(p && g && m && v && ws, NFDP_EINVAL, "fused_adam_step: null pointer");
    NFDP_REQUIRE(n > 0 && n_clip >= 0 && n_clip <= n, NFDP_ESHAPE, "fused_adam_step: bad sizes n=%ld n_clip=%ld", n, n_clip);
    hipStream_t s = as_stream(stream);
    if (hyper) hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, s, hyper);      // step count += 1 on the device
    double* partial = reinterpret_cast<double*>(ws);
    int nblk = 1;
    long scaled_upto = 0;                                  // gradient elements [0, scaled_upto) get grad_scale in the clip pass
    if (n_clip > 0 && (clip_value > 0.f || max_norm > 0.f)) {
        long gsz = cdiv(n_clip, (long)kBlock * 8);
        nblk = (int)(gsz > kMaxPartials ? kMaxPartials : (gsz < 1 ? 1 : gsz));
        hipLaunchKernelGGL(clip_value_sumsq_kernel, dim3(nblk), dim3(kBlock), 0, s, g, n_clip, clip_value, grad_scale, partial);
        scaled_upto = n_clip;
    } else {
        max_norm = 0.f;
    }
    const float bias_c1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bias_c2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adam_step_kernel, dim3(stream_grid(n, kBlock)), dim3(kBlock), 0, s, p, g, m, v, n, n_clip, partial, nblk,
                       max_norm, lr, beta1, beta2, eps, weight_decay, adamw, bias_c1, bias_c2_sqrt, norm_out, hyper, grad_scale,
                       scaled_upto);
    NFDP_CHECK_LAUNCH("fused_adam_step");
    return NFDP_OK;
}
