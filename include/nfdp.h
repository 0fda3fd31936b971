/* nfdp.h — C ABI of the MI355X (gfx950) kernel library for the Glow + diffusion-prior hot path.
 *
 * The reference (davitpapikyan/Normalizing-Flow-with-Diffusion-Prior-Model) has no FFI: its
 * boundary is a Python class surface whose methods dispatch torch ops.  Each entry point below
 * replaces the torch-op sequence of one reference method (cited as file:line relative to the
 * reference root).  Python binds these with ctypes; nothing here depends on
 * torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates, frees
 *     or retains memory and keeps no global mutable state (re-entrant per stream): every mode is a per-call argument (`compute`,
 *     `form`, `variant`), the shipped build reads no environment variable.  The only state is THREAD-local: the last-error text and
 *     the test hooks nfdp_debug_set_* (kernel-form overrides of the calling thread, used by the bit-identity tests);
 *   - activations are fp32 (or bf16 where stated), physical layout NHWC: element (b,h,w,c) of a
 *     [B,H,W,C] tensor is at ((b*H+h)*W+w)*ld + c with ld >= C the pixel stride in ELEMENTS
 *     (ld lets a kernel address a channel-half of a wider tensor in place);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously on it;
 *   - return value: NFDP_OK or a negative error; nfdp_last_error() gives thread-local text.
 *     Asynchronous HIP faults surface at the caller's next synchronisation.
 */
#ifndef NFDP_H
#define NFDP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFDP_OK 0
#define NFDP_EINVAL (-1)       /* null pointer / bad enum */
#define NFDP_ESHAPE (-2)       /* shape the kernel cannot take (odd H/W, C not divisible, ...) */
#define NFDP_EUNSUPPORTED (-3) /* valid request this build has no kernel for */
#define NFDP_EHIP (-4)         /* a HIP runtime call failed (launch error) */

#define NFDP_F32 0
#define NFDP_BF16 1
#define NFDP_F32X6 2 /* a `compute` value of the forward convs only: fp32 storage and fp32 packed weights like NFDP_F32, every product
                      * formed on the bf16 matrix cores from three-way splits of both operands (v = h + m + l exactly; six
                      * v_mfma_f32_16x16x32_bf16 per 32 k with fp32 accumulation: as close to the fp64 value as the fp32 FMA chain, 2e-7 vs
                      * 5e-7 at K = 576..4608, at 2.7x fewer matrix-pipe cycles).  Non-finite or > bf16-max operands travel in the high
                      * piece alone: outputs they reach are non-finite exactly where NFDP_F32's are and NaN stays NaN, but an inf may
                      * come out as NaN (inf times the other operand's signed residual pieces). */
#define NFDP_MLP_NO_K16 4 /* bit of the `form` argument of the tiled coupling-network entries (nfdp_coupling_mlp_image_kind) */
#define NFDP_PACK_JOB_WORDS 12 /* int64 words of one nfdp_conv_pack_weights_batch job */

typedef void* nfdp_stream_t;

int nfdp_version(void);
const char* nfdp_last_error(void);

/* ---------------------------------------------------------------- layout / index reshuffles (bit-exact) */
/* logical NCHW contiguous <-> NHWC contiguous (boundary conversion for reference-style tensors). */
int nfdp_nchw_to_nhwc_f32(const float* x, float* y, int B, int C, int H, int W, nfdp_stream_t stream);
int nfdp_nhwc_to_nchw_f32(const float* x, float* y, int B, int C, int H, int W, nfdp_stream_t stream);
/* Squeeze.transform, normalizing_flow/transforms.py:226:  y[b,h,w,4c+2h1+w1] = x[b,2h+h1,2w+w1,c].
 * x is [B,H,W,C] with pixel stride x_ld >= C (reads the kept half of a Split in place), H,W even;
 * y is [B,H/2,W/2,4C] contiguous, 16-byte aligned. */
int nfdp_squeeze_nhwc_f32(const float* x, int x_ld, float* y, int B, int C, int H, int W, nfdp_stream_t stream);
/* Squeeze.invert, transforms.py:238: y is [B,h,w,C4] (16-byte aligned) -> x [B,2h,2w,C4/4]. */
int nfdp_unsqueeze_nhwc_f32(const float* y, float* x, int B, int C4, int h, int w, nfdp_stream_t stream);
/* The coupling network (coupling_network utils.py:72-90: conv3x3 -> ActNorm -> ReLU -> conv1x1 -> ActNorm -> ReLU -> the nine tap
 * contributions of the ZeroConv2d) for images of ANY size, bf16 MFMA, hidden activations on chip: xcol = nfdp_im2col3x3_bf16 of the
 * first C/2 channels ([npix][xcol_ld] bf16, xcol_ld >= 9*C/2 rounded up to 16), packed = nfdp_coupling_fused_pack image,
 * T [npix][9*C] fp32 (n = tap*C + co) for nfdp_tap_shift_sum_f32 (add = bias, mul = logs, mul_exp = 3).  C in {4,8,16,12,24,48}. */
int nfdp_coupling_mlp_bf16(const void* xcol, int xcol_ld, long npix, const void* packed, float* T, int C, nfdp_stream_t stream);
/* The same without the im2col matrix: the kernel gathers the first conv's operands from the fp32 NHWC tensor (x: the C/2-channel half of
 * [B][H][W][*], pixel stride x_ld floats; zero padding, bf16 rounding as nfdp_im2col3x3_bf16).  One launch less per flow. */
int nfdp_coupling_mlp_gather_bf16(const float* x, int x_ld, int B, int H, int W, const void* packed, float* T, int C, int form,
                                  nfdp_stream_t stream);
/* The same two networks on v_mfma_f32_16x16x32_bf16 (C = 4, 12): packed_hi = the fold_scales = 2 image, packed_lo = NULL (bf16
 * operands) or the fold_scales = 4 image (low pieces: split products).  nfdp_coupling_mlp_image_kind(C, B*H*W, split) says which
 * image kind a call of that size and `form` takes (2: this entry; 1: nfdp_coupling_mlp_gather_bf16 / _bf16x3).
 * `form` - a PER-CALL argument, the library keeps no mode: bits 0-1 = kernel form of the gather entries: 0 (default) the row-split form
 * (32-pixel workgroups whose four waves split the weights; same images, bf16 results bit-identical) where the 128-pixel form would
 * leave most CUs idle (<= 16384 pixels; C = 12, 24, 48), 1 = never, 2 = always (the split-product form at C = 48 exists row-split
 * only); NFDP_MLP_NO_K16: keep C = 4, 12 on the 32x32x16 kernels (image kind 1). */
int nfdp_coupling_mlp_image_kind(int C, long npix, int split, int form);
int nfdp_coupling_mlp_gather_k16(const float* x, int x_ld, int B, int H, int W, const void* packed_hi, const void* packed_lo, float* T,
                                 int C, nfdp_stream_t stream);
/* Split-product form for the exact-fp32 mode: every fp32 product as hi*hi + hi*lo + lo*hi of two-piece bf16 splits of both operands
 * (three bf16 MFMAs against sixteen fp32-MFMA passes), fp32 accumulation; pixel error of a 48-flow inverse 4-6e-5.  packed_hi: the
 * nfdp_coupling_fused_pack image with fold_scales = 1, packed_lo: the same call with fold_scales = 3 (low pieces).  C <= 24 or C = 48
 * (row-split form only). */
int nfdp_coupling_mlp_gather_bf16x3(const float* x, int x_ld, int B, int H, int W, const void* packed_hi, const void* packed_lo, float* T,
                                    int C, int form, nfdp_stream_t stream);
/* The same network in EXACT fp32 (v_mfma_f32_32x32x2_f32: fp32 products and sums), hidden activations on chip: the fp32-accurate fast
 * path of compute="f32" (utils.py:83-89 is fp32 end to end).  Reads the C/2-channel half x [B,H,W,*] fp32 NHWC (pixel stride x_ld
 * floats) directly - the 3x3 gather of the first conv happens in the kernel - and writes T [B*H*W][9*C] fp32 (n = tap*C + co) for
 * nfdp_tap_shift_sum_f32.  packed: nfdp_coupling_mlp_f32_pack's image (nfdp_coupling_mlp_f32_packed_bytes(C) bytes, 0 = C unsupported)
 * of w1 [512][C/2][3][3], scale1/bias1 [512], w2 [512][512], scale2/bias2 [512], w3 [C][512][3][3]; the ZeroConv2d's bias / logs are
 * applied by the shift-sum.  C in {4,8,12,16,24,48}; any image size.  Bound: fp32 matrix pipe (157 TFLOP/s). */
size_t nfdp_coupling_mlp_f32_packed_bytes(int C);
int nfdp_coupling_mlp_f32_pack(const float* w1, const float* scale1, const float* bias1, const float* w2, const float* scale2,
                               const float* bias2, const float* w3, void* packed, int C, nfdp_stream_t stream);
int nfdp_coupling_mlp_f32(const float* x, int x_ld, int B, int H, int W, const void* packed, float* T, int C, nfdp_stream_t stream);
/* variant: 0 = chosen by the pixel count (what nfdp_coupling_mlp_f32 does), 1 = one wave per 32-pixel tile, 2 = four waves per tile
 * splitting the output rows of every GEMM (hidden blocks through LDS; for small pixel counts such as the 4x4 level).  The two differ in
 * fp32 summation order only. */
int nfdp_coupling_mlp_f32_ex(const float* x, int x_ld, int B, int H, int W, const void* packed, float* T, int C, int variant,
                             nfdp_stream_t stream);
/* Explicit im2col of a narrow tensor for a 3x3 "same" conv (training path of the C/2 -> 512 convs, utils.py:67): x [B,H,W,C] fp32
 * with pixel stride x_ld -> out [B*H*W][Kpad] bf16, out[p][tap*C + c] = x[p + shift(tap)][c] (zero outside the image and for
 * columns >= 9C); Kpad >= 9C, Kpad % 8 == 0, out 16-byte aligned. */
int nfdp_im2col3x3_bf16(const float* x, int x_ld, void* out, int B, int H, int W, int C, int Kpad, nfdp_stream_t stream);
/* channel slice copy: dst[p, 0:n) = src[p, off:off+n) for npix pixels (Split / chunk / concat plumbing). */
int nfdp_copy_channels_f32(const float* src, int src_ld, float* dst, int dst_ld, long npix, int n, nfdp_stream_t stream);
/* torch.cat((a, b), dim=1) of the U-Net's skip connections (denoising_diffusion_pytorch.py:478-486) for pixel-strided NHWC tensors:
 * out[p][0:ca] = a[p], out[p][ca:ca+cb] = b[p], one launch. */
int nfdp_concat_channels_f32(const float* a, int a_ld, int ca, const float* b, int b_ld, int cb, float* out, int out_ld, long npix,
                             nfdp_stream_t stream);

/* ---------------------------------------------------------------- ActNorm (transforms.py:56-94) */
/* Data-dependent init, transforms.py:74-78: scale[c] = -log(std_unbiased + 1e-6), bias[c] = -mean over
 * npix pixels.  fp64 two-pass reduction, deterministic.  ws: >= nfdp_actnorm_init_ws_bytes(C) bytes. */
size_t nfdp_actnorm_init_ws_bytes(int C);
int nfdp_actnorm_init_f32(const float* x, int ld, long npix, int C, float* scale, float* bias, void* ws,
                          nfdp_stream_t stream);
/* y = exp(scale)*(x+bias)  (transforms.py:80) and x = y*exp(-scale)-bias (transforms.py:93).  In place allowed. */
int nfdp_actnorm_fwd_f32(const float* x, float* y, long npix, int C, const float* scale, const float* bias,
                         nfdp_stream_t stream);
int nfdp_actnorm_inv_f32(const float* y, float* x, long npix, int C, const float* scale, const float* bias,
                         nfdp_stream_t stream);

/* ---------------------------------------------------------------- invertible 1x1 conv (transforms.py:116-145) */
/* For n matrices W_i [C,C] fp32 (row-major, out x in): logabsdet[i] = (float) log|det W_i| via fp64 elimination
 * (transforms.py:131) and, if winv != NULL, winv_i = (float) W_i^-1 (transforms.py:144).
 * ws: >= nfdp_invconv_prepare_ws_bytes(n, C) bytes. */
size_t nfdp_invconv_prepare_ws_bytes(int n, int C);
int nfdp_invconv_prepare_f32(const float* w, int n, int C, float* logabsdet, float* winv, void* ws,
                             nfdp_stream_t stream);
/* log|det W_i| (fp64 LU with partial pivoting, returned as fp32) of n matrices of possibly DIFFERENT sizes C[i] <= 96 in one
 * launch: every 1x1 convolution of a Glow forward pass (torch.slogdet(W.double())[1].float(), transforms.py:131, K*L times
 * per pass in the reference).  w, C: HOST arrays of n device pointers / sizes; logabsdet: device [n]. */
int nfdp_invconv_logdets_f32(const float* const* w, const int* C, int n, float* logabsdet, nfdp_stream_t stream);
/* W^-1 (and log|det W|) of n matrices of DIFFERENT sizes in one launch (the 1x1 convs of all levels of a model ahead of its inverse
 * pass, InvConv2d.invert transforms.py:139-146): w / C / winv are HOST arrays of device pointers / sizes / output pointers; sizes
 * from nfdp_invconv_inverses_supported (4, 8, 12, 16, 24, 48: one wave per matrix, fp64 Gauss-Jordan as nfdp_invconv_prepare_f32 -
 * same results). */
int nfdp_invconv_inverses_supported(int C);
int nfdp_invconv_inverses_f32(const float* const* w, const int* C, int n, float* logabsdet, float* const* winv, nfdp_stream_t stream);
/* y[p,:] = W x[p,:]   (conv2d with a [C,C,1,1] kernel, transforms.py:132 / :144 with W^-1). */
int nfdp_invconv_apply_f32(const float* x, float* y, long npix, int C, const float* w, nfdp_stream_t stream);
/* Fused flow head, glow.py:46-47: y = W (exp(scale)*(x+bias)). */
int nfdp_actnorm_invconv_fwd_f32(const float* x, float* y, long npix, int C, const float* scale, const float* bias,
                                 const float* w, nfdp_stream_t stream);
/* Same, and additionally writes the flow's two batch-independent log-det terms for nfdp_coupling_*'s ldj_const:
 * ldj_const_out[0] = hw*sum(scale) (transforms.py:81), [1] = hw*logabsdet[0] (transforms.py:131), fp32
 * ([1] = 0 when logabsdet is NULL: the caller adds the 1x1-conv log-dets itself, e.g. from a side stream).
 * C must be one of {4,8,12,16,24,48} (templated kernels); otherwise NFDP_EUNSUPPORTED (use nfdp_flow_ldj_consts_f32). */
int nfdp_flow_head_fwd_f32(const float* x, float* y, long npix, int C, const float* scale, const float* bias,
                           const float* w, const float* logabsdet, float hw, float* ldj_const_out, nfdp_stream_t stream);
/* Fused inverse, glow.py:61-62: x = (W^-1 y)*exp(-scale) - bias. */
int nfdp_invconv_actnorm_inv_f32(const float* y, float* x, long npix, int C, const float* scale, const float* bias,
                                 const float* winv, nfdp_stream_t stream);

/* ---------------------------------------------------------------- convolution (MFMA implicit GEMM) */
/* Replaces F.conv2d inside Conv2dActNorm / ZeroConv2d (normalizing_flow/utils.py:43-44, 67-69) and the U-Net
 * convs (diffusion_prior/gaussian_diffusion.py).  stride 1, zero padding (KH-1)/2.
 * Weights are first packed from torch's OIHW fp32 into [Cout_pad][K_pad], k = (ky*KW+kx)*Cin+ci, in the compute
 * dtype.  Epilogue per output channel n:  v = (acc + add[n]) * m[n]; if (relu) v = max(v,0)  (add/mul may be NULL),
 * with m[n] = mul[n] if mul_exp == 0 else exp(mul[n]*mul_exp) — mul_exp = 1 gives ActNorm's exp(scale)
 * (transforms.py:80), mul_exp = 3 ZeroConv2d's exp(3*logs) (utils.py:44) straight from the raw parameters.
 * compute = NFDP_F32 uses v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain); NFDP_BF16 uses v_mfma_f32_16x16x32_bf16
 * with fp32 accumulation; NFDP_F32X6 (above) is NFDP_F32 with split products.  x/y dtypes: NFDP_F32 or NFDP_BF16 storage (bf16 storage only with bf16 compute). */
size_t nfdp_conv_packed_bytes(int Cout, int Cin, int KH, int KW, int compute);
int nfdp_conv_pack_weights(const float* w_oihw, void* w_packed, int Cout, int Cin, int KH, int KW, int compute,
                           nfdp_stream_t stream);
/* The packed forms the training path derives from a parameter, in one launch from the parameter itself: flip = 1 packs the
 * data-gradient weight W'[i][o][ky][kx] = w[o][i][K-1-ky][K-1-kx]; taps = 1 packs the pointwise form of a KxK conv with few outputs
 * (rows tap*Cout' + co', columns ci': the GEMM in front of nfdp_tap_shift_sum_f32).  Sizes: nfdp_conv_packed_bytes_ex. */
size_t nfdp_conv_packed_bytes_ex(int Cout, int Cin, int KH, int KW, int compute, int flip, int taps);
int nfdp_conv_pack_weights_ex(const float* w_oihw, void* w_packed, int Cout, int Cin, int KH, int KW, int compute, int flip, int taps,
                              nfdp_stream_t stream);
int nfdp_conv2d_nhwc(const void* x, int x_dtype, int x_ld, const void* w_packed, void* y, int y_dtype, int y_ld,
                     int B, int H, int W, int Cin, int Cout, int KH, int KW, const float* add, const float* mul,
                     float mul_exp, int relu, int compute, nfdp_stream_t stream);
/* conv2d(torch.cat((x1, x2), dim=1)) without materialising the concatenation - the U-Net's skip connections
 * (diffusion_prior/gaussian_diffusion.py:479-486: `torch.cat((x, h.pop()), dim=1)` in front of every up-path ResnetBlock): exact fp32,
 * both inputs fp32 pixel-strided with 16-byte aligned rows, Cin1 % 8 == 0, Cin2 % 4 == 0; w_packed is the packed weight of the
 * (Cin1 + Cin2)-channel conv (nfdp_conv_pack_weights, compute NFDP_F32); `add` = bias or null; compute: NFDP_F32 or NFDP_F32X6. */
int nfdp_conv2d_cat_nhwc_f32(const float* x1, int x1_ld, int Cin1, const float* x2, int x2_ld, int Cin2, const void* w_packed, float* y,
                             int y_ld, int B, int H, int W, int Cout, int K, const float* add, int compute, nfdp_stream_t stream);

/* ---------------------------------------------------------------- level boundary / end of the forward pass as single launches */
/* Split.transform (transforms.py:270-289) + the next block's Squeeze (transforms.py:226) in ONE launch: x [B,H,W,C] (the level buffer
 * after its flows) -> z [B,H,W,C/2] (the dropped half: the latent part, contiguous), ynext [B,H/2,W/2,2C] (the kept half squeezed: the
 * next level's buffer, 16-byte aligned) and, when logp != NULL, logp[b] += log N(z_b; mean, exp(logsd)) with (mean | logsd) =
 * (conv3x3(kept half, w) + bias) * exp(3 logs) (ZeroConv2d, utils.py:43-44; w OIHW [C][C/2][3][3], zero padding; w = NULL: mean = logsd
 * = 0, learn_prior_mean_logs = False) - fp32 FMA chains in (tap, ci) order, the density terms in prior.py:36's op order, fp64 sum.
 * Replaces four launches (narrow conv, nfdp_gauss_logp_f32, nfdp_copy_channels_f32, nfdp_squeeze_nhwc_f32).  One workgroup per image
 * with the kept half in LDS: nfdp_split_prior_squeeze_supported (C/2 in {2,4,6,8,12,24}, even H, W, <= 96 KB of LDS). */
int nfdp_split_prior_squeeze_supported(int C, int H, int W);
int nfdp_split_prior_squeeze_f32(const float* x, int B, int H, int W, int C, const float* w, const float* bias, const float* logs, float* z,
                                 float* ynext, double* logp, nfdp_stream_t stream);
/* ldj[b] += sum over levels l of hw[l] * float(sum_i v[l][i]) (the weight-only 1x1-conv log-dets of every level, transforms.py:131),
 * per level the arithmetic of nfdp_ldj_add_scaled_f32, added in level order: bit-identical to `levels` calls of it.  v, n_sum, hw: HOST
 * arrays of `levels` <= 8 entries (v[l]: device pointers). */
int nfdp_ldj_add_levels_f32(double* ldj, int B, const float* const* v, const int* n_sum, const float* hw, int levels, nfdp_stream_t stream);
/* GaussianPrior.compute_log_prob (prior.py:70-83): logp[b] += log N(z_b; mean, exp(logsd)) with the spatially constant (mean | logsd) =
 * bias * exp(3 logs) of the ZeroConv2d on zeros formed in the kernel (bias, logs: [2*Cz]; NULL: zeros).  z [B,HW,*] pixel stride z_ld. */
int nfdp_gauss_logp_prior_f32(const float* z, int z_ld, const float* bias, const float* logs, int B, int HW, int Cz, double* logp,
                              nfdp_stream_t stream);

/* ---------------------------------------------------------------- affine coupling (transforms.py:166-201) */
/* x: [B,HW,C] flow tensor (x_a = channels [0,C/2), x_b = [C/2,C)); h: NN output [B,HW,C] (log_s | t).
 * fwd (transforms.py:181-184): s = sigmoid(log_s+2); x_b <- (x_b+t)*s in place; ldj[b] += sum log(s+1e-6) (fp64).
 * If ldj_const != NULL, ldj[b] first receives (double)ldj_const[0] + (double)ldj_const[1] — the batch-independent
 * ActNorm and 1x1-conv log-dets of the same flow step (transforms.py:81, :131), added in the reference's order.
 * ldj may be NULL (no accumulation).  inv (transforms.py:198-199): x_b <- y_b/(s+1e-6) - t. */
int nfdp_coupling_fwd_f32(float* x, const float* h, int B, int HW, int C, double* ldj, const float* ldj_const,
                          nfdp_stream_t stream);
int nfdp_coupling_inv_f32(float* y, const float* h, int B, int HW, int C, nfdp_stream_t stream);
/* The narrow 3x3 output conv's tap shift-sum (nfdp_tap_shift_sum_f32) and the coupling (nfdp_coupling_fwd_f32 / _inv_f32) in one pass:
 * T [B*H*W][9*C] tap contributions (n = tap*C + co); h = (sum of taps + add) * exp(f * mul) is consumed at once, x's second channel
 * half is updated in place; forward: ldj[b] (+ ldj_const[0..1]) += sum log(s + 1e-6).  ws: nfdp_tap_coupling_ws_bytes(B, H*W, C) bytes
 * when that is non-zero and ldj is given.  Replaces transforms.py:179-185 / :196-200 after utils.py:43-44 (reference). */
size_t nfdp_tap_coupling_ws_bytes(int B, int HW, int C);
int nfdp_tap_coupling_f32(float* x, const float* T, int B, int H, int W, int C, const float* add, const float* mul, float f, double* ldj,
                          const float* ldj_const, void* ws, int inverse, nfdp_stream_t stream);
/* nfdp_tap_coupling_f32 with the ADJOINING flow head applied to the finished pixel in the same pass (one launch and one pass over the
 * flow state per flow instead of two; bit-identical to them).  Forward (inverse = 0): the coupling of this flow, then the NEXT flow's
 * ActNorm + 1x1 conv (reference glow.py:40-45 order actnorm -> invconv -> coupling; head_scale / head_bias [C], head_w [C][C]), and the
 * next flow's two log-det constants to head_const_out [2] (may be NULL; head_logabsdet [1] or NULL).  Inverse: the coupling's inverse,
 * then THIS flow's inverse head (head_w = W^-1: x = (W^-1 y) e^-s - b, transforms.py:86-93, :139-146).  C in {4, 8, 12, 16, 24, 32, 48}
 * (nfdp_tap_coupling_head_supported); ws: nfdp_tap_coupling_head_ws_bytes (forward with ldj). */
int nfdp_tap_coupling_head_supported(int C);
size_t nfdp_tap_coupling_head_ws_bytes(int B, int HW, int C);
int nfdp_tap_coupling_head_f32(float* x, const float* T, int B, int H, int W, int C, const float* add, const float* mul, float f,
                               double* ldj, const float* ldj_const, void* ws, int inverse, const float* head_scale,
                               const float* head_bias, const float* head_w, const float* head_logabsdet, float* head_const_out,
                               int defer_ldj, nfdp_stream_t stream);
/* defer_ldj != 0 (forward with ldj): the launch leaves its chunk sums of log(s + 1e-6) in ws ([B][nfdp_tap_coupling_head_chunks]
 * doubles) and does not touch ldj; nfdp_ldj_finish_level_f64 then adds the sums of up to a level's launches to ldj in ONE launch - in
 * launch order, each with its two constants (consts[i] = that launch's ldj_const or NULL), i.e. bit-identical to the launches' own
 * finishes - instead of one small launch per flow. */
int nfdp_tap_coupling_head_chunks(int B, int HW, int C);
int nfdp_ldj_finish_level_f64(double* ldj, int B, const void* const* parts, const float* const* consts, int n, int nch,
                              nfdp_stream_t stream);
/* coupling_fwd for large images at small batch: the pixels of a sample spread over several workgroups, chunk sums of the log-det in
 * ws (nfdp_coupling_chunked_ws_bytes; 0 = use nfdp_coupling_fwd_f32), added to ldj in chunk order by a second launch. */
size_t nfdp_coupling_chunked_ws_bytes(int B, int HW, int C);
int nfdp_coupling_fwd_chunked_f32(float* x, const float* h, int B, int HW, int C, double* ldj, const float* ldj_const, void* ws,
                                  nfdp_stream_t stream);
/* Fused coupling layer (bf16 MFMA, fp32 accumulate, fp32 flow state): the whole coupling network
 * (utils.py:83-89) and the coupling epilogue above in ONE launch; the 512-wide hidden activations stay in
 * registers, weights stream through LDS from a pre-packed MFMA register image.  x: [B,H,W,C] contiguous, updated in
 * place (x_b half).  Geometry: n_features = 512, H = W in {16,8,4} with C in {4|12, 8|24, 16|48} respectively
 * (nfdp_coupling_fused_supported returns 1); anything else returns NFDP_EUNSUPPORTED and callers use
 * nfdp_conv2d_nhwc + nfdp_coupling_*_f32.  `packed` (nfdp_coupling_fused_packed_bytes(C) bytes) is built by
 * nfdp_coupling_fused_pack from the nine parameter tensors of one coupling network in the reference's layouts:
 * w1 [512,C/2,3,3], scale1/bias1 [512], w2 [512,512(,1,1)], scale2/bias2 [512], w3 [C,512,3,3], bias3 [C], logs3 [C]. */
int nfdp_coupling_fused_supported(int H, int W, int C, int n_features);
/* Inference-only geometries on top of those: the wide last levels of a 5-level model on 128x128x3 (H = W = 8 with C = 96, H = W = 4 with
 * C = 192; coupling_fused_wide_kernel: whole images per workgroup, the 3x3 output conv as one accumulation over (tap, hidden channel)
 * with the hidden tile read at the shifted pixel).  nfdp_coupling_fused_has_head: 1 when the launch includes the flow head (ActNorm +
 * 1x1 conv, glow.py:46-47 / :61-62), 0 when it is the coupling layer alone (C = 192) and nfdp_stepflow_fused_bf16 /
 * nfdp_level_fused_bf16 refuse the geometry: run nfdp_flow_head_fwd_f32 / nfdp_invconv_actnorm_inv_f32, then nfdp_coupling_fused_bf16. */
int nfdp_coupling_fused_infer_supported(int H, int W, int C, int n_features);
int nfdp_coupling_fused_has_head(int H, int W, int C);
/* Which image the fused kernels of a geometry read (0: geometry unsupported): 1 = nfdp_coupling_fused_pack / _pack_ex(fold_scales = 1),
 * 2 = _pack_ex(fold_scales = 2): the 16x16 levels run their GEMMs on v_mfma_f32_16x16x32_bf16 and read 16-row x 32-deep fragments.
 * nfdp_coupling_fused_bf16 / nfdp_stepflow_fused_bf16 / nfdp_level_fused_bf16 must be handed the kind this returns for their H, W, C;
 * nfdp_coupling_mlp_bf16 (any image size) reads kind 1. */
int nfdp_coupling_fused_image_kind(int H, int W, int C);
size_t nfdp_coupling_fused_packed_bytes(int C);
int nfdp_coupling_fused_pack(const float* w1, const float* scale1, const float* bias1, const float* w2,
                             const float* scale2, const float* bias2, const float* w3, const float* bias3,
                             const float* logs3, void* packed, int C, nfdp_stream_t stream);
/* fold_scales != 0: as nfdp_coupling_fused_pack (inference images: exp(scale) of each hidden ActNorm folded into the next GEMM's
 * weights).  fold_scales == 0: the image nfdp_level_fused_train_bf16 takes - plain weights, the kernel's epilogues compute
 * (acc + bias) * exp(scale), the hidden activation the backward pass reads (utils.py:67-69). */
int nfdp_coupling_fused_pack_ex(const float* w1, const float* scale1, const float* bias1, const float* w2,
                                const float* scale2, const float* bias2, const float* w3, const float* bias3,
                                const float* logs3, void* packed, int C, int fold_scales, nfdp_stream_t stream);
int nfdp_coupling_fused_bf16(float* x, const void* packed, int B, int H, int W, int C, double* ldj,
                             const float* ldj_const, int inverse, nfdp_stream_t stream);
/* One whole flow step in one launch (StepFlow.transform / invert, glow.py:46-48, :60-62): the flow head
 * y = W (exp(scale)(x+bias)) is applied to the tile in LDS before the coupling layer (forward), or
 * x = (W y) exp(-scale) - bias after it (inverse; pass W^-1 as `w`).  Same geometry/packing as above.
 * ldj[b] += H*W*sum(scale) (computed in-kernel, fp32) + (logdet_term ? logdet_term[0] : 0) + sum log(s+1e-6). */
int nfdp_stepflow_fused_bf16(float* x, const void* packed, int B, int H, int W, int C, const float* scale,
                             const float* bias, const float* w, double* ldj, const float* logdet_term, int inverse,
                             nfdp_stream_t stream);
/* n_flows consecutive flow steps of one Glow level (GlowBlock.transform / invert loop, glow.py:107-109, :128-130) in ONE
 * persistent launch: a workgroup owns whole images, so its flow-state tile stays in LDS across all the flows and only the
 * weights stream.  packed/scale/bias/w: HOST arrays of n_flows DEVICE pointers in forward flow order (the kernel receives
 * them by value); inverse walks them backwards and expects W^-1 in `w`.  ldj[b] += sum over the flows of
 * H*W*sum(scale) + sum log(s+1e-6) (the 1x1-conv log-dets are the caller's, nfdp_ldj_add_scaled_f32).  Results are
 * bit-identical to n_flows calls of nfdp_stepflow_fused_bf16. */
int nfdp_level_fused_bf16(float* x, int B, int H, int W, int C, int n_flows, const void* const* packed,
                          const float* const* scale, const float* const* bias, const float* const* w, double* ldj,
                          int inverse, nfdp_stream_t stream);
/* A 1x1 conv (split-product fp32 arithmetic, weights packed by nfdp_conv_pack_weights with compute = NFDP_F32: Cin % 32 == 0, Cin <= 128)
 * with the channel LayerNorm of its INPUT (pre_gain [Cin]: PreNorm -> to_qkv, gaussian_diffusion.py:212-220) and / or of its OUTPUT
 * (post_gain [Cout], Cout <= 128, then + residual: LinearAttention.to_out -> LayerNorm -> Residual, :160-166, :314-318) in one launch;
 * LayerNorm as nfdp_chan_layernorm_f32 ((x - mean) * rsqrt(var + eps) * gain, biased variance).  bias may be null.  With neither gain
 * it is the plain 1x1 conv whose epilogue adds the residual (Attention.to_out + x, :160-166, :356-358). */
int nfdp_conv1x1_ln_supported(long npix, int Cin, int Cout, int pre, int post);
/* PreNorm -> LinearAttention without the qkv tensor (:212-220, :305-333): the workgroup of a (sample, head) forms its own q / k / v rows
 * (LayerNorm of x with pre_gain, to_qkv: the products of nfdp_conv1x1_ln_f32 in the same order) and runs nfdp_linear_attention_f32's core on
 * them.  x [B][n][C] (rows x_ld), w_packed = nfdp_conv_pack_weights of to_qkv's [3*heads*32][C][1][1] weight (fp32), out [B][n][heads*32].
 * n <= 64 or a multiple of 256 (beyond 256: two launches over 256-pixel chunks as nfdp_linear_attention_chunked_f32, with a workspace of
 * nfdp_linear_attention_qkv_ws_bytes; ws may be NULL up to 256), C in {64, 128}, dim_head 32. */
int nfdp_linear_attention_qkv_supported(int n, int C, int heads, int dim_head);
size_t nfdp_linear_attention_qkv_ws_bytes(int B, int n, int heads);
/* ... and PreNorm -> Attention (:336-358: softmax(q k^T * scale) v) the same way, nfdp_attention_f32's core; n <= 64, C in {64, 128}. */
int nfdp_attention_qkv_supported(int n, int C, int heads, int dim_head);
int nfdp_attention_qkv_f32(const float* x, int x_ld, const float* pre_gain, const void* w_packed, float* out, int B, int n, int C, int heads,
                           int dim_head, float scale, float eps, nfdp_stream_t stream);
int nfdp_linear_attention_qkv_f32(const float* x, int x_ld, const float* pre_gain, const void* w_packed, float* out, int B, int n, int C,
                                  int heads, int dim_head, float scale, float eps, void* ws, nfdp_stream_t stream);
int nfdp_conv1x1_ln_f32(const float* x, int x_ld, const void* w_packed, float* y, int y_ld, long npix, int Cin, int Cout,
                        const float* bias, const float* pre_gain, const float* post_gain, const float* residual, int res_ld, float eps,
                        nfdp_stream_t stream);
/* The wide levels of a 5-level model (8x8 with C = 96, 4x4 with C = 192) with every pixel tile's GEMM rows dealt to FOUR workgroups and a
 * flow as three launches (hidden block rows of GEMM1 | GEMM2 + the output conv's K slice of the same hidden block | partial sums +
 * coupling): the one-workgroup-per-tile kernels above stream a flow's whole weight image through 16 / 32 of the 256 CUs.  Same packed
 * images and arguments as nfdp_level_fused_bf16 / nfdp_coupling_fused_bf16 (C = 192: the coupling layer alone, heads are launches of
 * their own), plus a workspace of nfdp_level_split_ws_bytes(B, H, W, C) bytes (0: geometry unsupported), 16-byte aligned, private to the
 * call's stream.  The output conv's sum is associated by hidden block: equal to the one-workgroup kernels up to fp32 summation order. */
size_t nfdp_level_split_ws_bytes(int B, int H, int W, int C);
int nfdp_coupling_fused_ws_bf16(float* x, const void* packed, int B, int H, int W, int C, double* ldj, const float* ldj_const,
                                int inverse, void* ws, size_t ws_bytes, nfdp_stream_t stream);
int nfdp_level_fused_ws_bf16(float* x, int B, int H, int W, int C, int n_flows, const void* const* packed,
                             const float* const* scale, const float* const* bias, const float* const* w, double* ldj,
                             int inverse, void* ws, size_t ws_bytes, nfdp_stream_t stream);
/* The same launch as the FORWARD of a training step (normalizing_flow/trainer.py:155: flow.transform under autograd): besides
 * updating x in place it writes, for every flow i of the level, what the backward pass reads instead of recomputing or keeping the
 * per-op intermediates of an unfused forward -
 *   xpre [n_flows][B][H*W][C] fp32   flow-state tile entering the flow (input of ActNorm, transforms.py:80)
 *   ypost[n_flows][B][H*W][C] fp32   after ActNorm + 1x1 conv: the coupling layer's input (transforms.py:179)
 *   hout [n_flows][B][H*W][C] fp32   output of the coupling network's ZeroConv2d = (log_s | t) (utils.py:43-44)
 *   y1, y2 [n_flows][Bpad * H*W pixels x 512] bf16   the two hidden activations after ActNorm + ReLU (utils.py:83-87) in the BLOCKED
 *          layout: 32-pixel tiles of [hidden block 16][k-step 2][half 2][pixel 32][8 channels], channel = 32 hb + 16 s + 8 g + e (the
 *          MFMA B-fragment image: every store is one contiguous KiB); Bpad = B rounded up to a multiple of
 *          nfdp_level_fused_train_group(H, W, C) (whole workgroups: tiles of the images past B hold unspecified finite values)
 *   xcol [n_flows][Bpad][H*W][nfdp_coupling_fused_im2col_cols(C, 0)] bf16   im2col rows of the coupling network's input x_a, columns
 *          k = tap * C/2 + ci (zero beyond 9 C/2): the x operand of the first conv's weight-gradient GEMM
 * - straight from the registers / LDS tile of the fused kernel; `packed` images must be built with fold_scales = 0 (hidden stores are issued between the MFMAs of the last GEMM and
 * counted in the weight ring's vmcnt waits).  ldj[b] += sum over the flows of sum log(s+1e-6) ONLY: the weight-only terms
 * (H*W*sum(scale), H*W*log|det W|) belong to the caller's autograd graph.  n_flows <= 32.  Forward only. */
int nfdp_level_fused_train_group(int H, int W, int C);
int nfdp_coupling_fused_im2col_cols(int C, int backward);
int nfdp_level_fused_train_bf16(float* x, int B, int H, int W, int C, int n_flows, const void* const* packed,
                                const float* const* scale, const float* const* bias, const float* const* w, double* ldj,
                                float* xpre, float* ypost, float* hout, void* y1, void* y2, void* xcol, nfdp_stream_t stream);
/* The same with keep_hidden = 0: all flows write y1 / y2 / xcol into ONE flow-sized slot (the buffers are one flow large) - the forward
 * of a step that recomputes the hidden activations in its backward pass (normalizing_flow.training.RECOMPUTE_FLOWS: per group of flows
 * the launch is repeated from the group's saved input state xpre[first flow] with keep_hidden = 1).
 * masks (nullable; levels with nfdp_level_fused_train_has_masks: the 16x16 / 8x8 ones): the ReLU SIGNS of y1 / y2 for the reverse network,
 * which needs the activations for nothing else - [n_flows][2 layers][Bpad*H*W / 32 tiles][16 hidden blocks][64 lanes] dwords (16 bits per
 * lane and block, in the order of the blocked layout's fragments), 1/8 of the activations' bytes; pass it to nfdp_coupling_bwd_fused_bf16
 * flow by flow (flow i at masks + i * 2 * Bpad*H*W * 128 bytes) in place of y1 / y2. */
int nfdp_level_fused_train_has_masks(int H, int W, int C);
int nfdp_level_fused_train_ex_bf16(float* x, int B, int H, int W, int C, int n_flows, const void* const* packed,
                                const float* const* scale, const float* const* bias, const float* const* w, double* ldj,
                                float* xpre, float* ypost, float* hout, void* y1, void* y2, void* xcol, void* masks, int keep_hidden,
                                nfdp_stream_t stream);
/* Backward of the coupling network's data path in one launch (autograd of utils.py:83-89 w.r.t. its input, training step
 * normalizing_flow/trainer.py:164): dc3 [B][H*W][C] fp32 = gradient at the ZeroConv2d's raw conv output (after its scale backward);
 * y1, y2 = the hidden activations saved by nfdp_level_fused_train_bf16 (blocked layout, Bpad images).  Writes dc2, dc1 (same layout as
 * y2, y1: gradients at the two hidden convs' raw outputs, i.e. after the ReLU mask and the ActNorm scale) for the weight-gradient
 * GEMMs, dc3col ([Bpad][H*W][nfdp_coupling_fused_im2col_cols(C, 1)] bf16: im2col rows of dc3, the x operand of the zero conv's
 * weight-gradient GEMM), and ADDS the data gradient of the network's input into dx[..., 0:C/2] (dx [B][H*W][C] fp32).  `packed`
 * (nfdp_coupling_bwd_fused_packed_bytes(C) bytes): nfdp_coupling_bwd_fused_pack of w1 [512,C/2,3,3], scale1 [512], w2 [512,512,1,1],
 * scale2 [512], w3 [C,512,3,3].  Geometries: those of nfdp_coupling_fused_supported; others return NFDP_EUNSUPPORTED. */
size_t nfdp_coupling_bwd_fused_packed_bytes(int C);
/* The same images for n (<= 32) networks in one launch: params = HOST array of 9 * n (forward: w1, scale1, bias1, w2, scale2, bias2, w3,
 * bias3, logs3 per network) or 5 * n (reverse: w1, scale1, w2, scale2, w3) device pointers, packed = HOST array of n output buffers.
 * A training step re-packs every network of a level (its weights have just changed): one launch per level instead of one per flow. */
int nfdp_coupling_fused_pack_batch(const float* const* params, void* const* packed, int n, int C, int fold_scales, nfdp_stream_t stream);
int nfdp_coupling_bwd_fused_pack_batch(const float* const* params, void* const* packed, int n, int C, nfdp_stream_t stream);
int nfdp_coupling_bwd_fused_pack(const float* w1, const float* scale1, const float* w2, const float* scale2, const float* w3,
                                 void* packed, int C, nfdp_stream_t stream);
/* masks (nullable): this flow's slice of nfdp_level_fused_train_ex_bf16's mask tensor; when given the kernel reads it INSTEAD of y1 / y2
 * (which may then be NULL) - the same result bit for bit at 1/8 of the saved-activation bytes. */
int nfdp_coupling_bwd_fused_bf16(const float* dc3, float* dx, const void* packed, const void* y1, const void* y2, void* dc1,
                                 void* dc2, void* dc3col, const void* masks, int B, int H, int W, int C, nfdp_stream_t stream);
/* The per-pixel link between two flow steps of the backward pass, in one launch: the flow head of step i backwards (1x1 conv
 * transforms.py:132, ActNorm :80) and the affine coupling of step i-1 backwards (transforms.py:179-185) with the scale backward of its
 * ZeroConv2d (utils.py:43-44).  All tensors [npix][C] fp32 rows.
 *   head (xpre != NULL):    a = e^scale (xpre + bias) -> a_out (optional);  da = W^T dx_in (w = the 1x1 weight, row-major [C][C]);  dy = da e^scale
 *                           (no head: dy = dx_in)
 *   coupling (ypost != NULL): on (ypost, hout = (log_s | t), dy, gl[b] = gradient of the per-sample log-det, optional):
 *                           dx_out = (dy_a | dy_b s),  dh = (ds s (1-s) | dy_b s) with ds = dy_b (x_b + t) + gl / (s + 1e-6),
 *                           dc3_out = dh e^{f3 logs3}            (no coupling: dx_out = dy)
 *   part (nfdp_flow_link_bwd_part_bytes: [workgroups][4][C] fp64, then [workgroups][C*C] fp32; 256 or 64 pixels per workgroup): per-workgroup sums of da a,
 *                           da e^scale, dh hout, dc3 and - head calls - the workgroup's partial of the 1x1 conv's weight gradient
 *                           dW[k][c] = sum_p dx_in[p][k] a[p][c] (transforms.py:132; exact-fp32 MFMA).
 * nfdp_flow_link_finish_f32 reduces the partials of n_calls calls (laid out back to back) in one launch: for call k
 * dscale[k] (+)= sum da a,  dbias[k] (+)= sum da e^scale,  dlogs3[k] (+)= f3 sum dh hout,  dbias3[k] (+)= sum dc3,  dw[k] ([C][C]) (+)= dW
 * (HOST arrays of n_calls device pointers, entries may be NULL).  C in {4, 8, 12, 16, 24, 48}; n_calls <= 33. */
size_t nfdp_flow_link_bwd_part_bytes(long npix, int C);
int nfdp_flow_link_bwd_f32(const float* dx_in, const float* xpre, const float* w, const float* scale, const float* bias, float* a_out,
                           const float* ypost, const float* hout, const double* gl, const float* logs3, float f3, float* dx_out,
                           float* dc3_out, void* part, long npix, int HW, int C, nfdp_stream_t stream);
int nfdp_flow_link_finish_f32(const void* part, int n_calls, long npix, int C, float f3, float* const* dscale, float* const* dbias,
                              float* const* dlogs3, float* const* dbias3, float* const* dw, int accumulate, nfdp_stream_t stream);
/* ldj[b] += (double)(hw * (float)sum(v[0..n_sum)))  (fp32 product): the standalone ActNorm / InvConv2d log-det terms. */
int nfdp_ldj_add_scaled_f32(double* ldj, int B, const float* v, int n_sum, float hw, nfdp_stream_t stream);
/* out2[0] = hw*sum(scale[0..C)) (transforms.py:81), out2[1] = hw*logabsdet[0] (transforms.py:131), fp32: the two
 * batch-independent log-det terms of one flow step, consumed by nfdp_coupling_fwd_f32's ldj_const. */
int nfdp_flow_ldj_consts_f32(const float* scale, int C, const float* logabsdet, float hw, float* out2,
                             nfdp_stream_t stream);

/* ---------------------------------------------------------------- Gaussian prior (normalizing_flow/prior.py) */
/* logp[b] += sum_{p,c} -0.5(log2pi + 2 logs + (z-mean)^2 exp(-2 logs))   (prior.py:36-37).
 * z: [B,HW,Cz] with pixel stride z_ld.  mean/logs come from h: mean = h[..., 0:Cz), logs = h[..., Cz:2Cz).
 * h_mode 0: h is [B,HW,2Cz] (Split prior, transforms.py:266-268); h_mode 1: h is [2Cz] broadcast over b,p
 * (GaussianPrior with zero input, prior.py:79-82). */
int nfdp_gauss_logp_f32(const float* z, int z_ld, const float* h, int h_mode, int B, int HW, int Cz, double* logp,
                        nfdp_stream_t stream);
/* out = mean + exp(logs)*temperature*eps  (prior.py:49-50); eps [B,HW,Cz] supplied by the caller. */
int nfdp_gauss_sample_f32(const float* h, int h_mode, const float* eps, float* out, int out_ld, int B, int HW, int Cz,
                          float temperature, nfdp_stream_t stream);
/* GaussianPrior's degenerate conv on zeros (prior.py:79-80): h[n] = bias[n]*exp(3*logs[n]). */
int nfdp_zeroconv_bias_only_f32(const float* bias, const float* logs, float* h, int n, nfdp_stream_t stream);

/* Block of the U-Net in ONE launch: WeightStandardizedConv2d (w_packed = nfdp_conv_pack_weights of the standardised weight, fp32) ->
 * GroupNorm(groups, gamma, beta, eps) -> x*(scale+1)+shift (scale_shift [B][2*Cout] rows, stride ss_ld; nullable) -> SiLU -> + residual
 * (nullable) - gaussian_diffusion.py:267-276, :301 - on the split-K fp32 conv with the normalisation in its epilogue.  Available when a
 * workgroup's pixel tile holds whole images and its 64 output channels whole groups (H*W divides 64, (Cout/groups) divides 64 and is a
 * multiple of 4, Cout % 64 == 0, Cin % 8 == 0, K*K*Cin >= 256): the 8x8 / 4x4 / 2x2 stages.  nfdp_conv2d_gn_silu_supported returns 1
 * there; elsewhere callers run nfdp_conv2d_nhwc + nfdp_groupnorm_silu_res_f32.  compute: NFDP_F32 or NFDP_F32X6. */
int nfdp_conv2d_gn_silu_supported(int B, int H, int W, int Cin, int Cout, int K, int groups);
int nfdp_conv2d_gn_silu_nhwc_f32(const float* x, int x_ld, const void* w_packed, float* y, int y_ld, int B, int H, int W, int Cin, int Cout,
                                 int K, const float* bias, int groups, const float* gamma, const float* beta, const float* scale_shift,
                                 int ss_ld, float eps, const float* residual, int res_ld, int compute, nfdp_stream_t stream);
/* The same Block on torch.cat((x1, x2), dim=1) (see nfdp_conv2d_cat_nhwc_f32); supported where nfdp_conv2d_gn_silu_supported(B, H, W,
 * Cin1 + Cin2, ...) is and Cin1 % 8 == 0, Cin2 % 8 == 0. */
int nfdp_conv2d_gn_silu_cat_nhwc_f32(const float* x1, int x1_ld, int Cin1, const float* x2, int x2_ld, int Cin2, const void* w_packed,
                                     float* y, int y_ld, int B, int H, int W, int Cout, int K, const float* bias, int groups,
                                     const float* gamma, const float* beta, const float* scale_shift, int ss_ld, float eps,
                                     const float* residual, int res_ld, int compute, nfdp_stream_t stream);

/* ResnetBlock.forward :297-302 with block1's GroupNorm -> x*(scale+1)+shift -> SiLU (:267-276) applied by block2's conv instead of a
 * launch between the two convs - for the stages where a conv workgroup's pixel tile is a PART of an image, so that the one-launch Block
 * above does not apply.  (1) nfdp_conv2d_stats_nhwc_f32 = nfdp_conv2d_nhwc / nfdp_conv2d_cat_nhwc_f32 (x2 = NULL, Cin2 = 0: one source)
 * on the split-K fp32 conv, which also writes sums[(tile * groups + g) * 2 + {0, 1}] = fp64 sum of the tile's outputs of group g / of their
 * squares; nfdp_conv2d_stats_tile returns the pixels per tile (a divisor of H*W; sums holds B*H*W/tile * groups * 2 doubles) or 0 when the
 * problem does not qualify.  (2) nfdp_conv2d_gnin_nhwc_f32: 3x3 conv (split products) over SiLU(GN(x)*(scale+1)+shift), x the RAW output of
 * (1), the statistics taken from in_sums (in_tile = (1)'s tile), the normalisation formed while the input tile is staged in LDS - the same
 * expressions in the same order as nfdp_groupnorm_silu_f32.  out_groups > 0: followed by block2's own Block epilogue (GroupNorm, optional
 * scale/shift, SiLU, + residual) as in nfdp_conv2d_gn_silu_nhwc_f32; out_groups = 0: the plain conv (+ bias).  nfdp_conv2d_gnin_supported
 * says whether (2) is available for the geometry. */
int nfdp_conv2d_stats_tile(int B, int H, int W, int Cin, int Cout, int K, int groups);
int nfdp_conv2d_stats_nhwc_f32(const float* x1, int x1_ld, int Cin1, const float* x2, int x2_ld, int Cin2, const void* w_packed, float* y,
                               int y_ld, int B, int H, int W, int Cout, int K, const float* bias, int groups, double* sums, int compute,
                               nfdp_stream_t stream);
/* ResnetBlock's res_conv (:288, :301: a 1x1 conv of the block's INPUT) in the launch of block1's 3x3 conv: y = conv3x3(x) + bias as
 * nfdp_conv2d_stats_nhwc_f32 (sums NULL: without the group sums), y_res = conv1x1(x) + bias_res from a second accumulator set over the
 * centre tap of the staged input tile.  w_res_packed = nfdp_conv_pack_weights of the [Cout][Cin][1][1] weight (fp32).  Split products. */
int nfdp_conv2d_res_supported(int B, int H, int W, int Cin, int Cout);
int nfdp_conv2d_res_nhwc_f32(const float* x1, int x1_ld, int Cin1, const float* x2, int x2_ld, int Cin2, const void* w_packed, float* y,
                             int y_ld, int B, int H, int W, int Cout, const float* bias, int groups, double* sums, const void* w_res_packed,
                             const float* bias_res, float* y_res, int y_res_ld, nfdp_stream_t stream);
/* Unet.init_conv :383 (Conv2d(channels, init_dim, 7, padding=3)) for 1-8 input channels: y [B*H*W][y_ld] = conv7x7(x [B][H][W][Cin], rows
 * x_ld) + bias as fp32 FMA chains over (tap, channel) on the vector ALU; W % 4 == 0, Cout % 4 == 0.  w_k7 = nfdp_conv_k7_pack_f32 of the
 * [Cout][Cin][7][7] parameter: [49][Cin][Cout] fp32. */
int nfdp_conv_k7_narrow_supported(int B, int H, int W, int Cin, int Cout);
int nfdp_conv_k7_pack_f32(const float* w_oihw, float* w_k7, int Cout, int Cin, nfdp_stream_t stream);
int nfdp_conv_k7_narrow_f32(const float* x, int x_ld, const float* w_k7, const float* bias, float* y, int y_ld, int B, int H, int W, int Cin,
                            int Cout, nfdp_stream_t stream);
/* Upsample :169-173 (nn.Upsample(scale_factor=2, mode="nearest") -> Conv2d 3x3 pad 1) in one launch: x [B][H/2][W/2][Cin] (rows x_ld) is
 * read through the index map of its upsampling; y [B][H][W][Cout]; H, W = OUTPUT sides (powers of two).  Split products. */
int nfdp_conv2d_up2_supported(int B, int H, int W, int Cin, int Cout);
/* Downsample :176-180 (Rearrange "b c (h p1) (w p2) -> b (c p1 p2) h w" -> Conv2d 1x1) in one launch: the 1x1 weight [Cout][4C][1][1]
 * viewed as [Cout][C][2][2] is a 2x2 conv with stride 2 on the un-rearranged tensor.  x [B][2H][2W][C] (rows x_ld), w_packed =
 * nfdp_conv_pack_weights(view, Cout, C, 2, 2, NFDP_F32), y [B][H][W][Cout]; H, W = OUTPUT sides.  compute: NFDP_F32 or NFDP_F32X6. */
int nfdp_conv2d_squeeze_supported(int B, int H, int W, int C, int Cout);
int nfdp_conv2d_squeeze_nhwc_f32(const float* x, int x_ld, const void* w_packed, float* y, int y_ld, int B, int H, int W, int C, int Cout,
                                 const float* bias, int compute, nfdp_stream_t stream);
int nfdp_conv2d_up2_nhwc_f32(const float* x, int x_ld, const void* w_packed, float* y, int y_ld, int B, int H, int W, int Cin, int Cout,
                             const float* bias, nfdp_stream_t stream);
/* ... and beside the one-launch Block (nfdp_conv2d_gn_silu_(cat_)nhwc_f32 without its residual; x2 = NULL, Cin2 = 0: one source): y = the
 * Block of block1, y_res = conv1x1(x) + bias_res. */
int nfdp_conv2d_gn_silu_res_supported(int B, int H, int W, int Cin, int Cout, int groups);
int nfdp_conv2d_gn_silu_res_nhwc_f32(const float* x1, int x1_ld, int Cin1, const float* x2, int x2_ld, int Cin2, const void* w_packed, float* y,
                                     int y_ld, int B, int H, int W, int Cout, const float* bias, int groups, const float* gamma,
                                     const float* beta, const float* scale_shift, int ss_ld, float eps, const void* w_res_packed,
                                     const float* bias_res, float* y_res, int y_res_ld, nfdp_stream_t stream);
int nfdp_conv2d_gnin_supported(int B, int H, int W, int Cin, int Cout, int in_groups, int out_groups);
int nfdp_conv2d_gnin_nhwc_f32(const float* x, int x_ld, const double* in_sums, int in_tile, int in_groups, const float* in_gamma,
                              const float* in_beta, const float* in_scale_shift, int in_ss_ld, float in_eps, const void* w_packed, float* y,
                              int y_ld, int B, int H, int W, int Cin, int Cout, const float* bias, int out_groups, const float* out_gamma,
                              const float* out_beta, const float* out_scale_shift, int out_ss_ld, float out_eps, const float* residual,
                              int res_ld, int sums_groups, double* sums_out, nfdp_stream_t stream);
/* Block.forward :267-276 after a conv that left its group sums (nfdp_conv2d_stats_nhwc_f32, or nfdp_conv2d_gnin_nhwc_f32 with out_groups = 0
 * and sums_out: the sums of ITS output in sums_groups groups, tile = nfdp_conv2d_stats_tile): GroupNorm -> x*(scale+1)+shift -> SiLU ->
 * + residual as nfdp_groupnorm_silu_res_f32, the statistics from the sums - no reduction, workgroups of whole pixel rows. */
int nfdp_groupnorm_silu_sums_supported(int B, int HW, int C, int groups, int tile);
int nfdp_groupnorm_silu_sums_f32(const float* x, float* y, const double* sums, int tile, int B, int HW, int C, int groups, const float* gamma,
                                 const float* beta, const float* scale_shift, int ss_ld, float eps, const float* residual,
                                 nfdp_stream_t stream);

/* ---------------------------------------------------------------- diffusion prior: U-Net blocks
 * (diffusion_prior/gaussian_diffusion.py; the convolutions themselves use nfdp_conv2d_nhwc) */
/* WeightStandardizedConv2d.forward :189-197: out[o] = (w[o]-mean_o) * rsqrt(var_o(biased) + eps) over fan_in. */
int nfdp_ws_standardize_f32(const float* w, float* out, int Cout, int fan_in, float eps, nfdp_stream_t stream);
/* Block.forward :267-276 after the conv: GroupNorm(groups) -> x*(scale+1)+shift -> SiLU.  scale_shift (or NULL):
 * row b holds (scale[C] | shift[C]) at scale_shift + b*ss_ld (ss_ld >= 2C lets all blocks' time-MLP outputs share
 * one [B][sum 2C] buffer). */
int nfdp_groupnorm_silu_f32(const float* x, float* y, int B, int HW, int C, int groups, const float* gamma,
                            const float* beta, const float* scale_shift, int ss_ld, float eps, nfdp_stream_t stream);
/* The same with ResnetBlock's residual add (:301) folded in: y = SiLU(...) + residual (residual [B,HW,C] or null). */
int nfdp_groupnorm_silu_res_f32(const float* x, float* y, int B, int HW, int C, int groups, const float* gamma,
                                const float* beta, const float* scale_shift, int ss_ld, float eps, const float* residual,
                                nfdp_stream_t stream);
/* LayerNorm.forward :205-209 (channel LayerNorm, gain only) + optional residual add (Residual :160-166). */
int nfdp_chan_layernorm_f32(const float* x, float* y, long npix, int C, const float* gain, float eps,
                            const float* residual, nfdp_stream_t stream);
/* nn.Linear with fused activations (time_mlp :407-412, ResnetBlock.mlp :282-285):
 * out = post(bias + pre(in) W^T); pre_act 0 none / 1 SiLU; post_act 0 none / 1 GELU(erf) / 2 SiLU; I <= 480. */
int nfdp_linear_f32(const float* in, const float* w, const float* bias, float* out, int B, int I, int O, int pre_act,
                    int post_act, nfdp_stream_t stream);
/* SinusoidalPosEmb.forward :231-237; t is int64 [B]. */
int nfdp_sinusoidal_emb_f32(const long* t, float* out, int B, int dim, nfdp_stream_t stream);
/* RandomOrLearnedSinusoidalPosEmb.forward (gaussian_diffusion.py:240-255): out [B][2*half + 1] = [t | sin(2 pi t w_j) | cos(2 pi t w_j)],
 * w [half] the module's `weights`; _bwd: dw[j] (+)= d loss / d w_j from dout [B][2*half + 1] (learned variant; the random one is frozen). */
int nfdp_learned_sinusoidal_emb_f32(const long* t, const float* w, float* out, int B, int half, nfdp_stream_t stream);
int nfdp_learned_sinusoidal_bwd_f32(const long* t, const float* w, const float* dout, float* dw, int B, int half, int accumulate,
                                    nfdp_stream_t stream);
/* LinearAttention.forward :320-332 between to_qkv and to_out; qkv [B][n][3*heads*32] -> out [B][n][heads*32]. */
int nfdp_linear_attention_f32(const float* qkv, float* out, int B, int n, int heads, int dim_head, float scale,
                              nfdp_stream_t stream);
/* The same for long sequences (n > 256: the 32x32 / 64x64 latent parts of the larger models): positions in chunks of 256 over
 * the grid, two launches, chunk partials in ws (nfdp_linear_attention_ws_bytes; 0 when n <= 256: use the entry above). */
size_t nfdp_linear_attention_ws_bytes(int B, int n, int heads);
int nfdp_linear_attention_chunked_f32(const float* qkv, float* out, int B, int n, int heads, int dim_head, float scale, void* ws,
                                      nfdp_stream_t stream);
/* Attention.forward :347-357 between to_qkv and to_out (softmax over keys); n > 256 pixels: queries chunked over the grid, keys
 * streamed through LDS. */
int nfdp_attention_f32(const float* qkv, float* out, int B, int n, int heads, int dim_head, float scale,
                       nfdp_stream_t stream);
/* nn.Upsample(scale_factor=2, mode='nearest') :171;  x [B,H,W,C] -> y [B,2H,2W,C]. */
int nfdp_upsample_nearest2x_f32(const float* x, float* y, int B, int C, int H, int W, nfdp_stream_t stream);
int nfdp_add_f32(const float* a, const float* b, float* out, long n, nfdp_stream_t stream);

/* ---------------------------------------------------------------- diffusion prior: Gaussian diffusion steps
 * The [T] fp32 schedule buffers (GaussianDiffusion.__init__ :603-634) stay on the device and are indexed by t[b]. */
/* q_sample :806-812. */
int nfdp_q_sample_f32(const float* x0, const float* noise, const long* t, const float* sqrt_ac, const float* sqrt_1mac,
                      float* out, int B, int per_sample, nfdp_stream_t stream);
/* p_sample :706-715 around the eps prediction: predict_start_from_noise :641-645, clamp :700-701, q_posterior
 * :665-672, + exp(0.5 logvar) * noise (noise NULL at t == 0).  x0_out may be NULL. */
int nfdp_p_sample_f32(const float* x, const float* eps_hat, const float* noise, const long* t,
                      const float* sqrt_recip_ac, const float* sqrt_recipm1_ac, const float* post_coef1,
                      const float* post_coef2, const float* post_logvar, float* out, float* x0_out, int B,
                      int per_sample, int clip_x0, nfdp_stream_t stream);
/* one ddim_sample iteration :749-773 with the step's scalars computed by the caller. */
int nfdp_ddim_step_f32(const float* x, const float* eps_hat, const float* noise, float* out, long n, float sr,
                       float srm1, float sqrt_alpha_next, float c, float sigma, int last, nfdp_stream_t stream);
/* The same with the step's scalars in a device table, row chosen by a device-side counter (one captured hipGraph serves every
 * step): table [S][8] fp32 rows (sr, srm1, sqrt_alpha_next, c, sigma, last (0/1), time of this step, unused); step: int64 [1];
 * t_next (nullable): int64 [B], receives the time of row step+1 for the next U-Net call.  noise must be valid (unused when last). */
int nfdp_ddim_step_table_f32(const float* x, const float* eps_hat, const float* noise, float* out, long n, const float* table,
                             const long* step, int S, long* t_next, int B, nfdp_stream_t stream);
/* The same two steps for every objective of GaussianDiffusion.model_predictions (gaussian_diffusion.py:674-694):
 *   x0 = clamp(x0_a x - x0_b model_out), x0_a / x0_b = (sr, srm1) for pred_noise, (0, -1) for pred_x0, (sqrt_ac, sqrt_one_minus_ac) for
 *   pred_v;  derive_noise != 0: the noise used by the update is (sr x - x0) / srm1 (predict_noise_from_start :647-651) instead of
 *   model_out;  x0_out (optional) receives x0 - the next step's self-conditioning input (:744-747).  Table rows of stride 12 carry
 *   (..., derive at [7], x0_a at [8], x0_b at [9]). */
int nfdp_ddim_step_ex_f32(const float* x, const float* model_out, const float* noise, float* out, float* x0_out, long n, float sr,
                          float srm1, float x0_a, float x0_b, int derive_noise, float sqrt_alpha_next, float c, float sigma, int last,
                          nfdp_stream_t stream);
int nfdp_ddim_step_table_ex_f32(const float* x, const float* model_out, const float* noise, float* out, float* x0_out, long n,
                                const float* table, int stride, const long* step, int S, long* t_next, int B, nfdp_stream_t stream);
/* out = (sqrt_recip_ac[t] x_t - x_start) / sqrt_recipm1_ac[t]   (predict_noise_from_start, gaussian_diffusion.py:647-651). */
int nfdp_noise_from_start_f32(const float* x_t, const float* x_start, const long* t, const float* sqrt_recip_ac,
                              const float* sqrt_recipm1_ac, float* out, int B, int per_sample, nfdp_stream_t stream);
/* out = a x + b  (add_first: (x + b) a): normalize_to_neg_one_to_one x*2-1 / unnormalize_to_zero_to_one (x+1)*0.5
 * (gaussian_diffusion.py:148-157), the auto_normalize option of GaussianDiffusion. */
int nfdp_axpb_f32(const float* x, float a, float b, float* out, long n, int add_first, nfdp_stream_t stream);
/* p_losses :855-859: out[0] = mean_b( mean_chw(|pred-target|^p) * p2_weight[t_b] ), p = 2 (l1=0) or 1 (l1=1);
 * per_sample_ws: B floats of workspace. */
int nfdp_diffusion_loss_f32(const float* pred, const float* target, const long* t, const float* p2_weight, int B,
                            int per_sample, int l1, float* per_sample_ws, float* out, nfdp_stream_t stream);
/* One term of the variational bound per sample, GaussianDiffusion._vb_terms_bpd :932-958 (with normal_kl :36-44,
 * gaussian_log_likelihood :47-74, q_posterior :665-672 and the x0 clamp of p_mean_variance :696-704) for the
 * fixed-variance eps-model:  out[b] = mean_chw( t_b == 0 ? decoder NLL : KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) ).
 * x_start, x_t, eps_hat: [B, per_sample] in one common layout; total (nullable): fp64 [B], total[b] += out[b]
 * (the running sum over timesteps of calc_neg_log_likelihood_loop :869-903). */
int nfdp_vb_term_f32(const float* x_start, const float* x_t, const float* eps_hat, const long* t,
                     const float* sqrt_recip_ac, const float* sqrt_recipm1_ac, const float* post_coef1,
                     const float* post_coef2, const float* post_logvar, int B, int per_sample, int clip_x0, float* out,
                     double* total, nfdp_stream_t stream);
/* GaussianDiffusion._prior_bpd :905-918: out[b] = mean_chw KL( q(x_T|x_0) || N(0,I) ), schedule rows read at t_index. */
int nfdp_prior_kl_f32(const float* x_start, const float* sqrt_ac, const float* log_one_minus_ac, long t_index, int B,
                      int per_sample, float* out, nfdp_stream_t stream);

/* ---------------------------------------------------------------- backward of the flow path (training step)
 * What torch autograd computes for loss.backward() in normalizing_flow/trainer.py:163-164, exact fp32. */
/* Weight gradient of a stride-1 'same' conv (F.conv2d in utils.py:43, :67; the 1x1 conv of InvConv2d, transforms.py:132):
 * dw[o][ci][ky][kx] (OIHW) = sum_{b,y,x} dy[b,y,x,o] * x[b,y+ky-k/2,x+kx-k/2,ci], odd k <= 7.  x, dy: NHWC with pixel
 * strides x_ld, dy_ld.  ws: nfdp_conv_wgrad_ws_bytes(...) bytes of workspace. */
size_t nfdp_conv_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int k);
int nfdp_conv_wgrad_f32(const float* x, int x_ld, const float* dy, int dy_ld, int B, int H, int W, int Cin, int Cout, int k,
                        float* dw, void* ws, nfdp_stream_t stream);
/* The same weight gradient with bf16 MFMA (operands rounded to bf16, fp32 accumulation): fp32 tiles are converted and
 * transposed on their way into LDS.  Needs Cin % 4 == 0, Cout % 4 == 0, 16-byte aligned operands with pixel strides that are
 * multiples of 4 (nfdp_conv_wgrad_bf16_supported); otherwise use nfdp_conv_wgrad_f32.  x_dtype / dy_dtype: NFDP_F32 or NFDP_BF16
 * storage of the operands (bf16 training mode keeps the hidden activations and their gradients in bf16). */
int nfdp_conv_wgrad_bf16_supported(int Cin, int Cout);
size_t nfdp_conv_wgrad_bf16_ws_bytes(int B, int H, int W, int Cin, int Cout, int k);
int nfdp_conv_wgrad_bf16(const void* x, int x_dtype, int x_ld, const void* dy, int dy_dtype, int dy_ld, int B, int H, int W,
                         int Cin, int Cout, int k, float* dw, void* ws, nfdp_stream_t stream);
/* Gradient accumulation in the reducing launch (what autograd's AccumulateGrad does with one more elementwise launch per parameter,
 * ~640 per training step of Glow L3 K16): accumulate != 0 adds the result to dw instead of storing it.  out_form selects where the
 * reduced GEMM lands, for the convs the training path runs as plain GEMMs over im2col rows (k must be 1 for forms 1, 2):
 *   0: dw[Cout][Cin][k][k] of the conv itself (as above);
 *   1: x = nfdp_im2col3x3_bf16 rows of an out_c-channel tensor (9*out_c <= Cin): the 3x3 weight gradient dw[Cout][out_c][3][3];
 *   2: x = im2col rows of the out_c-channel OUTPUT gradient of a 3x3 conv and dy = that conv's input: dw[out_c][Cout][3][3] with the
 *      taps mirrored (ZeroConv2d's weight gradient from the im2col matrix of its output gradient). */
int nfdp_conv_wgrad_f32_ex(const float* x, int x_ld, const float* dy, int dy_ld, int B, int H, int W, int Cin, int Cout, int k,
                           float* dw, void* ws, int out_form, int out_c, int accumulate, nfdp_stream_t stream);
int nfdp_conv_wgrad_bf16_ex(const void* x, int x_dtype, int x_ld, const void* dy, int dy_dtype, int dy_ld, int B, int H, int W,
                            int Cin, int Cout, int k, float* dw, void* ws, int out_form, int out_c, int accumulate, nfdp_stream_t stream);
/* The weight gradient of a 1x1 conv on bf16 operands (x [npix][Cin], dy [npix][Cout], rows 16-byte aligned, npix a multiple of 64) for
 * a layer y = act((conv(x, W) + bias) * exp(scale)) whose dy is the gradient at the conv's RAW output (mask and scale already applied):
 * besides dW (out_form 0: [Cout][Cin]; 1: x = im2col rows, dW stored as [Cout][out_c][3][3]) it yields the layer's ActNorm gradients
 * from the same GEMM, with no further pass over the activations (Conv2dActNorm backward, utils.py:67-69):
 *   dbias[m]  (+)= sum_p dy[p][m]                                   (ones-fragment MFMAs inside the GEMM)
 *   dscale[m] (+)= sum_n W[m][n] dW[m][n] + bias[m] * dbias_term     (= sum_p dy_out * y; identity in DESIGN.md)
 * w: the layer's weight in the layout dW is stored in; accumulate != 0 adds into dw / dscale / dbias.  w == NULL: dW only (any out_form).
 * layout bit 0 / bit 1: x / dy is a 512-channel tensor in the BLOCKED layout of the fused flow-step kernels (32-pixel tiles of
 * [hidden block 16][k-step 2][half 2][pixel 32][8 channels]; what nfdp_level_fused_train_bf16 and nfdp_coupling_bwd_fused_bf16
 * write) instead of [pixel][channel] rows.  ws: nfdp_conv_wgrad_bf16_ws_bytes(1, 1, npix, Cin, Cout, 1) bytes. */
int nfdp_conv_wgrad_bf16_affine(const void* x, int x_ld, const void* dy, int dy_ld, long npix, int Cin, int Cout, float* dw, void* ws,
                                int out_form, int out_c, int accumulate, int layout, const float* w, const float* bias, float* dscale,
                                float* dbias, nfdp_stream_t stream);
/* The weight gradients of ONE flow's coupling network (utils.py:72-90; autograd of trainer.py:163-164) as one GEMM launch and one
 * reduce launch: n_gemm (1..3) problems nfdp_conv_wgrad_bf16_affine would run one by one, over the same npix pixels (a multiple of 32)
 * and each with a 512-channel BLOCKED dy.  Per problem q: x[q] ([npix][x_ld[q]] bf16 rows, or BLOCKED when x_blocked[q]: then
 * cin[q] == x_ld[q] == 512), cin[q] (multiple of 8, at most 512), dw[q] / out_form[q] / out_c[q] as in nfdp_conv_wgrad_bf16_affine,
 * w[q] / bias[q] / dscale[q] / dbias[q]: the layer's ActNorm parameters and gradients (w[q] == NULL: dW only, any out_form).
 * The pixel slices are dealt to the problems in proportion to their cost, so they finish together on one chip-wide grid.
 * ws: nfdp_coupling_wgrad3_ws_bytes(npix, cin, n_gemm) bytes (0: unsupported shapes). */
size_t nfdp_coupling_wgrad3_ws_bytes(long npix, const int* cin, int n_gemm);
int nfdp_coupling_wgrad3_bf16(const void* const* x, const int* x_ld, const int* x_blocked, const void* const* dy, const int* cin,
                              float* const* dw, const int* out_form, const int* out_c, const float* const* w, const float* const* bias,
                              float* const* dscale, float* const* dbias, int n_gemm, long npix, void* ws, int accumulate,
                              nfdp_stream_t stream);
/* The same for up to nfdp_coupling_wgrad3_batch_max_flows() (16) flows of a level in ONE GEMM launch + ONE reduce launch: the GEMMs of one
 * flow at the 8x8 / 4x4 levels of a B = 512 step fill a fraction of the chip (66 / 52 us per flow against 165 us for the 16x larger
 * 16x16 flow); the flows' gradients are independent, so the backward pass keeps their operands and runs them here.  x / dy / dw / w / bias /
 * dscale / dbias: HOST tables [flows][n_gemm] row-major; x_ld / x_blocked / cin / out_form / out_c: [n_gemm], shared by the flows.
 * ws: flows * nfdp_coupling_wgrad3_batch_ws_bytes(npix, cin, n_gemm) bytes.  The slice plan does not depend on `flows`: a flow's sums are
 * the same bits whether it travels with 15 others or alone. */
size_t nfdp_coupling_wgrad3_batch_ws_bytes(long npix, const int* cin, int n_gemm);
int nfdp_coupling_wgrad3_batch_max_flows(void);
int nfdp_coupling_wgrad3_batch_bf16(const void* const* x, const int* x_ld, const int* x_blocked, const void* const* dy, const int* cin,
                                    float* const* dw, const int* out_form, const int* out_c, const float* const* w,
                                    const float* const* bias, float* const* dscale, float* const* dbias, int n_gemm, int flows, long npix,
                                    void* ws, int accumulate, nfdp_stream_t stream);
/* wt[i][o][ky][kx] = w[o][i][k-1-ky][k-1-kx]: nfdp_conv2d_nhwc on pack(wt) is the data gradient of the conv with w. */
int nfdp_conv_flip_weights_f32(const float* w, float* wt, int O, int I, int k, nfdp_stream_t stream);
/* Second half of a narrow 3x3 'same' conv computed as a pointwise GEMM to tap contributions T [B,H,W,9*Co] (column
 * tap*Co + co, taps row-major) plus this shift-sum: out[b,y,x,co] = act((sum_tap T[b,y+dy,x+dx,tap*Co+co] + add[co]) *
 * exp(mul_exp*mul[co])), zero outside the image; add, mul nullable.  Used by the training path for Cin >> Cout convs
 * (ZeroConv2d 512->C, utils.py:43, and the data gradient of the 3x3 C/2->512 conv, utils.py:67). */
int nfdp_tap_shift_sum_f32(const float* T, float* out, int B, int H, int W, int Co, const float* add, const float* mul,
                           float mul_exp, int relu, nfdp_stream_t stream);
/* Backward of y = act((u + add) * exp(f*mul)), act = ReLU (relu=1) or identity - ActNorm (transforms.py:80; f=1),
 * Conv2dActNorm+ReLU (utils.py:67-69, :83-87; f=1), ZeroConv2d (utils.py:43-44; f=3):
 *   dc = dy * [y>0] * exp(f*mul)   (gradient w.r.t. u; nullable),  dmul[c] = f * sum_p dy*y*[y>0],  dadd[c] = sum_p dc.
 * y (the saved OUTPUT) and mul nullable: with both null this is a plain per-channel sum of dy into dadd.
 * ws: nfdp_scale_bwd_ws_bytes(npix, C). */
size_t nfdp_scale_bwd_ws_bytes(long npix, int C);
/* The same with fp32 or bf16 storage of dy, y and dc; supported (dy, y, dc): (f32,f32,f32), (f32,bf16,bf16), (bf16,bf16,bf16). */
int nfdp_scale_bwd(const void* dy, int dy_dtype, int dy_ld, const void* y, int y_dtype, int y_ld, const float* mul, float f, int relu,
                   long npix, int C, void* dc, int dc_dtype, int dc_ld, float* dmul, float* dadd, void* ws, nfdp_stream_t stream);
int nfdp_scale_bwd_f32(const float* dy, int dy_ld, const float* y, int y_ld, const float* mul, float f, int relu, long npix,
                       int C, float* dc, int dc_ld, float* dmul, float* dadd, void* ws, nfdp_stream_t stream);
/* accumulate != 0: dmul / dadd are added to (the parameters' .grad in place) instead of stored. */
int nfdp_scale_bwd_ex(const void* dy, int dy_dtype, int dy_ld, const void* y, int y_dtype, int y_ld, const float* mul, float f, int relu,
                      long npix, int C, void* dc, int dc_dtype, int dc_ld, float* dmul, float* dadd, void* ws, int accumulate,
                      nfdp_stream_t stream);
/* Backward of AffineCoupling.transform (transforms.py:179-185).  x: tensor before the coupling [B,HW,C] dense,
 * h = (log_s | t) network output, dy: gradient w.r.t. the output, gl (nullable): fp64 [B] gradient w.r.t. the
 * per-sample log-det increment.  dx, dh: [B,HW,C] dense. */
int nfdp_coupling_bwd_f32(const float* x, const float* h, const float* dy, const double* gl, int B, int HW, int C, float* dx,
                          float* dh, nfdp_stream_t stream);
/* Backward of nfdp_gauss_logp_f32 (prior.py:36-37): gl fp64 [B] gradient w.r.t. logp; dz [B,HW,Cz] dense;
 * dh [B,HW,2Cz] dense per element for both h modes (mode 1 callers sum it over pixels with nfdp_scale_bwd_f32). */
int nfdp_gauss_logp_bwd_f32(const float* z, int z_ld, const float* h, int h_mode, const double* gl, int B, int HW, int Cz,
                            float* dz, float* dh, nfdp_stream_t stream);

/* Batched nfdp_conv_pack_weights_ex: one launch for every derived weight layout of a training step (forward, data-gradient and tap
 * forms of the coupling networks' convs: ~290 separate launches per step of Glow L3 K16 otherwise).  jobs_dev: n jobs of
 * NFDP_PACK_JOB_WORDS int64 words each in DEVICE memory, each filled on the host by nfdp_conv_pack_job (same arguments and packed
 * layout as nfdp_conv_pack_weights_ex; w_packed: nfdp_conv_packed_bytes_ex bytes); max_total: the largest job[10] (element count). */
int nfdp_conv_pack_job(const float* w_oihw, void* w_packed, int Cout, int Cin, int KH, int KW, int flip, int taps, long* job);
int nfdp_conv_pack_weights_batch(const long* jobs_dev, int n, long max_total, int compute, nfdp_stream_t stream);

/* Data gradient of a pointwise conv fused with the scale/ReLU backward of the layer in front of it (the two hidden layers of the
 * coupling network): dy = x W'^T stays in the accumulators; with the saved output y = relu((u + b) e^{f s}) of that layer (bf16) the
 * epilogue writes dc = [y > 0] dy e^{f s} (bf16) and returns dmul = f sum [y > 0] dy y and dadd = sum dc, i.e. what
 * nfdp_conv2d_nhwc + nfdp_scale_bwd produce in two passes with dy in memory.  x: [npix][x_ld] bf16, w_packed: nfdp_conv_pack_weights(_ex)
 * of the data-gradient weight (1x1, Cin -> Cout), Cin % 64 == 0, Cout % 8 == 0, npix >= 512 (nfdp_pointwise_dgrad_masked_supported). */
int nfdp_pointwise_dgrad_masked_supported(long npix, int Cin, int Cout);
size_t nfdp_pointwise_dgrad_masked_ws_bytes(long npix, int Cout);
int nfdp_pointwise_dgrad_masked_bf16(const void* x, int x_ld, const void* w_packed, const void* y_saved, int y_ld, const float* mul,
                                     float f, long npix, int Cin, int Cout, void* dc, int dc_ld, float* dmul, float* dadd, void* ws,
                                     nfdp_stream_t stream);
int nfdp_pointwise_dgrad_masked_bf16_ex(const void* x, int x_ld, const void* w_packed, const void* y_saved, int y_ld, const float* mul,
                                        float f, long npix, int Cin, int Cout, void* dc, int dc_ld, float* dmul, float* dadd, void* ws,
                                        int accumulate, nfdp_stream_t stream);

/* Fused optimizer step on flat fp32 buffers (trainer.py:165-167 + utils.py:120-137): for the first n_clip elements
 * g = clamp(g, +-clip_value) (clip_grad_value_; <= 0: off) then g *= min(1, max_norm / (||g[:n_clip]||_2 + 1e-6))
 * (clip_grad_norm_; <= 0: off), then torch.optim.Adam (adamw=0: weight decay added to g) or AdamW (adamw=1: decoupled
 * decay) on all n elements; step = 1-based iteration count.  The norm never leaves the device; norm_out (nullable)
 * receives it.  ws: nfdp_fused_adam_ws_bytes() bytes. */
size_t nfdp_fused_adam_ws_bytes(void);
/* The same with the learning rate and the step count on the device: hyper = double[2] {lr, steps taken so far}; the call increments
 * hyper[1] first and derives the bias corrections from it in the kernel, so a captured hipGraph of a training step replays a
 * correct Adam step every time (host-side scalars would freeze at their capture-time values). */
int nfdp_fused_adam_step_dev_f32(float* p, float* g, float* m, float* v, long n, long n_clip, float clip_value, float max_norm,
                                 double* hyper, float beta1, float beta2, float eps, float weight_decay, int adamw, void* ws,
                                 float* norm_out, nfdp_stream_t stream);
int nfdp_fused_adam_step_f32(float* p, float* g, float* m, float* v, long n, long n_clip, float clip_value, float max_norm,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int adamw, long step, void* ws,
                             float* norm_out, nfdp_stream_t stream);
/* Data-parallel form: g holds the all-reduced SUM over the ranks; grad_scale = 1/world (0 < grad_scale <= 1) is applied where the
 * clip pass / the Adam pass read the gradient anyway (g is left holding the averaged, clipped gradient, as torch's in-place clips
 * leave it) - no separate pass over the buffer.  Everything else as nfdp_fused_adam_step_dev_f32. */
int nfdp_fused_adam_step_dev_scaled_f32(float* p, float* g, float* m, float* v, long n, long n_clip, float clip_value, float max_norm,
                                        double* hyper, float beta1, float beta2, float eps, float weight_decay, int adamw,
                                        float grad_scale, void* ws, float* norm_out, nfdp_stream_t stream);

/* ---------------------------------------------------------------- backward of the U-Net glue ops (diffusion-prior training:
 * loss.backward() of diffusion_prior/trainer.py:134-150 through GaussianDiffusion.p_losses); the convolutions use
 * nfdp_conv_wgrad_f32 / nfdp_conv_flip_weights_f32 above. */
/* Block (gaussian_diffusion.py:267-276): x = input of the GroupNorm, dy = gradient of the SiLU output; scale_shift [B,2C]
 * (scale | shift, row stride ss_ld) nullable together with dscale_shift [B,2C] dense.  dgamma_b, dbeta_b: [B,C] per-sample
 * partial sums (sum them over B with nfdp_scale_bwd_f32's column-sum mode). */
int nfdp_groupnorm_silu_bwd_f32(const float* x, const float* dy, int B, int HW, int C, int groups, const float* gamma,
                                const float* beta, const float* scale_shift, int ss_ld, float eps, float* dx, float* dgamma_b,
                                float* dbeta_b, float* dscale_shift, nfdp_stream_t stream);
/* channel LayerNorm, gain only (:205-209): dx and dgx[p][c] = dy * xhat (column-sum it for the gain gradient). */
int nfdp_chan_layernorm_bwd_f32(const float* x, const float* dy, long npix, int C, const float* gain, float eps, float* dx,
                                float* dgx, nfdp_stream_t stream);
/* LinearAttention core (:320-332) and Attention core (:351-357): qkv [B,n,3*heads*32], dout [B,n,heads*32] -> dqkv; n <= 256. */
int nfdp_linear_attention_bwd_f32(const float* qkv, const float* dout, float* dqkv, int B, int n, int heads, int dim_head,
                                  float scale, nfdp_stream_t stream);
int nfdp_attention_bwd_f32(const float* qkv, const float* dout, float* dqkv, int B, int n, int heads, int dim_head, float scale,
                           nfdp_stream_t stream);
/* nn.Upsample(scale_factor=2, mode='nearest') (:169-173): dy [B,2H,2W,C] -> dx [B,H,W,C] (sum of each 2x2 block). */
int nfdp_upsample_nearest2x_bwd_f32(const float* dy, float* dx, int B, int C, int H, int W, nfdp_stream_t stream);
/* act: 1 = nn.GELU() (erf form, :409), 2 = nn.SiLU() (:283).  Forward y = act(x); backward dx = dy * act'(x). */
int nfdp_act_fwd_f32(const float* x, float* y, long n, int act, nfdp_stream_t stream);
int nfdp_act_bwd_f32(const float* x, const float* dy, float* dx, long n, int act, nfdp_stream_t stream);
/* Gradient of nfdp_diffusion_loss_f32 w.r.t. pred; grad_out: device fp32 scalar (upstream gradient of the loss). */
int nfdp_diffusion_loss_bwd_f32(const float* pred, const float* target, const long* t, const float* p2_weight,
                                const float* grad_out, int B, int per_sample, int l1, float* dpred, nfdp_stream_t stream);

/* ---------------------------------------------------------------- NLL plumbing (normalizing_flow/utils.py) */
/* out[0] = sum_b bpd_b, out[1] = B, with bpd_b = (log(n_bins)*n_pixel - (ldj_b+logp_b))*log2(e)/n_pixel
 * (utils.py:255-256; the mean is out[0]/out[1], left to the caller so ranks can all-reduce the pair). */
int nfdp_bpd_sum_f64(const double* ldj, const double* logp, int B, double n_bins, double n_pixel, double* out,
                     nfdp_stream_t stream);
/* preprocess_batch + dequantisation noise, utils.py:189-196 + trainer.py:155:
 * y = floor(u8 / 2^(8-n_bits)) / 2^n_bits - 0.5 + noise/2^n_bits   (noise may be NULL).  NCHW u8 in, NHWC f32 out. */
int nfdp_preprocess_u8_nchw_to_nhwc_f32(const uint8_t* u8, const float* noise_nchw, float* y, int B, int C, int H,
                                        int W, int n_bits, nfdp_stream_t stream);
/* Same arithmetic for a float batch in [0,1] (what the reference's loaders deliver), elementwise, any layout:
 * y = floor(x*255 / 2^(8-n_bits)) / 2^n_bits - 0.5 (+ noise/2^n_bits).  utils.py:189-196. */
int nfdp_preprocess_f32(const float* x, const float* noise, float* y, long n, int n_bits, nfdp_stream_t stream);
/* postprocess_batch, utils.py:210: clip(floor((x+.5)*n_bins)*256/n_bins, 0, 255) -> u8; NHWC f32 in, NCHW u8 out. */
int nfdp_postprocess_nhwc_to_u8_nchw(const float* x, uint8_t* u8, int B, int C, int H, int W, float n_bins,
                                     nfdp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NFDP_H */
