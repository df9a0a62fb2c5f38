/*
 * mi355seg.h -- C-ABI of libmi355seg.so: the MI355X (gfx950) hot path for 3D
 * segmentation training (U-Net family forward/backward + loss/Dice reductions).
 *
 * The reference (QingYunA/General-Medical-Image-Segmentation-CNN-Framework) has no
 * FFI: its seam for this path is torch.nn leaf modules.  Each entry point below names
 * the reference call it replaces (file:line under /root/reference) -- that is what a
 * ctypes binding on the reference side would bind.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - activations are NDHWC fp32: element (n,d,h,w,c) of a tensor with channel pitch
 *     `ld` lives at ((((n*D+d)*H+h)*W+w)*ld + c); ld >= C lets a producer write into a
 *     channel slice of a wider (concat) buffer;
 *   - weights keep the PyTorch logical layouts: Conv3d (Cout,Cin,kD,kH,kW),
 *     ConvTranspose3d (Cin,Cout,kD,kH,kW), contiguous;
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered, nothing
 *     synchronises, allocates or frees; scratch comes from the caller (`ws`);
 *   - return value: 0 on success, a negative MI355SEG_E* code otherwise;
 *     mi355seg_last_error() returns a static message for the calling thread.
 */
#ifndef MI355SEG_H
#define MI355SEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 16-bit brain-float storage element of the *_bf16 entry points (same bits as torch.bfloat16).  The HIP compiler sees its
 * native __bf16, a plain C / C++ host compiler an opaque 16-bit integer: one ABI, one storage format. */
#if defined(__HIP__) || defined(__HIPCC__)
typedef __bf16 mi355seg_bf16;
#else
typedef uint16_t mi355seg_bf16;
#endif

#define MI355SEG_OK 0
#define MI355SEG_EINVAL (-1)   /* bad shape / unsupported argument combination */
#define MI355SEG_EWORKSPACE (-2) /* workspace too small */
#define MI355SEG_EHIP (-3)     /* a HIP launch failed */

/* activation codes for the norm/activation entry points */
#define MI355SEG_ACT_NONE 0
#define MI355SEG_ACT_RELU 1   /* nn.ReLU            (unet3d.py:89,101)            */
#define MI355SEG_ACT_ELU 2    /* nn.ELU(alpha=1)    (vnet3d.py:14-18)             */
#define MI355SEG_ACT_LRELU 3  /* nn.LeakyReLU(slope) (residual_unet3d.py:17)      */
#define MI355SEG_ACT_SIGMOID 4 /* torch.sigmoid       (RE_net.py:160, ER_net.py)    */

const char* mi355seg_last_error(void);
int mi355seg_version(void);

/* Arithmetic of the MFMA convolutions (k3 / k5 Conv3d forward, input gradient and weight gradient) on fp32 tensors.  The
 * reference's arithmetic for these is ATen fp32 (unet3d.py:80-98 through nn.Conv3d); both modes accumulate in fp32:
 *   FP32   -- v_mfma_f32_32x32x2_f32, exact fp32 products;
 *   BF16X6 -- every fp32 operand is split into three bf16 parts (x = h + m + l, 24 mantissa bits) and six
 *             v_mfma_f32_32x32x16_bf16 (hh, hm, mh, mm, hl, lh) form each product: fp32-level accuracy (the dropped
 *             terms are below 2^-23 of a product) at 2.7x the fp32 matrix rate;
 *   F16X3  -- every fp32 operand, scaled by a per-tensor power of two 2^s that places the tensor's largest magnitude in
 *             [2^14, 2^15), is split into two fp16 parts (v 2^s = h + l: 11 + 11 mantissa bits and the sign of l) and three
 *             v_mfma_f32_16x16x32_f16 (hh, hl, lh) form each product; the result is scaled back by 2^-(sx + sw).  The dropped
 *             l*l term is <= 2^-22 of a product (2^-24.6 rms); values more than 2^18 below their tensor's maximum keep an
 *             absolute error of 2^-40 of that maximum.  Same fp64-graded accuracy tests as BF16X6 at half its MFMA count.
 *             The maxima are measured on the device (one pass per tensor) unless the caller hands them over (*_amax entry
 *             points).  Covers the k3 s1 forward / input gradient / weight gradient; layers without that form run BF16X6;
 * (bf16 TENSORS have their own entry points, *_bf16: there the products are plain bf16 MFMAs.)
 * Process-wide, read at each launch.  Initial value: environment MI355SEG_CONV_MATH = fp32 | bf16x6 | f16x3, else the default. */
#define MI355SEG_MATH_FP32 0
#define MI355SEG_MATH_BF16X6 2
#define MI355SEG_MATH_F16X3 3
#define MI355SEG_MATH_DEFAULT MI355SEG_MATH_F16X3
int mi355seg_set_conv_math(int mode);
int mi355seg_get_conv_math(void);
/* MFMA shape of the BF16X6 forward / input-gradient kernels: 16 = v_mfma_f32_16x16x32_bf16 (conv_x3s.hip; the default: the
 * chip holds a higher clock under it, +8-12 % on the cfg-2 layers), 32 = v_mfma_f32_32x32x16_bf16 (the generic kernel).  Same six
 * products per fp32 product either way; layers the 16-wide tiles do not cover use 32 regardless.  Process-wide, read at each
 * launch.  Initial value: environment MI355SEG_X3_SHAPE = 16 | 32. */
int mi355seg_set_x3_shape(int shape);
int mi355seg_get_x3_shape(void);
/* Tiling of the k3 / k5 stride-1 convolutions on bf16 TENSORS (forward and input gradient): 0 = automatic (the
 * v_mfma_f32_16x16x32_bf16 kernel of conv_b16s.hip -- eight 16-voxel lines x 32 channels per wave -- where a layer cuts enough
 * tiles to fill the chip, the generic 32x32x16 tiles with split-K otherwise), 1 = conv_b16s.hip wherever its geometry applies,
 * 2 = the generic tiles only.  Process-wide, read at each launch (A/B timing, tests of both kernels on one shape). */
int mi355seg_set_b16_tiles(int mode);
int mi355seg_get_b16_tiles(void);
/* The f16x3 weight gradient of k3 s1 convolutions with Cout % 64 == 0 has a second kernel (conv_wgrad_f16w_kernel: four waves, one per
 * SIMD, a 32 x 64 channel block per workgroup -- half the LDS fragment reads per MFMA).  0: never, 1 (default): where the strip of
 * tiles per workgroup is long enough for it to pay, 2: wherever the geometry allows.  Same results contract either way. */
int mi355seg_set_wgrad_wide(int mode);
int mi355seg_get_wgrad_wide(void);

/* ------------------------------------------------------------------ Conv3d
 * Replaces nn.Conv3d forward/backward (ATen convolution / convolution_backward):
 * unet3d.py:80-98 (k3 s1 p1), :46-48 (k1 head); vnet3d.py:25,47,111 (k5 p2), :65 (k2 s2);
 * residual_unet3d.py:22-79 (k3 s1/s2 p1, k1, bias=False); unetr.py:134 (k16 s16).
 * Cubic kernel k, isotropic stride/pad.  Output extent Do = (D + 2*pad - k)/stride + 1.
 */
size_t mi355seg_conv3d_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);

/* y = conv(x, w) + bias.  bias may be NULL.  If stats_sum/stats_sq are non-NULL they
 * receive, per output channel, sum(y) and sum(y*y) over all N*Do*Ho*Wo voxels as
 * doubles (the BatchNorm batch statistics, fused into the conv epilogue). */
int mi355seg_conv3d_fwd_f32(const float* x, int ldx, const float* w, const float* bias,
                            float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                            int k, int stride, int pad, double* stats_sum, double* stats_sq,
                            void* ws, size_t ws_bytes, void* stream);

/* F16X3 operand maxima.  The two-piece fp16 split needs an upper bound of max |.| of both operands of a convolution (a device
 * scalar each).  The plain entry points measure them (one pass over each operand per call); the *_ax forms take them from the
 * caller, who usually has them for free: mi355seg_norm_act_fwd_ax_f32 / mi355seg_norm_act_bwd_*_ax_f32 emit the maximum of the
 * tensor they write, a max-pool keeps its input's maximum, a parameter's maximum changes once per optimiser step
 * (mi355seg_amax_f32).  A bound above the true maximum by a factor 2^j costs j bits of headroom above the fp16 underflow; a
 * bound BELOW the true maximum by more than 2x overflows fp16 (non-finite results).  NULL = measure.  Ignored by the other maths.
 * mi355seg_conv_math_takes_amax() != 0 while the selected math uses them. */
int mi355seg_conv_math_takes_amax(void);
/* which passes of this layer read operand maxima under the selected math: bit 0 forward (x, w), bit 1 input gradient (dy, w),
 * bit 2 weight gradient (dy, x); 0 = none (hand nothing over, nothing is measured either) */
int mi355seg_conv3d_amax_use_f32(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int mi355seg_amax_f32(const float* x, int ld, long long rows, int C, float* amax /* max-combined into; zero it first */, void* stream);
int mi355seg_conv3d_fwd_ax_f32(const float* x, int ldx, const float* w, const float* bias,
                               float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                               int k, int stride, int pad, double* stats_sum, double* stats_sq, const float* x_amax, const float* w_amax,
                               void* ws, size_t ws_bytes, void* stream);
int mi355seg_conv3d_dgrad_ax_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                 int N, int D, int H, int W, int Cin, int Cout,
                                 int k, int stride, int pad, const float* dy_amax, const float* w_amax, void* ws, size_t ws_bytes, void* stream);
int mi355seg_conv3d_dgrad_bnsums_ax_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                        int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                        const float* bn_x, int ld_bnx, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                        int act, float slope, float* s1, float* s2, float* dgamma, float* dbeta, const float* dy_amax, const float* w_amax,
                                        void* ws, size_t ws_bytes, void* stream);
int mi355seg_conv3d_wgrad_ax_f32(const float* dy, int lddy, const float* x, int ldx,
                                 float* dw, float* db, int N, int D, int H, int W, int Cin, int Cout,
                                 int k, int stride, int pad, int accumulate, const float* dy_amax, const float* x_amax,
                                 void* ws, size_t ws_bytes, void* stream);

/* Inference forward (predict.py:79-81,133: model.eval()): eval-mode BatchNorm and the activation that follows it folded into the
 * convolution -- y = act(conv(x, w) * oscale[co] + oshift[co]).  oscale is multiplied into the packed weights, oshift takes the
 * bias slot and the activation runs in the MFMA kernel's epilogue: no normalise pass over y.  mi355seg_bn_fold_f32 builds the two
 * vectors from the BatchNorm parameters, running statistics and the convolution's bias.  Matrix-core layers only: ask
 * *_fused_supported_* first (non-zero = yes) and keep convolution + norm_act_fwd for the other layers. */
int mi355seg_bn_fold_f32(const float* gamma, const float* beta, const float* mean, const float* var, const float* conv_bias,
                         float eps, int C, float* scale, float* shift, void* stream);
int mi355seg_conv3d_fused_supported_f32(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy);
int mi355seg_conv3d_fwd_fused_f32(const float* x, int ldx, const float* w, const float* oscale, const float* oshift, int act, float slope,
                                  float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                  void* ws, size_t ws_bytes, void* stream);
int mi355seg_conv3d_fused_supported_bf16(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy);
int mi355seg_conv3d_fwd_fused_bf16(const mi355seg_bf16* x, int ldx, const float* w, const float* oscale, const float* oshift, int act, float slope,
                                   mi355seg_bf16* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                   void* ws, size_t ws_bytes, void* stream);

/* ---- conv2 of a double-conv block on conv1's RAW output (/root/reference/models/three_d/unet3d.py:80-101: conv1 -> norm1 -> relu1 ->
 * conv2): norm1 + relu1 run as a PROLOGUE of conv2's tile staging, in its forward and in its weight gradient, so the activation between
 * the two convolutions is never written or re-read.  f16x3 math, k3 s1 p1, act none / relu / leaky relu.
 *   mi355seg_conv3d_fwd_yamax_ax_f32   conv1's forward, also handing back max |y| (zeroed device scalar, max-combined into)
 *   mi355seg_norm_fold_f32             norm_stats_from_sums + the folded normalisation al = rstd gamma, be = beta - mean al (+ running
 *                                      statistics) + a_amax >= max |act(al x + be)| from x_amax = max |x|
 *   mi355seg_conv3d_fwd_pro_ax_f32     y = conv3d(act(pro_al[c] x + pro_be[c]), w) + bias (+ batch statistics of y); x_amax bounds the prologue's output
 *   mi355seg_conv3d_wgrad_pro_ax_f32   dw (+ db) of that convolution, the same prologue on its x operand
 * Same values as norm_act_fwd followed by the plain entry points (the prologue is the same fma + activation per element). */
int mi355seg_conv3d_pro_supported_f32(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int act);
int mi355seg_conv3d_fwd_yamax_ax_f32(const float* x, int ldx, const float* w, const float* bias,
                                     float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                                     int k, int stride, int pad, double* stats_sum, double* stats_sq, const float* x_amax, const float* w_amax,
                                     float* y_amax, void* ws, size_t ws_bytes, void* stream);
int mi355seg_norm_fold_f32(const double* sum, const double* sq, long long rows, int C, float eps, const float* gamma, const float* beta, int act,
                           float* mean, float* rstd, float* running_mean, float* running_var, float momentum,
                           const float* x_amax, float* al, float* be, float* a_amax, void* stream);
int mi355seg_conv3d_fwd_pro_ax_f32(const float* x, int ldx, const float* pro_al, const float* pro_be, int pro_act, float pro_slope,
                                   const float* w, const float* bias, float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                                   int k, int stride, int pad, double* stats_sum, double* stats_sq, const float* x_amax, const float* w_amax,
                                   void* ws, size_t ws_bytes, void* stream);
int mi355seg_conv3d_wgrad_pro_ax_f32(const float* dy, int lddy, const float* x, int ldx, const float* pro_al, const float* pro_be, int pro_act, float pro_slope,
                                     float* dw, float* db, int N, int D, int H, int W, int Cin, int Cout,
                                     int k, int stride, int pad, int accumulate, const float* dy_amax, const float* x_amax,
                                     void* ws, size_t ws_bytes, void* stream);

/* The 1-channel stem of a double-conv block (/root/reference/models/three_d/unet3d.py:80-89: enc1conv1 -> enc1norm1 -> enc1relu1) when the block's
 * input needs no gradient: dw (+ db) of the stem straight from da = d(activation) and the pre-norm tensor y.  The norm backward's apply half
 * (dy = rstd gamma (dz - s1 / rows - xhat s2 / rows), dz = da act'(z); s1 / s2 = its column sums, e.g. from mi355seg_conv3d_dgrad_bnsums_f32) is
 * formed inside the weight-gradient kernel: the same values as mi355seg_norm_act_bwd_apply_f32 + mi355seg_conv3d_wgrad_f32 without writing dy. */
int mi355seg_stem_wgrad_bnbwd_supported_f32(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int mi355seg_stem_wgrad_bnbwd_f32(const float* da, int ldda, const float* y, int ldy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  int act, float slope, const float* s1, const float* s2, const float* x, int ldx, float* dw, float* db,
                                  int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, void* ws, size_t ws_bytes, void* stream);

/* dx = conv_backward_input(dy, w).  D,H,W are the INPUT extents (of x/dx). */
int mi355seg_conv3d_dgrad_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                              int N, int D, int H, int W, int Cin, int Cout,
                              int k, int stride, int pad, void* ws, size_t ws_bytes, void* stream);

/* The same input gradient for a convolution whose input was act(BatchNorm(bn_x)) of the layer in front (conv2 of a double-conv
 * block, unet3d.py:92-101 behind :80-89), plus the two column sums that layer's norm backward starts with: s1[c] = sum dz,
 * s2[c] = sum dz * xhat with dz = dx * act'(z) (dgamma = s2, dbeta = s1; may be NULL).  On the bf16x6 k3 path the sums come out of
 * the input-gradient kernel's epilogue (one read of bn_x, no pass over dx); elsewhere this is conv3d_dgrad followed by
 * norm_act_bwd_sums.  Finish the layer in front with mi355seg_norm_act_bwd_apply_f32.  ws: max(conv3d_ws_bytes, norm_ws_bytes). */
int mi355seg_conv3d_dgrad_bnsums_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                     int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                     const float* bn_x, int ld_bnx, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     int act, float slope, float* s1, float* s2, float* dgamma, float* dbeta,
                                     void* ws, size_t ws_bytes, void* stream);

/* dw (Cout,Cin,k,k,k) and db (Cout, may be NULL) = conv_backward_weight(dy, x).
 * Deterministic (two-stage reduction, no atomics).  accumulate!=0 adds into dw/db
 * (weight sharing, residual_unet3d.py:126,128). */
int mi355seg_conv3d_wgrad_f32(const float* dy, int lddy, const float* x, int ldx,
                              float* dw, float* db, int N, int D, int H, int W, int Cin, int Cout,
                              int k, int stride, int pad, int accumulate,
                              void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ ConvTranspose3d k2 s2
 * Replaces nn.ConvTranspose3d(kernel_size=2, stride=2): unet3d.py:29-43, vnet3d.py:86,
 * unetr.py:11.  x is [N,D,H,W,Cin]; y is [N,2D,2H,2W,Cout]; w is (Cin,Cout,2,2,2).
 */
size_t mi355seg_convt3d_k2s2_ws_bytes(int N, int D, int H, int W, int Cin, int Cout);
int mi355seg_convt3d_k2s2_fwd_f32(const float* x, int ldx, const float* w, const float* bias,
                                  float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                                  void* ws, size_t ws_bytes, void* stream);
/* the same with max |y| as a by-product (F16X3 operand maxima: the up-convolution half of a decoder's concat buffer) */
int mi355seg_convt3d_k2s2_fwd_ax_f32(const float* x, int ldx, const float* w, const float* bias,
                                  float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                                  float* y_amax,
                                     void* ws, size_t ws_bytes, void* stream);
int mi355seg_convt3d_k2s2_dgrad_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                    int N, int D, int H, int W, int Cin, int Cout,
                                    void* ws, size_t ws_bytes, void* stream);
int mi355seg_convt3d_k2s2_wgrad_f32(const float* dy, int lddy, const float* x, int ldx,
                                    float* dw, float* db, int N, int D, int H, int W, int Cin, int Cout,
                                    void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ Batch / Instance norm
 * Replaces nn.BatchNorm3d in training mode (native_batch_norm, unet3d.py:88,100;
 * vnet3d.py:27,49,66,88,112) and nn.InstanceNorm3d (residual_unet3d.py:27..106), fused
 * with the activation that follows it and an optional residual add in front of the
 * activation (vnet3d.py:57-58,79,103).
 * rows = voxels per statistics group, groups = 1 (BatchNorm: rows = N*D*H*W) or N
 * (InstanceNorm: rows = D*H*W).  mean/rstd are [groups*C] floats.
 */
size_t mi355seg_norm_ws_bytes(long long rows, int groups, int C);

/* Batch statistics of x: mean and rstd = 1/sqrt(var_biased + eps).  If running_mean /
 * running_var are non-NULL (groups must be 1) they are updated in place with
 * `momentum`, running_var with the UNBIASED variance (PyTorch semantics). */
int mi355seg_norm_stats_f32(const float* x, int ldx, long long rows, int groups, int C, float eps,
                            float* mean, float* rstd, float* running_mean, float* running_var,
                            float momentum, void* ws, size_t ws_bytes, void* stream);

/* Same, but from per-channel sum / sum-of-squares (as produced by the conv epilogue). */
int mi355seg_norm_stats_from_sums_f32(const double* sum, const double* sq, long long rows, int C, float eps,
                                      float* mean, float* rstd, float* running_mean, float* running_var,
                                      float momentum, void* stream);

/* y = act( (x-mean)*rstd*gamma + beta  [+ res] ).  gamma/beta/res may be NULL. */
int mi355seg_norm_act_fwd_f32(const float* x, int ldx, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, const float* res, int ldres,
                              float* y, int ldy, long long rows, int groups, int C,
                              int act, float slope, void* stream);

/* Backward of the above.  Inputs: dy (grad of the activation output), x (the norm
 * input).  Outputs: dx; dgamma/dbeta [C] (NULL when the norm has no affine); dres (grad
 * of the residual = grad at the activation input; NULL if no residual). */
int mi355seg_norm_act_bwd_f32(const float* dy, int lddy, const float* x, int ldx,
                              const float* mean, const float* rstd, const float* gamma, const float* beta,
                              const float* res, int ldres,
                              float* dx, int lddx, float* dgamma, float* dbeta, float* dres, int lddres,
                              long long rows, int groups, int C, int act, float slope,
                              void* ws, size_t ws_bytes, void* stream);

/* Same, and additionally dx_colsum[c] = sum over rows of dx[r, c] (groups == 1): when the norm follows a
 * convolution that is exactly the convolution's bias gradient, so the separate reduction pass over dy is saved. */
int mi355seg_norm_act_bwd_colsum_f32(const float* dy, int lddy, const float* x, int ldx,
                                     const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     const float* res, int ldres,
                                     float* dx, int lddx, float* dgamma, float* dbeta, float* dres, int lddres, float* dx_colsum,
                                     long long rows, int groups, int C, int act, float slope,
                                     void* ws, size_t ws_bytes, void* stream);

/* BatchNorm + activation of the LAST double-conv block fused with the 1x1x1 output head (/root/reference/models/three_d/unet3d.py:46-48,
 * 68-71: ``self.conv(dec1)`` behind decoder1's norm2 (:100) + relu2 (:101); head = nn.Conv3d(features, out_channels, kernel_size=1)).
 * The activation a = act(BN(y)) has the head as its only consumer and is never written:
 *   fwd        logits[r, k] = bh[k] + sum_c wh[k, c] * act(gamma[c] * xhat[r, c] + beta[c])          (same bits as norm_act_fwd + conv3d_fwd k1)
 *   bwd_sums   one pass over (y, dlogits): s1 / s2 of the norm backward (dz = (sum_k dlogits[r, k] wh[k, c]) * act'(z)), dgamma = s2,
 *              dbeta = s1, AND the head's gradients dwh[k, c] = sum_r dlogits[r, k] a[r, c], dbh[k] = sum_r dlogits[r, k]
 *   bwd_apply  dy = rstd gamma (dz - s1 / rows - xhat s2 / rows), dy_colsum[c] = sum_r dy (bias gradient of the convolution in front),
 *              dy_amax (f16x3 operand maximum; max-combined into a zeroed scalar).  C a power of two in 4..256, K = 1..4, fp32.
 * ws: mi355seg_bn_act_head_ws_bytes(C, K). */
int mi355seg_bn_act_head_supported_f32(long long rows, int C, int K, int ldy);
size_t mi355seg_bn_act_head_ws_bytes(int C, int K);
int mi355seg_bn_act_head_fwd_f32(const float* y, int ldy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 int act, float slope, const float* wh, const float* bh, float* logits, int ldl,
                                 long long rows, int C, int K, void* stream);
int mi355seg_bn_act_head_bwd_sums_f32(const float* dlogits, int lddl, const float* y, int ldy, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, int act, float slope, const float* wh,
                                      float* s1, float* s2, float* dgamma, float* dbeta, float* dwh, float* dbh,
                                      long long rows, int C, int K, void* ws, size_t ws_bytes, void* stream);
int mi355seg_bn_act_head_bwd_apply_f32(const float* dlogits, int lddl, const float* y, int ldy, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, int act, float slope, const float* wh,
                                       const float* s1, const float* s2, float* dy, int lddy, float* dy_colsum, float* dy_amax,
                                       long long rows, int C, int K, void* ws, size_t ws_bytes, void* stream);

/* BatchNorm + activation of an ENCODER block fused with the MaxPool3d(2, 2) behind it (/root/reference/models/three_d/unet3d.py:19-25,
 * 51-58: ``self.pool1(enc1)`` with enc1 = relu2(norm2(.)) also kept for the skip concatenation :59-68).  fwd: one kernel writes the
 * activation a (pitch lda: a channel slice of the level's concat buffer), the pooled tensor [N, D/2, H/2, W/2, C] and the 3-bit argmax
 * codes (nn.MaxPool3d's first-maximum / NaN rule), + max |a| into a zeroed scalar (may be NULL).  bwd: the norm backward with
 * d(a) = dskip + maxpool_backward(dpooled, idx) formed on the fly -- no pool-backward pass, no d(a) tensor: s1 / s2 (dgamma = s2,
 * dbeta = s1; may be NULL), dy, dy_colsum (may be NULL), dy_amax (may be NULL).  D, H, W even, C a power of two in 4..256, fp32. */
int mi355seg_bn_act_pool_supported_f32(int N, int D, int H, int W, int C, int ldy);
size_t mi355seg_bn_act_pool_ws_bytes(int C);
int mi355seg_bn_act_pool_fwd_f32(const float* y, int ldy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 int act, float slope, float* a, int lda, float* pooled, unsigned char* idx, float* a_amax,
                                 int N, int D, int H, int W, int C, void* stream);
int mi355seg_bn_act_pool_bwd_f32(const float* dskip, int ldds, const float* dpooled, const unsigned char* idx, const float* y, int ldy,
                                 const float* mean, const float* rstd, const float* gamma, const float* beta, int act, float slope,
                                 float* s1, float* s2, float* dgamma, float* dbeta, float* dy, int lddy, float* dy_colsum, float* dy_amax,
                                 int N, int D, int H, int W, int C, void* ws, size_t ws_bytes, void* stream);

/* The same backward in its two halves (the conv -> BN -> ReLU -> conv chains of unet3d.py:73-104: the first half can ride in the
 * epilogue of the kernel that produces dy, see mi355seg_conv3d_dgrad_bnsums_f32):
 *   sums:  s1[g, c] = sum_rows dz, s2[g, c] = sum_rows dz * xhat, dz = dy * act'(z)  (+ dgamma = s2, dbeta = s1; groups == 1)
 *   apply: dx = rstd * gamma * (dz - s1 / rows - xhat * s2 / rows)  (+ dres = dz; + dx_colsum[c] = sum_rows dx[r, c]) */
int mi355seg_norm_act_bwd_sums_f32(const float* dy, int lddy, const float* x, int ldx,
                                   const float* mean, const float* rstd, const float* gamma, const float* beta,
                                   const float* res, int ldres, float* s1, float* s2, float* dgamma, float* dbeta,
                                   long long rows, int groups, int C, int act, float slope,
                                   void* ws, size_t ws_bytes, void* stream);
int mi355seg_norm_act_bwd_apply_f32(const float* dy, int lddy, const float* x, int ldx,
                                    const float* mean, const float* rstd, const float* gamma, const float* beta,
                                    const float* res, int ldres, const float* s1, const float* s2,
                                    float* dx, int lddx, float* dres, int lddres, float* dx_colsum,
                                    long long rows, int groups, int C, int act, float slope,
                                    void* ws, size_t ws_bytes, void* stream);

/* The same three with the maximum magnitude of the tensor they WRITE as a by-product (F16X3 operand maxima, see
 * mi355seg_conv3d_fwd_ax_f32): max |y| resp. max |dx| is max-combined into the device scalar the last pointer names (zeroed by the
 * caller; NULL = not wanted).  One compare per element in kernels that are HBM-bound, one atomic max per workgroup. */
int mi355seg_norm_act_fwd_ax_f32(const float* x, int ldx, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, const float* res, int ldres,
                                 float* y, int ldy, long long rows, int groups, int C,
                                 int act, float slope, float* y_amax, void* stream);
int mi355seg_norm_act_bwd_colsum_ax_f32(const float* dy, int lddy, const float* x, int ldx,
                                        const float* mean, const float* rstd, const float* gamma, const float* beta,
                                        const float* res, int ldres,
                                        float* dx, int lddx, float* dgamma, float* dbeta, float* dres, int lddres, float* dx_colsum, float* dx_amax,
                                        long long rows, int groups, int C, int act, float slope,
                                        void* ws, size_t ws_bytes, void* stream);
int mi355seg_norm_act_bwd_apply_ax_f32(const float* dy, int lddy, const float* x, int ldx,
                                       const float* mean, const float* rstd, const float* gamma, const float* beta,
                                       const float* res, int ldres, const float* s1, const float* s2,
                                       float* dx, int lddx, float* dres, int lddres, float* dx_colsum, float* dx_amax,
                                       long long rows, int groups, int C, int act, float slope,
                                       void* ws, size_t ws_bytes, void* stream);

/* Eval-mode BatchNorm (running stats) is norm_act_fwd with mean=running_mean and
 * rstd = 1/sqrt(running_var+eps) computed by: */
int mi355seg_rstd_from_var_f32(const float* var, float eps, float* rstd, int C, void* stream);

/* Stand-alone activation (residual_unet3d.py:112,116 LeakyReLU; vnet ELU): y = act(x [+ res]) */
/* dx = addend + dy * act'(x + res) (r5): an activation's backward together with the sum of a SECOND gradient of the same tensor -- the
 * residual forks of /root/reference/models/three_d/residual_unet3d.py:110-121 (the level's tensor feeds a LeakyReLU and, unchanged, the
 * block sum): one pass instead of the activation backward and autograd's add.  addend NULL: mi355seg_act_bwd. */
int mi355seg_act_bwd_add_f32(const float* dy, int lddy, const float* x, int ldx, const float* res, int ldres, const float* addend, int ldadd,
                             float* dx, int lddx, long long rows, int C, int act, float slope, void* stream);
int mi355seg_act_fwd_f32(const float* x, int ldx, const float* res, int ldres, float* y, int ldy,
                         long long rows, int C, int act, float slope, void* stream);
/* dx = dy * act'(x [+ res]) */
int mi355seg_act_bwd_f32(const float* dy, int lddy, const float* x, int ldx, const float* res, int ldres,
                         float* dx, int lddx, long long rows, int C, int act, float slope, void* stream);

/* nn.PReLU(C) -- V-Net's elu=False branch (vnet3d.py:14-18): y = max(z, 0) + slope[c] * min(z, 0), z = x [+ res].
 * Backward: dx = dy * (z > 0 ? 1 : slope[c]); dslope[c] = sum over rows of dy * min(z, 0) (two-stage, deterministic).
 * ws for the backward: mi355seg_norm_ws_bytes(rows, 1, C). */
int mi355seg_prelu_fwd_f32(const float* x, int ldx, const float* res, int ldres, const float* slope, float* y, int ldy,
                           long long rows, int C, void* stream);
int mi355seg_prelu_bwd_f32(const float* dy, int lddy, const float* x, int ldx, const float* res, int ldres, const float* slope,
                           float* dx, int lddx, float* dslope, long long rows, int C, void* ws, size_t ws_bytes, void* stream);

/* nn.Dropout3d (vnet3d.py:90,99; residual_unet3d.py:18): y[r,c] = x[r,c] * scale[g*C + c], the per-(sample,
 * channel) keep-mask / (1-p) being supplied by the caller (device RNG in production, the oracle's mask in
 * parity tests).  The backward is the same call on dy. */
int mi355seg_scale_channels_f32(const float* x, int ldx, const float* scale, float* y, int ldy,
                                long long rows, int groups, int C, void* stream);

/* ------------------------------------------------------------------ Pool / upsample
 * nn.MaxPool3d(2,2) (unet3d.py:19-25): y [N,D/2,H/2,W/2,C]; idx = 3-bit argmax code
 * (dz*4+dy*2+dx of the first maximum in PyTorch scan order) per output element. */
int mi355seg_maxpool2_fwd_f32(const float* x, int ldx, float* y, int ldy, uint8_t* idx,
                              int N, int D, int H, int W, int C, void* stream);
int mi355seg_maxpool2_bwd_f32(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx,
                              int N, int D, int H, int W, int C, void* stream);
/* dx = maxpool_backward(dy) + add: the encoder activation feeds both the pool and the skip connection
 * (unet3d.py:51-54,59-68); this folds autograd's gradient accumulation of the two branches into the pool backward. */
int mi355seg_maxpool2_bwd_add_f32(const float* dy, int lddy, const uint8_t* idx, const float* add, int ldadd, float* dx, int lddx,
                                  int N, int D, int H, int W, int C, void* stream);
/* nn.Upsample(scale_factor=2, mode='nearest') (residual_unet3d.py:19,103) */
int mi355seg_upsample2_fwd_f32(const float* x, int ldx, float* y, int ldy,
                               int N, int D, int H, int W, int C, void* stream);
int mi355seg_upsample2_bwd_f32(const float* dy, int lddy, float* dx, int lddx,
                               int N, int D, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------ Loss / argmax / Dice
 * All take NCDHW-contiguous logits/targets [N,K,S] (S = D*H*W), as train.py hands them.
 */
size_t mi355seg_loss_ws_bytes(long long numel);

/* nn.BCEWithLogitsLoss() mean (train.py:115,209; loss_function.py:19-41):
 * loss[0] = mean( max(x,0) - x*t + log1p(exp(-|x|)) ). */
int mi355seg_bce_logits_fwd_f32(const float* logits, const float* target, long long numel,
                                float* loss, void* ws, size_t ws_bytes, void* stream);
/* dlogits = (sigmoid(x) - t) * gscale[0] / numel   (gscale = upstream grad, device scalar) */
int mi355seg_bce_logits_bwd_f32(const float* logits, const float* target, const float* gscale,
                                long long numel, float* dlogits, void* stream);

/* pred.argmax(dim=1, keepdim=True) (train.py:204, predict.py:139): first max wins; int64 out [N,1,S] */
int mi355seg_argmax_ch_f32(const float* logits, long long N, int K, long long S, int64_t* mask, void* stream);

/* Integer counters of utils/metric.py:34-43 on two int64 label volumes:
 * counts[0]=sum(gt) counts[1]=sum(pred) counts[2]=nnz(gt&pred) counts[3]=nnz(gt|pred). */
int mi355seg_dice_counts_i64(const int64_t* gt, const int64_t* pred, long long numel,
                             int64_t* counts, void* ws, size_t ws_bytes, void* stream);

/* The 2-channel target of the live loop (train.py:190-193): out[n][0] = (gt[n] == 0), out[n][1] = gt[n], float [N,2,S]. */
int mi355seg_two_channel_gt_f32(const float* gt, float* out, long long N, long long S, void* stream);

/* Fused train-step tail (train.py:204,209,221 in one pass over the logits):
 * BCE mean loss, argmax mask (int64), gt.argmax and the four Dice counters.
 * target is the K-channel float one-hot [N,K,S]. */
int mi355seg_bce_argmax_dice_f32(const float* logits, const float* target, long long N, int K, long long S,
                                 float* loss, int64_t* mask, int64_t* counts,
                                 void* ws, size_t ws_bytes, void* stream);

/* Sum-type reductions for the library losses (loss_function.py:61-185):
 * out[0]=sum(a*b) out[1]=sum(a) out[2]=sum(b) out[3]=sum(a*a) out[4]=sum(b*b), a = sigmoid(x) if
 * apply_sigmoid else x.  Deterministic wavefront-shuffle + two-stage reduce. */
int mi355seg_dice_sums_f32(const float* x, const float* t, long long numel, int apply_sigmoid,
                           double* out5, void* ws, size_t ws_bytes, void* stream);

/* d/dx of  L(S0..S4)  with S = dice_sums(x, t): dx[i] = (g[0]*t + g[1] + 2*g[3]*a) * da/dx, a = sigmoid(x) or x.
 * g = dL/dS as 5 device doubles (g[2], g[4] multiply target-only sums and do not reach x). */
int mi355seg_dice_sums_bwd_f32(const float* x, const float* t, const double* g5, long long numel, int apply_sigmoid,
                               float* dx, void* stream);

/* The same five sums for `rows` contiguous rows of `len` elements in ONE launch: out[r*5 + j].  BinaryDiceLoss reduces per
 * sample (loss_function.py:78-83: predict.view(N, -1)), DiceLossss per (sample, class) (loss_function.py:160-183).  `p` is
 * BinaryDiceLoss's denominator exponent (loss_function.py:82, any value): out[r*5+3] = sum a^p, out[r*5+4] = sum b^p. */
size_t mi355seg_dice_rows_ws_bytes(long long rows, long long len);
int mi355seg_dice_rows_f32(const float* x, const float* t, long long rows, long long len, int apply_sigmoid, float p,
                           double* out, void* ws, size_t ws_bytes, void* stream);
/* dx[r][i] = (g[r*5+0]*t + g[r*5+1] + g[r*5+3] * p * a^(p-1)) * da/dx */
int mi355seg_dice_rows_bwd_f32(const float* x, const float* t, const double* g, long long rows, long long len, int apply_sigmoid,
                               float p, float* dx, void* stream);

/* tio.ZNormalization() of one volume (dataloader.py:94): y = (x - mean) / std over all n voxels, unbiased std, fp64 sums;
 * in place (y == x) allowed.  The device patch queue normalises each uploaded volume once with it. */
size_t mi355seg_znorm_ws_bytes(long long n);
int mi355seg_znorm_f32(const float* x, long long n, float* y, void* ws, size_t ws_bytes, void* stream);

/* softmax over the channel dim of an NCDHW tensor [N,K,S] (DiceLossss softmax=True, loss_function.py:170-171) */
int mi355seg_softmax_ch_f32(const float* x, float* y, long long N, int K, long long S, void* stream);
/* dx = y * (dy - sum_k dy*y) */
int mi355seg_softmax_ch_bwd_f32(const float* y, const float* dy, float* dx, long long N, int K, long long S, void* stream);

/* cross_entropy_3D (loss_function.py:8-16): loss[0] = sum_v w[label_v] * (logsumexp_k x[v,k] - x[v,label_v]),
 * divided by the voxel count when size_average.  logits NCDHW [N,K,S], labels int64 [N,S], weight [K] or NULL. */
int mi355seg_ce3d_fwd_f32(const float* logits, const int64_t* labels, const float* weight, long long N, int K, long long S,
                          int size_average, float* loss, void* ws, size_t ws_bytes, void* stream);
int mi355seg_ce3d_bwd_f32(const float* logits, const int64_t* labels, const float* weight, const float* gscale,
                          long long N, int K, long long S, int size_average, float* dlogits, void* stream);

/* ------------------------------------------------------------------ UNETR encoder (unetr.py:54-168)
 * Strided, batched fp32 GEMM on the MFMA:  C[b0,b1][m][n] (+)= alpha * sum_k A[..][m][k] * B[..][k][n]  (+ bias[n]) (relu).
 * Element strides: A(m,k) at a_rs*m + a_cs*k + a_b0*b0 + a_b1*b1, B(k,n) likewise, C(m,n) at c_rs*m + n + c_b0*b0 + c_b1*b1.
 * Replaces nn.Linear (unetr.py:61-66,120-121), torch.matmul of the attention (unetr.py:88,94) and their backward. */
int mi355seg_gemm_f32(const float* A, long long a_rs, long long a_cs, long long a_b0, long long a_b1,
                      const float* B, long long b_rs, long long b_cs, long long b_b0, long long b_b1,
                      float* C, long long c_rs, long long c_b0, long long c_b1, const float* bias,
                      int M, int N, int K, int nb0, int nb1, float alpha, int relu, int accumulate,
                      void* ws, size_t ws_bytes, void* stream);
/* The same GEMM with its products on the bf16 matrix cores: fp32 operands rounded to bf16 (RNE) in registers, v_mfma_f32_32x32x16_bf16,
 * fp32 accumulation and result -- the arithmetic of the reference's nn.Linear / matmul under torch.autocast(bfloat16)
 * (/root/reference/models/three_d/unetr.py:59-138), used by the token path inside functional.autocast(torch.bfloat16).  Few-hundred-row
 * shapes (the small-GEMM kernel); other shapes run the fp32 kernels of mi355seg_gemm_f32. */
int mi355seg_gemm_lowp_f32(const float* A, long long a_rs, long long a_cs, long long a_b0, long long a_b1,
                      const float* B, long long b_rs, long long b_cs, long long b_b0, long long b_b1,
                      float* C, long long c_rs, long long c_b0, long long c_b1, const float* bias,
                      int M, int N, int K, int nb0, int nb1, float alpha, int relu, int accumulate,
                      void* ws, size_t ws_bytes, void* stream);
/* nn.Linear forward with the element-wise operations that follow it in the token encoder in the GEMM's epilogue (r5;
 * /root/reference/models/three_d/unetr.py:98-100,120-138,159-166): y[M][N] = (relu?)(x W^T + b) * emul + eadd, emul = a dropout layer's
 * keep / (1 - p) factors, eadd = the residual stream the block adds its output to, both [M][N] at pitch N, either may be NULL.  One launch
 * on the few-hundred-row shapes (each of the two was a launch of its own); other shapes: the GEMM, then the element-wise kernels in place.
 * lowp != 0: products on the bf16 matrix cores (mi355seg_gemm_lowp_f32).  ws: mi355seg_gemm_ws_bytes(M, N, K, 1, 1). */
int mi355seg_linear_fwd_f32(int lowp, const float* x, int ldx, const float* w, const float* b, int relu, const float* emul, const float* eadd,
                            float* y, int M, int N, int K, void* ws, size_t ws_bytes, void* stream);
/* Weight and bias gradient of nn.Linear in one call (r5; /root/reference/models/three_d/unetr.py:61-66,120-121 backward):
 * dw[N][K] = dy^T x and db[n] = sum_m dy[m][n] -- on the few-hundred-row shapes of the token encoder the bias gradient is summed by the
 * weight-gradient GEMM's first column of tiles from the dy values it loads anyway (one launch instead of two per layer; fixed
 * summation order); other shapes run the GEMM and mi355seg_colsum_f32.  lowp != 0: products on the bf16 matrix cores
 * (mi355seg_gemm_lowp_f32).  db may be NULL.  ws: max(mi355seg_gemm_ws_bytes(N, K, M, 1, 1), mi355seg_norm_ws_bytes(M, 1, N)). */
int mi355seg_linear_wgrad_f32(int lowp, const float* dy, int lddy, const float* x, int ldx, float* dw, float* db, int M, int N, int K,
                              void* ws, size_t ws_bytes, void* stream);
/* The backward of nn.Linear as ONE launch (r6; /root/reference/models/three_d/unetr.py:61-66,98-100,120-138 backward): with
 * dyf = dy * dmul * [dgate > 0] (dmul: the keep / (1 - p) factors of the dropout layer behind the Linear, dgate: the ReLU's saved output; either
 * may be NULL) it writes dx[M][K] = dyf W, dw[N][K] = dyf^T x and db[n] = sum_m dyf[m][n] (db may be NULL).  The two GEMMs are independent and
 * latency-bound at the token encoder's sizes (216 rows): they run as two problems of one grid, the element-wise factors folded into their loads
 * of dy (they were an element-wise launch each), every result bit-identical to the separate launches'.  bf16 products (lowp != 0) on shapes
 * mi355seg_linear_bwd_supported_f32 accepts; the caller runs mi355seg_gemm_lowp_f32 + mi355seg_linear_wgrad_f32 otherwise. */
int mi355seg_linear_bwd_supported_f32(int lowp, int M, int N, int K);
int mi355seg_linear_bwd_f32(int lowp, const float* dy, int lddy, const float* dmul, const float* dgate, const float* x, int ldx, const float* w,
                            float* dx, float* dw, float* db, int M, int N, int K, void* stream);
/* Two independent batched small GEMMs C_p[b0][b1] = alpha_p A_p B_p (p = 0, 1; nb0 x nb1 batches each, own strides) in one launch (r6:
 * the pairs of attention's backward, dP = dO V^T with dV = Pd^T dO and dQ = dS K with dK = dS^T Q, unetr.py:74-98 backward).  bf16 products,
 * fp32 accumulation; each result bit-identical to mi355seg_gemm_lowp_f32's.  Shapes: mi355seg_gemm_pair_supported_f32. */
int mi355seg_gemm_pair_supported_f32(int M0, int N0, int K0, int M1, int N1, int K1, int nb0, int nb1);
int mi355seg_gemm_pair_lowp_f32(const float* A0, long long a0_rs, long long a0_cs, long long a0_b0, long long a0_b1,
                                const float* B0, long long b0_rs, long long b0_cs, long long b0_b0, long long b0_b1,
                                float* C0, long long c0_rs, long long c0_b0, long long c0_b1, int M0, int N0, int K0, float alpha0,
                                const float* A1, long long a1_rs, long long a1_cs, long long a1_b0, long long a1_b1,
                                const float* B1, long long b1_rs, long long b1_cs, long long b1_b0, long long b1_b1,
                                float* C1, long long c1_rs, long long c1_b0, long long c1_b1, int M1, int N1, int K1, float alpha1,
                                int nb0, int nb1, void* stream);
/* Scratch for the deterministic split-K path (single-batch GEMMs with too few 64x64 tiles to fill 256 CUs); 0 when
 * the shape is not split.  With a smaller / NULL workspace the GEMM runs unsplit. */
size_t mi355seg_gemm_ws_bytes(int M, int N, int K, int nb0, int nb1);
/* nn.LayerNorm(E, eps) over the last dim of [rows, E] (unetr.py:151-152); mean/rstd [rows] are saved for backward */
int mi355seg_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                               long long rows, int E, float eps, void* stream);
int mi355seg_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                               float* dx, float* dgamma, float* dbeta, long long rows, int E, void* stream);
/* The same with dx = addend + (LayerNorm backward of dy): `addend` is a second gradient of x -- the residual stream that bypasses the norm in a
 * pre-norm transformer block (x + f(LN(x)), /root/reference/models/three_d/unetr.py:159-166) -- summed in the kernel that writes dx instead of
 * by a separate add (r5).  addend NULL: mi355seg_layernorm_bwd_f32. */
int mi355seg_layernorm_bwd_add_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* addend,
                                   float* dx, float* dgamma, float* dbeta, long long rows, int E, void* stream);
/* nn.Softmax(dim=-1) on [rows, L] (unetr.py:71,90) and its backward dx = y * (dy - sum(dy*y)) */
int mi355seg_softmax_rows_f32(const float* x, float* y, long long rows, int L, void* stream);
int mi355seg_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, long long rows, int L, void* stream);
/* softmax over the last dim followed by an elementwise factor (the attention dropout of /root/reference/models/three_d/unetr.py:92-93,
 * keep = mask / (1 - p)): y = softmax(x) and yk = y * keep from one pass over x; backward of the pair from d(yk): dx = y (dy - sum y dy),
 * dy = dyk * keep (r5: three launches per attention layer and direction were softmax, mask product, [mask product]). */
int mi355seg_softmax_rows_keep_f32(const float* x, const float* keep, float* y, float* yk, long long rows, int L, void* stream);
int mi355seg_softmax_rows_keep_bwd_f32(const float* y, const float* dyk, const float* keep, float* dx, long long rows, int L, void* stream);
/* out[c] = sum over rows of x[r, c] (bias gradient of nn.Linear); ws >= mi355seg_norm_ws_bytes(rows, 1, C) */
int mi355seg_colsum_f32(const float* x, int ldx, long long rows, int C, float* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ In-library kernel timing
 * Optional HIP-event timing of the kernel families, on the stream each kernel is launched
 * on (used by bench.py for the live roofline numbers; off by default, zero cost when off).
 * Families: 0 conv implicit-GEMM (fwd+dgrad), 1 conv wgrad MFMA, 2 conv generic, 3 convT,
 * 4 norm/act, 5 pool/upsample, 6 loss/metric, 7 stem/head direct conv.
 * prof_read synchronises the recorded events and returns, per family f:
 *   out[4*f+0] = launches, out[4*f+1] = total milliseconds,
 *   out[4*f+2] = total algorithmic FLOPs, out[4*f+3] = total algorithmic bytes. */
#define MI355SEG_PROF_FAMILIES 8
/* on: 0 = off, 1 = every family, otherwise 2 * (bit mask of families): only those launches are bracketed, so the
 * event records do not serialise the many small kernels of a step that is being timed as a whole. */
int mi355seg_prof_enable(int on);
int mi355seg_prof_reset(void);
int mi355seg_prof_read(double* out_host, int n_doubles);
/* per-launch records: writes up to max_records rows of 4 doubles (family, ms, flops, bytes); returns the
 * number of rows through *n_host (call after prof_read, which synchronises). */
int mi355seg_prof_records(double* out_host, int max_records, int* n_host);

/* ------------------------------------------------------------------ Sliding-window inference (predict.py:98-147)
 * tio.inference.GridSampler patches of one volume [C,D,H,W] as a device-resident batch [count,C,pd,ph,pw], and
 * tio.inference.GridAggregator(overlap_mode='crop') of their int64 label maps [count,pd,ph,pw] into the output volume [D,H,W]: one
 * launch each.  table = int32 [P][9] on the device: patch origin z,y,x, then the crop window inside the patch, lo z,y,x and
 * hi z,y,x (exclusive); the launch handles patches first .. first+count-1.  (Overlapping crop windows are disjoint by
 * construction, so the paste needs no ordering.) */
int mi355seg_gather_patches_f32(const float* vol, int C, int D, int H, int W, const int* table, int first, int count,
                                int pd, int ph, int pw, float* out, void* stream);
int mi355seg_paste_labels_i64(const int64_t* labels, const int* table, int first, int count, int pd, int ph, int pw,
                              int64_t* out, int D, int H, int W, void* stream);

/* ------------------------------------------------------------------ Layout helpers */
int mi355seg_ncdhw_to_ndhwc_f32(const float* src, float* dst, int lddst, long long N, int C, long long S, void* stream);
int mi355seg_ndhwc_to_ncdhw_f32(const float* src, int ldsrc, float* dst, long long N, int C, long long S, void* stream);
/* strided channel-slice copy: dst[r, 0:C] = src[r, 0:C] */
int mi355seg_copy_rows_f32(const float* src, int ldsrc, float* dst, int lddst, long long rows, int C, void* stream);
/* dst[r, 0:C] += src[r, 0:C] */
int mi355seg_add_rows_f32(const float* src, int ldsrc, float* dst, int lddst, long long rows, int C, void* stream);
/* x.repeat(1, rep, 1, 1, 1) in channel-last form (vnet3d.py:55-56): y[r, j*C + c] = x[r, c], and its adjoint
 * dx[r, c] = sum_j dy[r, j*C + c]. */
int mi355seg_repeat_channels_f32(const float* x, int ldx, float* y, int ldy, long long rows, int C, int rep, void* stream);
int mi355seg_repeat_channels_bwd_f32(const float* dy, int lddy, float* dx, int lddx, long long rows, int C, int rep, void* stream);
/* Reverse-attention gate of RE_Net / ER_Net (RE_net.py:104-107,114-117,123-126):
 *   y[r, c] = enc[r, c] * (2 - sigmoid(t[r]))        (= (1 - sigmoid(t)).expand(C) * enc + enc; t has one channel)
 * backward: denc[r, c] = dy[r, c] * (2 - s_r),  dt[r] = -s_r (1 - s_r) * sum_c dy[r, c] enc[r, c].   C % 4 == 0. */
int mi355seg_gate_fwd_f32(const float* enc, int ldenc, const float* t, int ldt, float* y, int ldy, long long rows, int C, void* stream);
int mi355seg_gate_bwd_f32(const float* dy, int lddy, const float* enc, int ldenc, const float* t, int ldt,
                          float* denc, int lddenc, float* dt, long long rows, int C, void* stream);
/* Selective-fusion pieces of ER_Net's SFConv (ER_net.py:36-70); rows are grouped per sample (group g = rows [g*rows, (g+1)*rows)):
 *   group_sums:         out[g, c] (+)= alpha * sum_r x[g, r, c]           (fea_U.mean over the voxels; adjoint of the mix)
 *   mix_channels:       y[g, r, c] = x1[g, r, c] * a[g, c] + x2[g, r, c] * b[g, c]   ((feas * attention_vectors).sum(dim=1))
 *   broadcast_channels: y[g, r, c] = alpha * v[g, c]                      (adjoint of the spatial mean)
 * group_sums needs mi355seg_norm_ws_bytes(rows, 1, C) of workspace. */
int mi355seg_group_sums_f32(const float* x, int ldx, long long rows, int groups, int C, float alpha, float* out, int accumulate,
                            void* ws, size_t ws_bytes, void* stream);
int mi355seg_mix_channels_f32(const float* x1, int ld1, const float* a, const float* x2, int ld2, const float* b, float* y, int ldy,
                              long long rows, int groups, int C, void* stream);
int mi355seg_broadcast_channels_f32(const float* v, float alpha, float* y, int ldy, long long rows, int groups, int C, void* stream);
/* y[r, c] += bias[c] in place (bias of a ConvTranspose3d computed as the adjoint of a bias-free convolution). */
int mi355seg_add_bias_f32(float* y, int ldy, const float* bias, long long rows, int C, void* stream);
/* out[i] = a[i] * b[i] (attention-probability dropout mask, unetr.py:112; out may alias a). */
int mi355seg_mul_f32(const float* a, const float* b, float* out, long long n, void* stream);


/* ------------------------------------------------------------------ bf16 tensors (BASELINE cfg 3-5: V-Net / Residual U-Net / UNETR in bf16)
 * What torch autocast(bfloat16) does for the reference's modules (vnet3d.py:21-121, residual_unet3d.py:82-107,
 * unetr.py:8-51): activations and activation gradients are bf16 NDHWC tensors in HBM, parameters, parameter gradients,
 * norm statistics and the loss stay fp32, every convolution product is a bf16 MFMA with fp32 accumulation, every
 * elementwise / norm / pool kernel loads bf16, computes in fp32 registers and rounds once (RNE) on the store.
 * Each mi355seg_<op>_bf16 below has the contract of mi355seg_<op>_f32 above with the tensor pointers retyped; pitches
 * (ld*) count ELEMENTS.  Conv3d shapes without a native bf16 kernel run through the fp32 entry point on fp32 copies
 * made in the workspace (mi355seg_conv3d_ws_bytes_bf16 sizes for that). */
int mi355seg_norm_stats_bf16(const mi355seg_bf16* x, int ldx, long long rows, int groups, int C, float eps, float* mean, float* rstd, float* running_mean, float* running_var, float momentum, void* ws, size_t ws_bytes, void* stream);
int mi355seg_norm_act_fwd_ax_bf16(const mi355seg_bf16* x, int ldx, const float* mean, const float* rstd, const float* gamma, const float* beta, const mi355seg_bf16* res, int ldres, mi355seg_bf16* y, int ldy, long long rows, int groups, int C, int act, float slope, float* y_amax, void* stream);
int mi355seg_norm_act_bwd_colsum_ax_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, const float* mean, const float* rstd, const float* gamma, const float* beta, const mi355seg_bf16* res, int ldres, mi355seg_bf16* dx, int lddx, float* dgamma, float* dbeta, mi355seg_bf16* dres, int lddres, float* dx_colsum, float* dx_amax, long long rows, int groups, int C, int act, float slope, void* ws, size_t ws_bytes, void* stream);
int mi355seg_norm_act_bwd_apply_ax_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, const float* mean, const float* rstd, const float* gamma, const float* beta, const mi355seg_bf16* res, int ldres, const float* s1, const float* s2, mi355seg_bf16* dx, int lddx, mi355seg_bf16* dres, int lddres, float* dx_colsum, float* dx_amax, long long rows, int groups, int C, int act, float slope, void* ws, size_t ws_bytes, void* stream);
int mi355seg_norm_act_fwd_bf16(const mi355seg_bf16* x, int ldx, const float* mean, const float* rstd, const float* gamma, const float* beta, const mi355seg_bf16* res, int ldres, mi355seg_bf16* y, int ldy, long long rows, int groups, int C, int act, float slope, void* stream);
int mi355seg_norm_act_bwd_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, const float* mean, const float* rstd, const float* gamma, const float* beta, const mi355seg_bf16* res, int ldres, mi355seg_bf16* dx, int lddx, float* dgamma, float* dbeta, mi355seg_bf16* dres, int lddres, long long rows, int groups, int C, int act, float slope, void* ws, size_t ws_bytes, void* stream);
int mi355seg_norm_act_bwd_colsum_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, const float* mean, const float* rstd, const float* gamma, const float* beta, const mi355seg_bf16* res, int ldres, mi355seg_bf16* dx, int lddx, float* dgamma, float* dbeta, mi355seg_bf16* dres, int lddres, float* dx_colsum, long long rows, int groups, int C, int act, float slope, void* ws, size_t ws_bytes, void* stream);
int mi355seg_norm_act_bwd_sums_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, const float* mean, const float* rstd, const float* gamma, const float* beta, const mi355seg_bf16* res, int ldres, float* s1, float* s2, float* dgamma, float* dbeta, long long rows, int groups, int C, int act, float slope, void* ws, size_t ws_bytes, void* stream);
int mi355seg_norm_act_bwd_apply_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, const float* mean, const float* rstd, const float* gamma, const float* beta, const mi355seg_bf16* res, int ldres, const float* s1, const float* s2, mi355seg_bf16* dx, int lddx, mi355seg_bf16* dres, int lddres, float* dx_colsum, long long rows, int groups, int C, int act, float slope, void* ws, size_t ws_bytes, void* stream);
int mi355seg_scale_channels_bf16(const mi355seg_bf16* x, int ldx, const float* scale, mi355seg_bf16* y, int ldy, long long rows, int groups, int C, void* stream);
int mi355seg_act_fwd_bf16(const mi355seg_bf16* x, int ldx, const mi355seg_bf16* res, int ldres, mi355seg_bf16* y, int ldy, long long rows, int C, int act, float slope, void* stream);
int mi355seg_act_bwd_add_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, const mi355seg_bf16* res, int ldres, const mi355seg_bf16* addend, int ldadd, mi355seg_bf16* dx, int lddx, long long rows, int C, int act, float slope, void* stream);
int mi355seg_act_bwd_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, const mi355seg_bf16* res, int ldres, mi355seg_bf16* dx, int lddx, long long rows, int C, int act, float slope, void* stream);
int mi355seg_prelu_fwd_bf16(const mi355seg_bf16* x, int ldx, const mi355seg_bf16* res, int ldres, const float* slope, mi355seg_bf16* y, int ldy, long long rows, int C, void* stream);
int mi355seg_prelu_bwd_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, const mi355seg_bf16* res, int ldres, const float* slope, mi355seg_bf16* dx, int lddx, float* dslope, long long rows, int C, void* ws, size_t ws_bytes, void* stream);
int mi355seg_maxpool2_fwd_bf16(const mi355seg_bf16* x, int ldx, mi355seg_bf16* y, int ldy, uint8_t* idx, int N, int D, int H, int W, int C, void* stream);
int mi355seg_maxpool2_bwd_bf16(const mi355seg_bf16* dy, int lddy, const uint8_t* idx, mi355seg_bf16* dx, int lddx, int N, int D, int H, int W, int C, void* stream);
int mi355seg_maxpool2_bwd_add_bf16(const mi355seg_bf16* dy, int lddy, const uint8_t* idx, const mi355seg_bf16* add, int ldadd, mi355seg_bf16* dx, int lddx, int N, int D, int H, int W, int C, void* stream);
int mi355seg_upsample2_fwd_bf16(const mi355seg_bf16* x, int ldx, mi355seg_bf16* y, int ldy, int N, int D, int H, int W, int C, void* stream);
int mi355seg_upsample2_bwd_bf16(const mi355seg_bf16* dy, int lddy, mi355seg_bf16* dx, int lddx, int N, int D, int H, int W, int C, void* stream);
int mi355seg_convt3d_k2s2_fwd_ax_bf16(const mi355seg_bf16* x, int ldx, const float* w, const float* bias, mi355seg_bf16* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, float* y_amax, void* ws, size_t ws_bytes, void* stream);
int mi355seg_convt3d_k2s2_fwd_bf16(const mi355seg_bf16* x, int ldx, const float* w, const float* bias, mi355seg_bf16* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int mi355seg_convt3d_k2s2_dgrad_bf16(const mi355seg_bf16* dy, int lddy, const float* w, mi355seg_bf16* dx, int lddx, int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int mi355seg_convt3d_k2s2_wgrad_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, float* dw, float* db, int N, int D, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);
int mi355seg_ncdhw_f32_to_ndhwc_bf16(const float* src, mi355seg_bf16* dst, int lddst, long long N, int C, long long S, void* stream);
int mi355seg_ndhwc_bf16_to_ncdhw_f32(const mi355seg_bf16* src, int ldsrc, float* dst, long long N, int C, long long S, void* stream);
int mi355seg_copy_rows_bf16(const mi355seg_bf16* src, int ldsrc, mi355seg_bf16* dst, int lddst, long long rows, int C, void* stream);
int mi355seg_add_rows_bf16(const mi355seg_bf16* src, int ldsrc, mi355seg_bf16* dst, int lddst, long long rows, int C, void* stream);
int mi355seg_repeat_channels_bf16(const mi355seg_bf16* x, int ldx, mi355seg_bf16* y, int ldy, long long rows, int C, int rep, void* stream);
int mi355seg_repeat_channels_bwd_bf16(const mi355seg_bf16* dy, int lddy, mi355seg_bf16* dx, int lddx, long long rows, int C, int rep, void* stream);
int mi355seg_add_bias_bf16(mi355seg_bf16* y, int ldy, const float* bias, long long rows, int C, void* stream);
size_t mi355seg_conv3d_ws_bytes_bf16(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int mi355seg_conv3d_fwd_bf16(const mi355seg_bf16* x, int ldx, const float* w, const float* bias, mi355seg_bf16* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, double* stats_sum, double* stats_sq, void* ws, size_t ws_bytes, void* stream);
/* y = conv(x) + res on bf16 tensors, each of the two operations rounded to bf16 as the reference's `conv(...) + residual` under autocast
 * (/root/reference/models/three_d/residual_unet3d.py:121,140-168): the sum rides in the convolution's epilogue where the launch allows
 * (k3 / k5 s1 on the 16x16x32 tiles, whole-K), else the library adds it in place after the convolution.  res: y's geometry at pitch ldres. */
int mi355seg_conv3d_fwd_res_bf16(const mi355seg_bf16* x, int ldx, const float* w, const float* bias, const mi355seg_bf16* res, int ldres, mi355seg_bf16* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
int mi355seg_conv3d_dgrad_bf16(const mi355seg_bf16* dy, int lddy, const float* w, mi355seg_bf16* dx, int lddx, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
/* dx = conv3d_dgrad(dy) + res on bf16 tensors (r5): a convolution's input gradient together with the sum of a SECOND gradient of its input --
 * the residual blocks of /root/reference/models/three_d/vnet3d.py:61-104 (`down` / the concat feed the first k5 unit AND the block's final
 * sum) -- each of the two operations rounded to bf16 as autograd's own sum of the two gradients would be (bit-identical to
 * mi355seg_conv3d_dgrad_bf16 followed by the sum); in the input-gradient kernel's epilogue on the 16x16x32 tiles, in place after it elsewhere. */
int mi355seg_conv3d_dgrad_res_bf16(const mi355seg_bf16* dy, int lddy, const float* w, const mi355seg_bf16* res, int ldres, mi355seg_bf16* dx, int lddx, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
int mi355seg_conv3d_wgrad_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, float* dw, float* db, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int accumulate, void* ws, size_t ws_bytes, void* stream);
int mi355seg_cast_f32_to_bf16(const float* src, int ldsrc, mi355seg_bf16* dst, int lddst, long long rows, int C, void* stream);
int mi355seg_cast_bf16_to_f32(const mi355seg_bf16* src, int ldsrc, float* dst, int lddst, long long rows, int C, void* stream);

/* ---- every weight packing of a training step in ONE launch (r6; csrc/prepack.hip).  Every matrix-core convolution reads its fp32 master
 * weights (the reference's nn.Conv3d / nn.ConvTranspose3d parameters, /root/reference/models/three_d/unet3d.py:29-43,80-101) in a
 * packed form that its entry point builds right in front of the launch: 44 small launches per cfg-2 step.  A training step can have
 * them built all at once instead:
 *   record_begin ... one full step through the ordinary entry points ... record_end  -> plan id + arena size (once per model and shape);
 *   each later step: prepack_run(plan, arena, bytes, stream) at its top -- one memset, one launch measuring max |w| where a packing scales
 *   by it (f16x3) and ONE launch that forms every recorded packing in the caller's arena, all on `stream`; until prepack_done an entry
 *   point that finds its (weight pointer, layout) in the plan reads the arena copy and launches no packing of its own.
 * Anything not found packs as before; results are bit-identical either way.  The caller guarantees that the recorded weight
 * pointers are alive and that the weights do not change between prepack_run and their last use in the step.  (The first prepack_run
 * with an arena writes the job table into it with a blocking copy; called first inside a stream capture it leaves the plan inactive.) */
int mi355seg_prepack_record_begin(void);
int mi355seg_prepack_record_end(int* plan, size_t* arena_bytes);
int mi355seg_prepack_jobs(int plan);
int mi355seg_prepack_run(int plan, void* arena, size_t arena_bytes, void* stream);
int mi355seg_prepack_done(void* stream);
int mi355seg_prepack_active(void);      /* != 0 between a prepack_run that activated its plan and prepack_done */
int mi355seg_prepack_free(int plan);

#ifdef __cplusplus
}
#endif
#endif /* MI355SEG_H */
