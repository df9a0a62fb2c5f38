// internal.h -- cross-translation-unit declarations (not part of the C-ABI).
#pragma once
#include "common.h"

namespace seg {

// norm.hip
size_t colsum_ws_bytes(int C);
int channel_sums(const float* x, int ldx, long long rows, int C, double* sum, double* sq, float* fsum, int accumulate,
                 void* ws, size_t ws_bytes, hipStream_t st);
int channel_sums(const bf16* x, int ldx, long long rows, int C, double* sum, double* sq, float* fsum, int accumulate,
                 void* ws, size_t ws_bytes, hipStream_t st);

int finalize_channel_partials(const float* part, int nblk, int C, double* sum, double* sq, hipStream_t st);

// conv_small.hip -- direct kernels for Cin<=4 stems and Cout<=4 pointwise heads
bool stem_supported(int Cin, int Cout, int k, int stride, int pad, int ldy);
// conv_stem4_lowp.hip: Conv3d(4 -> 32 | 64, k3 p1) on bf16 tensors through the matrix cores ((tap, channel) is the GEMM axis)
bool stem4_lowp_supported(int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy);
size_t stem4_lowp_ws_bytes(int Cout);
int stem4_fwd_lowp(const bf16* x, const float* w, const float* bias, bf16* y, int ldy, int N, int D, int H, int W, int Cout,
                   void* ws, size_t ws_bytes, hipStream_t st);
int stem4_wgrad_lowp(const bf16* dy, int lddy, const bf16* x, float* dw, int N, int D, int H, int W, int Cout, int accumulate,
                     void* ws, size_t ws_bytes, hipStream_t st);
bool head_supported(int Cin, int Cout, int k, int stride, int pad, int ldx);
size_t small_ws_bytes(int Cin, int Cout, int k);
template <typename T> int stem_fwd(const T* x, int ldx, const float* w, const float* bias, T* y, int ldy, int N, int D, int H, int W, int Cin,
             int Cout, double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st, float* y_amax = nullptr, bool* amax_done = nullptr);
// conv_small.hip (r5): weight gradient (+ bias gradient) of the 1-channel stem with the norm backward's apply half as its prologue
bool stem_wgrad_bn_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int stem_wgrad_bn(const float* da, int ldda, const float* y, int ldy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  int act, float slope, const float* s1, const float* s2, const float* x, int ldx, float* dw, float* db,
                  int N, int D, int H, int W, int Cout, void* ws, size_t ws_bytes, hipStream_t st);
template <typename T> int stem_wgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout,
               int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
template <typename T> int head_fwd(const T* x, int ldx, const float* w, const float* bias, T* y, int ldy, int N, int D, int H, int W, int Cin,
             int Cout, hipStream_t st);
template <typename T> int head_dgrad(const T* dy, int lddy, const float* w, T* dx, int lddx, int N, int D, int H, int W, int Cin, int Cout,
               hipStream_t st);
bool tinypw_supported(int Cin, int Cout, int k, int stride, int pad);      // k1 with at most 4 channels on either side
template <typename T> int tinypw_fwd(const T* x, int ldx, const float* w, const float* bias, T* y, int ldy, long long nvox, int Cin, int Cout, hipStream_t st);
template <typename T> int tinypw_dgrad(const T* dy, int lddy, const float* w, T* dx, int lddx, long long nvox, int Cin, int Cout, hipStream_t st);
template <typename T> int tinypw_wgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, long long nvox, int Cin, int Cout, int accumulate, void* ws,
                                       size_t ws_bytes, hipStream_t st);
bool smallcin_wgrad_supported(int Cin, int Cout, int k);
bool smallcout_wgrad_supported(int Cin, int Cout, int k, int ldx);
size_t small_wgrad_ws_bytes(int Cin, int Cout, int k);
template <typename T> int smallcin_wgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout,
                   int k, int stride, int pad, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
template <typename T> int smallcout_wgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout,
                    int k, int stride, int pad, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
// conv_stem1k5_lowp.hip -- V-Net's one-channel k5 stem on the bf16 matrix cores (x-taps as the GEMM's narrow axis), bf16 tensors
bool stem1k5_lowp_supported(int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy);
size_t stem1k5_lowp_ws_bytes(int Cout);
int stem1k5_fwd_lowp(const bf16* x, const float* w, const float* bias, bf16* y, int ldy, int N, int D, int H, int W, int Cout,
                     void* ws, size_t ws_bytes, hipStream_t st);
int stem1k5_wgrad_lowp(const bf16* dy, int lddy, const bf16* x, float* dw, int N, int D, int H, int W, int Cout, int accumulate,
                       void* ws, size_t ws_bytes, hipStream_t st);
// conv_headpw_lowp.hip -- pointwise heads (k1, Cout = 2 | 4) for bf16 tensors: the matrix core contracts the channels
bool headpw_lowp_supported(int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy);
size_t headpw_lowp_ws_bytes(int Cin, int Cout);
int headpw_fwd_lowp(const bf16* x, int ldx, const float* w, const float* bias, bf16* y, int ldy, long long nvox, int Cin, int Cout, hipStream_t st);
int headpw_wgrad_lowp(const bf16* dy, int lddy, const bf16* x, int ldx, float* dw, long long nvox, int Cin, int Cout, int accumulate,
                      void* ws, size_t ws_bytes, hipStream_t st);
// conv_head2_lowp.hip -- the two-channel k5 head on the bf16 matrix cores ((dx, co) as the GEMM's narrow axis), bf16 tensors
bool head2_lowp_supported(int Cin, int Cout, int k, int stride, int pad, int ld_wide, int ld_narrow);
size_t head2_lowp_ws_bytes(int Cin);
int head2_fwd_lowp(const bf16* x, int ldx, const float* w, const float* bias, bf16* y, int ldy, int N, int D, int H, int W, int Cin,
                   void* ws, size_t ws_bytes, hipStream_t st);
int head2_dgrad_lowp(const bf16* dy, int lddy, const float* w, bf16* dx, int lddx, int N, int D, int H, int W, int Cin,
                     void* ws, size_t ws_bytes, hipStream_t st);
int head2_wgrad_lowp(const bf16* dy, int lddy, const bf16* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int accumulate,
                     void* ws, size_t ws_bytes, hipStream_t st);
// conv_headk.hip -- odd-kernel "same" convolutions with two output channels (V-Net head), z-marching VALU kernels
bool headk_supported(int Cin, int Cout, int k, int stride, int pad, int ldx_in, int ld_out, bool dgrad);
size_t headk_ws_bytes(int Cin, int Cout, int k);
int headk_conv(bool dgrad, const float* in, int ld_in, const float* w, const float* bias, float* out, int ld_out,
               int N, int D, int H, int W, int Cin, int k, void* ws, size_t ws_bytes, hipStream_t st);
bool stemk_supported(int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy);
int stemk_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
              void* ws, size_t ws_bytes, hipStream_t st);
bool headk_wgrad_supported(int Cin, int Cout, int k, int stride, int pad, int ldx, int lddy);
size_t headk_wgrad_ws_bytes(int N, int D, int H, int W, int Cin, int k);
int headk_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int k,
                int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
template <typename T> int head_wgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout,
               int accumulate, void* ws, size_t ws_bytes, hipStream_t st);

// conv_generic.hip
int f32_conv_policy();     // MATH_X3 when the bf16x6 conv math is selected, else MATH_F32
int x3_shape();            // 16 | 32: MFMA shape of the bf16x6 forward / dgrad kernels (mi355seg_set_x3_shape)
bool x3_f16();             // the split-precision kernels run the two-piece fp16 split (MI355SEG_MATH_F16X3) where they have that form
void pack_w_fwd(const float* w, float* wp, int Cout, int Cin, int T, hipStream_t st);
void pack_w_dgrad(const float* w, float* wd, int Cout, int Cin, int T, int flip, hipStream_t st);
void wgrad_reduce(const float* part, float* dw, int splits, int T, int Cin, int Cout, int accumulate, hipStream_t st);
void wgrad_reduce_swapped(const float* part, float* dw, int splits, int T, int Cin, int Cout, int accumulate, hipStream_t st);   // slabs [strip][T - 1 - tap][co][ci]

// conv_mfma.hip / igemm_kernel.h -- implicit-GEMM Conv3d / ConvTranspose3d on the matrix cores.  `math` is the arithmetic
// policy of igemm_kernel.h: MATH_F32 (0: fp32 tensors, fp32 MFMA), MATH_X3 (1: fp32 tensors, bf16x6 split) or MATH_B16
// (2: bf16 tensors, bf16 MFMA); x / y / dy / dx point at fp32 or bf16 NDHWC arrays accordingly, weights are always the
// fp32 masters in their PyTorch layout.
size_t conv_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);
bool conv_mfma_supported(int math, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy);
// dgrad != 0: `w` is still the (Cfwd_out=Cin_here ... ) torch weight of the FORWARD conv, i.e. shape (Cin, Cout, 27)
// seen from this call's Cin/Cout; taps are flipped while packing.
bool conv_gather_fwd_supported(int math, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy);
bool conv_gather_dgrad_supported(int math, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int lddy, int lddx);
size_t conv_gather_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int conv_gather_fwd_mfma(int math, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int N, int D, int H, int W,
                         int Cin, int Cout, int k, int stride, int pad, double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st);
int conv_gather_dgrad_mfma(int math, const void* dy, int lddy, const float* w, void* dx, int lddx, int N, int D, int H, int W,
                           int Cin, int Cout, int k, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t st);
// what the caller of an input-gradient launch hands over to get the BatchNorm-backward column sums of the layer in front out of the
// kernel's epilogue (done = 1 when the launch produced them; otherwise mi355seg_norm_act_bwd_sums_f32 has to)
struct BnBwdEpi {
    const float* x; int ldx; const float* mean; const float* rstd; const float* gamma; const float* beta; int act; float slope;
    float* s1; float* s2; float* dgamma; float* dbeta; int done;
};
// norm.hip: per-tile epilogue partials -> per-channel totals (two-stage when there are many tiles; tmp = part_reduce_ws_bytes(C) of scratch)
size_t part_reduce_ws_bytes(int C);
void norm_bwd_finalize(const float* part, int nblk, int C, float* s1, float* s2, float* dgamma, float* dbeta, double* tmp, hipStream_t st);
bool tile_stats_finalize2(const float* spart, int nM, int C, double* sum, double* sq, double* tmp, hipStream_t st);
// norm + activation prologue of a convolution's input (r5, f16x3 kernels): the operand is act(al[c] * x + be[c]) of the tensor handed
// over -- the folded BatchNorm of the layer in front (al = rstd gamma, be = beta - mean al) -- formed while the tiles are staged
struct ConvPro { const float* al; const float* be; int act; float slope; };
bool conv_pro_act_ok(int act);
int conv_fwd_mfma(int math, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int N, int D, int H, int W,
                  int Cin, int Cout, int k, int dgrad, double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st,
                  const float* oscale = nullptr, int act = 0, float slope = 0.f, BnBwdEpi* bne = nullptr,
                  const float* x_amax = nullptr, const float* w_amax = nullptr, const void* res = nullptr, int ldres = 0, int* res_fused = nullptr,
                  const ConvPro* pro = nullptr, float* y_amax = nullptr);
// (pro: fp32 tensors under f16x3 on the conv_x3s kernels only -- ask conv_fwd_takes_amax; x_amax must then bound the prologue's OUTPUT.
//  y_amax: max |y| max-combined into this zeroed device scalar: from the kernel's epilogue on whole-K conv_x3s launches, by a pass over y otherwise)
// (res: a tensor of y's geometry to add to the result; *res_fused = 1 when the launch took it into its epilogue -- bf16 16x16x32 tiles,
// whole-K, no statistics --, else 0 and the caller adds it)
// max |x| of a rows x C tensor at pitch ld (times |rowscale[row]| when given), max-combined into the zeroed device scalar *slot
void tensor_amax(const float* x, int ld, long long rows, int C, const float* rowscale, float* slot, hipStream_t st);
bool conv_fwd_takes_amax(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);
bool convt_mfma_supported(int math, int N, int D, int H, int W, int Cin, int Cout, int ldx, int ldy);
int convt_fwd_mfma(int math, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int N, int D, int H, int W,
                   int Cin, int Cout, void* ws, size_t ws_bytes, hipStream_t st);
int convt_dgrad_mfma(int math, const void* dy, int lddy, const float* w, void* dx, int lddx, int N, int D, int H, int W,
                     int Cin, int Cout, void* ws, size_t ws_bytes, hipStream_t st);
// conv_gwgrad.hip -- MFMA gather-wgrad for any cubic kernel / stride / padding
size_t gwgrad_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);
bool gwgrad_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int lddy);
template <typename T> int conv_gwgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout,
                int k, int stride, int pad, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
// conv_pw_wgrad.hip -- K = voxels GEMM wgrads (T = 1: Conv3d k1, T = 8: ConvTranspose3d k2 s2)
// convt_direct.hip: ConvTranspose3d k2 s2 forward (gather = false) / input gradient (gather = true) as one GEMM on the bf16 matrix
// cores (fp32 tensors: bf16x6 planes); coarse = the low-resolution tensor, fine = the 2x one
bool convt_direct_supported(int elem_bytes, bool gather, int N, int D, int H, int W, int Cin, int Cout, int ld_coarse, int ld_fine);
size_t convt_direct_ws_bytes(int Cin, int Cout);
size_t convt_direct_slab_bytes(long long nvox, int Cin, int Cout);      // split-K slabs of the GEMM-form input gradient (fp32, deep levels), 0 when unsplit
template <typename TT> int convt_direct(bool gather, const TT* x, int ldx, const float* w, const float* bias, TT* y, int ldy, int N, int D, int H, int W,
                                        int Cin, int Cout, void* ws, size_t ws_bytes, hipStream_t st, float* y_amax = nullptr);
// convt_wgrad_lowp.hip: ConvTranspose3d k2 s2 weight gradient on the bf16 matrix cores (fp32 tensors: bf16x6 planes)
void convt_wgrad_reduce(const float* part, float* dw, int splits, int Cin, int Cout, hipStream_t st);   // convt.hip: dw[ci][co][tap] = sum of the slabs
size_t convt_wgrad_lowp_ws_bytes(long long nvox, int Cin, int Cout);
bool convt_wgrad_lowp_supported(long long nvox, int Cin, int Cout, int ldx, int lddy, int elem_bytes);
template <typename IN_T> int convt_wgrad_lowp(const IN_T* dy, int lddy, const IN_T* x, int ldx, int N, int D, int H, int W, int Cin, int Cout,
                                              float** part_out, int* nstrips_out, void* ws, size_t ws_bytes, hipStream_t st);
size_t pw_wgrad_lowp_ws_bytes(long long nvox, int Cin, int Cout);                 // convt_wgrad_lowp.hip: k1 weight gradient, same operand path
bool pw_wgrad_lowp_supported(long long nvox, int Cin, int Cout, int ldx, int lddy, int elem_bytes);
template <typename IN_T> int pw_wgrad_lowp(const IN_T* dy, int lddy, const IN_T* x, int ldx, int N, int D, int H, int W, int Cin, int Cout,
                                           float** part_out, int* nstrips_out, void* ws, size_t ws_bytes, hipStream_t st);
size_t pw_wgrad_ws_bytes(long long nvox, int Cin, int Cout, int T);
bool pw_wgrad_supported(long long nvox, int Cin, int Cout, int T, int ldx, int lddy);
template <typename IN_T> int pw_wgrad_mfma(const IN_T* dy, int lddy, const IN_T* x, int ldx, int N, int D, int H, int W, int Cin, int Cout, int T,
                  float** part_out, int* nstrips_out, void* ws, size_t ws_bytes, hipStream_t st);
// conv_wgrad_lowp.hip -- k3 / k5 wgrad on the bf16 matrix cores (MATH_X3: fp32 tensors, bf16x6 split; MATH_B16: bf16 tensors)
bool wgrad_lowp_supported(int math, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int lddy);
size_t wgrad_lowp_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k);
size_t wgrad_lowp_ws_bytes_geom(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);
int conv_wgrad_lowp(int math, const void* dy, int lddy, const void* x, int ldx, float* dw, int N, int D, int H, int W, int Cin,
                    int Cout, int k, int stride, int accumulate, void* ws, size_t ws_bytes, hipStream_t st,
                    const float* x_amax = nullptr, const float* dy_amax = nullptr, const ConvPro* pro = nullptr);
size_t wgrad_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k);
bool wgrad_mfma_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int lddy);
int conv_wgrad_mfma(const float* dy, int lddy, const float* x, int ldx, float* dw, int N, int D, int H, int W, int Cin,
                    int Cout, int k, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace seg
