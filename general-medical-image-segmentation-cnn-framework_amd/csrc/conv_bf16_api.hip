// conv_bf16_api.hip -- Conv3d forward / input gradient / weight gradient on bf16 NDHWC tensors (the bf16 configurations:
// V-Net vnet3d.py:21-121, Residual U-Net residual_unet3d.py:22-107, UNETR decoder unetr.py:8-51).  Activations and
// activation gradients are bf16 in HBM, weights / bias / weight gradients stay fp32 masters (what torch autocast does for
// the reference), every product is a bf16 MFMA with fp32 accumulation.
//
// Native bf16 kernels: the implicit-GEMM kernel (k1 / k3 / k5 stride 1, gather mode for strided / even kernels), the
// transposing-read wgrad (k3 / k5), the K = voxels wgrads (k1, any-geometry gather wgrad: bf16 loads, fp32 MFMA) and the
// HBM-bound small-channel stems / heads.  Any other shape (V-Net's two-channel k5 head and one-channel k5 stem, tiny
// or ragged channel counts) runs through the fp32 entry point on fp32 copies made in the workspace: one extra read +
// write of tensors that are a few channels wide -- correct for every geometry, never the hot layers.
#include "common.h"
#include "internal.h"
#include "igemm_kernel.h"

namespace seg {

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_rows_kernel(const TS* __restrict__ src, int lds, TD* __restrict__ dst, int ldd, long long rows, int C) {
    const bool v = (C % 4 == 0) && (lds % 4 == 0) && (ldd % 4 == 0) && ((uintptr_t)src % (4 * sizeof(TS)) == 0) && ((uintptr_t)dst % (4 * sizeof(TD)) == 0);
    const int cw = v ? C / 4 : C;
    const long long total = rows * cw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cw;
        const int c = (int)(i - r * cw);
        if (v) st4(dst + r * ldd + c * 4, ld4(src + r * lds + c * 4));
        else st1(dst + r * ldd + c, ld1(src + r * lds + c));
    }
}

template <typename TS, typename TD>
static void cast_rows(const TS* src, int lds, TD* dst, int ldd, long long rows, int C, hipStream_t st) {
    long long b = (rows * C / 4 + 255) / 256;
    const int grid = (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
    hipLaunchKernelGGL((cast_rows_kernel<TS, TD>), dim3(grid), dim3(256), 0, st, src, lds, dst, ldd, rows, C);
}

static int oext(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

// which native bf16 kernel serves a shape (0 = none: fp32 fallback)
enum { NB_NONE = 0, NB_IGEMM, NB_GATHER, NB_STEM1K5, NB_PWL, NB_HEADPW, NB_K2S2W, NB_HEAD2, NB_STEM4, NB_STEM, NB_HEAD, NB_LOWP, NB_PW, NB_SMALLCIN, NB_SMALLCOUT, NB_GW, NB_TINY, NB_CONVT };
static int native_fwd(int N, int D, int H, int W, int Cin, int Cout, int k, int s, int p, int ldx, int ldy) {
    if (conv_mfma_supported(MATH_B16, N, D, H, W, Cin, Cout, k, s, p, ldx, ldy)) return NB_IGEMM;
    if (conv_gather_fwd_supported(MATH_B16, N, D, H, W, Cin, Cout, k, s, p, ldx, ldy)) return NB_GATHER;
    if (head2_lowp_supported(Cin, Cout, k, s, p, ldx, ldy)) return NB_HEAD2;
    if (stem1k5_lowp_supported(Cin, Cout, k, s, p, ldx, ldy)) return NB_STEM1K5;
    if (stem4_lowp_supported(Cin, Cout, k, s, p, ldx, ldy)) return NB_STEM4;
    if (stem_supported(Cin, Cout, k, s, p, ldy)) return NB_STEM;
    if (headpw_lowp_supported(Cin, Cout, k, s, p, ldx, ldy)) return NB_HEADPW;
    if (head_supported(Cin, Cout, k, s, p, ldx)) return NB_HEAD;
    if (tinypw_supported(Cin, Cout, k, s, p)) return NB_TINY;
    return NB_NONE;
}
static int native_dgrad(int N, int D, int H, int W, int Cin, int Cout, int k, int s, int p, int lddy, int lddx) {
    if (conv_mfma_supported(MATH_B16, N, D, H, W, Cout, Cin, k, s, p, lddy, lddx)) return NB_IGEMM;
    if (conv_gather_dgrad_supported(MATH_B16, N, D, H, W, Cin, Cout, k, s, p, lddy, lddx)) return NB_GATHER;
    if (head2_lowp_supported(Cin, Cout, k, s, p, lddx, lddy)) return NB_HEAD2;
    if (head_supported(Cin, Cout, k, s, p, lddx)) return NB_HEAD;
    if (tinypw_supported(Cin, Cout, k, s, p)) return NB_TINY;
    // k2 s2 p0: the input gradient is the forward of ConvTranspose3d k2 s2 with the same weight tensor (conv_generic.hip)
    if (k == 2 && s == 2 && p == 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && convt_mfma_supported(MATH_B16, N, D / 2, H / 2, W / 2, Cout, Cin, lddy, lddx))
        return NB_CONVT;
    return NB_NONE;
}
static int native_wgrad(int N, int D, int H, int W, int Cin, int Cout, int k, int s, int p, int ldx, int lddy) {
    if (wgrad_lowp_supported(MATH_B16, N, D, H, W, Cin, Cout, k, s, p, ldx, lddy)) return NB_LOWP;
    // V-Net's two-channel k5 head: the fp32 z-marching kernel (conv_headk.hip) behind the cast fall-back is 8x faster than the
    // generic small-channel wgrad below (the one-channel k5 stem has its own LDS-tiled kernel inside smallcin_wgrad)
    if (head2_lowp_supported(Cin, Cout, k, s, p, ldx, lddy)) return NB_HEAD2;
    if (stem1k5_lowp_supported(Cin, Cout, k, s, p, ldx, lddy)) return NB_STEM1K5;
    // k2 s2 down-convolution (V-Net): its weight gradient IS a ConvTranspose k2 s2 weight gradient with the roles swapped
    // (base voxels = the coarse dy, children = the fine x), and (Cout, Cin, 2, 2, 2) is that kernel's output layout
    if (k == 2 && s == 2 && p == 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0 &&
        convt_wgrad_lowp_supported((long long)N * (D / 2) * (H / 2) * (W / 2), Cout, Cin, lddy, ldx, 2)) return NB_K2S2W;
    if (headk_wgrad_supported(Cin, Cout, k, s, p, Cin, Cout)) return NB_NONE;
    if (tinypw_supported(Cin, Cout, k, s, p)) return NB_TINY;
    if (k == 1 && s == 1 && p == 0 && pw_wgrad_lowp_supported((long long)N * D * H * W, Cin, Cout, ldx, lddy, 2)) return NB_PWL;
    if (k == 1 && s == 1 && p == 0 && pw_wgrad_supported((long long)N * D * H * W, Cin, Cout, 1, ldx, lddy)) return NB_PW;
    if (stem4_lowp_supported(Cin, Cout, k, s, p, ldx, lddy)) return NB_STEM4;
    if (stem_supported(Cin, Cout, k, s, p, lddy)) return NB_STEM;
    if (headpw_lowp_supported(Cin, Cout, k, s, p, ldx, lddy)) return NB_HEADPW;
    if (head_supported(Cin, Cout, k, s, p, ldx)) return NB_HEAD;
    if (smallcin_wgrad_supported(Cin, Cout, k)) return NB_SMALLCIN;
    if (smallcout_wgrad_supported(Cin, Cout, k, ldx)) return NB_SMALLCOUT;
    if (gwgrad_supported(N, D, H, W, Cin, Cout, k, s, p, ldx, lddy)) return NB_GW;
    return NB_NONE;
}

}  // namespace seg

using namespace seg;

extern "C" {

int mi355seg_set_b16_tiles(int mode) {
    SEG_CHECK_ARG(mode >= 0 && mode <= 2, "set_b16_tiles: 0 (auto), 1 (16x16x32 tiles wherever possible) or 2 (generic tiles), got %d", mode);
    set_b16_tiles(mode);
    return MI355SEG_OK;
}
int mi355seg_get_b16_tiles(void) { return get_b16_tiles(); }

int mi355seg_set_wgrad_wide(int mode) {
    SEG_CHECK_ARG(mode >= 0 && mode <= 2, "set_wgrad_wide: 0 (never), 1 (where it pays) or 2 (wherever the geometry allows), got %d", mode);
    set_wgrad_wide(mode);
    return MI355SEG_OK;
}
int mi355seg_get_wgrad_wide(void) { return get_wgrad_wide(); }

// workspace of the three bf16 Conv3d entry points for one layer geometry (contiguous tensors assumed for the fallback test)
size_t mi355seg_conv3d_ws_bytes_bf16(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    size_t base = mi355seg_conv3d_ws_bytes(N, D, H, W, Cin, Cout, k, stride, pad);
    if (D + 2 * pad < k || H + 2 * pad < k || W + 2 * pad < k) return base;
    const int Do = oext(D, k, stride, pad), Ho = oext(H, k, stride, pad), Wo = oext(W, k, stride, pad);
    const bool fb = !native_fwd(N, D, H, W, Cin, Cout, k, stride, pad, Cin, Cout) || !native_dgrad(N, D, H, W, Cin, Cout, k, stride, pad, Cout, Cin) ||
                    !native_wgrad(N, D, H, W, Cin, Cout, k, stride, pad, Cin, Cout);
    if (stem4_lowp_supported(Cin, Cout, k, stride, pad, Cin, Cout) && base < stem4_lowp_ws_bytes(Cout)) base = stem4_lowp_ws_bytes(Cout);
    if (head2_lowp_supported(Cin, Cout, k, stride, pad, Cin, Cout) && base < head2_lowp_ws_bytes(Cin)) base = head2_lowp_ws_bytes(Cin);
    if (headpw_lowp_supported(Cin, Cout, k, stride, pad, Cin, Cout) && base < headpw_lowp_ws_bytes(Cin, Cout)) base = headpw_lowp_ws_bytes(Cin, Cout);
    if (stem1k5_lowp_supported(Cin, Cout, k, stride, pad, Cin, Cout) && base < stem1k5_lowp_ws_bytes(Cout)) base = stem1k5_lowp_ws_bytes(Cout);
    if (k == 1 && stride == 1 && pad == 0 && base < pw_wgrad_lowp_ws_bytes((long long)N * D * H * W, Cin, Cout)) base = pw_wgrad_lowp_ws_bytes((long long)N * D * H * W, Cin, Cout);
    if (k == 2 && stride == 2 && pad == 0 && base < convt_wgrad_lowp_ws_bytes((long long)N * (D / 2) * (H / 2) * (W / 2), Cout, Cin))
        base = convt_wgrad_lowp_ws_bytes((long long)N * (D / 2) * (H / 2) * (W / 2), Cout, Cin);
    // the fp32 staging copies of the fall-back: for shapes without a native kernel, and -- while they stay under 512 MB -- for
    // every shape, because an entry point also falls back when a POINTER is not 16-byte aligned (a bf16 channel slice at an
    // 8-byte offset) or a k2 s2 weight gradient is asked not to accumulate, which this query cannot see
    const size_t stage = align_up((size_t)N * D * H * W * Cin * 4, 256) + align_up((size_t)N * Do * Ho * Wo * Cout * 4, 256) + 512;
    if (fb || stage <= ((size_t)512 << 20)) base += stage;
    return base;
}

int mi355seg_conv3d_fused_supported_bf16(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    return native_fwd(N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy) == NB_IGEMM;
}
int mi355seg_conv3d_fwd_fused_bf16(const mi355seg_bf16* x, int ldx, const float* w, const float* oscale, const float* oshift, int act, float slope,
                                   mi355seg_bf16* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                   void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(x && w && y && oscale && oshift && N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ldx >= Cin && ldy >= Cout, "conv3d_fwd_fused_bf16: bad arguments");
    SEG_CHECK_ARG(mi355seg_conv3d_fused_supported_bf16(N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy) && ((uintptr_t)x % 16) == 0,
                  "conv3d_fwd_fused_bf16: no fused form for this shape / alignment (ask mi355seg_conv3d_fused_supported_bf16)");
    return conv_fwd_mfma(MATH_B16, x, ldx, w, oshift, y, ldy, N, D, H, W, Cin, Cout, k, /*dgrad=*/0, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream, oscale, act, slope);
}

// y = conv(x) + res as the reference's two bf16 operations give it (each rounded to bf16): the sum rides in the convolution's epilogue
// where the launch allows (conv_b16s tiles, whole-K), else the activation kernel adds it in place
int mi355seg_conv3d_fwd_res_bf16(const mi355seg_bf16* x, int ldx, const float* w, const float* bias, const mi355seg_bf16* res, int ldres,
                                 mi355seg_bf16* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                 void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(res && ldres >= Cout, "conv3d_fwd_res_bf16: null residual or pitch < channels");
    hipStream_t st = (hipStream_t)stream;
    const int Do = oext(D, k, stride, pad), Ho = oext(H, k, stride, pad), Wo = oext(W, k, stride, pad);
    if (x && y && native_fwd(N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy) == NB_IGEMM && ((uintptr_t)x % 16) == 0) {
        int fused = 0;
        int rc = conv_fwd_mfma(MATH_B16, x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, /*dgrad=*/0, nullptr, nullptr, ws, ws_bytes, st,
                               nullptr, 0, 0.f, nullptr, nullptr, nullptr, res, ldres, &fused);
        if (rc || fused) return rc;
    } else {
        int rc = mi355seg_conv3d_fwd_bf16(x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, stride, pad, nullptr, nullptr, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    return mi355seg_act_fwd_bf16(y, ldy, res, ldres, y, ldy, (long long)N * Do * Ho * Wo, Cout, 0, 0.f, stream);
}

int mi355seg_conv3d_fwd_bf16(const mi355seg_bf16* x, int ldx, const float* w, const float* bias, mi355seg_bf16* y, int ldy,
                             int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                             double* stats_sum, double* stats_sq, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(x && w && y && N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && ldx >= Cin && ldy >= Cout, "conv3d_fwd_bf16: bad arguments");
    SEG_CHECK_ARG(k >= 1 && k <= 16 && stride >= 1 && pad >= 0 && D + 2 * pad >= k && H + 2 * pad >= k && W + 2 * pad >= k,
                  "conv3d_fwd_bf16: bad k/stride/pad %d/%d/%d", k, stride, pad);
    SEG_CHECK_ARG((stats_sum == nullptr) == (stats_sq == nullptr), "conv3d_fwd_bf16: stats_sum/stats_sq must come together");
    hipStream_t st = (hipStream_t)stream;
    const int Do = oext(D, k, stride, pad), Ho = oext(H, k, stride, pad), Wo = oext(W, k, stride, pad);
    const long long vout = (long long)N * Do * Ho * Wo;
    const bool al = ((uintptr_t)x % 16) == 0;
    const int nb = native_fwd(N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy);
    if (nb == NB_IGEMM && al)
        return conv_fwd_mfma(MATH_B16, x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, /*dgrad=*/0, stats_sum, stats_sq, ws, ws_bytes, st);
    if (nb == NB_GATHER && al)
        return conv_gather_fwd_mfma(MATH_B16, x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, stride, pad, stats_sum, stats_sq, ws, ws_bytes, st);
    if (nb == NB_HEAD2 && al && ((uintptr_t)y % 4) == 0) {
        int rc = head2_fwd_lowp(x, ldx, w, bias, y, ldy, N, D, H, W, Cin, ws, ws_bytes, st);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, vout, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    if (nb == NB_STEM1K5 && ((uintptr_t)y % 8) == 0) {
        int rc = stem1k5_fwd_lowp(x, w, bias, y, ldy, N, D, H, W, Cout, ws, ws_bytes, st);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, vout, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    if (nb == NB_STEM4 && ((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 8) == 0) {
        int rc = stem4_fwd_lowp(x, w, bias, y, ldy, N, D, H, W, Cout, ws, ws_bytes, st);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, vout, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    if ((nb == NB_STEM || nb == NB_STEM4) && ((uintptr_t)y % 8) == 0)
        return stem_fwd(x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, stats_sum, stats_sq, ws, ws_bytes, st);
    if (nb == NB_HEADPW && al && ((uintptr_t)y % (2 * Cout)) == 0) {
        int rc = headpw_fwd_lowp(x, ldx, w, bias, y, ldy, vout, Cin, Cout, st);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, vout, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    if ((nb == NB_HEAD || nb == NB_HEADPW) && ((uintptr_t)x % 8) == 0) {
        int rc = head_fwd(x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, st);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, vout, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    if (nb == NB_TINY) {
        int rc = tinypw_fwd(x, ldx, w, bias, y, ldy, vout, Cin, Cout, st);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, vout, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    // fp32 fallback
    Carver cv(ws);
    float* xf = cv.take<float>((size_t)N * D * H * W * Cin);
    float* yf = cv.take<float>((size_t)vout * Cout);
    const size_t used = cv.used();
    SEG_CHECK_WS(used + mi355seg_conv3d_ws_bytes(N, D, H, W, Cin, Cout, k, stride, pad), ws_bytes);
    cast_rows(x, ldx, xf, Cin, (long long)N * D * H * W, Cin, st);
    SEG_CHECK_LAUNCH();
    int rc = mi355seg_conv3d_fwd_f32(xf, Cin, w, bias, yf, Cout, N, D, H, W, Cin, Cout, k, stride, pad, stats_sum, stats_sq, (char*)ws + used, ws_bytes - used, stream);
    if (rc) return rc;
    cast_rows(yf, Cout, y, ldy, vout, Cout, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_conv3d_dgrad_bf16(const mi355seg_bf16* dy, int lddy, const float* w, mi355seg_bf16* dx, int lddx,
                               int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                               void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(dy && w && dx && N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && lddy >= Cout && lddx >= Cin, "conv3d_dgrad_bf16: bad arguments");
    SEG_CHECK_ARG(k >= 1 && k <= 16 && stride >= 1 && pad >= 0 && D + 2 * pad >= k && H + 2 * pad >= k && W + 2 * pad >= k,
                  "conv3d_dgrad_bf16: bad k/stride/pad %d/%d/%d", k, stride, pad);
    hipStream_t st = (hipStream_t)stream;
    const int Do = oext(D, k, stride, pad), Ho = oext(H, k, stride, pad), Wo = oext(W, k, stride, pad);
    const long long vout = (long long)N * Do * Ho * Wo, vin = (long long)N * D * H * W;
    const bool al = ((uintptr_t)dy % 16) == 0;
    const int nb = native_dgrad(N, D, H, W, Cin, Cout, k, stride, pad, lddy, lddx);
    if (nb == NB_IGEMM && al)
        return conv_fwd_mfma(MATH_B16, dy, lddy, w, nullptr, dx, lddx, N, D, H, W, Cout, Cin, k, /*dgrad=*/1, nullptr, nullptr, ws, ws_bytes, st);
    if (nb == NB_GATHER && al)
        return conv_gather_dgrad_mfma(MATH_B16, dy, lddy, w, dx, lddx, N, D, H, W, Cin, Cout, k, stride, pad, ws, ws_bytes, st);
    if (nb == NB_HEAD2 && ((uintptr_t)dy % 4) == 0 && ((uintptr_t)dx % 8) == 0)
        return head2_dgrad_lowp(dy, lddy, w, dx, lddx, N, D, H, W, Cin, ws, ws_bytes, st);
    if (nb == NB_HEAD && ((uintptr_t)dx % 8) == 0)
        return head_dgrad(dy, lddy, w, dx, lddx, N, D, H, W, Cin, Cout, st);
    if (nb == NB_TINY) return tinypw_dgrad(dy, lddy, w, dx, lddx, vin, Cin, Cout, st);
    if (nb == NB_CONVT && al) return convt_fwd_mfma(MATH_B16, dy, lddy, w, nullptr, dx, lddx, N, D / 2, H / 2, W / 2, Cout, Cin, ws, ws_bytes, st);
    Carver cv(ws);
    float* dxf = cv.take<float>((size_t)vin * Cin);
    float* dyf = cv.take<float>((size_t)vout * Cout);
    const size_t used = cv.used();
    SEG_CHECK_WS(used + mi355seg_conv3d_ws_bytes(N, D, H, W, Cin, Cout, k, stride, pad), ws_bytes);
    cast_rows(dy, lddy, dyf, Cout, vout, Cout, st);
    SEG_CHECK_LAUNCH();
    int rc = mi355seg_conv3d_dgrad_f32(dyf, Cout, w, dxf, Cin, N, D, H, W, Cin, Cout, k, stride, pad, (char*)ws + used, ws_bytes - used, stream);
    if (rc) return rc;
    cast_rows(dxf, Cin, dx, lddx, vin, Cin, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// dx = conv3d_dgrad(dy) + res on bf16 tensors, each of the two operations rounded to bf16 (what autograd's sum of the two gradients of a
// forked tensor computes): the sum rides in the input-gradient kernel's epilogue on the k3 / k5 stride-1 16x16x32 tiles (whole-K launches),
// else the library adds it in place after the input gradient.  res: dx's geometry at pitch ldres.
int mi355seg_conv3d_dgrad_res_bf16(const mi355seg_bf16* dy, int lddy, const float* w, const mi355seg_bf16* res, int ldres, mi355seg_bf16* dx, int lddx,
                                   int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                   void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(dy && w && dx && res && N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && lddy >= Cout && lddx >= Cin && ldres >= Cin,
                  "conv3d_dgrad_res_bf16: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1 && 2 * pad == k - 1 && ((uintptr_t)dy % 16) == 0 && native_dgrad(N, D, H, W, Cin, Cout, k, stride, pad, lddy, lddx) == NB_IGEMM) {
        int fused = 0;
        int rc = conv_fwd_mfma(MATH_B16, dy, lddy, w, nullptr, dx, lddx, N, D, H, W, Cout, Cin, k, /*dgrad=*/1, nullptr, nullptr, ws, ws_bytes, st,
                               nullptr, 0, 0.f, nullptr, nullptr, nullptr, res, ldres, &fused);
        if (rc || fused) return rc;
    } else {
        int rc = mi355seg_conv3d_dgrad_bf16(dy, lddy, w, dx, lddx, N, D, H, W, Cin, Cout, k, stride, pad, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    return mi355seg_act_fwd_bf16(dx, lddx, res, ldres, dx, lddx, (long long)N * D * H * W, Cin, 0, 0.f, stream);
}

int mi355seg_conv3d_wgrad_bf16(const mi355seg_bf16* dy, int lddy, const mi355seg_bf16* x, int ldx, float* dw, float* db,
                               int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int accumulate,
                               void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(dy && x && dw && N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && lddy >= Cout && ldx >= Cin, "conv3d_wgrad_bf16: bad arguments");
    SEG_CHECK_ARG(k >= 1 && k <= 16 && stride >= 1 && pad >= 0 && D + 2 * pad >= k && H + 2 * pad >= k && W + 2 * pad >= k,
                  "conv3d_wgrad_bf16: bad k/stride/pad %d/%d/%d", k, stride, pad);
    hipStream_t st = (hipStream_t)stream;
    const int Do = oext(D, k, stride, pad), Ho = oext(H, k, stride, pad), Wo = oext(W, k, stride, pad);
    const long long vout = (long long)N * Do * Ho * Wo, vin = (long long)N * D * H * W;
    if (db) {
        int rc = channel_sums(dy, lddy, vout, Cout, nullptr, nullptr, db, accumulate, ws, ws_bytes, st);
        if (rc) return rc;
    }
    const bool al16 = ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, al8 = ((uintptr_t)x % 8) == 0 && ((uintptr_t)dy % 8) == 0;
    const int nb = native_wgrad(N, D, H, W, Cin, Cout, k, stride, pad, ldx, lddy);
    if (nb == NB_K2S2W && al16 && !accumulate) {
        float* part; int nstrips;
        int rc = convt_wgrad_lowp(x, ldx, dy, lddy, N, D / 2, H / 2, W / 2, Cout, Cin, &part, &nstrips, ws, ws_bytes, st);
        if (rc) return rc;
        convt_wgrad_reduce(part, dw, nstrips, Cout, Cin, st);
        SEG_CHECK_LAUNCH();
        return MI355SEG_OK;
    }
    if (nb == NB_STEM1K5 && ((uintptr_t)dy % 16) == 0)
        return stem1k5_wgrad_lowp(dy, lddy, x, dw, N, D, H, W, Cout, accumulate, ws, ws_bytes, st);
    if (nb == NB_HEAD2 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 4) == 0)
        return head2_wgrad_lowp(dy, lddy, x, ldx, dw, N, D, H, W, Cin, accumulate, ws, ws_bytes, st);
    if (nb == NB_LOWP && al16) return conv_wgrad_lowp(MATH_B16, dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, k, stride, accumulate, ws, ws_bytes, st);
    if (nb == NB_PWL && al16) {
        float* part; int nstrips;
        int rc = pw_wgrad_lowp(dy, lddy, x, ldx, N, D, H, W, Cin, Cout, &part, &nstrips, ws, ws_bytes, st);
        if (rc) return rc;
        wgrad_reduce(part, dw, nstrips, 1, Cin, Cout, accumulate, st);
        SEG_CHECK_LAUNCH();
        return MI355SEG_OK;
    }
    if ((nb == NB_PW || (nb == NB_PWL && pw_wgrad_supported((long long)N * D * H * W, Cin, Cout, 1, ldx, lddy))) && al8) {
        float* part; int nstrips;
        int rc = pw_wgrad_mfma(dy, lddy, x, ldx, N, D, H, W, Cin, Cout, 1, &part, &nstrips, ws, ws_bytes, st);
        if (rc) return rc;
        wgrad_reduce(part, dw, nstrips, 1, Cin, Cout, accumulate, st);
        SEG_CHECK_LAUNCH();
        return MI355SEG_OK;
    }
    if (nb == NB_TINY) return tinypw_wgrad(dy, lddy, x, ldx, dw, vin, Cin, Cout, accumulate, ws, ws_bytes, st);
    if (nb == NB_STEM4 && al16) return stem4_wgrad_lowp(dy, lddy, x, dw, N, D, H, W, Cout, accumulate, ws, ws_bytes, st);
    if ((nb == NB_STEM || nb == NB_STEM4) && al8) return stem_wgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, accumulate, ws, ws_bytes, st);
    if (nb == NB_HEADPW && al16) return headpw_wgrad_lowp(dy, lddy, x, ldx, dw, vin, Cin, Cout, accumulate, ws, ws_bytes, st);
    if ((nb == NB_HEAD || nb == NB_HEADPW) && al8) return head_wgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, accumulate, ws, ws_bytes, st);
    if (nb == NB_SMALLCIN && al8) return smallcin_wgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, k, stride, pad, accumulate, ws, ws_bytes, st);
    if (nb == NB_SMALLCOUT && al8) return smallcout_wgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, k, stride, pad, accumulate, ws, ws_bytes, st);
    if (nb == NB_GW && al8) return conv_gwgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, k, stride, pad, accumulate, ws, ws_bytes, st);
    Carver cv(ws);
    float* xf = cv.take<float>((size_t)vin * Cin);
    float* dyf = cv.take<float>((size_t)vout * Cout);
    const size_t used = cv.used();
    SEG_CHECK_WS(used + mi355seg_conv3d_ws_bytes(N, D, H, W, Cin, Cout, k, stride, pad), ws_bytes);
    cast_rows(x, ldx, xf, Cin, vin, Cin, st);
    cast_rows(dy, lddy, dyf, Cout, vout, Cout, st);
    SEG_CHECK_LAUNCH();
    return mi355seg_conv3d_wgrad_f32(dyf, Cout, xf, Cin, dw, nullptr, N, D, H, W, Cin, Cout, k, stride, pad, accumulate, (char*)ws + used, ws_bytes - used, stream);
}

// dtype casts of [rows, C] matrices with row pitches (autocast boundaries: fp32 <-> bf16 activations)
int mi355seg_cast_f32_to_bf16(const float* src, int ldsrc, mi355seg_bf16* dst, int lddst, long long rows, int C, void* stream) {
    SEG_CHECK_ARG(src && dst && rows > 0 && C > 0 && ldsrc >= C && lddst >= C, "cast_f32_to_bf16: bad arguments");
    cast_rows(src, ldsrc, dst, lddst, rows, C, (hipStream_t)stream);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_cast_bf16_to_f32(const mi355seg_bf16* src, int ldsrc, float* dst, int lddst, long long rows, int C, void* stream) {
    SEG_CHECK_ARG(src && dst && rows > 0 && C > 0 && ldsrc >= C && lddst >= C, "cast_bf16_to_f32: bad arguments");
    cast_rows(src, ldsrc, dst, lddst, rows, C, (hipStream_t)stream);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // extern "C"
