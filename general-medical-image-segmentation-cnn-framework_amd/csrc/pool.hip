// pool.hip -- MaxPool3d(2,2) with 3-bit argmax codes and nearest-neighbour x2 upsampling,
// forward and backward, NDHWC fp32.  Pure streaming kernels (HBM-bound), 16 B per lane.
// Reference: nn.MaxPool3d(kernel_size=2, stride=2) unet3d.py:19-25;
//            nn.Upsample(scale_factor=2, mode='nearest') residual_unet3d.py:19,103.
#include "common.h"

namespace seg {

template <bool VEC>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
        uint8_t* __restrict__ idx, int N, int D, int H, int W, int C) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2;
    const int cw = VEC ? C / 4 : C;
    const long long total = (long long)N * Do * Ho * Wo * cw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cc = (int)(i % cw); long long v = i / cw;
        int ow = (int)(v % Wo); long long r = v / Wo;
        int oh = (int)(r % Ho); r /= Ho;
        int od = (int)(r % Do); int n = (int)(r / Do);
        const int c = VEC ? cc * 4 : cc;
        constexpr int NJ = VEC ? 4 : 1;
        float best[NJ]; int code[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { best[j] = -INFINITY; code[j] = 0; }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            int dz = t >> 2, dy = (t >> 1) & 1, dx = t & 1;
            const float* p = x + ((((long long)n * D + 2 * od + dz) * H + 2 * oh + dy) * W + 2 * ow + dx) * ldx + c;
            float vals[NJ];
            if (VEC) { float4 q = *reinterpret_cast<const float4*>(p); vals[0] = q.x; if (NJ > 1) { vals[1] = q.y; vals[2] = q.z; vals[3] = q.w; } }
            else vals[0] = *p;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // PyTorch: (val > max) || isnan(val); the first maximum in scan order wins
                if (t == 0 || vals[j] > best[j] || vals[j] != vals[j]) { best[j] = vals[j]; code[j] = t; }
            }
        }
        if (VEC) {
            *reinterpret_cast<float4*>(y + v * ldy + c) = make_float4(best[0], best[NJ > 1 ? 1 : 0], best[NJ > 1 ? 2 : 0], best[NJ > 1 ? 3 : 0]);
            uchar4 cd = make_uchar4((uint8_t)code[0], (uint8_t)code[NJ > 1 ? 1 : 0], (uint8_t)code[NJ > 1 ? 2 : 0], (uint8_t)code[NJ > 1 ? 3 : 0]);
            *reinterpret_cast<uchar4*>(idx + v * C + c) = cd;
        } else {
            y[v * ldy + c] = best[0];
            idx[v * C + c] = (uint8_t)code[0];
        }
    }
}

// one thread per INPUT element: gradient flows to the window's arg-max position only
template <bool VEC>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dy, int lddy, const uint8_t* __restrict__ idx,
        float* __restrict__ dx, int lddx, int N, int D, int H, int W, int C, const float* __restrict__ add, int ldadd) {
    const int Do = D / 2, Ho = H / 2, Wo = W / 2;
    const int cw = VEC ? C / 4 : C;
    const long long total = (long long)N * D * H * W * cw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cc = (int)(i % cw); long long v = i / cw;
        int iw = (int)(v % W); long long r = v / W;
        int ih = (int)(r % H); r /= H;
        int id = (int)(r % D); int n = (int)(r / D);
        const int c = VEC ? cc * 4 : cc;
        int od = id >> 1, oh = ih >> 1, ow = iw >> 1;
        bool inside = od < Do && oh < Ho && ow < Wo;
        int me = ((id & 1) << 2) | ((ih & 1) << 1) | (iw & 1);
        long long ov = (((long long)n * Do + od) * Ho + oh) * Wo + ow;
        if (VEC) {
            float4 o = make_float4(0, 0, 0, 0);
            if (inside) {
                float4 g = *reinterpret_cast<const float4*>(dy + ov * lddy + c);
                uchar4 cd = *reinterpret_cast<const uchar4*>(idx + ov * C + c);
                o.x = cd.x == me ? g.x : 0.f; o.y = cd.y == me ? g.y : 0.f;
                o.z = cd.z == me ? g.z : 0.f; o.w = cd.w == me ? g.w : 0.f;
            }
            if (add) { float4 q = *reinterpret_cast<const float4*>(add + v * ldadd + c); o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
            *reinterpret_cast<float4*>(dx + v * lddx + c) = o;
        } else {
            float o = 0.f;
            if (inside && idx[ov * C + c] == me) o = dy[ov * lddy + c];
            if (add) o += add[v * ldadd + c];
            dx[v * lddx + c] = o;
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
        int N, int D, int H, int W, int C) {
    const int cw = VEC ? C / 4 : C;
    const int D2 = 2 * D, H2 = 2 * H, W2 = 2 * W;
    const long long total = (long long)N * D2 * H2 * W2 * cw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cc = (int)(i % cw); long long v = i / cw;
        int ow = (int)(v % W2); long long r = v / W2;
        int oh = (int)(r % H2); r /= H2;
        int od = (int)(r % D2); int n = (int)(r / D2);
        long long iv = (((long long)n * D + (od >> 1)) * H + (oh >> 1)) * W + (ow >> 1);
        if (VEC) *reinterpret_cast<float4*>(y + v * ldy + cc * 4) = *reinterpret_cast<const float4*>(x + iv * ldx + cc * 4);
        else y[v * ldy + cc] = x[iv * ldx + cc];
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
        int N, int D, int H, int W, int C) {
    const int cw = VEC ? C / 4 : C;
    const int H2 = 2 * H, W2 = 2 * W, D2 = 2 * D;
    const long long total = (long long)N * D * H * W * cw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cc = (int)(i % cw); long long v = i / cw;
        int iw = (int)(v % W); long long r = v / W;
        int ih = (int)(r % H); r /= H;
        int id = (int)(r % D); int n = (int)(r / D);
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            int dz = t >> 2, dyy = (t >> 1) & 1, dxx = t & 1;
            const float* p = dy + ((((long long)n * D2 + 2 * id + dz) * H2 + 2 * ih + dyy) * W2 + 2 * iw + dxx) * lddy + (VEC ? cc * 4 : cc);
            if (VEC) { float4 q = *reinterpret_cast<const float4*>(p); acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
            else acc.x += *p;
        }
        if (VEC) *reinterpret_cast<float4*>(dx + v * lddx + cc * 4) = acc;
        else dx[v * lddx + cc] = acc.x;
    }
}

static int pgrid(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
}  // namespace seg

using namespace seg;

#define POOL_DISPATCH(KERN, vec, total, ...)                                                                   \
    do {                                                                                                       \
        if (vec) hipLaunchKernelGGL((KERN<true>), dim3(pgrid(total)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERN<false>), dim3(pgrid(total)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);    \
        SEG_CHECK_LAUNCH();                                                                                    \
    } while (0)

extern "C" {

int mi355seg_maxpool2_fwd_f32(const float* x, int ldx, float* y, int ldy, uint8_t* idx,
                              int N, int D, int H, int W, int C, void* stream) {
    SEG_CHECK_ARG(x && y && idx && N > 0 && D >= 2 && H >= 2 && W >= 2 && C > 0 && ldx >= C && ldy >= C, "maxpool2_fwd: bad arguments");
    bool v = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0);
    long long total = (long long)N * (D / 2) * (H / 2) * (W / 2) * (v ? C / 4 : C);
    POOL_DISPATCH(maxpool2_fwd_kernel, v, total, x, ldx, y, ldy, idx, N, D, H, W, C);
    return MI355SEG_OK;
}
int mi355seg_maxpool2_bwd_f32(const float* dy, int lddy, const uint8_t* idx, float* dx, int lddx,
                              int N, int D, int H, int W, int C, void* stream) {
    SEG_CHECK_ARG(dy && dx && idx && N > 0 && D >= 2 && H >= 2 && W >= 2 && C > 0 && lddy >= C && lddx >= C, "maxpool2_bwd: bad arguments");
    bool v = (C % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0);
    long long total = (long long)N * D * H * W * (v ? C / 4 : C);
    POOL_DISPATCH(maxpool2_bwd_kernel, v, total, dy, lddy, idx, dx, lddx, N, D, H, W, C, (const float*)nullptr, 0);
    return MI355SEG_OK;
}
int mi355seg_maxpool2_bwd_add_f32(const float* dy, int lddy, const uint8_t* idx, const float* add, int ldadd, float* dx, int lddx,
                                  int N, int D, int H, int W, int C, void* stream) {
    SEG_CHECK_ARG(dy && dx && idx && add && N > 0 && D >= 2 && H >= 2 && W >= 2 && C > 0 && lddy >= C && lddx >= C && ldadd >= C,
                  "maxpool2_bwd_add: bad arguments");
    bool v = (C % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0) && (ldadd % 4 == 0) && ((uintptr_t)add % 16) == 0;
    long long total = (long long)N * D * H * W * (v ? C / 4 : C);
    POOL_DISPATCH(maxpool2_bwd_kernel, v, total, dy, lddy, idx, dx, lddx, N, D, H, W, C, add, ldadd);
    return MI355SEG_OK;
}
int mi355seg_upsample2_fwd_f32(const float* x, int ldx, float* y, int ldy, int N, int D, int H, int W, int C, void* stream) {
    SEG_CHECK_ARG(x && y && N > 0 && D > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && ldy >= C, "upsample2_fwd: bad arguments");
    bool v = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0);
    long long total = (long long)N * D * H * W * 8 * (v ? C / 4 : C);
    POOL_DISPATCH(upsample2_fwd_kernel, v, total, x, ldx, y, ldy, N, D, H, W, C);
    return MI355SEG_OK;
}
int mi355seg_upsample2_bwd_f32(const float* dy, int lddy, float* dx, int lddx, int N, int D, int H, int W, int C, void* stream) {
    SEG_CHECK_ARG(dy && dx && N > 0 && D > 0 && H > 0 && W > 0 && C > 0 && lddy >= C && lddx >= C, "upsample2_bwd: bad arguments");
    bool v = (C % 4 == 0) && (lddy % 4 == 0) && (lddx % 4 == 0);
    long long total = (long long)N * D * H * W * (v ? C / 4 : C);
    POOL_DISPATCH(upsample2_bwd_kernel, v, total, dy, lddy, dx, lddx, N, D, H, W, C);
    return MI355SEG_OK;
}

}  // extern "C"
