// pool.hip -- MaxPool3d(2,2) with 3-bit argmax codes and nearest-neighbour x2 upsampling,
// forward and backward, NDHWC fp32 or bf16 (template on the storage type).  Pure streaming kernels (HBM-bound), 16 B per lane.
// Reference: nn.MaxPool3d(kernel_size=2, stride=2) unet3d.py:19-25;
//            nn.Upsample(scale_factor=2, mode='nearest') residual_unet3d.py:19,103.
#include "common.h"

namespace seg {

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
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
            const T* p = x + ((((long long)n * D + 2 * od + dz) * H + 2 * oh + dy) * W + 2 * ow + dx) * ldx + c;
            float vals[NJ];
            if (VEC) { float4 q = ldf4(p); vals[0] = q.x; if (NJ > 1) { vals[1] = q.y; vals[2] = q.z; vals[3] = q.w; } }
            else vals[0] = ld1(p);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                // PyTorch: (val > max) || isnan(val); the first maximum in scan order wins
                if (t == 0 || vals[j] > best[j] || vals[j] != vals[j]) { best[j] = vals[j]; code[j] = t; }
            }
        }
        if (VEC) {
            stf4(y + v * ldy + c, make_float4(best[0], best[NJ > 1 ? 1 : 0], best[NJ > 1 ? 2 : 0], best[NJ > 1 ? 3 : 0]));
            uchar4 cd = make_uchar4((uint8_t)code[0], (uint8_t)code[NJ > 1 ? 1 : 0], (uint8_t)code[NJ > 1 ? 2 : 0], (uint8_t)code[NJ > 1 ? 3 : 0]);
            *reinterpret_cast<uchar4*>(idx + v * C + c) = cd;
        } else {
            st1(y + v * ldy + c, best[0]);
            idx[v * C + c] = (uint8_t)code[0];
        }
    }
}

// one thread per INPUT element: gradient flows to the window's arg-max position only
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ dy, int lddy, const uint8_t* __restrict__ idx,
        T* __restrict__ dx, int lddx, int N, int D, int H, int W, int C, const T* __restrict__ add, int ldadd) {
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
                float4 g = ldf4(dy + ov * lddy + c);
                uchar4 cd = *reinterpret_cast<const uchar4*>(idx + ov * C + c);
                o.x = cd.x == me ? g.x : 0.f; o.y = cd.y == me ? g.y : 0.f;
                o.z = cd.z == me ? g.z : 0.f; o.w = cd.w == me ? g.w : 0.f;
            }
            if (add) { float4 q = ldf4(add + v * ldadd + c); o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
            stf4(dx + v * lddx + c, o);
        } else {
            float o = 0.f;
            if (inside && idx[ov * C + c] == me) o = ld1(dy + ov * lddy + c);
            if (add) o += ld1(add + v * ldadd + c);
            st1(dx + v * lddx + c, o);
        }
    }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
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
        if (VEC) stf4(y + v * ldy + cc * 4, ldf4(x + iv * ldx + cc * 4));
        else st1(y + v * ldy + cc, ld1(x + iv * ldx + cc));
    }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx,
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
            const T* p = dy + ((((long long)n * D2 + 2 * id + dz) * H2 + 2 * ih + dyy) * W2 + 2 * iw + dxx) * lddy + (VEC ? cc * 4 : cc);
            if (VEC) { float4 q = ldf4(p); acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
            else acc.x += ld1(p);
        }
        if (VEC) stf4(dx + v * lddx + cc * 4, acc);
        else st1(dx + v * lddx + cc, acc.x);
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
        if (vec) hipLaunchKernelGGL((KERN<TT, true>), dim3(pgrid(total)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERN<TT, false>), dim3(pgrid(total)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);    \
        SEG_CHECK_LAUNCH();                                                                                    \
    } while (0)

extern "C" {

#define TT float
#define FN(name) mi355seg_##name##_f32
#include "pool_api.inc"
#undef TT
#undef FN
#define TT bf16
#define FN(name) mi355seg_##name##_bf16
#include "pool_api.inc"
#undef TT
#undef FN

}  // extern "C"
