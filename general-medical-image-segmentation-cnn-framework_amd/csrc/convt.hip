// convt.hip -- ConvTranspose3d(kernel_size=2, stride=2) forward / dgrad / wgrad, NDHWC fp32.
// kernel == stride => the 8 taps write disjoint output voxels: the op is a GEMM
// [voxels, Cin] x [Cin, 8*Cout] whose columns are scattered to the 2x2x2 children.
// Reference: nn.ConvTranspose3d unet3d.py:29-43, vnet3d.py:86, unetr.py:11.
// Weight layout (PyTorch): (Cin, Cout, 2, 2, 2).
#include "common.h"
#include "internal.h"

namespace seg {

// w (Cin,Cout,8) -> wp[8][Cin][Cout]  and  wd[8][Cout][Cin]
__global__ void convt_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ wd, int Cin, int Cout) {
    long long total = (long long)Cin * Cout * 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int t = (int)(i % 8); long long r = i / 8;
        int co = (int)(r % Cout); int ci = (int)(r / Cout);
        float v = w[i];
        if (wp) wp[((long long)t * Cin + ci) * Cout + co] = v;
        if (wd) wd[((long long)t * Cout + co) * Cin + ci] = v;
    }
}

// thread per (output voxel, cout)
template <typename T>
__global__ __launch_bounds__(256) void convt_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ wp,
        const float* __restrict__ bias, T* __restrict__ y, int ldy, int N, int D, int H, int W, int Cin, int Cout) {
    const int D2 = 2 * D, H2 = 2 * H, W2 = 2 * W;
    const long long total = (long long)N * D2 * H2 * W2 * Cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int co = (int)(i % Cout); long long v = i / Cout;
        int ow = (int)(v % W2); long long r = v / W2;
        int oh = (int)(r % H2); r /= H2;
        int od = (int)(r % D2); int n = (int)(r / D2);
        int t = ((od & 1) << 2) | ((oh & 1) << 1) | (ow & 1);
        const T* xp = x + ((((long long)n * D + (od >> 1)) * H + (oh >> 1)) * W + (ow >> 1)) * ldx;
        const float* wq = wp + (long long)t * Cin * Cout + co;
        float acc = bias ? bias[co] : 0.f;
        int ci = 0;
        for (; ci + 4 <= Cin; ci += 4) {
            acc = fmaf(xp[ci], wq[(long long)ci * Cout], acc);
            acc = fmaf(xp[ci + 1], wq[(long long)(ci + 1) * Cout], acc);
            acc = fmaf(xp[ci + 2], wq[(long long)(ci + 2) * Cout], acc);
            acc = fmaf(xp[ci + 3], wq[(long long)(ci + 3) * Cout], acc);
        }
        for (; ci < Cin; ++ci) acc = fmaf(xp[ci], wq[(long long)ci * Cout], acc);
        y[v * ldy + co] = acc;
    }
}

// thread per (input voxel, cin): dx = sum_{t,co} dy[child t, co] * w[ci][co][t]
template <typename T>
__global__ __launch_bounds__(256) void convt_dgrad_kernel(const T* __restrict__ dy, int lddy, const float* __restrict__ wd,
        T* __restrict__ dx, int lddx, int N, int D, int H, int W, int Cin, int Cout) {
    const int H2 = 2 * H, W2 = 2 * W, D2 = 2 * D;
    const long long total = (long long)N * D * H * W * Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int ci = (int)(i % Cin); long long v = i / Cin;
        int iw = (int)(v % W); long long r = v / W;
        int ih = (int)(r % H); r /= H;
        int id = (int)(r % D); int n = (int)(r / D);
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const T* dp = dy + ((((long long)n * D2 + 2 * id + (t >> 2)) * H2 + 2 * ih + ((t >> 1) & 1)) * W2 + 2 * iw + (t & 1)) * lddy;
            const float* wq = wd + (long long)t * Cout * Cin + ci;
            for (int co = 0; co < Cout; ++co) acc = fmaf(dp[co], wq[(long long)co * Cin], acc);
        }
        dx[v * lddx + ci] = acc;
    }
}

// grid = (pair blocks, 8 taps, splits); thread = (ci, co); partial[split][t][ci][co]
template <typename T>
__global__ __launch_bounds__(256) void convt_wgrad_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
        float* __restrict__ part, int N, int D, int H, int W, int Cin, int Cout, long long vps) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y, split = blockIdx.z;
    const long long nvox = (long long)N * D * H * W;
    long long v0 = (long long)split * vps, v1 = v0 + vps;
    if (v1 > nvox) v1 = nvox;
    if (pair >= Cin * Cout) return;
    const int co = pair % Cout, ci = pair / Cout;
    const int H2 = 2 * H, W2 = 2 * W, D2 = 2 * D;
    float acc = 0.f;
    for (long long v = v0; v < v1; ++v) {
        int iw = (int)(v % W); long long r = v / W;
        int ih = (int)(r % H); r /= H;
        int id = (int)(r % D); int n = (int)(r / D);
        long long ov = (((long long)n * D2 + 2 * id + (t >> 2)) * H2 + 2 * ih + ((t >> 1) & 1)) * W2 + 2 * iw + (t & 1);
        acc = fmaf(x[v * ldx + ci], dy[ov * lddy + co], acc);
    }
    part[(((long long)split * 8 + t) * Cin + ci) * Cout + co] = acc;
}

// dw[ci][co][t] = sum_split part[split][t][ci][co].  The slab stack is a [splits][8 * pairs] matrix: a workgroup owns 32 consecutive
// columns (128-byte row pieces), its 8 row lanes walk the splits 8 at a time (independent loads, coalesced), LDS combines the lanes in
// a fixed order.  (One thread per (ci, co) pair looping over all splits was 0.17 ms for 64 -> 32 @ 64^3: 8 workgroups, 512 serial
// strided reads each.)
__global__ __launch_bounds__(256) void convt_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits, int Cin, int Cout) {
    __shared__ float sh[8][32];
    const long long pairs = (long long)Cin * Cout, total = 8 * pairs;
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const long long col = (long long)blockIdx.x * 32 + c;
    float s0 = 0.f, s1 = 0.f;
    if (col < total) {
        const float* p = part + col;
        int k = q;
        for (; k + 8 < splits; k += 16) { s0 += p[(long long)k * total]; s1 += p[(long long)(k + 8) * total]; }
        if (k < splits) s0 += p[(long long)k * total];
    }
    sh[q][c] = s0 + s1;
    __syncthreads();
    if (q == 0 && col < total) {
        float s = sh[0][c];
#pragma unroll
        for (int j = 1; j < 8; ++j) s += sh[j][c];
        const long long t = col / pairs, pr = col - t * pairs;
        dw[pr * 8 + t] = s;
    }
}

void convt_wgrad_reduce(const float* part, float* dw, int splits, int Cin, int Cout, hipStream_t st) {
    const long long b = ((long long)8 * Cin * Cout + 31) / 32;
    hipLaunchKernelGGL(convt_wgrad_reduce_kernel, dim3((unsigned)b), dim3(256), 0, st, part, dw, splits, Cin, Cout);
}

static int convt_splits(long long nvox, int Cin, int Cout) {
    int pairblocks = cdiv((long long)Cin * Cout, 256);
    long long want = 4096 / ((long long)pairblocks * 8) + 1;
    long long maxs = nvox / 64 + 1;
    if (want > maxs) want = maxs;
    if (want > 128) want = 128;
    return (int)(want < 1 ? 1 : want);
}
static int tgrid(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}
}  // namespace seg

using namespace seg;

extern "C" {


size_t mi355seg_convt3d_k2s2_ws_bytes(int N, int D, int H, int W, int Cin, int Cout) {
    size_t wb = align_up((size_t)8 * Cin * Cout * sizeof(float), 256);
    if (convt_direct_ws_bytes(Cin, Cout) > wb) wb = convt_direct_ws_bytes(Cin, Cout);
    size_t part = align_up((size_t)convt_splits((long long)N * D * H * W, Cin, Cout) * 8 * Cin * Cout * sizeof(float), 256);
    size_t red = colsum_ws_bytes(Cout);
    size_t pw = pw_wgrad_ws_bytes((long long)N * D * H * W, Cin, Cout, 8);
    if (pw > part) part = pw;
    const size_t lw = convt_wgrad_lowp_ws_bytes((long long)N * D * H * W, Cin, Cout);
    if (lw > part) part = lw;
    const size_t slab = convt_direct_slab_bytes((long long)N * D * H * W, Cin, Cout);
    if (slab > part) part = slab;
    return wb + (part > red ? part : red) + 1024;
}


#define TT float
#define TT_MATH MATH_F32
#define FN(name) mi355seg_##name##_f32
#include "convt_api.inc"
#undef TT
#undef TT_MATH
#undef FN
#define TT bf16
#define TT_MATH MATH_B16
#define FN(name) mi355seg_##name##_bf16
#include "convt_api.inc"
#undef TT
#undef TT_MATH
#undef FN

}  // extern "C"
