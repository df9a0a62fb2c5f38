// loss.hip -- BCE-with-logits, channel argmax, integer Dice counters and the sum-type
// reductions of the library Dice losses.  All are single-pass streaming reductions:
// 16 B/lane loads, wavefront shuffle (DPP) sums, one LDS hop across the block's 4 waves,
// per-block partials and a fixed-order fp64 / int64 finalise -- deterministic, no atomics.
//
// Reference: nn.BCEWithLogitsLoss train.py:115,209 (== Binary_Loss, loss_function.py:19-41);
// pred.argmax(dim=1, keepdim=True) train.py:204; metric() utils/metric.py:20-75;
// DiceLoss / BinaryDiceLoss / DiceLossss loss_function.py:61-185.
#include "common.h"

namespace seg {

constexpr int kLossThreads = 256;
constexpr int kLossMaxBlocks = 2048;

static int loss_grid(long long items) {
    long long b = (items + kLossThreads - 1) / kLossThreads;
    return (int)(b < 1 ? 1 : (b > kLossMaxBlocks ? kLossMaxBlocks : b));
}

// block-wide sum of NV doubles; result valid in thread 0
template <int NV, typename T>
__device__ __forceinline__ void block_sum(T (&v)[NV], T* sh) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = wave_sum(v[j]);
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) sh[wid * NV + j] = v[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            T s = sh[j];
            for (int w = 1; w < nw; ++w) s += sh[w * NV + j];
            v[j] = s;
        }
    }
}

__device__ __forceinline__ float bce_term(float x, float t) {
    // max(x,0) - x*t + log1p(exp(-|x|))
    return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(kLossThreads) void bce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                                long long numel, double* __restrict__ part) {
    __shared__ double sh[4];
    double acc[1] = {0.0};
    const long long n4 = numel / 4;
    float local = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(x)[i];
        float4 b = reinterpret_cast<const float4*>(t)[i];
        local = bce_term(a.x, b.x) + bce_term(a.y, b.y) + bce_term(a.z, b.z) + bce_term(a.w, b.w);
        acc[0] += (double)local;
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 3)) {
        long long i = n4 * 4 + threadIdx.x;
        acc[0] += (double)bce_term(x[i], t[i]);
    }
    block_sum<1>(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(256) void bce_finalize_kernel(const double* __restrict__ part, int nblk, double numel, float* __restrict__ loss) {
    __shared__ double sh[4];
    double acc[1] = {0.0};
    for (int i = threadIdx.x; i < nblk; i += 256) acc[0] += part[i];
    block_sum<1>(acc, sh);
    if (threadIdx.x == 0) loss[0] = (float)(acc[0] / numel);
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                       const float* __restrict__ gscale, long long numel, float* __restrict__ dx) {
    const float sc = gscale[0] / (float)numel;
    const long long n4 = numel / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(x)[i];
        float4 b = reinterpret_cast<const float4*>(t)[i];
        float4 o;
        o.x = (1.f / (1.f + expf(-a.x)) - b.x) * sc;
        o.y = (1.f / (1.f + expf(-a.y)) - b.y) * sc;
        o.z = (1.f / (1.f + expf(-a.z)) - b.z) * sc;
        o.w = (1.f / (1.f + expf(-a.w)) - b.w) * sc;
        reinterpret_cast<float4*>(dx)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (numel & 3)) {
        long long i = n4 * 4 + threadIdx.x;
        dx[i] = (1.f / (1.f + expf(-x[i])) - t[i]) * sc;
    }
}

__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ x, long long N, int K, long long S,
                                                      int64_t* __restrict__ mask) {
    const long long total = N * S;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long n = i / S, s = i % S;
        const float* p = x + n * K * S + s;
        float best = p[0]; int bi = 0;
        for (int k = 1; k < K; ++k) { float v = p[(long long)k * S]; if (v > best) { best = v; bi = k; } }
        mask[i] = bi;
    }
}

__global__ __launch_bounds__(kLossThreads) void dice_counts_kernel(const int64_t* __restrict__ gt, const int64_t* __restrict__ pr,
                                                                    long long numel, long long* __restrict__ part) {
    __shared__ long long sh[16];
    long long acc[4] = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (long long)gridDim.x * blockDim.x) {
        long long g = gt[i], p = pr[i];
        acc[0] += g; acc[1] += p;
        acc[2] += ((g & p) != 0); acc[3] += ((g | p) != 0);
    }
    block_sum<4>(acc, sh);
    if (threadIdx.x == 0) for (int j = 0; j < 4; ++j) part[(long long)blockIdx.x * 4 + j] = acc[j];
}

__global__ __launch_bounds__(256) void counts_finalize_kernel(const long long* __restrict__ part, int nblk, int64_t* __restrict__ counts) {
    __shared__ long long sh[16];
    long long acc[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblk; i += 256)
        for (int j = 0; j < 4; ++j) acc[j] += part[(long long)i * 4 + j];
    block_sum<4>(acc, sh);
    if (threadIdx.x == 0) for (int j = 0; j < 4; ++j) counts[j] = acc[j];
}

// train.py:190-193: gt_back = (gt == 0); gt = cat([gt_back, gt], dim=1) as float, one pass
__global__ __launch_bounds__(256) void two_channel_gt_kernel(const float* __restrict__ gt, float* __restrict__ out, long long N, long long S) {
    const long long total = N * S;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / S, s = i - n * S;
        const float v = gt[i];
        out[(2 * n) * S + s] = v == 0.f ? 1.f : 0.f;
        out[(2 * n + 1) * S + s] = v;
    }
}

// one pass over logits + one-hot targets: BCE sum, argmax(pred), argmax(gt), Dice counters
__global__ __launch_bounds__(kLossThreads) void bce_argmax_dice_kernel(const float* __restrict__ x, const float* __restrict__ t,
        long long N, int K, long long S, int64_t* __restrict__ mask, double* __restrict__ lpart, long long* __restrict__ cpart) {
    __shared__ double shd[4];
    __shared__ long long shc[16];
    double lacc[1] = {0.0};
    long long acc[4] = {0, 0, 0, 0};
    const long long total = N * S;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long n = i / S, s = i % S;
        const float* px = x + n * K * S + s;
        const float* pt = t + n * K * S + s;
        float bx = px[0], bt = pt[0]; int ix = 0, it = 0;
        float l = bce_term(bx, bt);
        for (int k = 1; k < K; ++k) {
            float vx = px[(long long)k * S], vt = pt[(long long)k * S];
            l += bce_term(vx, vt);
            if (vx > bx) { bx = vx; ix = k; }
            if (vt > bt) { bt = vt; it = k; }
        }
        lacc[0] += (double)l;
        mask[i] = ix;
        acc[0] += it; acc[1] += ix;
        acc[2] += ((it & ix) != 0); acc[3] += ((it | ix) != 0);
    }
    block_sum<1>(lacc, shd);
    block_sum<4>(acc, shc);
    if (threadIdx.x == 0) {
        lpart[blockIdx.x] = lacc[0];
        for (int j = 0; j < 4; ++j) cpart[(long long)blockIdx.x * 4 + j] = acc[j];
    }
}

__global__ __launch_bounds__(kLossThreads) void dice_sums_kernel(const float* __restrict__ x, const float* __restrict__ t,
        long long numel, int apply_sigmoid, double* __restrict__ part) {
    __shared__ double sh[20];
    double acc[5] = {0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (long long)gridDim.x * blockDim.x) {
        float a = x[i], b = t[i];
        if (apply_sigmoid) a = 1.f / (1.f + expf(-a));
        acc[0] += (double)(a * b); acc[1] += (double)a; acc[2] += (double)b;
        acc[3] += (double)(a * a); acc[4] += (double)(b * b);
    }
    block_sum<5>(acc, sh);
    if (threadIdx.x == 0) for (int j = 0; j < 5; ++j) part[(long long)blockIdx.x * 5 + j] = acc[j];
}
__global__ __launch_bounds__(256) void sums_finalize5_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    __shared__ double sh[20];
    double acc[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblk; i += 256)
        for (int j = 0; j < 5; ++j) acc[j] += part[(long long)i * 5 + j];
    block_sum<5>(acc, sh);
    if (threadIdx.x == 0) for (int j = 0; j < 5; ++j) out[j] = acc[j];
}

__global__ __launch_bounds__(256) void dice_sums_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
        const double* __restrict__ g5, long long numel, int apply_sigmoid, float* __restrict__ dx) {
    const float g0 = (float)g5[0], g1 = (float)g5[1], g3 = (float)g5[3];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (long long)gridDim.x * blockDim.x) {
        float a = x[i], d = 1.f;
        if (apply_sigmoid) { a = 1.f / (1.f + expf(-a)); d = a * (1.f - a); }
        dx[i] = (g0 * t[i] + g1 + 2.f * g3 * a) * d;
    }
}

// ---- row-segmented form: R contiguous rows of L elements, all rows in ONE launch (BinaryDiceLoss: row = sample,
// loss_function.py:78-83; DiceLossss: row = (sample, class), loss_function.py:160-183).  Denominator exponent p
// (loss_function.py:82 takes any p): a^p = a*a for p == 2 (the bits of the single-row kernel), a for p == 1, powf otherwise.
__device__ __forceinline__ float pow_p(float a, float p) { return p == 2.f ? a * a : (p == 1.f ? a : powf(a, p)); }
__device__ __forceinline__ float dpow_p(float a, float p) {     // d a^p / da, torch's pow backward: p * a^(p-1), 0 for p == 0
    return p == 2.f ? 2.f * a : (p == 1.f ? 1.f : (p == 0.f ? 0.f : p * powf(a, p - 1.f)));
}
__device__ __forceinline__ void dice_acc(double (&acc)[5], float a, float b, int apply_sigmoid, float p) {
    if (apply_sigmoid) a = 1.f / (1.f + expf(-a));
    acc[0] += (double)(a * b); acc[1] += (double)a; acc[2] += (double)b;
    acc[3] += (double)pow_p(a, p); acc[4] += (double)pow_p(b, p);
}
// grid = (blocks per row, R); part[(row * gridDim.x + block) * 5 + j]
__global__ __launch_bounds__(kLossThreads) void dice_rows_kernel(const float* __restrict__ x, const float* __restrict__ t,
        long long L, int apply_sigmoid, float p, int vec, double* __restrict__ part) {
    __shared__ double sh[20];
    double acc[5] = {0, 0, 0, 0, 0};
    const float* xr = x + (long long)blockIdx.y * L;
    const float* tr = t + (long long)blockIdx.y * L;
    const long long stride = (long long)gridDim.x * blockDim.x, first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {                                        // rows 16-byte aligned: 16 B per lane
        const long long n4 = L / 4;
        for (long long i = first; i < n4; i += stride) {
            const float4 a = reinterpret_cast<const float4*>(xr)[i], b = reinterpret_cast<const float4*>(tr)[i];
            dice_acc(acc, a.x, b.x, apply_sigmoid, p); dice_acc(acc, a.y, b.y, apply_sigmoid, p);
            dice_acc(acc, a.z, b.z, apply_sigmoid, p); dice_acc(acc, a.w, b.w, apply_sigmoid, p);
        }
        if (blockIdx.x == 0 && threadIdx.x < (L & 3)) dice_acc(acc, xr[n4 * 4 + threadIdx.x], tr[n4 * 4 + threadIdx.x], apply_sigmoid, p);
    } else {
        for (long long i = first; i < L; i += stride) dice_acc(acc, xr[i], tr[i], apply_sigmoid, p);
    }
    block_sum<5>(acc, sh);
    if (threadIdx.x == 0) for (int j = 0; j < 5; ++j) part[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 5 + j] = acc[j];
}
// one block per row, fixed order
__global__ __launch_bounds__(256) void dice_rows_finalize_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    __shared__ double sh[20];
    double acc[5] = {0, 0, 0, 0, 0};
    const double* pr = part + (long long)blockIdx.x * nblk * 5;
    for (int i = threadIdx.x; i < nblk; i += 256)
        for (int j = 0; j < 5; ++j) acc[j] += pr[(long long)i * 5 + j];
    block_sum<5>(acc, sh);
    if (threadIdx.x == 0) for (int j = 0; j < 5; ++j) out[(long long)blockIdx.x * 5 + j] = acc[j];
}
// dx[r][i] = (g[r][0]*t + g[r][1] + g[r][3] * d(a^p)/da) * da/dx
__global__ __launch_bounds__(256) void dice_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
        const double* __restrict__ g5, long long L, int apply_sigmoid, float p, float* __restrict__ dx) {
    const double* g = g5 + (long long)blockIdx.y * 5;
    const float g0 = (float)g[0], g1 = (float)g[1], g3 = (float)g[3];
    const long long base = (long long)blockIdx.y * L;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long long)gridDim.x * blockDim.x) {
        float a = x[base + i], d = 1.f;
        if (apply_sigmoid) { a = 1.f / (1.f + expf(-a)); d = a * (1.f - a); }
        dx[base + i] = (g0 * t[base + i] + g1 + g3 * dpow_p(a, p)) * d;
    }
}

// ---- ZNormalization of one volume (dataloader.py:94, torchio: (x - mean) / std over all voxels, unbiased std).
// Sums of d = x - x[0] and d^2 in fp64 (the pivot keeps sum d^2 - (sum d)^2 / n well conditioned for CT-like offsets),
// two-stage and fixed-order; the apply pass turns them into mean and 1 / std.
__global__ __launch_bounds__(kLossThreads) void znorm_sums_kernel(const float* __restrict__ x, long long n, double* __restrict__ part) {
    __shared__ double sh[8];
    const float pv = x[0];
    double acc[2] = {0.0, 0.0};
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float a = v.x - pv, b = v.y - pv, c = v.z - pv, d = v.w - pv;
        acc[0] += (double)a + (double)b + (double)c + (double)d;
        acc[1] += (double)a * a + (double)b * b + (double)c * c + (double)d * d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float a = x[n4 * 4 + threadIdx.x] - pv; acc[0] += a; acc[1] += (double)a * a; }
    block_sum<2>(acc, sh);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = acc[0]; part[2 * blockIdx.x + 1] = acc[1]; }
}
__global__ __launch_bounds__(256) void znorm_finalize_kernel(const double* __restrict__ part, int nblk, const float* __restrict__ x, long long n, float* __restrict__ mr) {
    __shared__ double sh[8];
    double acc[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < nblk; i += 256) { acc[0] += part[2 * i]; acc[1] += part[2 * i + 1]; }
    block_sum<2>(acc, sh);
    if (threadIdx.x == 0) {
        const double md = acc[0] / (double)n, var = (acc[1] - acc[0] * md) / (double)(n - 1);
        mr[0] = (float)((double)x[0] + md);
        mr[1] = (float)(1.0 / sqrt(var));
    }
}
__global__ __launch_bounds__(256) void znorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mr, long long n, float* __restrict__ y) {
    const float m = mr[0], r = mr[1];
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = (v.x - m) * r; v.y = (v.y - m) * r; v.z = (v.z - m) * r; v.w = (v.w - m) * r;
        reinterpret_cast<float4*>(y)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[n4 * 4 + threadIdx.x] = (x[n4 * 4 + threadIdx.x] - m) * r;
}

constexpr int kMaxClasses = 16;

__global__ __launch_bounds__(256) void softmax_ch_kernel(const float* __restrict__ x, float* __restrict__ y, long long N, int K, long long S) {
    const long long total = N * S;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / S, s = i % S;
        const float* p = x + n * K * S + s;
        float v[kMaxClasses], m = -INFINITY;
        for (int k = 0; k < K; ++k) { v[k] = p[(long long)k * S]; m = fmaxf(m, v[k]); }
        float sum = 0.f;
        for (int k = 0; k < K; ++k) { v[k] = expf(v[k] - m); sum += v[k]; }
        const float inv = 1.f / sum;
        float* q = y + n * K * S + s;
        for (int k = 0; k < K; ++k) q[(long long)k * S] = v[k] * inv;
    }
}
__global__ __launch_bounds__(256) void softmax_ch_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                                              long long N, int K, long long S) {
    const long long total = N * S;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / S, s = i % S, base = n * K * S + s;
        float dot = 0.f;
        for (int k = 0; k < K; ++k) dot += dy[base + (long long)k * S] * y[base + (long long)k * S];
        for (int k = 0; k < K; ++k) dx[base + (long long)k * S] = y[base + (long long)k * S] * (dy[base + (long long)k * S] - dot);
    }
}

__global__ __launch_bounds__(kLossThreads) void ce3d_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ lab,
        const float* __restrict__ w, long long N, int K, long long S, double* __restrict__ part) {
    __shared__ double sh[4];
    double acc[1] = {0.0};
    const long long total = N * S;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / S, s = i % S;
        const float* p = x + n * K * S + s;
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) m = fmaxf(m, p[(long long)k * S]);
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += expf(p[(long long)k * S] - m);
        const int l = (int)lab[i];
        const float nll = (m + logf(sum)) - p[(long long)l * S];
        acc[0] += (double)(w ? w[l] * nll : nll);
    }
    block_sum<1>(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}
__global__ __launch_bounds__(256) void ce3d_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ lab, const float* __restrict__ w,
        const float* __restrict__ gscale, long long N, int K, long long S, float inv_count, float* __restrict__ dx) {
    const long long total = N * S;
    const float g = gscale[0] * inv_count;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / S, s = i % S, base = n * K * S + s;
        float m = -INFINITY;
        for (int k = 0; k < K; ++k) m = fmaxf(m, x[base + (long long)k * S]);
        float sum = 0.f;
        for (int k = 0; k < K; ++k) sum += expf(x[base + (long long)k * S] - m);
        const int l = (int)lab[i];
        const float sc = g * (w ? w[l] : 1.f), inv = 1.f / sum;
        for (int k = 0; k < K; ++k)
            dx[base + (long long)k * S] = sc * (expf(x[base + (long long)k * S] - m) * inv - (k == l ? 1.f : 0.f));
    }
}

}  // namespace seg

using namespace seg;

extern "C" {

size_t mi355seg_loss_ws_bytes(long long numel) {
    (void)numel;
    return (size_t)kLossMaxBlocks * (8 * sizeof(double)) + 1024;
}

int mi355seg_bce_logits_fwd_f32(const float* logits, const float* target, long long numel,
                                float* loss, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(logits && target && loss && numel > 0, "bce_logits_fwd: bad arguments");
    SEG_CHECK_ARG(((uintptr_t)logits % 16) == 0 && ((uintptr_t)target % 16) == 0, "bce_logits_fwd: pointers must be 16-byte aligned");
    ProfScope ps(PF_LOSS, 0.0, 8.0 * numel, (hipStream_t)stream);
    int nblk = loss_grid(numel / 4 + 1);
    SEG_CHECK_WS((size_t)nblk * sizeof(double), ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(nblk), dim3(kLossThreads), 0, st, logits, target, numel, (double*)ws);
    SEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(bce_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nblk, (double)numel, loss);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_bce_logits_bwd_f32(const float* logits, const float* target, const float* gscale,
                                long long numel, float* dlogits, void* stream) {
    SEG_CHECK_ARG(logits && target && gscale && dlogits && numel > 0, "bce_logits_bwd: bad arguments");
    SEG_CHECK_ARG(((uintptr_t)logits % 16) == 0 && ((uintptr_t)target % 16) == 0 && ((uintptr_t)dlogits % 16) == 0,
                  "bce_logits_bwd: pointers must be 16-byte aligned");
    ProfScope ps(PF_LOSS, 0.0, 12.0 * numel, (hipStream_t)stream);
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(loss_grid(numel / 4 + 1) * 2), dim3(256), 0, (hipStream_t)stream, logits, target,
                       gscale, numel, dlogits);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_argmax_ch_f32(const float* logits, long long N, int K, long long S, int64_t* mask, void* stream) {
    SEG_CHECK_ARG(logits && mask && N > 0 && K > 0 && S > 0, "argmax_ch: bad arguments");
    ProfScope ps(PF_LOSS, 0.0, (4.0 * K + 8.0) * N * S, (hipStream_t)stream);
    hipLaunchKernelGGL(argmax_kernel, dim3(loss_grid(N * S) * 2), dim3(256), 0, (hipStream_t)stream, logits, N, K, S, mask);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_dice_counts_i64(const int64_t* gt, const int64_t* pred, long long numel,
                             int64_t* counts, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(gt && pred && counts && numel > 0, "dice_counts: bad arguments");
    ProfScope ps(PF_LOSS, 0.0, 16.0 * numel, (hipStream_t)stream);
    int nblk = loss_grid(numel);
    SEG_CHECK_WS((size_t)nblk * 4 * sizeof(long long), ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dice_counts_kernel, dim3(nblk), dim3(kLossThreads), 0, st, gt, pred, numel, (long long*)ws);
    SEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(counts_finalize_kernel, dim3(1), dim3(256), 0, st, (const long long*)ws, nblk, counts);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_two_channel_gt_f32(const float* gt, float* out, long long N, long long S, void* stream) {
    SEG_CHECK_ARG(gt && out && N > 0 && S > 0, "two_channel_gt: bad arguments");
    ProfScope ps(PF_LOSS, 0.0, 12.0 * N * S, (hipStream_t)stream);
    hipLaunchKernelGGL(two_channel_gt_kernel, dim3(loss_grid(N * S) * 2), dim3(256), 0, (hipStream_t)stream, gt, out, N, S);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_bce_argmax_dice_f32(const float* logits, const float* target, long long N, int K, long long S,
                                 float* loss, int64_t* mask, int64_t* counts,
                                 void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(logits && target && loss && mask && counts && N > 0 && K > 0 && S > 0, "bce_argmax_dice: bad arguments");
    ProfScope ps(PF_LOSS, 0.0, (8.0 * K + 8.0) * N * S, (hipStream_t)stream);
    int nblk = loss_grid(N * S);
    SEG_CHECK_WS((size_t)nblk * 5 * sizeof(double), ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    double* lpart = (double*)ws;
    long long* cpart = (long long*)((char*)ws + (size_t)nblk * sizeof(double));
    hipLaunchKernelGGL(bce_argmax_dice_kernel, dim3(nblk), dim3(kLossThreads), 0, st, logits, target, N, K, S, mask, lpart, cpart);
    SEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(bce_finalize_kernel, dim3(1), dim3(256), 0, st, lpart, nblk, (double)(N * K * S), loss);
    SEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(counts_finalize_kernel, dim3(1), dim3(256), 0, st, cpart, nblk, counts);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_dice_sums_f32(const float* x, const float* t, long long numel, int apply_sigmoid,
                           double* out5, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(x && t && out5 && numel > 0, "dice_sums: bad arguments");
    ProfScope ps(PF_LOSS, 0.0, 8.0 * numel, (hipStream_t)stream);
    int nblk = loss_grid(numel);
    SEG_CHECK_WS((size_t)nblk * 5 * sizeof(double), ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dice_sums_kernel, dim3(nblk), dim3(kLossThreads), 0, st, x, t, numel, apply_sigmoid, (double*)ws);
    SEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(sums_finalize5_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nblk, out5);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_dice_sums_bwd_f32(const float* x, const float* t, const double* g5, long long numel, int apply_sigmoid,
                               float* dx, void* stream) {
    SEG_CHECK_ARG(x && t && g5 && dx && numel > 0, "dice_sums_bwd: bad arguments");
    ProfScope ps(PF_LOSS, 0.0, 12.0 * numel, (hipStream_t)stream);
    hipLaunchKernelGGL(dice_sums_bwd_kernel, dim3(loss_grid(numel) * 2), dim3(256), 0, (hipStream_t)stream, x, t, g5, numel, apply_sigmoid, dx);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

static int dice_rows_grid(long long rows, long long L) {
    long long per = (L + (long long)kLossThreads * 16 - 1) / ((long long)kLossThreads * 16);      // >= 16 elements per thread
    long long cap = kLossMaxBlocks / rows;
    if (per > cap) per = cap;
    return (int)(per < 1 ? 1 : per);
}
size_t mi355seg_dice_rows_ws_bytes(long long rows, long long len) {
    if (rows < 1 || len < 1) return 0;
    return (size_t)rows * dice_rows_grid(rows, len) * 5 * sizeof(double);
}
int mi355seg_dice_rows_f32(const float* x, const float* t, long long rows, long long len, int apply_sigmoid, float p,
                           double* out, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(x && t && out && rows > 0 && rows <= 65535 && len > 0, "dice_rows: bad arguments (1 <= rows <= 65535)");
    ProfScope ps(PF_LOSS, 0.0, 8.0 * rows * len, (hipStream_t)stream);
    const int nblk = dice_rows_grid(rows, len);
    SEG_CHECK_WS((size_t)rows * nblk * 5 * sizeof(double), ws_bytes);
    const int vec = ((uintptr_t)x % 16) == 0 && ((uintptr_t)t % 16) == 0 && (rows == 1 || len % 4 == 0);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dice_rows_kernel, dim3(nblk, (unsigned)rows), dim3(kLossThreads), 0, st, x, t, len, apply_sigmoid, p, vec, (double*)ws);
    SEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(dice_rows_finalize_kernel, dim3((unsigned)rows), dim3(256), 0, st, (const double*)ws, nblk, out);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_dice_rows_bwd_f32(const float* x, const float* t, const double* g, long long rows, long long len, int apply_sigmoid,
                               float p, float* dx, void* stream) {
    SEG_CHECK_ARG(x && t && g && dx && rows > 0 && rows <= 65535 && len > 0, "dice_rows_bwd: bad arguments (1 <= rows <= 65535)");
    ProfScope ps(PF_LOSS, 0.0, 12.0 * rows * len, (hipStream_t)stream);
    long long per = (len + 256 * 8 - 1) / (256 * 8);
    if (per > 4096) per = 4096;
    hipLaunchKernelGGL(dice_rows_bwd_kernel, dim3((unsigned)per, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, t, g, len, apply_sigmoid, p, dx);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

size_t mi355seg_znorm_ws_bytes(long long n) { (void)n; return (size_t)kLossMaxBlocks * 2 * sizeof(double) + 64; }
int mi355seg_znorm_f32(const float* x, long long n, float* y, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(x && y && n > 1, "znorm: bad arguments (n > 1)");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "znorm: pointers must be 16-byte aligned");
    SEG_CHECK_WS(mi355seg_znorm_ws_bytes(n), ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(PF_LOSS, 0.0, 12.0 * n, st);
    float* mr = (float*)ws;                                    // mean, 1 / std; the block partials behind them
    double* part = (double*)((char*)ws + 64);
    const int nblk = loss_grid(n / 4 + 1);
    hipLaunchKernelGGL(znorm_sums_kernel, dim3(nblk), dim3(kLossThreads), 0, st, x, n, part);
    SEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(znorm_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)part, nblk, x, n, mr);
    SEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(znorm_apply_kernel, dim3(loss_grid(n / 4 + 1) * 2), dim3(256), 0, st, x, (const float*)mr, n, y);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_softmax_ch_f32(const float* x, float* y, long long N, int K, long long S, void* stream) {
    SEG_CHECK_ARG(x && y && N > 0 && K > 0 && K <= kMaxClasses && S > 0, "softmax_ch: bad arguments (K <= %d)", kMaxClasses);
    ProfScope ps(PF_LOSS, 0.0, 8.0 * N * K * S, (hipStream_t)stream);
    hipLaunchKernelGGL(softmax_ch_kernel, dim3(loss_grid(N * S) * 2), dim3(256), 0, (hipStream_t)stream, x, y, N, K, S);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_softmax_ch_bwd_f32(const float* y, const float* dy, float* dx, long long N, int K, long long S, void* stream) {
    SEG_CHECK_ARG(y && dy && dx && N > 0 && K > 0 && S > 0, "softmax_ch_bwd: bad arguments");
    ProfScope ps(PF_LOSS, 0.0, 12.0 * N * K * S, (hipStream_t)stream);
    hipLaunchKernelGGL(softmax_ch_bwd_kernel, dim3(loss_grid(N * S) * 2), dim3(256), 0, (hipStream_t)stream, y, dy, dx, N, K, S);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_ce3d_fwd_f32(const float* logits, const int64_t* labels, const float* weight, long long N, int K, long long S,
                          int size_average, float* loss, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(logits && labels && loss && N > 0 && K > 0 && S > 0, "ce3d_fwd: bad arguments");
    ProfScope ps(PF_LOSS, 0.0, (4.0 * K + 8.0) * N * S, (hipStream_t)stream);
    int nblk = loss_grid(N * S);
    SEG_CHECK_WS((size_t)nblk * sizeof(double), ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ce3d_fwd_kernel, dim3(nblk), dim3(kLossThreads), 0, st, logits, labels, weight, N, K, S, (double*)ws);
    SEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(bce_finalize_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nblk, size_average ? (double)(N * S) : 1.0, loss);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_ce3d_bwd_f32(const float* logits, const int64_t* labels, const float* weight, const float* gscale,
                          long long N, int K, long long S, int size_average, float* dlogits, void* stream) {
    SEG_CHECK_ARG(logits && labels && gscale && dlogits && N > 0 && K > 0 && S > 0, "ce3d_bwd: bad arguments");
    ProfScope ps(PF_LOSS, 0.0, (8.0 * K + 8.0) * N * S, (hipStream_t)stream);
    hipLaunchKernelGGL(ce3d_bwd_kernel, dim3(loss_grid(N * S) * 2), dim3(256), 0, (hipStream_t)stream, logits, labels, weight, gscale,
                       N, K, S, size_average ? 1.f / (float)(N * S) : 1.f, dlogits);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // extern "C"
