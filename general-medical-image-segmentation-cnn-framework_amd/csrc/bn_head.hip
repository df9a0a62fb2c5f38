// bn_head.hip -- the LAST double-conv block's BatchNorm + activation fused with the 1x1x1 output head (unet3d.py:46-48,68-71:
// ``self.conv(dec1)`` behind decoder1's norm2 + relu2).  The activation a2 = act(BN(y2)) has exactly one consumer -- the head, K = 2..4
// output channels, recomputable from K floats per voxel -- so it is never written:
//
//   forward   logits[v][k] = bh[k] + sum_c Wh[k][c] * act(al[c] y2[v][c] + be[c])                      (reads y2, writes logits)
//   backward  da2[v][c] = sum_k dlogits[v][k] Wh[k][c];  dz = da2 * act'(z);  one pass over (y2, dlogits) reduces the BatchNorm
//             backward's two column sums s1 = sum dz, s2 = sum dz * xhat TOGETHER WITH the head's weight / bias gradients
//             dWh[k][c] = sum_v dlogits[v][k] a2[v][c], dbh[k] = sum_v dlogits[v][k]; a second pass writes
//             dy2 = rstd * gamma * (dz - s1 / n - xhat * s2 / n) (+ its column sums = conv2's bias gradient, + max |dy2| for f16x3).
//
// Against the unfused chain (norm_act_fwd, head_fwd, head_dgrad, head_wgrad, norm_act_bwd sums + apply) on [2, 32, 128^3]: 5.6 GB
// of HBM traffic -> 2.4 GB and 537 MB less memory; HBM-bound streaming kernels, thread = (voxel, channel quad), C / 4 lanes per
// voxel (16-byte accesses, a wavefront covers 64 consecutive 16-byte pieces of the NDHWC stream), two voxels per trip.
// The forward reproduces norm_act_fwd_kernel's and head_fwd_kernel's arithmetic operation by operation: same logits, bit for bit.
#include "common.h"
#include "internal.h"

namespace seg {

namespace {

constexpr int kBhMaxBlocks = 1024;

struct BhArgs {
    const float* y; int ldy;                        // pre-norm tensor [rows][C]
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    const float* wh; const float* bh;               // head weights [K][C], bias [K]
    const float* dl; int lddl;                      // d(logits) [rows][K]  (backward)
    float* out; int ldo;                            // logits (forward) / dy2 (apply)
    const float* s1; const float* s2;               // apply
    float* part;                                    // per-block partials
    unsigned* amax;                                 // apply: max |dy2|
    long long rows; int C; int act; float slope;
};

// z = gamma xhat + beta from the backward's point of view (norm.hip BwdF::one), a2 from the forward's (norm_act_fwd_kernel)
template <int K>
__global__ __launch_bounds__(256) void bn_head_fwd_kernel(BhArgs a) {
    const int LPV = a.C / 4, VPB = 256 / LPV;
    const int c4 = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    f32x4_t wr[K], al, be;
#pragma unroll
    for (int k = 0; k < K; ++k) wr[k] = ld4(a.wh + (long long)k * a.C + c4 * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        al[j] = a.rstd[c4 * 4 + j] * (a.gamma ? a.gamma[c4 * 4 + j] : 1.f);
        be[j] = (a.beta ? a.beta[c4 * 4 + j] : 0.f) - a.mean[c4 * 4 + j] * al[j];
    }
    float bias[K];
#pragma unroll
    for (int k = 0; k < K; ++k) bias[k] = a.bh ? a.bh[k] : 0.f;
    const long long stride = (long long)gridDim.x * VPB;
    for (long long v = (long long)blockIdx.x * VPB + vl; v < a.rows; v += 2 * stride) {
        const bool two = v + stride < a.rows;
        const float4 x0 = ldf4(a.y + v * a.ldy + c4 * 4);
        const float4 x1 = two ? ldf4(a.y + (v + stride) * a.ldy + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float4 x = u ? x1 : x0;
            f32x4_t o;
            o[0] = act_apply(fmaf(x.x, al[0], be[0]), a.act, a.slope); o[1] = act_apply(fmaf(x.y, al[1], be[1]), a.act, a.slope);
            o[2] = act_apply(fmaf(x.z, al[2], be[2]), a.act, a.slope); o[3] = act_apply(fmaf(x.w, al[3], be[3]), a.act, a.slope);
            float r[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float s = o[0] * wr[k][0] + o[1] * wr[k][1] + o[2] * wr[k][2] + o[3] * wr[k][3];
                for (int off = LPV >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                r[k] = s + bias[k];
            }
            if (c4 == 0 && (u == 0 || two)) {
                float* dst = a.out + (v + u * stride) * a.ldo;
#pragma unroll
                for (int k = 0; k < K; ++k) dst[k] = r[k];
            }
        }
    }
}

// per-voxel backward quantities of one channel quad
template <int K>
struct BhBwd {
    f32x4_t wr[K], m, rs, ga, be, al, bf;
    __device__ __forceinline__ void init(const BhArgs& a, int c4) {
#pragma unroll
        for (int k = 0; k < K; ++k) wr[k] = ld4(a.wh + (long long)k * a.C + c4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[j] = a.mean[c4 * 4 + j]; rs[j] = a.rstd[c4 * 4 + j];
            ga[j] = a.gamma ? a.gamma[c4 * 4 + j] : 1.f; be[j] = a.beta ? a.beta[c4 * 4 + j] : 0.f;
            al[j] = rs[j] * ga[j];                                             // the forward's folded form (bn_head_fwd_kernel)
            bf[j] = be[j] - m[j] * al[j];
        }
    }
    // dz, xhat (and the activation value when WANT_A) of the quad at one voxel
    template <bool WANT_A>
    __device__ __forceinline__ void eval(const BhArgs& a, const float4 x, const float (&dl)[K], f32x4_t& dz, f32x4_t& xh, f32x4_t& av) const {
        f32x4_t da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; ++k) da += dl[k] * wr[k];                       // head_dgrad_kernel's accumulation order
        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh[j] = (xv[j] - m[j]) * rs[j];
            const float z = fmaf(xh[j], ga[j], be[j]);
            dz[j] = da[j] * act_grad(z, a.act, a.slope);
            if (WANT_A) av[j] = act_apply(fmaf(xv[j], al[j], bf[j]), a.act, a.slope);   // the forward's a2
        }
    }
};

// partials of one block: part[blk][ (2 + K) * C + K ] = s1[C] | s2[C] | dWh[K][C] | dbh[K]
template <int K>
__global__ __launch_bounds__(256) void bn_head_bwd_sums_kernel(BhArgs a) {
    constexpr int NS = 8 + 4 * K;                       // floats per thread: s1 quad, s2 quad, K dWh quads
    __shared__ float sred[4 * 64 * (NS + K)];
    const int LPV = a.C / 4, VPB = 256 / LPV;
    const int c4 = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    BhBwd<K> q;
    q.init(a, c4);
    f32x4_t s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, dw[K];
    float db[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { dw[k] = s1; db[k] = 0.f; }
    const long long stride = (long long)gridDim.x * VPB;
    for (long long v = (long long)blockIdx.x * VPB + vl; v < a.rows; v += 2 * stride) {
        const bool two = v + stride < a.rows;
        const float4 x0 = ldf4(a.y + v * a.ldy + c4 * 4);
        const float4 x1 = two ? ldf4(a.y + (v + stride) * a.ldy + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float d0[K], d1[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { d0[k] = a.dl[v * a.lddl + k]; d1[k] = two ? a.dl[(v + stride) * a.lddl + k] : 0.f; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            f32x4_t dz, xh, av;
            q.template eval<true>(a, u ? x1 : x0, u ? d1 : d0, dz, xh, av);
            s1 += dz; s2 += dz * xh;
#pragma unroll
            for (int k = 0; k < K; ++k) { const float d = u ? d1[k] : d0[k]; dw[k] += d * av; db[k] += d; }
        }
    }
    // lanes of a wave that own the same channel quad: xor-shuffle tree over the voxel slots, then the four waves through LDS
    float vals[NS + K];
#pragma unroll
    for (int j = 0; j < 4; ++j) { vals[j] = s1[j]; vals[4 + j] = s2[j]; }
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) vals[8 + 4 * k + j] = dw[k][j];
        vals[NS + k] = db[k];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS + K; ++i) {
        float s = vals[i];
        for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
        vals[i] = s;
    }
    const int nl = LPV < 64 ? LPV : 64;              // lanes [0, nl) of each wave hold that wave's totals of quad `lane`
    if (lane < nl) {
#pragma unroll
        for (int i = 0; i < NS + K; ++i) sred[(wave * 64 + lane) * (NS + K) + i] = vals[i];
    }
    __syncthreads();
    // LPV == 64: a wave covers one voxel per trip and lane == c4; LPV < 64: lane < LPV == c4
    const int ncol = (2 + K) * a.C + K;
    float* dst = a.part + (long long)blockIdx.x * ncol;
    for (int i = threadIdx.x; i < ncol; i += 256) {
        float s = 0.f;
        if (i < (2 + K) * a.C) {
            const int which = i / a.C, c = i % a.C;                       // 0: s1, 1: s2, 2 + k: dWh[k]
            const int slot = which < 2 ? which * 4 + c % 4 : 8 + 4 * (which - 2) + c % 4;
            for (int w = 0; w < 4; ++w) s += sred[(w * 64 + (c / 4) % 64) * (NS + K) + slot];
        } else {
            const int k = i - (2 + K) * a.C;                              // every quad's lanes saw every voxel of their slots: take quad 0's
            for (int w = 0; w < 4; ++w) s += sred[(w * 64 + 0) * (NS + K) + NS + k];
        }
        dst[i] = s;
    }
}

// out[j] = sum over the blocks of part[b][j], fp64, one wavefront per column, fixed order
__global__ __launch_bounds__(64) void bh_finalize_kernel(const float* __restrict__ part, int nblk, int ncol, float* __restrict__ o0, int n0,
                                                        float* __restrict__ o1, int n1, float* __restrict__ o2, int n2, float* __restrict__ o3,
                                                        float* __restrict__ alias0, float* __restrict__ alias1) {
    const int j = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)part[(long long)b * ncol + j];
    s = wave_sum(s);
    if (threadIdx.x != 0) return;
    const float v = (float)s;
    if (j < n0) { o0[j] = v; if (alias0) alias0[j] = v; }
    else if (j < n0 + n1) { o1[j - n0] = v; if (alias1) alias1[j - n0] = v; }
    else if (j < n0 + n1 + n2) { if (o2) o2[j - n0 - n1] = v; }
    else if (o3) o3[j - n0 - n1 - n2] = v;
}

// dy2 = rstd gamma (dz - s1 / n - xhat s2 / n); per-block column sums of dy2 into part[blk][C]; max |dy2|
template <int K>
__global__ __launch_bounds__(256) void bn_head_bwd_apply_kernel(BhArgs a) {
    __shared__ float sred[4 * 64 * 4];
    const int LPV = a.C / 4, VPB = 256 / LPV;
    const int c4 = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    BhBwd<K> q;
    q.init(a, c4);
    const float invM = 1.f / (float)a.rows;
    f32x4_t k1, k2, sc, col = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) { k1[j] = a.s1[c4 * 4 + j] * invM; k2[j] = a.s2[c4 * 4 + j] * invM; sc[j] = q.ga[j] * q.rs[j]; }
    float amax = 0.f;
    const long long stride = (long long)gridDim.x * VPB;
    for (long long v = (long long)blockIdx.x * VPB + vl; v < a.rows; v += 2 * stride) {
        const bool two = v + stride < a.rows;
        const float4 x0 = ldf4(a.y + v * a.ldy + c4 * 4);
        const float4 x1 = two ? ldf4(a.y + (v + stride) * a.ldy + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float d0[K], d1[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { d0[k] = a.dl[v * a.lddl + k]; d1[k] = two ? a.dl[(v + stride) * a.lddl + k] : 0.f; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            f32x4_t dz, xh, av;
            q.template eval<false>(a, u ? x1 : x0, u ? d1 : d0, dz, xh, av);
            f32x4_t od;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                od[j] = sc[j] * (dz[j] - k1[j] - xh[j] * k2[j]);               // norm_act_bwd_apply_kernel's expression
                amax = fmaxf(amax, fabsf(od[j]));
            }
            col += od;
            stf4(a.out + (v + u * stride) * a.ldo + c4 * 4, make_float4(od[0], od[1], od[2], od[3]));
        }
    }
    if (a.part) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = col[j];
            for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
            col[j] = s;
        }
        if (lane < (LPV < 64 ? LPV : 64)) st4(sred + (wave * 64 + lane) * 4, col);
        __syncthreads();
        for (int c = threadIdx.x; c < a.C; c += 256) {
            float s = 0.f;
            for (int w = 0; w < 4; ++w) s += sred[(w * 64 + (c / 4) % 64) * 4 + c % 4];
            a.part[(long long)blockIdx.x * a.C + c] = s;
        }
    }
    if (a.amax) block_amax_commit(amax, a.amax);
}

int bh_grid(long long rows, int vpb) {
    long long b = (rows + vpb - 1) / vpb;
    b = (b + 15) / 16;                                  // >= 8 two-voxel trips per block
    return (int)(b < 1 ? 1 : (b > kBhMaxBlocks ? kBhMaxBlocks : b));
}

bool bh_ok(long long rows, int C, int K, int ldy) {
    return rows > 0 && C >= 4 && C <= 256 && (C & (C - 1)) == 0 && K >= 1 && K <= 4 && (ldy % 4) == 0;
}

}  // namespace

}  // namespace seg

using namespace seg;

#define BH_DISPATCH(kern, K, grid, st, args)                                                                  \
    switch (K) {                                                                                              \
        case 1: hipLaunchKernelGGL((kern<1>), dim3(grid), dim3(256), 0, st, args); break;                     \
        case 2: hipLaunchKernelGGL((kern<2>), dim3(grid), dim3(256), 0, st, args); break;                     \
        case 3: hipLaunchKernelGGL((kern<3>), dim3(grid), dim3(256), 0, st, args); break;                     \
        default: hipLaunchKernelGGL((kern<4>), dim3(grid), dim3(256), 0, st, args); break;                    \
    }

extern "C" {

int mi355seg_bn_act_head_supported_f32(long long rows, int C, int K, int ldy) { return bh_ok(rows, C, K, ldy) ? 1 : 0; }

size_t mi355seg_bn_act_head_ws_bytes(int C, int K) {
    return align_up((size_t)kBhMaxBlocks * ((size_t)(2 + K) * C + K) * sizeof(float), 256) + 1024;
}

int mi355seg_bn_act_head_fwd_f32(const float* y, int ldy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 int act, float slope, const float* wh, const float* bh, float* logits, int ldl,
                                 long long rows, int C, int K, void* stream) {
    SEG_CHECK_ARG(y && mean && rstd && wh && logits && bh_ok(rows, C, K, ldy) && ldl >= K && act >= 0 && act <= 4, "bn_act_head_fwd: bad arguments");
    SEG_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)wh % 16) == 0, "bn_act_head_fwd: y and the head weights must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    BhArgs a{y, ldy, mean, rstd, gamma, beta, wh, bh, nullptr, 0, logits, ldl, nullptr, nullptr, nullptr, nullptr, rows, C, act, slope};
    const int grid = bh_grid(rows, 256 / (C / 4)) * 2;
    ProfScope ps(PF_NORM, 2.0 * rows * C * K, 4.0 * rows * (C + K), st);
    BH_DISPATCH(bn_head_fwd_kernel, K, grid, st, a);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

/* One pass over (y, dlogits): s1[c] = sum dz, s2[c] = sum dz * xhat (dgamma = s2, dbeta = s1 when given), dwh[k][c], dbh[k]. */
int mi355seg_bn_act_head_bwd_sums_f32(const float* dlogits, int lddl, const float* y, int ldy, const float* mean, const float* rstd,
                                      const float* gamma, const float* beta, int act, float slope, const float* wh,
                                      float* s1, float* s2, float* dgamma, float* dbeta, float* dwh, float* dbh,
                                      long long rows, int C, int K, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(dlogits && y && mean && rstd && wh && s1 && s2 && bh_ok(rows, C, K, ldy) && lddl >= K && act >= 0 && act <= 4, "bn_act_head_bwd_sums: bad arguments");
    SEG_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)wh % 16) == 0, "bn_act_head_bwd_sums: y and the head weights must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int grid = bh_grid(rows, 256 / (C / 4));
    const int ncol = (2 + K) * C + K;
    SEG_CHECK_WS((size_t)grid * ncol * sizeof(float), ws_bytes);
    BhArgs a{y, ldy, mean, rstd, gamma, beta, wh, nullptr, dlogits, lddl, nullptr, 0, nullptr, nullptr, (float*)ws, nullptr, rows, C, act, slope};
    {
        ProfScope ps(PF_NORM, 2.0 * rows * C * K * 2, 4.0 * rows * (C + K), st);
        BH_DISPATCH(bn_head_bwd_sums_kernel, K, grid, st, a);
        SEG_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(bh_finalize_kernel, dim3(ncol), dim3(64), 0, st, (const float*)ws, grid, ncol, s1, C, s2, C, dwh, K * C, dbh, dbeta, dgamma);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

/* dy = rstd gamma (dz - s1 / rows - xhat s2 / rows); dy_colsum[c] = sum_rows dy (the bias gradient of the convolution in front; may be
 * NULL); dy_amax: max |dy| max-combined into a zeroed device scalar (may be NULL). */
int mi355seg_bn_act_head_bwd_apply_f32(const float* dlogits, int lddl, const float* y, int ldy, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, int act, float slope, const float* wh,
                                       const float* s1, const float* s2, float* dy, int lddy, float* dy_colsum, float* dy_amax,
                                       long long rows, int C, int K, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(dlogits && y && mean && rstd && wh && s1 && s2 && dy && bh_ok(rows, C, K, ldy) && lddl >= K && (lddy % 4) == 0 && act >= 0 && act <= 4,
                  "bn_act_head_bwd_apply: bad arguments");
    SEG_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)wh % 16) == 0 && ((uintptr_t)dy % 16) == 0, "bn_act_head_bwd_apply: y, dy and the head weights must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int grid = bh_grid(rows, 256 / (C / 4)) * 2;
    if (dy_colsum) SEG_CHECK_WS((size_t)grid * C * sizeof(float), ws_bytes);
    BhArgs a{y, ldy, mean, rstd, gamma, beta, wh, nullptr, dlogits, lddl, dy, lddy, s1, s2, dy_colsum ? (float*)ws : nullptr,
             (unsigned*)dy_amax, rows, C, act, slope};
    {
        ProfScope ps(PF_NORM, 2.0 * rows * C * K, 4.0 * rows * (2.0 * C + K), st);
        BH_DISPATCH(bn_head_bwd_apply_kernel, K, grid, st, a);
        SEG_CHECK_LAUNCH();
    }
    if (dy_colsum) {
        hipLaunchKernelGGL(bh_finalize_kernel, dim3(C), dim3(64), 0, st, (const float*)ws, grid, C, dy_colsum, C, (float*)nullptr, 0, (float*)nullptr, 0,
                           (float*)nullptr, (float*)nullptr, (float*)nullptr);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

}  // extern "C"
