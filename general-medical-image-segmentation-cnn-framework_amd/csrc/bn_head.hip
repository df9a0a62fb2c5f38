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
template <int K, int ACT>
__device__ __forceinline__ void bn_head_fwd_body(const BhArgs& a) {
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
            o[0] = act_apply(fmaf(x.x, al[0], be[0]), (ACT >= 0 ? ACT : a.act), a.slope); o[1] = act_apply(fmaf(x.y, al[1], be[1]), (ACT >= 0 ? ACT : a.act), a.slope);
            o[2] = act_apply(fmaf(x.z, al[2], be[2]), (ACT >= 0 ? ACT : a.act), a.slope); o[3] = act_apply(fmaf(x.w, al[3], be[3]), (ACT >= 0 ? ACT : a.act), a.slope);
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
// (r6) the training launches of the U-Net run ReLU: that instantiation has no per-element activation switch
template <int K>
__global__ __launch_bounds__(256) void bn_head_fwd_kernel(BhArgs a) {
    if (a.act == MI355SEG_ACT_RELU) bn_head_fwd_body<K, MI355SEG_ACT_RELU>(a); else bn_head_fwd_body<K, -1>(a);
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
    template <bool WANT_A, int ACT>
    __device__ __forceinline__ void eval(const BhArgs& a, const float4 x, const float (&dl)[K], f32x4_t& dz, f32x4_t& xh, f32x4_t& av) const {
        f32x4_t da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < K; ++k) da += dl[k] * wr[k];                       // head_dgrad_kernel's accumulation order
        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh[j] = (xv[j] - m[j]) * rs[j];
            const float z = fmaf(xh[j], ga[j], be[j]);
            dz[j] = da[j] * act_grad(z, (ACT >= 0 ? ACT : a.act), a.slope);
            if (WANT_A) av[j] = act_apply(fmaf(xv[j], al[j], bf[j]), (ACT >= 0 ? ACT : a.act), a.slope);   // the forward's a2
        }
    }
};

// partials of one block: part[blk][ (2 + K) * C + K ] = s1[C] | s2[C] | dWh[K][C] | dbh[K]
template <int K, int ACT>
__device__ __forceinline__ void bn_head_bwd_sums_body(const BhArgs& a) {
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
            q.template eval<true, ACT>(a, u ? x1 : x0, u ? d1 : d0, dz, xh, av);
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
// (r6) the training launches of the U-Net run ReLU: that instantiation has no per-element activation switch
template <int K>
__global__ __launch_bounds__(256) void bn_head_bwd_sums_kernel(BhArgs a) {
    if (a.act == MI355SEG_ACT_RELU) bn_head_bwd_sums_body<K, MI355SEG_ACT_RELU>(a); else bn_head_bwd_sums_body<K, -1>(a);
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
template <int K, int ACT>
__device__ __forceinline__ void bn_head_bwd_apply_body(const BhArgs& a) {
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
            q.template eval<false, ACT>(a, u ? x1 : x0, u ? d1 : d0, dz, xh, av);
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
// (r6) the training launches of the U-Net run ReLU: that instantiation has no per-element activation switch
template <int K>
__global__ __launch_bounds__(256) void bn_head_bwd_apply_kernel(BhArgs a) {
    if (a.act == MI355SEG_ACT_RELU) bn_head_bwd_apply_body<K, MI355SEG_ACT_RELU>(a); else bn_head_bwd_apply_body<K, -1>(a);
}

// ---------------------------------------------------------------- BatchNorm + activation + MaxPool3d(2, 2) of an ENCODER block
// unet3d.py:19-25,51-58: ``pool1(enc1)`` where enc1 = relu2(norm2(.)) also feeds the skip concatenation.  Forward: thread = (pooled voxel,
// channel quad) normalises its eight children, writes them into the skip tensor (a channel slice of the level's concat buffer) AND
// their maximum + 3-bit argmax code: the activation is not re-read by a pooling pass.  Backward: d(act) = d(skip) + scatter(d(pooled))
// is formed on the fly in both passes of the norm backward (no pool-backward pass, no d(act) tensor).
struct BpArgs {
    const float* y; int ldy;                        // pre-norm tensor [N, D, H, W, C]
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    float* a; int lda;                              // forward: activation out (skip)
    float* p; uint8_t* idx;                         // forward: pooled out [N, D/2, H/2, W/2, C] + codes
    const float* ds; int ldds;                      // backward: d(skip)
    const float* dp;                                // backward: d(pooled), pitch C
    const uint8_t* cidx;
    float* dy; int lddy;                            // apply: d(pre-norm)
    const float* s1; const float* s2;
    float* part; unsigned* amax;
    int N, D, H, W, C, act; float slope;
};

__device__ __forceinline__ long long bp_child(const BpArgs& a, int n, int od, int oh, int ow, int t) {
    return (((long long)n * a.D + 2 * od + (t >> 2)) * a.H + 2 * oh + ((t >> 1) & 1)) * a.W + 2 * ow + (t & 1);
}

template <int ACT>
__device__ __forceinline__ void bn_pool_fwd_body(const BpArgs& a) {
    const int LPV = a.C / 4, VPB = 256 / LPV;
    const int c4 = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    const int Do = a.D / 2, Ho = a.H / 2, Wo = a.W / 2;
    const long long npool = (long long)a.N * Do * Ho * Wo;
    f32x4_t al, be;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        al[j] = a.rstd[c4 * 4 + j] * (a.gamma ? a.gamma[c4 * 4 + j] : 1.f);
        be[j] = (a.beta ? a.beta[c4 * 4 + j] : 0.f) - a.mean[c4 * 4 + j] * al[j];
    }
    float amax = 0.f;
    for (long long pv = (long long)blockIdx.x * VPB + vl; pv < npool; pv += (long long)gridDim.x * VPB) {
        const int ow = (int)(pv % Wo); long long r = pv / Wo;
        const int oh = (int)(r % Ho); r /= Ho;
        const int od = (int)(r % Do); const int n = (int)(r / Do);
        float4 x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = ldf4(a.y + bp_child(a, n, od, oh, ow, t) * a.ldy + c4 * 4);
        float best[4]; int code[4];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float o[4];
            o[0] = act_apply(fmaf(x[t].x, al[0], be[0]), (ACT >= 0 ? ACT : a.act), a.slope); o[1] = act_apply(fmaf(x[t].y, al[1], be[1]), (ACT >= 0 ? ACT : a.act), a.slope);
            o[2] = act_apply(fmaf(x[t].z, al[2], be[2]), (ACT >= 0 ? ACT : a.act), a.slope); o[3] = act_apply(fmaf(x[t].w, al[3], be[3]), (ACT >= 0 ? ACT : a.act), a.slope);
            stf4(a.a + bp_child(a, n, od, oh, ow, t) * a.lda + c4 * 4, make_float4(o[0], o[1], o[2], o[3]));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                amax = fmaxf(amax, fabsf(o[j]));
                if (t == 0 || o[j] > best[j] || o[j] != o[j]) { best[j] = o[j]; code[j] = t; }       // maxpool2_fwd_kernel's rule (PyTorch's)
            }
        }
        stf4(a.p + pv * a.C + c4 * 4, make_float4(best[0], best[1], best[2], best[3]));
        *reinterpret_cast<uchar4*>(a.idx + pv * a.C + c4 * 4) = make_uchar4((uint8_t)code[0], (uint8_t)code[1], (uint8_t)code[2], (uint8_t)code[3]);
    }
    if (a.amax) block_amax_commit(amax, a.amax);
}
// (r6) the training launches of the U-Net run ReLU: that instantiation has no per-element activation switch
__global__ __launch_bounds__(256) void bn_pool_fwd_kernel(BpArgs a) {
    if (a.act == MI355SEG_ACT_RELU) bn_pool_fwd_body<MI355SEG_ACT_RELU>(a); else bn_pool_fwd_body<-1>(a);
}

// APPLY = false: per-block partials part[blk][2 C] = s1 | s2;  APPLY = true: dy written, part[blk][C] = column sums of dy, max |dy|
template <bool APPLY, int ACT>
__device__ __forceinline__ void bn_pool_bwd_body(const BpArgs& a) {
    __shared__ float sred[4 * 64 * 8];
    const int LPV = a.C / 4, VPB = 256 / LPV;
    const int c4 = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    const int Do = a.D / 2, Ho = a.H / 2, Wo = a.W / 2;
    const long long npool = (long long)a.N * Do * Ho * Wo;
    f32x4_t m, rs, ga, be, k1 = {0.f, 0.f, 0.f, 0.f}, k2 = k1, sc = k1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = a.mean[c4 * 4 + j]; rs[j] = a.rstd[c4 * 4 + j];
        ga[j] = a.gamma ? a.gamma[c4 * 4 + j] : 1.f; be[j] = a.beta ? a.beta[c4 * 4 + j] : 0.f;
    }
    if (APPLY) {
        const float invM = 1.f / (float)((long long)a.N * a.D * a.H * a.W);
#pragma unroll
        for (int j = 0; j < 4; ++j) { k1[j] = a.s1[c4 * 4 + j] * invM; k2[j] = a.s2[c4 * 4 + j] * invM; sc[j] = ga[j] * rs[j]; }
    }
    f32x4_t acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = acc1;
    float amax = 0.f;
    for (long long pv = (long long)blockIdx.x * VPB + vl; pv < npool; pv += (long long)gridDim.x * VPB) {
        const int ow = (int)(pv % Wo); long long r = pv / Wo;
        const int oh = (int)(r % Ho); r /= Ho;
        const int od = (int)(r % Do); const int n = (int)(r / Do);
        const float4 g = ldf4(a.dp + pv * a.C + c4 * 4);
        const uchar4 cd = *reinterpret_cast<const uchar4*>(a.cidx + pv * a.C + c4 * 4);
        float4 x[8], d[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const long long v = bp_child(a, n, od, oh, ow, t);
            x[t] = ldf4(a.y + v * a.ldy + c4 * 4);
            d[t] = ldf4(a.ds + v * a.ldds + c4 * 4);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            // maxpool2_bwd_kernel's merge: the pooled gradient goes to the arg-max child, then the skip gradient is added
            const float dv[4] = {(cd.x == t ? g.x : 0.f) + d[t].x, (cd.y == t ? g.y : 0.f) + d[t].y, (cd.z == t ? g.z : 0.f) + d[t].z, (cd.w == t ? g.w : 0.f) + d[t].w};
            const float xv[4] = {x[t].x, x[t].y, x[t].z, x[t].w};
            f32x4_t od4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (xv[j] - m[j]) * rs[j];
                const float dz = dv[j] * act_grad(fmaf(xh, ga[j], be[j]), (ACT >= 0 ? ACT : a.act), a.slope);
                if (APPLY) {
                    od4[j] = sc[j] * (dz - k1[j] - xh * k2[j]);
                    acc1[j] += od4[j];
                    amax = fmaxf(amax, fabsf(od4[j]));
                } else {
                    acc1[j] += dz; acc2[j] = fmaf(dz, xh, acc2[j]);
                }
            }
            if (APPLY) stf4(a.dy + bp_child(a, n, od, oh, ow, t) * a.lddy + c4 * 4, make_float4(od4[0], od4[1], od4[2], od4[3]));
        }
    }
    if (a.part) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float vals[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { vals[j] = acc1[j]; vals[4 + j] = acc2[j]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float s = vals[i];
            for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
            vals[i] = s;
        }
        if (lane < (LPV < 64 ? LPV : 64)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sred[(wave * 64 + lane) * 8 + i] = vals[i];
        }
        __syncthreads();
        const int ncol = APPLY ? a.C : 2 * a.C;
        for (int i = threadIdx.x; i < ncol; i += 256) {
            const int which = i / a.C, c = i % a.C;
            float s = 0.f;
            for (int w = 0; w < 4; ++w) s += sred[(w * 64 + (c / 4) % 64) * 8 + which * 4 + c % 4];
            a.part[(long long)blockIdx.x * ncol + i] = s;
        }
    }
    if (APPLY && a.amax) block_amax_commit(amax, a.amax);
}
// (r6) the training launches of the U-Net run ReLU: that instantiation has no per-element activation switch
template <bool APPLY>
__global__ __launch_bounds__(256) void bn_pool_bwd_kernel(BpArgs a) {
    if (a.act == MI355SEG_ACT_RELU) bn_pool_bwd_body<APPLY, MI355SEG_ACT_RELU>(a); else bn_pool_bwd_body<APPLY, -1>(a);
}

bool bp_ok(int N, int D, int H, int W, int C, int ldy) {
    return N > 0 && D >= 2 && H >= 2 && W >= 2 && (D % 2) == 0 && (H % 2) == 0 && (W % 2) == 0 && C >= 4 && C <= 256 && (C & (C - 1)) == 0 && (ldy % 4) == 0;
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

int mi355seg_bn_act_pool_supported_f32(int N, int D, int H, int W, int C, int ldy) { return bp_ok(N, D, H, W, C, ldy) ? 1 : 0; }

size_t mi355seg_bn_act_pool_ws_bytes(int C) { return align_up((size_t)2 * kBhMaxBlocks * 2 * C * sizeof(float), 256) + 1024; }

int mi355seg_bn_act_pool_fwd_f32(const float* y, int ldy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 int act, float slope, float* a, int lda, float* pooled, unsigned char* idx, float* a_amax,
                                 int N, int D, int H, int W, int C, void* stream) {
    SEG_CHECK_ARG(y && mean && rstd && a && pooled && idx && bp_ok(N, D, H, W, C, ldy) && (lda % 4) == 0 && act >= 0 && act <= 4, "bn_act_pool_fwd: bad arguments");
    SEG_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)pooled % 16) == 0 && ((uintptr_t)idx % 4) == 0, "bn_act_pool_fwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    BpArgs b{};
    b.y = y; b.ldy = ldy; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.beta = beta; b.a = a; b.lda = lda; b.p = pooled; b.idx = idx;
    b.amax = (unsigned*)a_amax; b.N = N; b.D = D; b.H = H; b.W = W; b.C = C; b.act = act; b.slope = slope;
    const long long npool = (long long)N * (D / 2) * (H / 2) * (W / 2);
    const int vpb = 256 / (C / 4);
    long long nb = (npool + vpb - 1) / vpb;
    nb = (nb + 3) / 4;
    const int grid = (int)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb));
    ProfScope ps(PF_NORM, 0.0, 4.0 * npool * C * (16.0 + 1.25), st);
    hipLaunchKernelGGL(bn_pool_fwd_kernel, dim3(grid), dim3(256), 0, st, b);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

/* The norm backward of that block with d(act) = dskip + maxpool_backward(dpooled, idx) formed on the fly: s1 / s2 (dgamma = s2,
 * dbeta = s1), then dy = rstd gamma (dz - s1 / rows - xhat s2 / rows) with its column sums (dy_colsum, may be NULL) and maximum
 * (dy_amax, may be NULL).  ws: mi355seg_bn_act_pool_ws_bytes(C). */
int mi355seg_bn_act_pool_bwd_f32(const float* dskip, int ldds, const float* dpooled, const unsigned char* idx, const float* y, int ldy,
                                 const float* mean, const float* rstd, const float* gamma, const float* beta, int act, float slope,
                                 float* s1, float* s2, float* dgamma, float* dbeta, float* dy, int lddy, float* dy_colsum, float* dy_amax,
                                 int N, int D, int H, int W, int C, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(dskip && dpooled && idx && y && mean && rstd && s1 && s2 && dy && bp_ok(N, D, H, W, C, ldy) && (ldds % 4) == 0 && (lddy % 4) == 0 && act >= 0 && act <= 4,
                  "bn_act_pool_bwd: bad arguments");
    SEG_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)dskip % 16) == 0 && ((uintptr_t)dpooled % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)idx % 4) == 0,
                  "bn_act_pool_bwd: tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    BpArgs b{};
    b.y = y; b.ldy = ldy; b.mean = mean; b.rstd = rstd; b.gamma = gamma; b.beta = beta; b.ds = dskip; b.ldds = ldds; b.dp = dpooled; b.cidx = idx;
    b.dy = dy; b.lddy = lddy; b.N = N; b.D = D; b.H = H; b.W = W; b.C = C; b.act = act; b.slope = slope;
    const long long npool = (long long)N * (D / 2) * (H / 2) * (W / 2);
    const int vpb = 256 / (C / 4);
    long long nb = (npool + vpb - 1) / vpb;
    nb = (nb + 3) / 4;
    const int grid = (int)(nb < 1 ? 1 : (nb > 2 * kBhMaxBlocks ? 2 * kBhMaxBlocks : nb));
    SEG_CHECK_WS((size_t)grid * 2 * C * sizeof(float), ws_bytes);
    b.part = (float*)ws;
    {
        ProfScope ps(PF_NORM, 0.0, 4.0 * npool * C * (16.0 + 1.25), st);
        hipLaunchKernelGGL(bn_pool_bwd_kernel<false>, dim3(grid), dim3(256), 0, st, b);
        SEG_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(bh_finalize_kernel, dim3(2 * C), dim3(64), 0, st, (const float*)ws, grid, 2 * C, s1, C, s2, C, (float*)nullptr, 0, (float*)nullptr, dbeta, dgamma);
    SEG_CHECK_LAUNCH();
    b.s1 = s1; b.s2 = s2; b.amax = (unsigned*)dy_amax; b.part = dy_colsum ? (float*)ws : nullptr;
    {
        ProfScope ps(PF_NORM, 0.0, 4.0 * npool * C * (24.0 + 1.25), st);
        hipLaunchKernelGGL(bn_pool_bwd_kernel<true>, dim3(grid), dim3(256), 0, st, b);
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
