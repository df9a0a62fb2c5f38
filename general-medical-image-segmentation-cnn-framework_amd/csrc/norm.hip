// norm.hip -- BatchNorm3d(train) / InstanceNorm3d statistics, fused normalise +
// residual + activation, and their backward, on NDHWC fp32 tensors.  All HBM-bound:
// vectorised 16 B/lane accesses, per-channel reductions as per-thread register sums ->
// LDS -> per-block partials -> deterministic fp64 finalise (no atomics).
//
// Reference semantics: nn.BatchNorm3d training mode (unet3d.py:88,100; vnet3d.py:27,...)
// = biased variance for the output, unbiased for running_var, momentum 0.1, eps 1e-5;
// nn.InstanceNorm3d (residual_unet3d.py:27...) = the same per (n, c), no affine.
#include "common.h"
#include "internal.h"
#include <initializer_list>

namespace seg {

constexpr int kRedThreads = 256;
constexpr int kMaxRedBlocks = 1024;

struct RedPlan {
    bool vec;       // float4 path
    int lanes;      // threads across channels per row
    int rpi;        // rows per block-iteration
    int nblk;       // blocks per group
};

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static bool red_plan(long long rows, int C, int ld, RedPlan* p) {
    if (is_pow2(C) && C >= 4 && C <= 1024 && (ld % 4) == 0) {
        p->vec = true;
        p->lanes = C / 4;
    } else if (C <= kRedThreads) {
        p->vec = false;
        p->lanes = C;
    } else {
        return false;
    }
    p->rpi = kRedThreads / p->lanes;
    long long iters = (rows + p->rpi - 1) / p->rpi;
    long long nb = (iters + 7) / 8;
    p->nblk = (int)(nb < 1 ? 1 : (nb > kMaxRedBlocks ? kMaxRedBlocks : nb));
    return true;
}

// ---------------------------------------------------------------------------------------
// Generic two-sum column reduction.  F::eval(row, c, ...) returns the two addends of an
// element.  Partials: part[((g*nblk + b)*C + c)*2 + {0,1}].
// ---------------------------------------------------------------------------------------
// pivot of channel c in group g: the mean of 8 samples spread over the group's rows.  Sums are taken of
// (x - pivot), so var = E[(x-p)^2] - E[x-p]^2 does not cancel when |mean| >> std (nearly constant channels).
template <typename T>
__device__ __forceinline__ float stats_pivot(const T* __restrict__ x, int ldx, long long rbase, long long rows, int c) {
    float p = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) p += ld1(x + (rbase + (rows * k) / 8) * ldx + c);
    return p * 0.125f;
}

template <typename T>
struct StatsF {     // sum (x - pivot), sum (x - pivot)^2 ; pivots fetched once per thread
    const T* x; int ldx; long long rows;
    struct State { float p[4]; };
    __device__ __forceinline__ State prepC(int g, int c, int nj, int C) const { return prep(g, c, nj); }
    __device__ __forceinline__ State prep(int g, int c, int nj) const {
        State s;
        for (int j = 0; j < 4; ++j) s.p[j] = j < nj ? stats_pivot(x, ldx, (long long)g * rows, rows, c + j) : 0.f;
        return s;
    }
    __device__ __forceinline__ void eval(const State& st, long long r, int g, int c, int C, float& a, float& b) const {
        float v = ld1(x + r * ldx + c) - st.p[0];
        a = v; b = v * v;
    }
    __device__ __forceinline__ void eval4(const State& st, long long r, int g, int c, int C, float4& a, float4& b) const {
        float4 v = ldf4(x + r * ldx + c);
        v.x -= st.p[0]; v.y -= st.p[1]; v.z -= st.p[2]; v.w -= st.p[3];
        a = v; b = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
    }
};

template <typename T, int ACT = -1>      // ACT >= 0: the activation as a compile-time constant (else the run-time switch, per element)
struct BwdF {       // sum dz, sum dz*xhat  with dz = dy*act'(z), z = xhat*gamma+beta (+res)
    const T* dy; int lddy; const T* x; int ldx; const float* mean; const float* rstd;
    const float* gamma; const float* beta; const T* res; int ldres; int act; float slope;
    struct State { float m[4], rs[4], ga[4], be[4]; };
    __device__ __forceinline__ State prep(int g, int c, int nj) const { return State(); }
    __device__ __forceinline__ State prepC(int g, int c, int nj, int C) const {
        State s;
        for (int j = 0; j < 4; ++j) {
            const bool ok = j < nj;
            s.m[j] = ok ? mean[g * C + c + j] : 0.f; s.rs[j] = ok ? rstd[g * C + c + j] : 0.f;
            s.ga[j] = (ok && gamma) ? gamma[c + j] : 1.f; s.be[j] = (ok && beta) ? beta[c + j] : 0.f;
        }
        return s;
    }
    __device__ __forceinline__ void one(const State& st, int j, float dyv, float xv, float rv, float& a, float& b) const {
        float xh = (xv - st.m[j]) * st.rs[j];
        float z = fmaf(xh, st.ga[j], st.be[j]) + rv;
        float dz = dyv * act_grad(z, ACT >= 0 ? ACT : act, slope);
        a = dz; b = dz * xh;
    }
    __device__ __forceinline__ void eval(const State& st, long long r, int g, int c, int C, float& a, float& b) const {
        one(st, 0, ld1(dy + r * lddy + c), ld1(x + r * ldx + c), res ? ld1(res + r * ldres + c) : 0.f, a, b);
    }
    __device__ __forceinline__ void eval4(const State& st, long long r, int g, int c, int C, float4& a, float4& b) const {
        float4 d = ldf4(dy + r * lddy + c);
        float4 v = ldf4(x + r * ldx + c);
        float4 rr = res ? ldf4(res + r * ldres + c) : make_float4(0, 0, 0, 0);
        one(st, 0, d.x, v.x, rr.x, a.x, b.x);
        one(st, 1, d.y, v.y, rr.y, a.y, b.y);
        one(st, 2, d.z, v.z, rr.z, a.z, b.z);
        one(st, 3, d.w, v.w, rr.w, a.w, b.w);
    }
};

template <typename T>
struct PreluF {     // column sum of dy * min(z, 0), z = x (+ res): the gradient of a PReLU slope (second sum unused)
    const T* dy; int lddy; const T* x; int ldx; const T* res; int ldres;
    struct State { };
    __device__ __forceinline__ State prep(int g, int c, int nj) const { return State(); }
    __device__ __forceinline__ State prepC(int g, int c, int nj, int C) const { return State(); }
    __device__ __forceinline__ void eval(const State&, long long r, int g, int c, int C, float& a, float& b) const {
        const float z = ld1(x + r * ldx + c) + (res ? ld1(res + r * ldres + c) : 0.f);
        a = ld1(dy + r * lddy + c) * fminf(z, 0.f); b = 0.f;
    }
    __device__ __forceinline__ void eval4(const State&, long long r, int g, int c, int C, float4& a, float4& b) const {
        const float4 d = ldf4(dy + r * lddy + c);
        float4 v = ldf4(x + r * ldx + c);
        if (res) { const float4 q = ldf4(res + r * ldres + c); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        a = make_float4(d.x * fminf(v.x, 0.f), d.y * fminf(v.y, 0.f), d.z * fminf(v.z, 0.f), d.w * fminf(v.w, 0.f));
        b = make_float4(0.f, 0.f, 0.f, 0.f);
    }
};

template <typename F, bool VEC>
__global__ __launch_bounds__(kRedThreads) void colreduce2_kernel(F f, long long rows, int C, int lanes, int rpi,
                                                                 float* __restrict__ part) {
    __shared__ float sh[kRedThreads * 8];
    const int t = threadIdx.x;
    const int g = blockIdx.y;
    const int nblk = gridDim.x;
    const int lane = t % lanes;       // channel slot
    const int rsub = t / lanes;       // row slot
    const bool active = rsub < rpi;
    const long long rbase = (long long)g * rows;
    if (VEC) {
        float4 sa = make_float4(0, 0, 0, 0), sb = sa;
        if (active) {
            const typename F::State fst = f.prepC(g, lane * 4, 4, C);
            // two rows per trip: both rows' loads are issued before either is reduced (twice the bytes in flight per wave)
            const long long stride = (long long)nblk * rpi;
            long long r = (long long)blockIdx.x * rpi + rsub;
            for (; r + stride < rows; r += 2 * stride) {
                float4 a0, b0, a1, b1;
                f.eval4(fst, rbase + r, g, lane * 4, C, a0, b0);
                f.eval4(fst, rbase + r + stride, g, lane * 4, C, a1, b1);
                sa.x += a0.x; sa.y += a0.y; sa.z += a0.z; sa.w += a0.w;
                sb.x += b0.x; sb.y += b0.y; sb.z += b0.z; sb.w += b0.w;
                sa.x += a1.x; sa.y += a1.y; sa.z += a1.z; sa.w += a1.w;
                sb.x += b1.x; sb.y += b1.y; sb.z += b1.z; sb.w += b1.w;
            }
            if (r < rows) {
                float4 a, b;
                f.eval4(fst, rbase + r, g, lane * 4, C, a, b);
                sa.x += a.x; sa.y += a.y; sa.z += a.z; sa.w += a.w;
                sb.x += b.x; sb.y += b.y; sb.z += b.z; sb.w += b.w;
            }
        }
        float* s = sh + t * 8;
        s[0] = sa.x; s[1] = sa.y; s[2] = sa.z; s[3] = sa.w;
        s[4] = sb.x; s[5] = sb.y; s[6] = sb.z; s[7] = sb.w;
        __syncthreads();
        if (t < lanes) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int q = 0; q < rpi; ++q) {
                const float* o = sh + (q * lanes + t) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += o[j];
            }
            float* dst = part + (((long long)g * nblk + blockIdx.x) * C + t * 4) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) { dst[j * 2] = acc[j]; dst[j * 2 + 1] = acc[4 + j]; }
        }
    } else {
        float sa = 0.f, sb = 0.f;
        if (active) {
            const typename F::State fst = f.prepC(g, lane, 1, C);
            for (long long r = (long long)blockIdx.x * rpi + rsub; r < rows; r += (long long)nblk * rpi) {
                float a, b;
                f.eval(fst, rbase + r, g, lane, C, a, b);
                sa += a; sb += b;
            }
        }
        sh[t * 2] = sa; sh[t * 2 + 1] = sb;
        __syncthreads();
        if (t < lanes) {
            float a = 0.f, b = 0.f;
            for (int q = 0; q < rpi; ++q) { a += sh[(q * lanes + t) * 2]; b += sh[(q * lanes + t) * 2 + 1]; }
            float* dst = part + (((long long)g * nblk + blockIdx.x) * C + t) * 2;
            dst[0] = a; dst[1] = b;
        }
    }
}

template <typename F>
static int launch_colreduce2(const F& f, long long rows, int groups, int C, int ld_for_plan, float* part,
                             RedPlan* plan_out, hipStream_t st) {
    RedPlan p;
    if (!red_plan(rows, C, ld_for_plan, &p)) {
        set_error("per-channel reduction: unsupported channel count C=%d", C);
        return MI355SEG_EINVAL;
    }
    dim3 grid(p.nblk, groups);
    if (p.vec)
        hipLaunchKernelGGL((colreduce2_kernel<F, true>), grid, dim3(kRedThreads), 0, st, f, rows, C, p.lanes, p.rpi, part);
    else
        hipLaunchKernelGGL((colreduce2_kernel<F, false>), grid, dim3(kRedThreads), 0, st, f, rows, C, p.lanes, p.rpi, part);
    *plan_out = p;
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// finalise: one wavefront per (group, channel); lanes stride over the per-block partials, fp64
// shuffle reduction -> fixed summation order (deterministic).
template <typename T>
__global__ __launch_bounds__(64) void stats_finalize_kernel(const float* __restrict__ part, int nblk, int C, int groups, double rows,
                                      float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                      float* __restrict__ rmean, float* __restrict__ rvar, float momentum,
                                      const T* __restrict__ x, int ldx) {
    const int i = blockIdx.x;
    const int g = i / C, c = i % C;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) {
        const float* p = part + (((long long)g * nblk + b) * C + c) * 2;
        s += (double)p[0]; q += (double)p[1];
    }
    s = wave_sum(s); q = wave_sum(q);
    if (threadIdx.x != 0) return;
    double m = s / rows;                              // mean of (x - pivot)
    double var = q / rows - m * m;
    if (var < 0.0) var = 0.0;
    m += (double)stats_pivot(x, ldx, (long long)g * (long long)rows, (long long)rows, c);
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        double unb = rows > 1.0 ? var * rows / (rows - 1.0) : var;
        rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
        rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
    }
}

__global__ void stats_from_sums_kernel(const double* __restrict__ sum, const double* __restrict__ sq, int C, double rows,
                                       float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                       float* __restrict__ rmean, float* __restrict__ rvar, float momentum) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double m = sum[c] / rows;
    double var = sq[c] / rows - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        double unb = rows > 1.0 ? var * rows / (rows - 1.0) : var;
        rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
        rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
    }
}

__global__ __launch_bounds__(64) void bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, int groups,
                                    float* __restrict__ s1, float* __restrict__ s2,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    // s1/s2: per (group, channel) sums of dz and dz*xhat; one wavefront per (group, channel).
    const int i = blockIdx.x;
    const int g = i / C, c = i % C;
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) {
        const float* p = part + (((long long)g * nblk + k) * C + c) * 2;
        a += (double)p[0]; b += (double)p[1];
    }
    a = wave_sum(a); b = wave_sum(b);
    if (threadIdx.x != 0) return;
    s1[i] = (float)a; s2[i] = (float)b;
    if (dgamma && groups == 1) { dgamma[c] = (float)b; dbeta[c] = (float)a; }
}

// ---------------------------------------------------------------------------------------
// elementwise kernels.  A thread owns one fixed channel quad (or channel) and walks rows:
// per-channel constants are hoisted out of the loop, there is no per-element integer
// division, and a wavefront touches 64 consecutive 16-byte pieces of the NDHWC stream.
// grid = (row blocks, groups).
// ---------------------------------------------------------------------------------------
struct RowMap {
    int lanes, rpi;     // threads across channels, rows per block iteration
};

template <typename T, bool VEC, int ACT = -1>
__global__ __launch_bounds__(256) void norm_act_fwd_kernel(const T* __restrict__ x, int ldx,
        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
        const float* __restrict__ beta, const T* __restrict__ res, int ldres, T* __restrict__ y, int ldy,
        long long rows, int C, int lanes, int rpi, int act, float slope, unsigned* __restrict__ amax_out) {
    const int ACTV = ACT >= 0 ? ACT : act;            // (a compile-time constant in the per-activation instantiations: the per-element switch folds)
    constexpr int NJ = VEC ? 4 : 1;
    const int g = blockIdx.y;
    const int cw = VEC ? C / 4 : C;
    const int rsub = threadIdx.x / lanes;
    const bool active = rsub < rpi;
    if (!active && !amax_out) return;
    const long long rbase = (long long)g * rows;
    float amax = 0.f;                   // max |y| of this thread's outputs (amax_out: the f16x3 scale of the convolution that reads y)
    for (int cc = threadIdx.x % lanes; active && cc < cw; cc += lanes) {
        const int c = cc * NJ;
        float al[NJ], be[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            al[j] = rstd[g * C + c + j] * (gamma ? gamma[c + j] : 1.f);
            be[j] = (beta ? beta[c + j] : 0.f) - mean[g * C + c + j] * al[j];
        }
        if (VEC) {          // two adjacent rows per trip: both rows' loads are issued before either is normalised
            for (long long r = 2 * ((long long)blockIdx.x * rpi + rsub); r < rows; r += 2 * (long long)gridDim.x * rpi) {
                const int nr = r + 1 < rows ? 2 : 1;
                float4 v[2], rr[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (u < nr) {
                        v[u] = ldf4(x + (rbase + r + u) * ldx + c);
                        rr[u] = res ? ldf4(res + (rbase + r + u) * ldres + c) : make_float4(0, 0, 0, 0);
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (u < nr) {
                        float4 o;
                        o.x = act_apply(fmaf(v[u].x, al[0], be[0]) + rr[u].x, ACTV, slope);
                        o.y = act_apply(fmaf(v[u].y, al[NJ > 1 ? 1 : 0], be[NJ > 1 ? 1 : 0]) + rr[u].y, ACTV, slope);
                        o.z = act_apply(fmaf(v[u].z, al[NJ > 1 ? 2 : 0], be[NJ > 1 ? 2 : 0]) + rr[u].z, ACTV, slope);
                        o.w = act_apply(fmaf(v[u].w, al[NJ > 1 ? 3 : 0], be[NJ > 1 ? 3 : 0]) + rr[u].w, ACTV, slope);
                        stf4(y + (rbase + r + u) * ldy + c, o);
                        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
                    }
            }
            continue;
        }
        for (long long r = (long long)blockIdx.x * rpi + rsub; r < rows; r += (long long)gridDim.x * rpi) {
            const long long row = rbase + r;
            {
                const float rv = res ? ld1(res + row * ldres + c) : 0.f;
                const float o = act_apply(fmaf(ld1(x + row * ldx + c), al[0], be[0]) + rv, ACTV, slope);
                st1(y + row * ldy + c, o);
                amax = fmaxf(amax, fabsf(o));
            }
        }
    }
    if (amax_out) block_amax_commit(amax, amax_out);
}

// eight channels per thread: 16-byte accesses on bf16 tensors (launched for bf16 only, see norm_api.inc)
template <typename T, int ACT = -1>
__global__ __launch_bounds__(256) void norm_act_fwd8_kernel(const T* __restrict__ x, int ldx,
        const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
        const float* __restrict__ beta, const T* __restrict__ res, int ldres, T* __restrict__ y, int ldy,
        long long rows, int C, int lanes, int rpi, int act, float slope) {
    const int ACTV = ACT >= 0 ? ACT : act;            // (a compile-time constant in the per-activation instantiations: the per-element switch folds)
    const int g = blockIdx.y;
    const int cw = C / 8;
    const int rsub = threadIdx.x / lanes;
    if (rsub >= rpi) return;
    const long long rbase = (long long)g * rows;
    for (int cc = threadIdx.x % lanes; cc < cw; cc += lanes) {
        const int c = cc * 8;
        float al[8], be[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            al[j] = rstd[g * C + c + j] * (gamma ? gamma[c + j] : 1.f);
            be[j] = (beta ? beta[c + j] : 0.f) - mean[g * C + c + j] * al[j];
        }
        for (long long r = (long long)blockIdx.x * rpi + rsub; r < rows; r += (long long)gridDim.x * rpi) {
            const long long row = rbase + r;
            float v[8], rr[8], o[8];
            ld8(x + row * ldx + c, v);
            if (res) ld8(res + row * ldres + c, rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = act_apply(fmaf(v[j], al[j], be[j]) + (res ? rr[j] : 0.f), ACTV, slope);
            st8(y + row * ldy + c, o);
        }
    }
}

template <typename T, bool VEC, int ACT = -1>
__global__ __launch_bounds__(256) void norm_act_bwd_apply_kernel(const T* __restrict__ dy, int lddy,
        const T* __restrict__ x, int ldx, const float* __restrict__ mean, const float* __restrict__ rstd,
        const float* __restrict__ gamma, const float* __restrict__ beta, const T* __restrict__ res, int ldres,
        const float* __restrict__ s1, const float* __restrict__ s2, T* __restrict__ dx, int lddx,
        T* __restrict__ dres, int lddres, long long rows, int C, int lanes, int rpi, int act, float slope,
        float* __restrict__ dxpart, unsigned* __restrict__ amax_out) {
    const int ACTV = ACT >= 0 ? ACT : act;            // (a compile-time constant in the per-activation instantiations: the per-element switch folds)
    constexpr int NJ = VEC ? 4 : 1;
    __shared__ float shs[256 * 4];
    const int g = blockIdx.y;
    const int cw = VEC ? C / 4 : C;
    const int rsub = threadIdx.x / lanes;
    const bool active = rsub < rpi;
    if (!active && !dxpart && !amax_out) return;
    float amax = 0.f;                   // max |dx| of this thread's outputs
    const long long rbase = (long long)g * rows;
    const float invM = 1.f / (float)rows;
    float colsum[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) colsum[j] = 0.f;
    for (int cc = threadIdx.x % lanes; active && cc < cw; cc += lanes) {
        const int c = cc * NJ;
        float m[NJ], rs[NJ], ga[NJ], be[NJ], k1[NJ], k2[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            m[j] = mean[g * C + c + j]; rs[j] = rstd[g * C + c + j];
            ga[j] = gamma ? gamma[c + j] : 1.f; be[j] = beta ? beta[c + j] : 0.f;
            k1[j] = s1[g * C + c + j] * invM; k2[j] = s2[g * C + c + j] * invM;
        }
        for (long long r = (long long)blockIdx.x * rpi + rsub; r < rows; r += (long long)gridDim.x * rpi) {
            const long long row = rbase + r;
            float dv[NJ], xv[NJ], rv[NJ], od[NJ], oz[NJ];
            if (VEC) {
                float4 d = ldf4(dy + row * lddy + c);
                float4 v = ldf4(x + row * ldx + c);
                dv[0] = d.x; xv[0] = v.x;
                if (NJ > 1) { dv[1] = d.y; dv[2] = d.z; dv[3] = d.w; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w; }
                float4 q = res ? ldf4(res + row * ldres + c) : make_float4(0, 0, 0, 0);
                rv[0] = q.x;
                if (NJ > 1) { rv[1] = q.y; rv[2] = q.z; rv[3] = q.w; }
            } else {
                dv[0] = ld1(dy + row * lddy + c); xv[0] = ld1(x + row * ldx + c);
                rv[0] = res ? ld1(res + row * ldres + c) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float xh = (xv[j] - m[j]) * rs[j];
                const float z = fmaf(xh, ga[j], be[j]) + rv[j];
                const float dz = dv[j] * act_grad(z, ACTV, slope);
                oz[j] = dz;
                od[j] = ga[j] * rs[j] * (dz - k1[j] - xh * k2[j]);
                colsum[j] += od[j];
                amax = fmaxf(amax, fabsf(od[j]));
            }
            if (VEC) {
                stf4(dx + row * lddx + c, make_float4(od[0], od[NJ > 1 ? 1 : 0], od[NJ > 1 ? 2 : 0], od[NJ > 1 ? 3 : 0]));
                if (dres) stf4(dres + row * lddres + c, make_float4(oz[0], oz[NJ > 1 ? 1 : 0], oz[NJ > 1 ? 2 : 0], oz[NJ > 1 ? 3 : 0]));
            } else {
                st1(dx + row * lddx + c, od[0]);
                if (dres) st1(dres + row * lddres + c, oz[0]);
            }
        }
    }
    if (dxpart) {       // per-block column sums of dx (= the bias gradient of the convolution feeding this norm);
                        // only used when lanes == cw (one channel slot per thread), see the launcher
#pragma unroll
        for (int j = 0; j < NJ; ++j) shs[threadIdx.x * NJ + j] = colsum[j];
        __syncthreads();
        if (threadIdx.x < lanes) {
            float acc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
            for (int q = 0; q < rpi; ++q)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] += shs[(q * lanes + threadIdx.x) * NJ + j];
            float* dst = dxpart + ((long long)blockIdx.x * C + threadIdx.x * NJ) * 2;
#pragma unroll
            for (int j = 0; j < NJ; ++j) { dst[j * 2] = acc[j]; dst[j * 2 + 1] = 0.f; }
        }
    }
    if (amax_out) block_amax_commit(amax, amax_out);
}

template <typename T, bool VEC, bool BWD, int ACT = -1>
__global__ __launch_bounds__(256) void act_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
        const T* __restrict__ res, int ldres, T* __restrict__ out, int ldo, long long rows, int C, int lanes, int rpi,
        int act, float slope, const float* __restrict__ slope_c) {
    const int ACTV = ACT >= 0 ? ACT : act;            // (a compile-time constant in the per-activation instantiations: the per-element switch folds)
    // slope_c != null: PReLU -- the leaky slope of channel c is slope_c[c] (act = LRELU)
    const int cw = VEC ? C / 4 : C;
    const int rsub = threadIdx.x / lanes;
    if (rsub >= rpi) return;
    for (int cc = threadIdx.x % lanes; cc < cw; cc += lanes) {
        const int c = cc * (VEC ? 4 : 1);
        float4 sl = make_float4(slope, slope, slope, slope);
        if (slope_c) { sl.x = slope_c[c]; if (VEC) { sl.y = slope_c[c + 1]; sl.z = slope_c[c + 2]; sl.w = slope_c[c + 3]; } }
        for (long long r = (long long)blockIdx.x * rpi + rsub; r < rows; r += (long long)gridDim.x * rpi) {
            if (VEC) {
                float4 v = ldf4(x + r * ldx + c);
                if (res) {
                    float4 q = ldf4(res + r * ldres + c);
                    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
                }
                float4 o;
                if (BWD) {
                    float4 d = ldf4(dy + r * lddy + c);
                    o = make_float4(d.x * act_grad(v.x, ACTV, sl.x), d.y * act_grad(v.y, ACTV, sl.y),
                                    d.z * act_grad(v.z, ACTV, sl.z), d.w * act_grad(v.w, ACTV, sl.w));
                } else {
                    o = make_float4(act_apply(v.x, ACTV, sl.x), act_apply(v.y, ACTV, sl.y),
                                    act_apply(v.z, ACTV, sl.z), act_apply(v.w, ACTV, sl.w));
                }
                stf4(out + r * ldo + c, o);
            } else {
                float v = ld1(x + r * ldx + c) + (res ? ld1(res + r * ldres + c) : 0.f);
                st1(out + r * ldo + c, BWD ? ld1(dy + r * lddy + c) * act_grad(v, ACTV, sl.x) : act_apply(v, ACTV, sl.x));
            }
        }
    }
}

// eight channels per thread (16-byte accesses on bf16 tensors, 32-byte on fp32), no per-channel slopes; BWD: out = addend + dy * act'(x + res)
// -- `addend` (optional) is a second gradient of the same tensor: a residual fork's sum d(x) = d_skip + d_act * act'(x) without a separate add
// pass (residual_unet3d.py:110-121: the level-1 tensor feeds a LeakyReLU and, unchanged, a later block sum)
template <typename T, bool BWD, int ACT = -1>
__global__ __launch_bounds__(256) void act8_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
        const T* __restrict__ res, int ldres, const T* __restrict__ addend, int ldadd, T* __restrict__ out, int ldo, long long rows, int C,
        int lanes, int rpi, int act, float slope) {
    const int ACTV = ACT >= 0 ? ACT : act;            // (a compile-time constant in the per-activation instantiations: the per-element switch folds)
    const int cw = C / 8;
    const int rsub = threadIdx.x / lanes;
    if (rsub >= rpi) return;
    for (int cc = threadIdx.x % lanes; cc < cw; cc += lanes) {
        const int c = cc * 8;
        for (long long r = (long long)blockIdx.x * rpi + rsub; r < rows; r += (long long)gridDim.x * rpi) {
            float v[8], q[8], d[8], ad[8], o[8];
            ld8(x + r * ldx + c, v);
            if (res) {
                ld8(res + r * ldres + c, q);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += q[j];
            }
            if (BWD) {
                ld8(dy + r * lddy + c, d);
                if (addend) ld8(addend + r * ldadd + c, ad);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (addend ? ad[j] : 0.f) + d[j] * act_grad(v[j], ACTV, slope);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = act_apply(v[j], ACTV, slope);
            }
            st8(out + r * ldo + c, o);
        }
    }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void scale_channels_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ scale,
        T* __restrict__ y, int ldy, long long rows, int C, int lanes, int rpi) {
    const int g = blockIdx.y;
    const int cw = VEC ? C / 4 : C;
    const int rsub = threadIdx.x / lanes;
    if (rsub >= rpi) return;
    const long long rbase = (long long)g * rows;
    for (int cc = threadIdx.x % lanes; cc < cw; cc += lanes) {
        const int c = cc * (VEC ? 4 : 1);
        float4 sc = make_float4(scale[g * C + c], 0, 0, 0);
        if (VEC) { sc.y = scale[g * C + c + 1]; sc.z = scale[g * C + c + 2]; sc.w = scale[g * C + c + 3]; }
        for (long long r = (long long)blockIdx.x * rpi + rsub; r < rows; r += (long long)gridDim.x * rpi) {
            const long long row = rbase + r;
            if (VEC) {
                float4 v = ldf4(x + row * ldx + c);
                stf4(y + row * ldy + c, make_float4(v.x * sc.x, v.y * sc.y, v.z * sc.z, v.w * sc.w));
            } else {
                st1(y + row * ldy + c, ld1(x + row * ldx + c) * sc.x);
            }
        }
    }
}

__global__ void rstd_from_var_kernel(const float* __restrict__ var, float eps, float* __restrict__ rstd, int C) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) rstd[c] = (float)(1.0 / sqrt((double)var[c] + (double)eps));
}

// row-block grid for the elementwise kernels: ~8 rows per thread, at most 8192 blocks
static int row_grid(long long rows, int rpi) {
    long long b = (rows + (long long)rpi * 8 - 1) / ((long long)rpi * 8);
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
static RowMap ew_map(int C, bool vec) {
    int cw = vec ? C / 4 : C;
    int lanes = cw > 256 ? 256 : cw;
    // lanes must divide 256 for a clean (row, lane) split; fall back to one row per iteration otherwise
    RowMap m;
    if (256 % lanes == 0) { m.lanes = lanes; m.rpi = 256 / lanes; }
    else { m.lanes = lanes; m.rpi = 1; }
    return m;
}

static bool vec8_ok(int C, std::initializer_list<int> lds) {
    if (C % 8) return false;
    for (int l : lds) if (l % 8) return false;
    return true;
}
static RowMap ew_map8(int C) {
    int cw = C / 8;
    int lanes = cw > 256 ? 256 : cw;
    RowMap m;
    if (256 % lanes == 0) { m.lanes = lanes; m.rpi = 256 / lanes; }
    else { m.lanes = lanes; m.rpi = 1; }
    return m;
}
static bool vec_ok(int C, std::initializer_list<int> lds) {
    if (C % 4) return false;
    for (int l : lds) if (l % 4) return false;
    return true;
}

// channel sums of a [rows, C] matrix as doubles (used for dbias and conv-epilogue stats
// on the generic path).  part must hold nblk*C*2 floats.
template <typename T>
struct SumF {        // sums of (x - pivot) and (x - pivot)^2; pivot = 0 when only the plain sum is wanted
    const T* x; int ldx; long long rows; int use_pivot;
    struct State { float p[4]; };
    __device__ __forceinline__ State prepC(int g, int c, int nj, int C) const {
        State s;
        for (int j = 0; j < 4; ++j) s.p[j] = (use_pivot && j < nj) ? stats_pivot(x, ldx, 0, rows, c + j) : 0.f;
        return s;
    }
    __device__ __forceinline__ void eval(const State& st, long long r, int g, int c, int C, float& a, float& b) const {
        float v = ld1(x + r * ldx + c) - st.p[0]; a = v; b = v * v;
    }
    __device__ __forceinline__ void eval4(const State& st, long long r, int g, int c, int C, float4& a, float4& b) const {
        float4 v = ldf4(x + r * ldx + c);
        v.x -= st.p[0]; v.y -= st.p[1]; v.z -= st.p[2]; v.w -= st.p[3];
        a = v; b = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
    }
};

template <typename T>
__global__ __launch_bounds__(64) void sums_finalize_kernel(const float* __restrict__ part, int nblk, int C, double* __restrict__ sum,
                                     double* __restrict__ sq, float* __restrict__ fsum, int accumulate,
                                     const T* __restrict__ x, int ldx, long long rows, int use_pivot) {
    const int c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) {
        a += (double)part[((long long)k * C + c) * 2];
        b += (double)part[((long long)k * C + c) * 2 + 1];
    }
    a = wave_sum(a); b = wave_sum(b);
    if (threadIdx.x != 0) return;
    if (use_pivot) {            // undo the shift in fp64: sum x = S1 + N p, sum x^2 = S2 + 2 p S1 + N p^2
        const double pv = (double)stats_pivot(x, ldx, 0, rows, c), n = (double)rows;
        b = b + 2.0 * pv * a + n * pv * pv;
        a = a + n * pv;
    }
    if (sum) sum[c] = a;
    if (sq) sq[c] = b;
    if (fsum) fsum[c] = accumulate ? fsum[c] + (float)a : (float)a;
}

size_t colsum_ws_bytes(int C) { return align_up((size_t)kMaxRedBlocks * C * 2 * sizeof(float), 256); }

int finalize_channel_partials(const float* part, int nblk, int C, double* sum, double* sq, hipStream_t st) {
    hipLaunchKernelGGL(sums_finalize_kernel<float>, dim3(C), dim3(64), 0, st, part, nblk, C, sum, sq, (float*)nullptr, 0,
                       (const float*)nullptr, 0, 0LL, 0);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

template <typename T>
static int channel_sums_chunk(const T* x, int ldx, long long rows, int C, double* sum, double* sq, float* fsum, int accumulate,
                              void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_WS(colsum_ws_bytes(C), ws_bytes);
    float* part = (float*)ws;
    RedPlan p;
    const int use_pivot = sq != nullptr;          // second moments: shift by a data pivot against cancellation
    SumF<T> f{x, ldx, rows, use_pivot};
    int rc = launch_colreduce2(f, rows, 1, C, ldx, part, &p, st);
    if (rc) return rc;
    hipLaunchKernelGGL(sums_finalize_kernel<T>, dim3(C), dim3(64), 0, st, part, p.nblk, C, sum, sq, fsum, accumulate, x, ldx, rows, use_pivot);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// exported to the other translation units.  The reducer handles power-of-two widths up to 1024 (4 elements per lane) or
// any width up to 256, so wider / ragged channel counts are walked in such chunks (e.g. 768 = 512 + 256).
template <typename T>
static int channel_sums_t(const T* x, int ldx, long long rows, int C, double* sum, double* sq, float* fsum, int accumulate,
                          void* ws, size_t ws_bytes, hipStream_t st) {
    int c0 = 0;
    while (c0 < C) {
        int rem = C - c0, take;
        if (rem >= 4 && (ldx % 4) == 0 && (c0 % 4) == 0 && ((uintptr_t)x % (4 * sizeof(T))) == 0) {
            take = 4;
            while (take * 2 <= rem && take * 2 <= 1024) take *= 2;
        } else {
            take = rem < 256 ? rem : 256;
        }
        int rc = channel_sums_chunk(x + c0, ldx, rows, take, sum ? sum + c0 : nullptr, sq ? sq + c0 : nullptr, fsum ? fsum + c0 : nullptr,
                                    accumulate, ws, ws_bytes, st);
        if (rc) return rc;
        c0 += take;
    }
    return MI355SEG_OK;
}
int channel_sums(const float* x, int ldx, long long rows, int C, double* sum, double* sq, float* fsum, int accumulate,
                 void* ws, size_t ws_bytes, hipStream_t st) {
    return channel_sums_t(x, ldx, rows, C, sum, sq, fsum, accumulate, ws, ws_bytes, st);
}
int channel_sums(const bf16* x, int ldx, long long rows, int C, double* sum, double* sq, float* fsum, int accumulate,
                 void* ws, size_t ws_bytes, hipStream_t st) {
    return channel_sums_t(x, ldx, rows, C, sum, sq, fsum, accumulate, ws, ws_bytes, st);
}

// ---- per-tile partials of a convolution's epilogue (thousands of M-tiles at full resolution) -> per-channel totals, two stages:
// stage 1: part[k][c][NV] as a [nblk][C * NV] matrix, a workgroup owns 32 channels x one slice of the tiles (8 row lanes, fp64, LDS
// combine in a fixed order) -> tmp[slice][c][NV]; stage 2: one thread per channel walks the <= 64 slices.  (One wavefront per
// channel over all tiles was 20-50 us per layer at 16384 tiles.)
// MODE 1: BatchNorm statistics triples {sum, M2 about the tile mean, n}: emits {sum, M2 + sum^2 / n} = {sum y, sum y^2} of the tile
template <int NV, int MODE>
__global__ __launch_bounds__(256) void part_reduce_kernel(const float* __restrict__ part, int nblk, int C, int R, double* __restrict__ tmp) {
    constexpr int NO = MODE == 1 ? 2 : NV;
    __shared__ double sh[8][32][NO];
    const int cl = threadIdx.x & 31, q = threadIdx.x >> 5, c = blockIdx.x * 32 + cl, rb = blockIdx.y;
    const int rpb = (nblk + R - 1) / R, k0 = rb * rpb, k1 = min(nblk, k0 + rpb);
    double a[NO];
#pragma unroll
    for (int v = 0; v < NO; ++v) a[v] = 0.0;
    if (c < C)
        for (int k = k0 + q; k < k1; k += 8) {
            const float* p = part + ((long long)k * C + c) * NV;
            if (MODE == 1) {
                const double s = (double)p[0], n = (double)p[2];
                a[0] += s; a[1] += (double)p[1] + (n > 0.0 ? s * s / n : 0.0);
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) a[v] += (double)p[v];
            }
        }
#pragma unroll
    for (int v = 0; v < NO; ++v) sh[q][cl][v] = a[v];
    __syncthreads();
    if (q == 0 && c < C) {
#pragma unroll
        for (int v = 0; v < NO; ++v) {
            double t = sh[0][cl][v];
#pragma unroll
            for (int j = 1; j < 8; ++j) t += sh[j][cl][v];
            tmp[((long long)rb * C + c) * NO + v] = t;
        }
    }
}

// stage 2; kind 0: BatchNorm backward (s1, s2, dgamma = s2, dbeta = s1 as floats), kind 1: statistics (sum, sq as doubles).
// A workgroup owns 32 channels; its eight row lanes each walk every eighth slice (independent loads in flight instead of one
// thread's serial walk over all slices: 18 us -> 3 us per layer at 64 slices), LDS combine in a fixed order.
__global__ __launch_bounds__(256) void part_finalize_kernel(const double* __restrict__ tmp, int R, int C, int kind, float* __restrict__ s1, float* __restrict__ s2,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, double* __restrict__ sum, double* __restrict__ sq) {
    __shared__ double sh[8][32][2];
    const int cl = threadIdx.x & 31, q = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int r = q; r < R; r += 8) { a += tmp[((long long)r * C + c) * 2]; b += tmp[((long long)r * C + c) * 2 + 1]; }
    sh[q][cl][0] = a; sh[q][cl][1] = b;
    __syncthreads();
    if (q != 0 || c >= C) return;
#pragma unroll
    for (int j = 1; j < 8; ++j) { a += sh[j][cl][0]; b += sh[j][cl][1]; }
    if (kind == 0) {
        s1[c] = (float)a; s2[c] = (float)b;
        if (dgamma) { dgamma[c] = (float)b; dbeta[c] = (float)a; }
    } else {
        sum[c] = a; sq[c] = b;
    }
}

constexpr int kPartSlices = 256;
static int part_slices(int nblk, int C) {
    int R = 2048 / ((C + 31) / 32);                     // about eight workgroups per CU in stage 1 (64 slices: 11 us per layer at 16384 tiles x 32 channels)
    if (R > kPartSlices) R = kPartSlices;
    if (R > nblk / 16) R = nblk / 16;
    return R < 1 ? 1 : R;
}
size_t part_reduce_ws_bytes(int C) { return align_up((size_t)kPartSlices * C * 2 * sizeof(double), 256); }

// s1 / s2 (+ dgamma / dbeta) from per-tile {sum dz, sum dz * xhat} pairs produced by a convolution's epilogue (conv_x3s.hip);
// tmp: part_reduce_ws_bytes(C) of scratch (null: the one-stage kernel)
void norm_bwd_finalize(const float* part, int nblk, int C, float* s1, float* s2, float* dgamma, float* dbeta, double* tmp, hipStream_t st) {
    if (tmp && nblk > 512) {
        const int R = part_slices(nblk, C);
        hipLaunchKernelGGL((part_reduce_kernel<2, 0>), dim3((C + 31) / 32, R), dim3(256), 0, st, part, nblk, C, R, tmp);
        hipLaunchKernelGGL(part_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, st, tmp, R, C, 0, s1, s2, dgamma, dbeta, (double*)nullptr, (double*)nullptr);
        return;
    }
    hipLaunchKernelGGL(bwd_finalize_kernel, dim3(C), dim3(64), 0, st, part, nblk, C, 1, s1, s2, dgamma, dbeta);
}

// sum y / sum y^2 per channel from the per-tile {sum, M2, n} triples of the forward epilogues; false: too few tiles, use the caller's
// one-stage kernel
bool tile_stats_finalize2(const float* spart, int nM, int C, double* sum, double* sq, double* tmp, hipStream_t st) {
    if (!tmp || nM <= 512) return false;
    const int R = part_slices(nM, C);
    hipLaunchKernelGGL((part_reduce_kernel<3, 1>), dim3((C + 31) / 32, R), dim3(256), 0, st, spart, nM, C, R, tmp);
    hipLaunchKernelGGL(part_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, st, tmp, R, C, 1, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, sum, sq);
    return true;
}

}  // namespace seg

using namespace seg;

extern "C" {
size_t mi355seg_norm_ws_bytes(long long rows, int groups, int C) {
    (void)rows;
    size_t g = (size_t)(groups < 1 ? 1 : groups);
    return align_up(g * kMaxRedBlocks * (size_t)C * 2 * sizeof(float), 256) + 2 * align_up(g * C * sizeof(float), 256) +
           align_up((size_t)8192 * C * 2 * sizeof(float), 256) + 1024;
}

int mi355seg_norm_stats_from_sums_f32(const double* sum, const double* sq, long long rows, int C, float eps,
                                      float* mean, float* rstd, float* running_mean, float* running_var,
                                      float momentum, void* stream) {
    SEG_CHECK_ARG(sum && sq && mean && rstd && rows > 0 && C > 0, "norm_stats_from_sums: bad arguments");
    hipLaunchKernelGGL(stats_from_sums_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, sum, sq, C,
                       (double)rows, eps, mean, rstd, running_mean, running_var, momentum);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// mi355seg_norm_stats_from_sums_f32 + the FOLDED form of the normalisation, al = rstd gamma, be = beta - mean al (what
// norm_act_fwd_kernel computes per thread), + an upper bound of max |act(al x + be)| from max |x|: bound = max_c (|al_c| X + |be_c|)
// (ReLU: max_c max(0, |al_c| X + be_c)) -- the operand maximum of a convolution that applies the norm + activation as a prologue
__global__ __launch_bounds__(256) void norm_fold_kernel(const double* __restrict__ sum, const double* __restrict__ sq, int C, double rows, float eps,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                       float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ rmean, float* __restrict__ rvar,
                                                       float momentum, const float* __restrict__ x_amax, float* __restrict__ al, float* __restrict__ be,
                                                       float* __restrict__ a_amax) {
    __shared__ float sh[4];
    const float X = x_amax ? *x_amax : 0.f;
    float bound = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        double m = sum[c] / rows;
        double var = sq[c] / rows - m * m;
        if (var < 0.0) var = 0.0;
        const float mf = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
        mean[c] = mf; rstd[c] = rs;
        if (rmean) {
            double unb = rows > 1.0 ? var * rows / (rows - 1.0) : var;
            rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
            rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
        }
        const float a = rs * (gamma ? gamma[c] : 1.f);
        const float b = (beta ? beta[c] : 0.f) - mf * a;
        al[c] = a; be[c] = b;
        const float hi = fabsf(a) * X;
        bound = fmaxf(bound, act == MI355SEG_ACT_RELU ? fmaxf(hi + b, 0.f) : hi + fabsf(b));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = bound;
    __syncthreads();
    if (threadIdx.x == 0 && a_amax) *a_amax = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) * 1.0000002f;      // (one ulp up: al x + be is rounded)
}
int mi355seg_norm_fold_f32(const double* sum, const double* sq, long long rows, int C, float eps, const float* gamma, const float* beta, int act,
                           float* mean, float* rstd, float* running_mean, float* running_var, float momentum,
                           const float* x_amax, float* al, float* be, float* a_amax, void* stream) {
    SEG_CHECK_ARG(sum && sq && mean && rstd && al && be && rows > 0 && C > 0 && act >= 0 && act <= 4, "norm_fold: bad arguments");
    hipLaunchKernelGGL(norm_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sum, sq, C, (double)rows, eps, gamma, beta, act, mean, rstd,
                       running_mean, running_var, momentum, x_amax, al, be, a_amax);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// eval-mode BatchNorm as a per-channel affine map of the convolution's raw output: scale = gamma / sqrt(var + eps),
// shift = beta + (conv_bias - mean) * scale
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                               const float* __restrict__ var, const float* __restrict__ cbias, float eps, int C,
                               float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = (gamma ? gamma[c] : 1.f) * (float)(1.0 / sqrt((double)var[c] + (double)eps));     // the rstd of rstd_from_var_kernel
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) + ((cbias ? cbias[c] : 0.f) - mean[c]) * sc;
}
int mi355seg_bn_fold_f32(const float* gamma, const float* beta, const float* mean, const float* var, const float* conv_bias,
                         float eps, int C, float* scale, float* shift, void* stream) {
    SEG_CHECK_ARG(mean && var && scale && shift && C > 0, "bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, gamma, beta, mean, var, conv_bias, eps, C, scale, shift);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_rstd_from_var_f32(const float* var, float eps, float* rstd, int C, void* stream) {
    SEG_CHECK_ARG(var && rstd && C > 0, "rstd_from_var: bad arguments");
    hipLaunchKernelGGL(rstd_from_var_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, var, eps, rstd, C);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

#define TT float
#define FN(name) mi355seg_##name##_f32
#define NORM_IMPL norm_act_bwd_impl_f32
#include "norm_api.inc"
#undef TT
#undef FN
#undef NORM_IMPL
#define TT bf16
#define FN(name) mi355seg_##name##_bf16
#define NORM_IMPL norm_act_bwd_impl_bf16
#include "norm_api.inc"
#undef TT
#undef FN
#undef NORM_IMPL

}  // extern "C"
