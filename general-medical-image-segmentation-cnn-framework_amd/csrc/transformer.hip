// transformer.hip -- the UNETR encoder's dense ops (unetr.py:54-168): Linear / attention matmuls as one
// strided, batched fp32-MFMA GEMM, LayerNorm and row softmax with their backward.  At 96^3 the encoder is
// 216 tokens x 768 (0.09 TFLOP per step vs 4.6 TFLOP of decoder convs), so these kernels are written for
// low latency at few-hundred-row sizes: 64x64 tiles, K-step 32 with register prefetch, float4 global reads
// along whichever operand index is contiguous, deterministic split-K when the tile count cannot fill the chip; a wavefront per row for LayerNorm / softmax (DPP shuffles).
#include "common.h"
#include "internal.h"
#include <algorithm>

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    long long a_rs, a_cs, a_b0, a_b1, b_rs, b_cs, b_b0, b_b1, c_rs, c_b0, c_b1;
    int M, N, K, nb1;
    float alpha; int relu, accumulate;
    int S, kchunk, avec, bvec;      // split-K factor, K range per split, float4 global reads allowed for A / B
    float* slabs;                   // [S][M][N] partial products when S > 1
    float* arowsum;                 // (direct kernels, one batch) arowsum[m] = sum_k A[m][k]: a Linear layer's bias gradient out of its weight-gradient GEMM
    const float* emul; const float* eadd; long long e_rs;      // (direct kernels, one batch) after bias / ReLU: C = C * emul[m][n] + eadd[m][n] (dropout mask, residual)
};

constexpr int GT = 64, GK = 32, LDA_S = GK + 1, LDB_S = GT + 1;

// One operand tile (64 "outer" rows x GK k-columns) moves global -> 8 registers per thread -> LDS.  ``ofast``: the outer
// index is the unit-stride one.  ``vec``: 16-byte reads along the unit-stride index (host-verified alignment), with a
// per-group scalar fallback at the M/N/K edges.  LDS element (o, k) lives at o * so + k * sk.
struct TileMover {
    const float* base; long long os, ks; int o0, olim, klim; bool ofast, vec;
    __device__ __forceinline__ void coords(int slot, int tid, int& o, int& k) const {
        if (vec) {
            const int q = slot * 256 + tid;                  // 512 float4 groups
            if (ofast) { k = q >> 4; o = (q & 15) << 2; } else { o = q >> 3; k = (q & 7) << 2; }
        } else {
            const int q = slot * 256 + tid;                  // 2048 scalars
            if (ofast) { k = q >> 6; o = q & 63; } else { o = q >> 5; k = q & 31; }
        }
    }
    __device__ __forceinline__ void load(int k0, int tid, float (&r)[8]) const {
        if (vec) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int o, k; coords(e, tid, o, k);
                const float* p = base + (long long)(o0 + o) * os + (long long)(k0 + k) * ks;
                const bool full = ofast ? (o0 + o + 3 < olim && k0 + k < klim) : (o0 + o < olim && k0 + k + 3 < klim);
                if (full) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    r[4 * e] = v.x; r[4 * e + 1] = v.y; r[4 * e + 2] = v.z; r[4 * e + 3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int oo = o + (ofast ? j : 0), kk = k + (ofast ? 0 : j);
                        r[4 * e + j] = (o0 + oo < olim && k0 + kk < klim) ? p[j] : 0.f;
                    }
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int o, k; coords(e, tid, o, k);
                r[e] = (o0 + o < olim && k0 + k < klim) ? base[(long long)(o0 + o) * os + (long long)(k0 + k) * ks] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float* lds, int so, int sk, int tid, const float (&r)[8]) const {
        if (vec) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int o, k; coords(e, tid, o, k);
#pragma unroll
                for (int j = 0; j < 4; ++j) lds[(o + (ofast ? j : 0)) * so + (k + (ofast ? 0 : j)) * sk] = r[4 * e + j];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { int o, k; coords(e, tid, o, k); lds[o * so + k * sk] = r[e]; }
        }
    }
};

// C[b][M,N] (+)= alpha * A[b][M,K] B[b][K,N] + bias, arbitrary strides, 64x64 tile per workgroup (4 waves, one 32x32 fp32
// MFMA accumulator each), K walked in steps of 32 with the next step's operands prefetched into registers while the
// current one is multiplied.  blockIdx.z = batch * S + split; with S > 1 the raw partial tile goes to its slab and
// gemm_splitk_epilogue applies alpha / bias / accumulate / ReLU after a fixed-order sum (deterministic).
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ float As[GT * LDA_S];
    __shared__ float Bs[GK * LDB_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const int split = blockIdx.z % g.S, bz = blockIdx.z / g.S;
    const int b0 = bz / g.nb1, b1 = bz % g.nb1;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int kbeg = split * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const TileMover ta{g.A + b0 * g.a_b0 + b1 * g.a_b1, g.a_rs, g.a_cs, m0, g.M, kend, g.a_rs == 1 && g.a_cs != 1, g.avec != 0};
    const TileMover tb{g.B + b0 * g.b_b0 + b1 * g.b_b1, g.b_cs, g.b_rs, n0, g.N, kend, g.b_cs == 1, g.bvec != 0};
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    float ra[8], rb[8];
    ta.load(kbeg, tid, ra);
    tb.load(kbeg, tid, rb);
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        ta.store(As, LDA_S, 1, tid, ra);
        tb.store(Bs, 1, LDB_S, tid, rb);
        __syncthreads();
        if (k0 + GK < kend) { ta.load(k0 + GK, tid, ra); tb.load(k0 + GK, tid, rb); }
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
            const float av = As[(wr * 32 + i) * LDA_S + kk * 2 + h];
            const float bv = Bs[(kk * 2 + h) * LDB_S + wc * 32 + i];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wc * 32 + i;
    if (col >= g.N) return;
    if (g.S > 1) {
        float* slab = g.slabs + (long long)split * g.M * g.N;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + wr * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
            if (row < g.M) slab[(long long)row * g.N + col] = acc[v];
        }
        return;
    }
    float* C = g.C + b0 * g.c_b0 + b1 * g.c_b1;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int row = m0 + wr * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (row < g.M) {
            float* dst = C + (long long)row * g.c_rs + col;
            float val = g.alpha * acc[v] + bv;
            if (g.accumulate) val += *dst;
            if (g.relu) val = val > 0.f ? val : 0.f;
            *dst = val;
        }
    }
}

// ---- the few-hundred-row GEMMs of the token encoder (M = 216 at 96^3), without LDS staging and without a second kernel.
// A 32x32x2 fp32 MFMA takes ONE float per lane per operand, A[m = i][k] and B[k][n = i] with k = the lane half's own index:
// which k each half multiplies is free as long as both operands agree, so half h of MFMA e in an 8-deep k-block takes
// k = 8j + 4h + e.  An operand that is contiguous along k then is one float4 per lane per four MFMAs straight from global
// memory (L2: these matrices are a few MB), an operand contiguous along its outer index is four dword loads, each coalesced
// over the 32 lanes.  The four waves of a workgroup split K (k-block j goes to wave j % 4), add their 32x32 accumulators
// through LDS in wave order (deterministic) and apply alpha / bias / accumulate / ReLU in the same kernel: one launch of
// ceil(M/32) x ceil(N/32) workgroups instead of a 64x64-tile kernel plus a split-K epilogue (19.8 + 4.4 us at 216x768x768).
constexpr int GD_WAVES = 4;
template <bool AK, bool BK>     // operand is contiguous along k (else along its outer index)
__global__ __launch_bounds__(GD_WAVES * 64) void gemm_direct_kernel(GemmArgs g) {
    __shared__ float red[GD_WAVES * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;
    const int b0 = blockIdx.z / g.nb1, b1 = blockIdx.z % g.nb1;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int arow = min(m0 + i, g.M - 1), bcol = min(n0 + i, g.N - 1);          // edge tiles: clamp the loads, mask the stores
    const float* __restrict__ ap = g.A + b0 * g.a_b0 + b1 * g.a_b1 + (long long)arow * g.a_rs;
    const float* __restrict__ bp = g.B + b0 * g.b_b0 + b1 * g.b_b1 + (long long)bcol * g.b_cs;
    const int nblk = g.K / 8;                                                      // host guarantees K % 8 == 0
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    auto lda = [&](int j, float (&r)[4]) {
        const int k = 8 * j + 4 * h;
        if constexpr (AK) { const float4 q = *reinterpret_cast<const float4*>(ap + k); r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w; }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = ap[(long long)(k + e) * g.a_cs];
        }
    };
    auto ldb = [&](int j, float (&r)[4]) {
        const int k = 8 * j + 4 * h;
        if constexpr (BK) { const float4 q = *reinterpret_cast<const float4*>(bp + k); r[0] = q.x; r[1] = q.y; r[2] = q.z; r[3] = q.w; }
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = bp[(long long)(k + e) * g.b_rs];
        }
    };
    // U k-blocks of this wave per trip; the operands of the next two trips are in flight while a trip's 4U MFMAs issue
    // (the weights come from HBM once per GEMM: a trip is ~0.4 us of MFMAs, a miss 1-2 us)
    constexpr int U = 4;
    float ra[3][U][4], rb[3][U][4];
    const int mine = nblk > wave ? (nblk - wave + GD_WAVES - 1) / GD_WAVES : 0;    // k-blocks of this wave: wave, wave + 4, ...
    const int trips = (mine + U - 1) / U;
    auto fetch = [&](int jt, float (&xa)[U][4], float (&xb)[U][4]) {
        if (jt >= trips) return;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = min(wave + GD_WAVES * (jt * U + u), nblk - 1);          // past the end: a valid address, the MFMA is skipped
            lda(j, xa[u]); ldb(j, xb[u]);
        }
    };
    const bool rowsum = g.arowsum != nullptr && blockIdx.x == 0;                  // (workgroup-uniform) the first column of tiles also sums A's rows
    float rs = 0.f;
    auto mm = [&](int jt, const float (&xa)[U][4], const float (&xb)[U][4]) {
        if (jt >= trips) return;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (jt * U + u < mine) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u][e], xb[u][e], acc, 0, 0, 0);
                if (rowsum) rs += (xa[u][0] + xa[u][1]) + (xa[u][2] + xa[u][3]);
            }
    };
    fetch(0, ra[0], rb[0]);
    fetch(1, ra[1], rb[1]);
    for (int jt = 0; jt < trips; jt += 3) {
        fetch(jt + 2, ra[2], rb[2]);
        mm(jt, ra[0], rb[0]);
        fetch(jt + 3, ra[0], rb[0]);
        mm(jt + 1, ra[1], rb[1]);
        fetch(jt + 4, ra[1], rb[1]);
        mm(jt + 2, ra[2], rb[2]);
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) red[wave * 1024 + ((v & 3) + 8 * (v >> 2) + 4 * h) * 32 + i] = acc[v];
    __syncthreads();
    float* C = g.C + b0 * g.c_b0 + b1 * g.c_b1;
#pragma unroll
    for (int q = 0; q < 1024 / (GD_WAVES * 64); ++q) {
        const int e = q * GD_WAVES * 64 + tid, r = e >> 5, c = e & 31;
        if (m0 + r < g.M && n0 + c < g.N) {
            float sum = red[e];
#pragma unroll
            for (int w = 1; w < GD_WAVES; ++w) sum += red[w * 1024 + e];           // wave order: deterministic
            float* dst = C + (long long)(m0 + r) * g.c_rs + n0 + c;
            float val = g.alpha * sum + (g.bias ? g.bias[n0 + c] : 0.f);
            if (g.accumulate) val += *dst;
            if (g.relu) val = val > 0.f ? val : 0.f;
            if (g.emul) val *= g.emul[(long long)(m0 + r) * g.e_rs + n0 + c];
            if (g.eadd) val += g.eadd[(long long)(m0 + r) * g.e_rs + n0 + c];
            *dst = val;
        }
    }
    if (rowsum) {                                   // lane (h, i) of wave w summed row m0 + i over its share of k: fixed order over (w, h)
        __syncthreads();
        red[wave * 64 + lane] = rs;
        __syncthreads();
        if (tid < 32 && m0 + tid < g.M) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < GD_WAVES; ++w) t += red[w * 64 + tid] + red[w * 64 + 32 + tid];
            g.arowsum[m0 + tid] = t;
        }
    }
}

// ---- the same GEMMs on the bf16 matrix cores (mi355seg_gemm_lowp_f32: the token path under bf16 autocast, where the reference's own
// nn.Linear / matmul run in bf16, unetr.py:59-138 under torch.autocast): fp32 operands in memory, rounded to bf16 (RNE) in registers,
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 result.  A lane holds EIGHT consecutive k of its row / column per MFMA
// (k = 16 j + 8 h + e): two float4 loads where the operand is contiguous along k, eight coalesced dword loads otherwise; one MFMA
// does the work of eight fp32 ones.  K % 8 == 0 (host): a lane's group of eight is inside K or wholly past it (zeros).
// MUL (r6): the A operand is A * amul * [agate > 0] (same indexing as A; either may be null) -- a dropout layer's factors and the ReLU
// backward's gate folded into the loads of the gradient a Linear layer's two backward GEMMs read (an element-wise launch each before)
template <bool AK, bool BK, bool MUL>
__device__ __forceinline__ void gemm_direct_lowp_body(const GemmArgs& g, int zb, const float* __restrict__ amul, const float* __restrict__ agate, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;
    const int b0 = zb / g.nb1, b1 = zb % g.nb1;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int arow = min(m0 + i, g.M - 1), bcol = min(n0 + i, g.N - 1);
    const long long aoff = b0 * g.a_b0 + b1 * g.a_b1 + (long long)arow * g.a_rs;
    const float* __restrict__ ap = g.A + aoff;
    const float* __restrict__ bp = g.B + b0 * g.b_b0 + b1 * g.b_b1 + (long long)bcol * g.b_cs;
    const int nblk = (g.K + 15) / 16;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    auto ld8 = [&](const float* __restrict__ base, long long kstride, bool contiguous, int j, float (&r)[8]) {
        const int k = 16 * j + 8 * h;
        if (k >= g.K) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = 0.f;
            return;
        }
        if (contiguous) {
            const float4 q0 = *reinterpret_cast<const float4*>(base + k), q1 = *reinterpret_cast<const float4*>(base + k + 4);
            r[0] = q0.x; r[1] = q0.y; r[2] = q0.z; r[3] = q0.w; r[4] = q1.x; r[5] = q1.y; r[6] = q1.z; r[7] = q1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = base[(long long)(k + e) * kstride];
        }
    };
    constexpr int U = 2;
    float ra[3][U][8], rb[3][U][8];
    const int mine = nblk > wave ? (nblk - wave + GD_WAVES - 1) / GD_WAVES : 0;    // 16-deep k-blocks of this wave: wave, wave + 4, ...
    const int trips = (mine + U - 1) / U;
    auto fetch = [&](int jt, float (&xa)[U][8], float (&xb)[U][8]) {
        if (jt >= trips) return;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = min(wave + GD_WAVES * (jt * U + u), nblk - 1);
            ld8(ap, g.a_cs, AK, j, xa[u]); ld8(bp, g.b_rs, BK, j, xb[u]);
            if constexpr (MUL) {
                float f[8];
                if (amul) {
                    ld8(amul + aoff, g.a_cs, AK, j, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) xa[u][e] *= f[e];
                }
                if (agate) {
                    ld8(agate + aoff, g.a_cs, AK, j, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) xa[u][e] = f[e] > 0.f ? xa[u][e] : 0.f;
                }
            }
        }
    };
    const bool rowsum = g.arowsum != nullptr && blockIdx.x == 0;                  // (workgroup-uniform) the first column of tiles also sums A's rows
    float rs = 0.f;                                                                // (of the fp32 values, before the bf16 rounding)
    auto mm = [&](int jt, const float (&xa)[U][8], const float (&xb)[U][8]) {
        if (jt >= trips) return;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (jt * U + u < mine) {
                bf16x8_t va, vb;
#pragma unroll
                for (int e = 0; e < 8; ++e) { va[e] = (bf16)xa[u][e]; vb[e] = (bf16)xb[u][e]; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
                if (rowsum) rs += ((xa[u][0] + xa[u][1]) + (xa[u][2] + xa[u][3])) + ((xa[u][4] + xa[u][5]) + (xa[u][6] + xa[u][7]));
            }
    };
    fetch(0, ra[0], rb[0]);
    fetch(1, ra[1], rb[1]);
    for (int jt = 0; jt < trips; jt += 3) {
        fetch(jt + 2, ra[2], rb[2]);
        mm(jt, ra[0], rb[0]);
        fetch(jt + 3, ra[0], rb[0]);
        mm(jt + 1, ra[1], rb[1]);
        fetch(jt + 4, ra[1], rb[1]);
        mm(jt + 2, ra[2], rb[2]);
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) red[wave * 1024 + ((v & 3) + 8 * (v >> 2) + 4 * h) * 32 + i] = acc[v];
    __syncthreads();
    float* C = g.C + b0 * g.c_b0 + b1 * g.c_b1;
#pragma unroll
    for (int q = 0; q < 1024 / (GD_WAVES * 64); ++q) {
        const int e = q * GD_WAVES * 64 + tid, r = e >> 5, c = e & 31;
        if (m0 + r < g.M && n0 + c < g.N) {
            float sum = red[e];
#pragma unroll
            for (int w = 1; w < GD_WAVES; ++w) sum += red[w * 1024 + e];           // wave order: deterministic
            float* dst = C + (long long)(m0 + r) * g.c_rs + n0 + c;
            float val = g.alpha * sum + (g.bias ? g.bias[n0 + c] : 0.f);
            if (g.accumulate) val += *dst;
            if (g.relu) val = val > 0.f ? val : 0.f;
            if (g.emul) val *= g.emul[(long long)(m0 + r) * g.e_rs + n0 + c];
            if (g.eadd) val += g.eadd[(long long)(m0 + r) * g.e_rs + n0 + c];
            *dst = val;
        }
    }
    if (rowsum) {                                   // lane (h, i) of wave w summed row m0 + i over its share of k: fixed order over (w, h)
        __syncthreads();
        red[wave * 64 + lane] = rs;
        __syncthreads();
        if (tid < 32 && m0 + tid < g.M) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < GD_WAVES; ++w) t += red[w * 64 + tid] + red[w * 64 + 32 + tid];
            g.arowsum[m0 + tid] = t;
        }
    }
}


__global__ __launch_bounds__(256) void gemm_splitk_epilogue(GemmArgs g) {
    const long long total = (long long)g.M * g.N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int row = (int)(e / g.N), col = (int)(e - (long long)row * g.N);
        float s = 0.f;
        for (int k = 0; k < g.S; ++k) s += g.slabs[(long long)k * total + e];
        float* dst = g.C + (long long)row * g.c_rs + col;
        float val = g.alpha * s + (g.bias ? g.bias[col] : 0.f);
        if (g.accumulate) val += *dst;
        if (g.relu) val = val > 0.f ? val : 0.f;
        *dst = val;
    }
}

// Split K only for single-batch GEMMs that would leave most of the 256 CUs idle (UNETR's 216-token Linears).
static void gemm_plan(int M, int N, int K, int nb, int& S, int& kchunk) {
    const long long tiles = (long long)cdiv(M, GT) * cdiv(N, GT) * nb;
    S = 1; kchunk = (int)cdiv(K, GK) * GK;
    if (nb != 1 || tiles >= 192 || K < 128) return;
    const int want = (int)cdiv(384, tiles);
    int chunk = (int)cdiv(cdiv(K, want), GK) * GK;
    if (chunk < 64) chunk = 64;
    S = (int)cdiv(K, chunk);
    if (S > 1) kchunk = chunk; else S = 1;
}

// ---------------------------------------------------------------- LayerNorm (one wavefront per row)
constexpr int LN_MAXJ = 32;     // E <= 64 * 32

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
        const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
        long long rows, int E, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * E;
    float v[LN_MAXJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; v[j] = c < E ? xr[c] : 0.f; s += v[j]; }
    const float m = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; const float d = c < E ? v[j] - m : 0.f; q += d * d; }
    const float rs = 1.f / sqrtf(wave_sum(q) / (float)E + eps);
    float* yr = y + row * E;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; if (c < E) yr[c] = (v[j] - m) * rs * gamma[c] + beta[c]; }
    if (lane == 0) { mean[row] = m; rstd[row] = rs; }
}

__device__ __forceinline__ void layernorm_bwd_dx_part(int blk, const float* __restrict__ dy, const float* __restrict__ x,
        const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
        float* __restrict__ dx, long long rows, int E, const float* __restrict__ addend) {
    // addend (optional): a second gradient of x (the residual stream that bypasses the norm, unetr.py:160,166): dx = addend + LN backward
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blk * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float m = mean[row], rs = rstd[row];
    float g[LN_MAXJ], xh[LN_MAXJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int c = lane + 64 * j;
        g[j] = c < E ? dy[row * E + c] * gamma[c] : 0.f;
        xh[j] = c < E ? (x[row * E + c] - m) * rs : 0.f;
        s1 += g[j]; s2 += g[j] * xh[j];
    }
    s1 = wave_sum(s1) / (float)E; s2 = wave_sum(s2) / (float)E;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int c = lane + 64 * j;
        if (c < E) dx[row * E + c] = rs * (g[j] - s1 - xh[j] * s2) + (addend ? addend[row * E + c] : 0.f);
    }
}

// dgamma[c] = sum_rows dy * xhat, dbeta[c] = sum_rows dy.  Block = 32 columns x 8 row groups (rows r = g, g+8, ...);
// the eight partial sums of a column are added in group order through LDS (fixed order -> reproducible).
__device__ __forceinline__ void layernorm_bwd_affine_part(int blk, const float* __restrict__ dy, const float* __restrict__ x,
        const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
        long long rows, int E) {
    __shared__ double sa[8][32], sb[8][32];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blk * 32 + cl;
    double a = 0.0, b = 0.0;
    if (c < E) {
        for (long long r = g; r < rows; r += 8) {
            const float d = dy[r * E + c];
            a += (double)(d * (x[r * E + c] - mean[r]) * rstd[r]);
            b += (double)d;
        }
    }
    sa[g][cl] = a; sb[g][cl] = b;
    __syncthreads();
    if (g == 0 && c < E) {
        double ta = 0.0, tb = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { ta += sa[k][cl]; tb += sb[k][cl]; }
        dgamma[c] = (float)ta; dbeta[c] = (float)tb;
    }
}

// (r6) both halves of the LayerNorm backward in ONE launch: they read the same tensors and do not depend on each other -- workgroups
// [0, ndx) write dx (a wavefront per row), the rest reduce dgamma / dbeta (32 columns each); 48 launches per UNETR step fewer
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
        const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
        long long rows, int E, const float* __restrict__ addend, int ndx) {
    if ((int)blockIdx.x < ndx) layernorm_bwd_dx_part(blockIdx.x, dy, x, gamma, mean, rstd, dx, rows, E, addend);
    else layernorm_bwd_affine_part((int)blockIdx.x - ndx, dy, x, mean, rstd, dgamma, dbeta, rows, E);
}

// ---------------------------------------------------------------- row softmax (one wavefront per row)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int L) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[LN_MAXJ];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; v[j] = c < L ? x[row * L + c] : -INFINITY; mx = fmaxf(mx, v[j]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; v[j] = c < L ? expf(v[j] - mx) : 0.f; s += v[j]; }
    const float inv = 1.f / wave_sum(s);
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; if (c < L) y[row * L + c] = v[j] * inv; }
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
        float* __restrict__ dx, long long rows, int L) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float dot = 0.f;
    for (int c = lane; c < L; c += 64) dot += y[row * L + c] * dy[row * L + c];
    dot = wave_sum(dot);
    for (int c = lane; c < L; c += 64) dx[row * L + c] = y[row * L + c] * (dy[row * L + c] - dot);
}
// softmax followed by an elementwise factor (attention dropout: keep / (1 - p)): y = softmax(x) and yk = y * keep from one pass, and the
// backward of the pair: dy = dyk * keep, dx = y (dy - sum y dy) -- the same products, in the same order, as the two-kernel chains
__global__ __launch_bounds__(256) void softmax_rows_keep_kernel(const float* __restrict__ x, const float* __restrict__ keep, float* __restrict__ y,
        float* __restrict__ yk, long long rows, int L) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[LN_MAXJ];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; v[j] = c < L ? x[row * L + c] : -INFINITY; mx = fmaxf(mx, v[j]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; v[j] = c < L ? expf(v[j] - mx) : 0.f; s += v[j]; }
    const float inv = 1.f / wave_sum(s);
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int c = lane + 64 * j;
        if (c < L) { const float pv = v[j] * inv; y[row * L + c] = pv; yk[row * L + c] = pv * keep[row * L + c]; }
    }
}
__global__ __launch_bounds__(256) void softmax_rows_keep_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dyk, const float* __restrict__ keep,
        float* __restrict__ dx, long long rows, int L) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float dot = 0.f;
    for (int c = lane; c < L; c += 64) dot += y[row * L + c] * (dyk[row * L + c] * keep[row * L + c]);
    dot = wave_sum(dot);
    for (int c = lane; c < L; c += 64) dx[row * L + c] = y[row * L + c] * (dyk[row * L + c] * keep[row * L + c] - dot);
}

template <bool AK, bool BK>
__global__ __launch_bounds__(GD_WAVES * 64) void gemm_direct_lowp_kernel(GemmArgs g) {
    __shared__ float red[GD_WAVES * 1024];
    gemm_direct_lowp_body<AK, BK, false>(g, blockIdx.z, nullptr, nullptr, red);
}

// ---- (r6) TWO independent small GEMMs in ONE launch.  The token encoder's backward is pairs of few-hundred-row GEMMs that read the same
// gradient and do not depend on each other -- a Linear layer's dX = dY W and dW = dY^T X (unetr.py:61-66,120-121), attention's dP = dO V^T and
// dV = Pd^T dO, dQ = dS K and dK = dS^T Q (unetr.py:74-98) -- each a 12-us launch at MFMA busy 0.01: latency, not work.  blockIdx.z picks the
// problem (z < nz0: problem 0, batch z; else problem 1, batch z - nz0), the x / y grid is the larger of the two tilings and a workgroup
// outside its problem's tiling leaves at once.  The body is gemm_direct_lowp_kernel's own, instantiated per problem: each result is
// bit-identical to its own launch's.
struct GemmPair { GemmArgs g[2]; const float* amul; const float* agate; int nz0; };

template <bool AK0, bool BK0, bool AK1, bool BK1, bool MUL>
__global__ __launch_bounds__(GD_WAVES * 64) void gemm_direct_lowp_pair_kernel(GemmPair gp) {
    __shared__ float red[GD_WAVES * 1024];
    if ((int)blockIdx.z < gp.nz0) {
        if ((int)blockIdx.y * 32 >= gp.g[0].M || (int)blockIdx.x * 32 >= gp.g[0].N) return;                 // (workgroup-uniform)
        gemm_direct_lowp_body<AK0, BK0, MUL>(gp.g[0], blockIdx.z, gp.amul, gp.agate, red);
    } else {
        if ((int)blockIdx.y * 32 >= gp.g[1].M || (int)blockIdx.x * 32 >= gp.g[1].N) return;
        gemm_direct_lowp_body<AK1, BK1, MUL>(gp.g[1], (int)blockIdx.z - gp.nz0, gp.amul, gp.agate, red);
    }
}

// an operand of a small-GEMM problem: contiguous along k (16-byte aligned rows) or along its outer index
static bool pair_operand_ok(const float* p, long long kfast_other, long long kstride, long long b0s, long long b1s, bool* kcontig) {
    *kcontig = kstride == 1;
    if (*kcontig) return kfast_other % 4 == 0 && b0s % 4 == 0 && b1s % 4 == 0 && (uintptr_t)p % 16 == 0;
    return kfast_other == 1;
}
static bool pair_problem_ok(const GemmArgs& g, int nb0, bool* ak, bool* bk) {
    if (!(g.A && g.B && g.C && g.M > 0 && g.N > 0 && g.K > 0 && g.K % 8 == 0)) return false;
    if (!pair_operand_ok(g.A, g.a_rs, g.a_cs, g.a_b0, g.a_b1, ak) || !pair_operand_ok(g.B, g.b_cs, g.b_rs, g.b_b0, g.b_b1, bk)) return false;
    return (long long)cdiv(g.M, 32) * cdiv(g.N, 32) * nb0 * g.nb1 <= 4096;
}
static int launch_pair(GemmPair& gp, int nb0a, int nb0b, void* stream) {
    bool ak0, bk0, ak1, bk1;
    SEG_CHECK_ARG(pair_problem_ok(gp.g[0], nb0a, &ak0, &bk0), "gemm pair: problem 0 is outside the small-GEMM kernel's range");
    SEG_CHECK_ARG(pair_problem_ok(gp.g[1], nb0b, &ak1, &bk1), "gemm pair: problem 1 is outside the small-GEMM kernel's range");
    gp.nz0 = nb0a * gp.g[0].nb1;
    const int nz1 = nb0b * gp.g[1].nb1;
    SEG_CHECK_ARG(gp.nz0 + nz1 < 65536, "gemm pair: too many batches");
    const int tx = (int)std::max(cdiv(gp.g[0].N, 32), cdiv(gp.g[1].N, 32)), ty = (int)std::max(cdiv(gp.g[0].M, 32), cdiv(gp.g[1].M, 32));
    const dim3 grid(tx, ty, gp.nz0 + nz1), block(GD_WAVES * 64);
    const bool mul = gp.amul || gp.agate;
    // the operand layouts of the pairs this is built for: (row-major A, row-major B^T or B) with (A^T, row-major B)
    if (ak0 && !bk0 && !ak1 && !bk1) {
        if (mul) hipLaunchKernelGGL((gemm_direct_lowp_pair_kernel<true, false, false, false, true>), grid, block, 0, (hipStream_t)stream, gp);
        else hipLaunchKernelGGL((gemm_direct_lowp_pair_kernel<true, false, false, false, false>), grid, block, 0, (hipStream_t)stream, gp);
    } else if (ak0 && bk0 && !ak1 && !bk1 && !mul) {
        hipLaunchKernelGGL((gemm_direct_lowp_pair_kernel<true, true, false, false, false>), grid, block, 0, (hipStream_t)stream, gp);
    } else {
        set_error("gemm pair: operand layouts (%d %d | %d %d) are not among the built instantiations", (int)ak0, (int)bk0, (int)ak1, (int)bk1);
        return MI355SEG_EINVAL;
    }
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // namespace seg

using namespace seg;

// column sums of a short matrix (the bias gradient of a token Linear): 64 columns per workgroup, four row groups whose
// partial sums are added in fixed order
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* __restrict__ x, int ldx, int rows, int C, float* __restrict__ out) {
    __shared__ float sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int r = rg; r < rows; r += 4) s += x[(long long)r * ldx + c];
    sh[rg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rg == 0 && c < C) out[c] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

extern "C" {

size_t mi355seg_gemm_ws_bytes(int M, int N, int K, int nb0, int nb1) {
    if (M <= 0 || N <= 0 || K <= 0 || nb0 <= 0 || nb1 <= 0) return 0;
    int S, kchunk;
    gemm_plan(M, N, K, nb0 * nb1, S, kchunk);
    return S > 1 ? (size_t)S * M * N * sizeof(float) : 0;
}

static int gemm_impl(int lowp, const float* A, long long a_rs, long long a_cs, long long a_b0, long long a_b1,
                     const float* B, long long b_rs, long long b_cs, long long b_b0, long long b_b1,
                     float* C, long long c_rs, long long c_b0, long long c_b1, const float* bias,
                     int M, int N, int K, int nb0, int nb1, float alpha, int relu, int accumulate,
                     void* ws, size_t ws_bytes, void* stream, float* arowsum = nullptr, int* arowsum_done = nullptr,
                     const float* emul = nullptr, const float* eadd = nullptr, long long e_rs = 0, int* epi_done = nullptr) {
    SEG_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && nb0 > 0 && nb1 > 0 && (long long)nb0 * nb1 < 4096, "gemm: bad arguments");
    if (arowsum_done) *arowsum_done = 0;
    if (epi_done) *epi_done = 0;
    int S, kchunk;
    gemm_plan(M, N, K, nb0 * nb1, S, kchunk);
    if (S > 1 && (!ws || ws_bytes < (size_t)S * M * N * sizeof(float))) { S = 1; kchunk = (int)cdiv(K, GK) * GK; }   // no room: unsplit
    {   // small problems (a few hundred rows or columns): the LDS-free 32x32-tile kernel, K split inside the workgroup
        const bool ak = a_cs == 1, bk = b_rs == 1;                                 // contiguous along k
        const bool a_ok = ak ? (a_rs % 4 == 0 && a_b0 % 4 == 0 && a_b1 % 4 == 0 && (uintptr_t)A % 16 == 0) : a_rs == 1;
        const bool b_ok = bk ? (b_cs % 4 == 0 && b_b0 % 4 == 0 && b_b1 % 4 == 0 && (uintptr_t)B % 16 == 0) : b_cs == 1;
        const long long tiles32 = (long long)cdiv(M, 32) * cdiv(N, 32) * nb0 * nb1;
        if (K % 8 == 0 && a_ok && b_ok && tiles32 <= 4096 && (long long)nb0 * nb1 < 65536) {
            GemmArgs g{A, B, C, bias, a_rs, a_cs, a_b0, a_b1, b_rs, b_cs, b_b0, b_b1, c_rs, c_b0, c_b1, M, N, K, nb1, alpha, relu, accumulate,
                       1, K, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
            if (arowsum && nb0 * nb1 == 1) { g.arowsum = arowsum; if (arowsum_done) *arowsum_done = 1; }
            if ((emul || eadd) && nb0 * nb1 == 1) { g.emul = emul; g.eadd = eadd; g.e_rs = e_rs; if (epi_done) *epi_done = 1; }
            dim3 grid(cdiv(N, 32), cdiv(M, 32), nb0 * nb1);
            if (lowp) {
                if (ak && bk) hipLaunchKernelGGL((gemm_direct_lowp_kernel<true, true>), grid, dim3(GD_WAVES * 64), 0, (hipStream_t)stream, g);
                else if (ak) hipLaunchKernelGGL((gemm_direct_lowp_kernel<true, false>), grid, dim3(GD_WAVES * 64), 0, (hipStream_t)stream, g);
                else if (bk) hipLaunchKernelGGL((gemm_direct_lowp_kernel<false, true>), grid, dim3(GD_WAVES * 64), 0, (hipStream_t)stream, g);
                else hipLaunchKernelGGL((gemm_direct_lowp_kernel<false, false>), grid, dim3(GD_WAVES * 64), 0, (hipStream_t)stream, g);
            }
            else if (ak && bk) hipLaunchKernelGGL((gemm_direct_kernel<true, true>), grid, dim3(GD_WAVES * 64), 0, (hipStream_t)stream, g);
            else if (ak) hipLaunchKernelGGL((gemm_direct_kernel<true, false>), grid, dim3(GD_WAVES * 64), 0, (hipStream_t)stream, g);
            else if (bk) hipLaunchKernelGGL((gemm_direct_kernel<false, true>), grid, dim3(GD_WAVES * 64), 0, (hipStream_t)stream, g);
            else hipLaunchKernelGGL((gemm_direct_kernel<false, false>), grid, dim3(GD_WAVES * 64), 0, (hipStream_t)stream, g);
            SEG_CHECK_LAUNCH();
            return MI355SEG_OK;
        }
    }
    auto vec_ok = [](const float* p, long long fast, long long other, long long b0s, long long b1s) {
        return fast == 1 && other % 4 == 0 && b0s % 4 == 0 && b1s % 4 == 0 && (uintptr_t)p % 16 == 0;
    };
    const bool a_ofast = a_rs == 1 && a_cs != 1, b_ofast = b_cs == 1;
    const int avec = vec_ok(A, a_ofast ? a_rs : a_cs, a_ofast ? a_cs : a_rs, a_b0, a_b1);
    const int bvec = vec_ok(B, b_ofast ? b_cs : b_rs, b_ofast ? b_rs : b_cs, b_b0, b_b1);
    GemmArgs g{A, B, C, bias, a_rs, a_cs, a_b0, a_b1, b_rs, b_cs, b_b0, b_b1, c_rs, c_b0, c_b1, M, N, K, nb1, alpha, relu, accumulate,
               S, kchunk, avec, bvec, (float*)ws, nullptr, nullptr, nullptr, 0};
    dim3 grid(cdiv(N, GT), cdiv(M, GT), nb0 * nb1 * S);
    hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
    SEG_CHECK_LAUNCH();
    if (S > 1) {
        const long long blocks = cdiv((long long)M * N, 256);
        hipLaunchKernelGGL(gemm_splitk_epilogue, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, (hipStream_t)stream, g);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

int mi355seg_gemm_f32(const float* A, long long a_rs, long long a_cs, long long a_b0, long long a_b1,
                      const float* B, long long b_rs, long long b_cs, long long b_b0, long long b_b1,
                      float* C, long long c_rs, long long c_b0, long long c_b1, const float* bias,
                      int M, int N, int K, int nb0, int nb1, float alpha, int relu, int accumulate,
                      void* ws, size_t ws_bytes, void* stream) {
    return gemm_impl(0, A, a_rs, a_cs, a_b0, a_b1, B, b_rs, b_cs, b_b0, b_b1, C, c_rs, c_b0, c_b1, bias, M, N, K, nb0, nb1, alpha, relu, accumulate, ws, ws_bytes, stream);
}
// fp32 tensors, products on the bf16 matrix cores (operands rounded to bf16 in registers, fp32 accumulation and result): the small-GEMM
// kernel only; shapes outside its range run the fp32 kernels
int mi355seg_gemm_lowp_f32(const float* A, long long a_rs, long long a_cs, long long a_b0, long long a_b1,
                           const float* B, long long b_rs, long long b_cs, long long b_b0, long long b_b1,
                           float* C, long long c_rs, long long c_b0, long long c_b1, const float* bias,
                           int M, int N, int K, int nb0, int nb1, float alpha, int relu, int accumulate,
                           void* ws, size_t ws_bytes, void* stream) {
    return gemm_impl(1, A, a_rs, a_cs, a_b0, a_b1, B, b_rs, b_cs, b_b0, b_b1, C, c_rs, c_b0, c_b1, bias, M, N, K, nb0, nb1, alpha, relu, accumulate, ws, ws_bytes, stream);
}

// dW[N][K] = dY^T X of a Linear layer together with its bias gradient db[n] = sum_m dY[m][n] -- the row sums of the GEMM's A operand
// (A(n, m) = dY[m][n]: a_rs = 1, a_cs = ldy), taken by the first column of tiles of the small-GEMM kernels from the values they load
// anyway (fixed summation order); other shapes: the GEMM, then mi355seg_colsum_f32 on dY.  lowp: products on the bf16 matrix cores.
int mi355seg_colsum_f32(const float* x, int ldx, long long rows, int C, float* out, void* ws, size_t ws_bytes, void* stream);
int mi355seg_linear_wgrad_f32(int lowp, const float* dy, int lddy, const float* x, int ldx, float* dw, float* db, int M, int N, int K,
                              void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(dy && x && dw && M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K, "linear_wgrad: bad arguments");
    int done = 0;
    int rc = gemm_impl(lowp ? 1 : 0, dy, 1, lddy, 0, 0, x, ldx, 1, 0, 0, dw, K, 0, 0, nullptr, N, K, M, 1, 1, 1.f, 0, 0, ws, ws_bytes, stream, db, &done);
    if (rc || !db || done) return rc;
    return mi355seg_colsum_f32(dy, lddy, M, N, db, ws, ws_bytes, stream);
}

// nn.Linear forward with what follows it in the token encoder folded into the GEMM's epilogue (unetr.py:98-100,120-138,159-166):
// y = (relu?)(x W^T + b) * emul + eadd -- emul: an element-wise dropout factor keep / (1 - p), eadd: the residual stream; both [M][N] at
// pitch N, either may be NULL.  Small-GEMM shapes: one launch; other shapes: the GEMM, then the two element-wise kernels in place.
int mi355seg_mul_f32(const float* a, const float* b, float* out, long long n, void* stream);
int mi355seg_act_fwd_f32(const float* x, int ldx, const float* res, int ldres, float* y, int ldy, long long rows, int C, int act, float slope, void* stream);
int mi355seg_linear_fwd_f32(int lowp, const float* x, int ldx, const float* w, const float* b, int relu, const float* emul, const float* eadd,
                            float* y, int M, int N, int K, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0 && ldx >= K, "linear_fwd: bad arguments");
    int done = 0;
    int rc = gemm_impl(lowp ? 1 : 0, x, ldx, 1, 0, 0, w, 1, K, 0, 0, y, N, 0, 0, b, M, N, K, 1, 1, 1.f, relu, 0, ws, ws_bytes, stream, nullptr, nullptr, emul, eadd, N, &done);
    if (rc || done) return rc;
    if (emul) { rc = mi355seg_mul_f32(y, emul, y, (long long)M * N, stream); if (rc) return rc; }
    if (eadd) rc = mi355seg_act_fwd_f32(y, N, eadd, N, y, N, M, N, MI355SEG_ACT_NONE, 0.f, stream);
    return rc;
}

// The backward of nn.Linear as ONE launch (r6): dyf = dy * dmul * [dgate > 0] (either may be NULL), dx[M][K] = dyf W, dw[N][K] = dyf^T x,
// db[n] = sum_m dyf[m][n].  bf16 products (the token path under autocast) on the few-hundred-row shapes; mi355seg_linear_bwd_supported_f32
// says whether a shape is taken (the caller runs the separate GEMMs otherwise).
int mi355seg_linear_bwd_supported_f32(int lowp, int M, int N, int K) {
    if (!lowp || M <= 0 || N <= 0 || K <= 0 || N % 8 || M % 8 || K % 4) return 0;
    return ((long long)cdiv(M, 32) * cdiv(K, 32) <= 4096 && (long long)cdiv(N, 32) * cdiv(K, 32) <= 4096) ? 1 : 0;
}
int mi355seg_linear_bwd_f32(int lowp, const float* dy, int lddy, const float* dmul, const float* dgate, const float* x, int ldx, const float* w,
                            float* dx, float* dw, float* db, int M, int N, int K, void* stream) {
    SEG_CHECK_ARG(dy && x && w && dx && dw && lddy >= N && ldx >= K && mi355seg_linear_bwd_supported_f32(lowp, M, N, K), "linear_bwd: unsupported arguments (see mi355seg_linear_bwd_supported_f32)");
    GemmPair gp{};
    // dx = dyf W: A(m, n) = dyf[m][n] (k = n contiguous), B(n, k) = w[n][k] (outer index k contiguous)
    gp.g[0] = GemmArgs{dy, w, dx, nullptr, lddy, 1, 0, 0, K, 1, 0, 0, K, 0, 0, M, K, N, 1, 1.f, 0, 0, 1, N, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
    // dw = dyf^T x: A(n, m) = dyf[m][n] (outer index n contiguous), B(m, k) = x[m][k] (outer index k contiguous); db = A's row sums
    gp.g[1] = GemmArgs{dy, x, dw, nullptr, 1, lddy, 0, 0, ldx, 1, 0, 0, K, 0, 0, N, K, M, 1, 1.f, 0, 0, 1, M, 0, 0, nullptr, db, nullptr, nullptr, 0};
    gp.amul = dmul; gp.agate = dgate;
    return launch_pair(gp, 1, 1, stream);
}

// Two independent batched small GEMMs C_p[b0][b1] = alpha_p A_p B_p in one launch (r6: attention's backward pairs, unetr.py:74-98), bf16
// products, fp32 accumulation; both problems on nb0 x nb1 batches with their own strides.
int mi355seg_gemm_pair_supported_f32(int M0, int N0, int K0, int M1, int N1, int K1, int nb0, int nb1) {
    if (M0 <= 0 || N0 <= 0 || K0 <= 0 || M1 <= 0 || N1 <= 0 || K1 <= 0 || nb0 <= 0 || nb1 <= 0 || K0 % 8 || K1 % 8) return 0;
    return ((long long)cdiv(M0, 32) * cdiv(N0, 32) * nb0 * nb1 <= 4096 && (long long)cdiv(M1, 32) * cdiv(N1, 32) * nb0 * nb1 <= 4096 && 2ll * nb0 * nb1 < 65536) ? 1 : 0;
}
int mi355seg_gemm_pair_lowp_f32(const float* A0, long long a0_rs, long long a0_cs, long long a0_b0, long long a0_b1,
                                const float* B0, long long b0_rs, long long b0_cs, long long b0_b0, long long b0_b1,
                                float* C0, long long c0_rs, long long c0_b0, long long c0_b1, int M0, int N0, int K0, float alpha0,
                                const float* A1, long long a1_rs, long long a1_cs, long long a1_b0, long long a1_b1,
                                const float* B1, long long b1_rs, long long b1_cs, long long b1_b0, long long b1_b1,
                                float* C1, long long c1_rs, long long c1_b0, long long c1_b1, int M1, int N1, int K1, float alpha1,
                                int nb0, int nb1, void* stream) {
    SEG_CHECK_ARG(nb0 > 0 && nb1 > 0, "gemm_pair: bad batch counts");
    GemmPair gp{};
    gp.g[0] = GemmArgs{A0, B0, C0, nullptr, a0_rs, a0_cs, a0_b0, a0_b1, b0_rs, b0_cs, b0_b0, b0_b1, c0_rs, c0_b0, c0_b1, M0, N0, K0, nb1, alpha0, 0, 0, 1, K0, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
    gp.g[1] = GemmArgs{A1, B1, C1, nullptr, a1_rs, a1_cs, a1_b0, a1_b1, b1_rs, b1_cs, b1_b0, b1_b1, c1_rs, c1_b0, c1_b1, M1, N1, K1, nb1, alpha1, 0, 0, 1, K1, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
    return launch_pair(gp, nb0, nb0, stream);
}

int mi355seg_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                               long long rows, int E, float eps, void* stream) {
    SEG_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0 && E > 0 && E <= 64 * LN_MAXJ, "layernorm_fwd: bad arguments (E <= %d)", 64 * LN_MAXJ);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, rows, E, eps);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                               float* dx, float* dgamma, float* dbeta, long long rows, int E, void* stream) {
    SEG_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && rows > 0 && E > 0 && E <= 64 * LN_MAXJ, "layernorm_bwd: bad arguments");
    const int ndx = (int)cdiv(rows, 4);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(ndx + (int)cdiv(E, 32)), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, E, (const float*)nullptr, ndx);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_layernorm_bwd_add_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* addend,
                                   float* dx, float* dgamma, float* dbeta, long long rows, int E, void* stream) {
    SEG_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && rows > 0 && E > 0 && E <= 64 * LN_MAXJ, "layernorm_bwd_add: bad arguments");
    const int ndx = (int)cdiv(rows, 4);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(ndx + (int)cdiv(E, 32)), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, E, addend, ndx);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_softmax_rows_f32(const float* x, float* y, long long rows, int L, void* stream) {
    SEG_CHECK_ARG(x && y && rows > 0 && L > 0 && L <= 64 * LN_MAXJ, "softmax_rows: bad arguments (L <= %d)", 64 * LN_MAXJ);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, L);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, long long rows, int L, void* stream) {
    SEG_CHECK_ARG(y && dy && dx && rows > 0 && L > 0, "softmax_rows_bwd: bad arguments");
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, rows, L);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_softmax_rows_keep_f32(const float* x, const float* keep, float* y, float* yk, long long rows, int L, void* stream) {
    SEG_CHECK_ARG(x && keep && y && yk && rows > 0 && L > 0 && L <= 64 * LN_MAXJ, "softmax_rows_keep: bad arguments (L <= %d)", 64 * LN_MAXJ);
    hipLaunchKernelGGL(softmax_rows_keep_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, keep, y, yk, rows, L);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_softmax_rows_keep_bwd_f32(const float* y, const float* dyk, const float* keep, float* dx, long long rows, int L, void* stream) {
    SEG_CHECK_ARG(y && dyk && keep && dx && rows > 0 && L > 0, "softmax_rows_keep_bwd: bad arguments");
    hipLaunchKernelGGL(softmax_rows_keep_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, dyk, keep, dx, rows, L);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
/* out[c] = sum_rows x[r, c]  (bias gradients of Linear layers) */
int mi355seg_colsum_f32(const float* x, int ldx, long long rows, int C, float* out, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(x && out && rows > 0 && C > 0 && ldx >= C, "colsum: bad arguments");
    if (rows <= 1024) {                        // token matrices (a few hundred rows): one launch, fixed summation order
        hipLaunchKernelGGL(colsum_small_kernel, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, x, ldx, (int)rows, C, out);
        SEG_CHECK_LAUNCH();
        return MI355SEG_OK;
    }
    return channel_sums(x, ldx, rows, C, nullptr, nullptr, out, 0, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
