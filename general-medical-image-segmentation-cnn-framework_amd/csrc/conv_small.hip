// conv_small.hip -- the HBM-bound ends of the network as direct (non-GEMM) kernels:
//   * stem:  Conv3d k3 s1 p1 with Cin <= 4   (unet3d.py enc1conv1 1->32; residual_unet3d.py conv3d_c1_1 4->32)
//   * heads: Conv3d k1 with Cout <= 4        (unet3d.py `conv` 32->2; residual_unet3d.py conv3d_l4 / ds*_1x1)
// Arithmetic intensity is 1-13 flop/B, so these are laid out for
// bandwidth, not for MFMA: every lane moves 16 B, a wavefront's stores are one contiguous
// 1 KiB run of NDHWC, weights sit in LDS / registers, per-channel reductions are register
// sums -> DPP shuffles -> one LDS hop -> per-block partials -> fixed-order second stage.
#include "common.h"
#include "internal.h"

namespace seg {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct SmallGeom { int N, D, H, W, Cin, Cout, ldx, ldy; };

// ---------------------------------------------------------------- stem forward
// thread = (voxel, output-channel quad); LPV = Cout/4 lanes per voxel
template <typename T, int CIN>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
        const float* __restrict__ bias, T* __restrict__ y, float* __restrict__ spart, SmallGeom g) {
    extern __shared__ __attribute__((aligned(16))) float sw[];      // [27][CIN][Cout] then reduction scratch
    const int Cout = g.Cout, LPV = Cout / 4;
    for (int i = threadIdx.x; i < 27 * CIN * Cout; i += 256) {
        int co = i % Cout, r = i / Cout, ci = r % CIN, tap = r / CIN;
        sw[i] = w[((long long)co * CIN + ci) * 27 + tap];
    }
    __syncthreads();
    const int cq = threadIdx.x % LPV, vl = threadIdx.x / LPV, VPB = 256 / LPV;
    const long long nvox = (long long)g.N * g.D * g.H * g.W;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = ld4(bias + cq * 4);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    for (long long v = (long long)blockIdx.x * VPB + vl; v < nvox; v += (long long)gridDim.x * VPB) {
        int xw = (int)(v % g.W); long long r = v / g.W;
        int yh = (int)(r % g.H); r /= g.H;
        int zd = (int)(r % g.D); int n = (int)(r / g.D);
        f32x4 acc = bv;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
            const int iz = zd + dz, iy = yh + dy, ix = xw + dx;
            if ((unsigned)iz < (unsigned)g.D && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) {
                const T* xp = x + ((((long long)n * g.D + iz) * g.H + iy) * g.W + ix) * g.ldx;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const float xv = xp[ci];
                    const f32x4 wv = ld4(sw + (tap * CIN + ci) * Cout + cq * 4);
                    acc += xv * wv;
                }
            }
        }
        st4(y + v * g.ldy + cq * 4, acc);
        s1 += acc; s2 += acc * acc;
    }
    if (spart) {          // per-channel sum / sum of squares of this block (BatchNorm statistics)
        __syncthreads();
        float* red = sw;  // reuse
        for (int j = 0; j < 4; ++j) { red[(threadIdx.x * 2) * 4 + j] = s1[j]; red[(threadIdx.x * 2 + 1) * 4 + j] = s2[j]; }
        __syncthreads();
        if (threadIdx.x < Cout) {
            const int c = threadIdx.x, q = c / 4, j = c % 4;
            float a = 0.f, b = 0.f;
            for (int k = 0; k < VPB; ++k) { a += red[((k * LPV + q) * 2) * 4 + j]; b += red[((k * LPV + q) * 2 + 1) * 4 + j]; }
            spart[((long long)blockIdx.x * Cout + c) * 2] = a;
            spart[((long long)blockIdx.x * Cout + c) * 2 + 1] = b;
        }
    }
}

// ---------------------------------------------------------------- stem wgrad
// grid = (blocks, CIN).  thread = (voxel, cout quad) keeps 27 x 4 accumulators:
// acc[tap][j] += x[v + tap][ci] * dy[v][cq*4 + j].  part[blk][tap][ci][co].
template <typename T, int DUMMY>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
        float* __restrict__ part, SmallGeom g, int lddy) {
    extern __shared__ __attribute__((aligned(16))) float sred[];
    const int Cout = g.Cout, LPV = Cout / 4, VPB = 256 / LPV;
    const int cq = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    const int ci = blockIdx.y, CIN = gridDim.y;
    const long long nvox = (long long)g.N * g.D * g.H * g.W;
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long v = (long long)blockIdx.x * VPB + vl; v < nvox; v += (long long)gridDim.x * VPB) {
        int xw = (int)(v % g.W); long long r = v / g.W;
        int yh = (int)(r % g.H); r /= g.H;
        int zd = (int)(r % g.D); int n = (int)(r / g.D);
        const f32x4 d = ld4(dy + v * lddy + cq * 4);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9 - 1, dyy = (tap / 3) % 3 - 1, dx = tap % 3 - 1;
            const int iz = zd + dz, iy = yh + dyy, ix = xw + dx;
            float xv = 0.f;
            if ((unsigned)iz < (unsigned)g.D && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
                xv = x[((((long long)n * g.D + iz) * g.H + iy) * g.W + ix) * g.ldx + ci];
            acc[tap] += xv * d;
        }
    }
    // reduce over the voxel lanes that share cq: DPP/shuffle inside the wave, LDS across the 4 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = acc[t][j];
            for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
            acc[t][j] = s;
        }
    }
    if (lane < LPV) {
#pragma unroll
        for (int t = 0; t < 27; ++t) st4(sred + ((wave * 27 + t) * LPV + lane) * 4, acc[t]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * Cout; i += 256) {
        const int t = i / Cout, co = i % Cout;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += sred[((w * 27 + t) * LPV + co / 4) * 4 + co % 4];
        part[(((long long)blockIdx.x * 27 + t) * CIN + ci) * Cout + co] = s;
    }
}

// ---------------------------------------------------------------- Cin == 4 stem wgrad (Residual U-Net conv3d_c1_1, residual_unet3d.py:22)
// grid = (blocks, 3 dz planes).  thread = (voxel, cout quad) keeps 9 taps x 4 input channels x 4 output channels of
// accumulators and fetches the four input channels of a tap with ONE 4-element load (the generic kernel above issues one
// scalar load per (tap, channel) and re-reads dy once per input channel): 27 vector loads per voxel instead of 108 scalar ones.
template <typename T>
__global__ __launch_bounds__(256) void stem4_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
        float* __restrict__ part, SmallGeom g, int lddy) {
    extern __shared__ __attribute__((aligned(16))) float sred[];     // [4 waves][9 taps][4 ci][LPV][4]
    const int Cout = g.Cout, LPV = Cout / 4, VPB = 256 / LPV;
    const int cq = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    const int dz = (int)blockIdx.y - 1;
    const long long nvox = (long long)g.N * g.D * g.H * g.W;
    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) acc[t][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long v = (long long)blockIdx.x * VPB + vl; v < nvox; v += (long long)gridDim.x * VPB) {
        int xw = (int)(v % g.W); long long r = v / g.W;
        int yh = (int)(r % g.H); r /= g.H;
        int zd = (int)(r % g.D); int n = (int)(r / g.D);
        const int iz = zd + dz;
        if ((unsigned)iz >= (unsigned)g.D) continue;
        const f32x4 d = ld4(dy + v * lddy + cq * 4);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = yh + t / 3 - 1, ix = xw + t % 3 - 1;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
                xv = ld4(x + ((((long long)n * g.D + iz) * g.H + iy) * g.W + ix) * g.ldx);
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) acc[t][ci] += xv[ci] * d;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = acc[t][ci][j];
                for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
                acc[t][ci][j] = s;
            }
    if (lane < LPV) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) st4(sred + (((wave * 9 + t) * 4 + ci) * LPV + lane) * 4, acc[t][ci]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 4 * Cout; i += 256) {
        const int co = i % Cout, ci = (i / Cout) % 4, t = i / (4 * Cout);
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += sred[(((w * 9 + t) * 4 + ci) * LPV + co / 4) * 4 + co % 4];
        part[(((long long)blockIdx.x * 27 + (dz + 1) * 9 + t) * 4 + ci) * Cout + co] = s;
    }
}

// ---------------------------------------------------------------- Cin == 1 stem, LDS-tiled
// The 1-channel NDHWC input is a plain 3-D volume: a (2+2) x (4+2) x (TX+2) halo tile sits in LDS,
// every tap is a fixed LDS offset (no per-tap bounds / address arithmetic), and the 27 x 4 weights of
// the lane's output-channel quad live in registers.  thread = (x position, channel quad); it walks the
// tile's 8 x-lines.  TX = 256 / (Cout/4).
constexpr int S1_TZ = 2, S1_TY = 4;

template <typename T>
__device__ __forceinline__ void s1_stage(float* tile, const T* __restrict__ x, int ldx, int n, int z0, int y0, int x0, int TX,
                                         int D, int H, int W) {
    const int HX = TX + 2, HY = S1_TY + 2, HZ = S1_TZ + 2;
    for (int p = threadIdx.x; p < HX * HY * HZ; p += 256) {
        const int hx = p % HX, r = p / HX, hy = r % HY, hz = r / HY;
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        float v = 0.f;
        if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
            v = ld1(x + ((((long long)n * D + gz) * H + gy) * W + gx) * ldx);
        tile[p] = v;
    }
}

// STATS (r5): BatchNorm statistics of y from the kernel itself instead of a pass over y (94 us on the 128^3 stem).  Plain fp32 sums of
// y and y^2 cancel when |mean| >> std, so every thread sums (y - pivot) and (y - pivot)^2 about ITS OWN first output (a value of the
// channel's distribution: |mean - pivot| ~ std), turns them into (n, mean, M2 about its mean) at the end, and the block merges its
// threads' triples pairwise-exactly (Chan et al.) into spart[blk][c] = {sum, M2, n} -- the format of the MFMA kernels' per-tile
// triples, finalised by the same two-stage fp64 reduction (tile_stats_finalize2).
template <typename T, bool STATS>
__global__ __launch_bounds__(256, 3) void stem1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
        const float* __restrict__ bias, T* __restrict__ y, float* __restrict__ spart, SmallGeom g, int ntiles, unsigned* __restrict__ amax_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Cout = g.Cout, LPV = Cout / 4, TX = 256 / LPV, HX = TX + 2, HY = S1_TY + 2;
    const int cq = threadIdx.x % LPV, xs = threadIdx.x / LPV;
    f32x4 wr[27];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[t][j] = w[(long long)(cq * 4 + j) * 27 + t];
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = ld4(bias + cq * 4);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, piv = s1;
    float ymax = 0.f;                    // max |y| of this thread's outputs (STATS launches: the next layer's prologue bound)
    int cnt = 0;
    const int ntx = g.W / TX, nty = g.H / S1_TY, ntz = g.D / S1_TZ;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int mt = tile;
        const int txi = mt % ntx; mt /= ntx;
        const int tyi = mt % nty; mt /= nty;
        const int tzi = mt % ntz; const int n = mt / ntz;
        const int x0 = txi * TX, y0 = tyi * S1_TY, z0 = tzi * S1_TZ;
        __syncthreads();
        s1_stage(sm, x, g.ldx, n, z0, y0, x0, TX, g.D, g.H, g.W);
        __syncthreads();
        // (two lines per trip, three workgroups per CU: fully unrolled the eight lines' 216 LDS reads are hoisted and the kernel takes 256+
        // registers -- one or two waves per SIMD on a kernel that waits on HBM stores)
#pragma unroll 2
        for (int line = 0; line < S1_TZ * S1_TY; ++line) {
            const int lz = line / S1_TY, ly = line % S1_TY;
            const float* tp = sm + (lz * HY + ly) * HX + xs;
            f32x4 acc = bv;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
                acc += tp[(dz * HY + dy) * HX + dx] * wr[t];
            }
            const long long v = (((long long)n * g.D + z0 + lz) * g.H + y0 + ly) * g.W + x0 + xs;
            st4(y + v * g.ldy + cq * 4, acc);
            if (STATS) {
                if (cnt == 0 && line == 0) piv = acc;
                const f32x4 d = acc - piv;
                s1 += d; s2 += d * d;
                ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(acc[0]), fabsf(acc[1]))), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
            }
        }
        cnt += S1_TZ * S1_TY;
    }
    if (STATS) {
        __syncthreads();
        float* red = sm;                   // [256][9]: mean quad, M2 quad, n
        const float nt = (float)cnt, inv = cnt > 0 ? 1.f / nt : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dm = s1[j] * inv;
            red[threadIdx.x * 9 + j] = piv[j] + dm;
            red[threadIdx.x * 9 + 4 + j] = fmaxf(s2[j] - s1[j] * dm, 0.f);
        }
        red[threadIdx.x * 9 + 8] = nt;
        __syncthreads();
        if (threadIdx.x < Cout) {
            const int c = threadIdx.x, q = c / 4, j = c % 4;
            double na = 0.0, ma = 0.0, m2a = 0.0;
            for (int k = 0; k < TX; ++k) {
                const float* e = red + (k * LPV + q) * 9;
                const double nb = (double)e[8];
                if (nb <= 0.0) continue;
                const double mb = (double)e[j], m2b = (double)e[4 + j], nab = na + nb, dl = mb - ma;
                ma += dl * nb / nab;
                m2a += m2b + dl * dl * na * nb / nab;
                na = nab;
            }
            float* dst = spart + ((long long)blockIdx.x * Cout + c) * 3;
            dst[0] = (float)(na * ma); dst[1] = (float)m2a; dst[2] = (float)na;
        }
        if (amax_out) { __syncthreads(); block_amax_commit(ymax, amax_out); }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void stem1_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
        float* __restrict__ part, SmallGeom g, int lddy, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Cout = g.Cout, LPV = Cout / 4, TX = 256 / LPV, HX = TX + 2, HY = S1_TY + 2;
    const int cq = threadIdx.x % LPV, xs = threadIdx.x / LPV;
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntx = g.W / TX, nty = g.H / S1_TY, ntz = g.D / S1_TZ;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int mt = tile;
        const int txi = mt % ntx; mt /= ntx;
        const int tyi = mt % nty; mt /= nty;
        const int tzi = mt % ntz; const int n = mt / ntz;
        const int x0 = txi * TX, y0 = tyi * S1_TY, z0 = tzi * S1_TZ;
        // the tile's dy (8 x 16 bytes per thread) is requested before the halo staging so both latencies overlap
        f32x4 dreg[S1_TZ * S1_TY];
#pragma unroll
        for (int line = 0; line < S1_TZ * S1_TY; ++line) {
            const int lz = line / S1_TY, ly = line % S1_TY;
            const long long v = (((long long)n * g.D + z0 + lz) * g.H + y0 + ly) * g.W + x0 + xs;
            dreg[line] = ld4(dy + v * lddy + cq * 4);
        }
        __syncthreads();
        s1_stage(sm, x, g.ldx, n, z0, y0, x0, TX, g.D, g.H, g.W);
        __syncthreads();
#pragma unroll
        for (int line = 0; line < S1_TZ * S1_TY; ++line) {
            const int lz = line / S1_TY, ly = line % S1_TY;
            const float* tp = sm + (lz * HY + ly) * HX + xs;
            const f32x4 d = dreg[line];
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int dz = t / 9, dyy = (t / 3) % 3, dx = t % 3;
                acc[t] += tp[(dz * HY + dyy) * HX + dx] * d;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = acc[t][j];
            for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
            acc[t][j] = s;
        }
    }
    __syncthreads();
    if (lane < LPV) {
#pragma unroll
        for (int t = 0; t < 27; ++t) st4(sm + ((wave * 27 + t) * LPV + lane) * 4, acc[t]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * Cout; i += 256) {
        const int t = i / Cout, co = i % Cout;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += sm[((w * 27 + t) * LPV + co / 4) * 4 + co % 4];
        part[((long long)blockIdx.x * 27 + t) * Cout + co] = s;
    }
}

// ---------------------------------------------------------------- Cin == 1 stem wgrad with the norm backward's APPLY half as its prologue (r5)
// unet3d.py:80-89 (enc1conv1 -> enc1norm1 -> enc1relu1): the stem's input needs no gradient, so d(conv1 output) = rstd gamma (dz - s1 / n -
// xhat s2 / n) has ONE consumer, this weight gradient.  It is formed here from d(activation) and the pre-norm tensor (norm_act_bwd_apply_kernel's
// expression, operation for operation) instead of being written (537 MB at 2 x 128^3) and read back; its column sums (conv1's bias gradient)
// come out of the same pass.  part[blk][tap][co] as stem1_wgrad_kernel, cpart[blk][co].
// floats ahead of the constant table: the staged halo tile and the end-of-kernel reduction scratch share this region, so it holds the larger
// of the two (for Cout <= 8 the tile -- 130 or 258 voxels wide -- is the larger one)
__host__ __device__ static inline int stem_bn_scratch_floats(int Cout) {
    const int TX = 256 / (Cout / 4);
    const int tile = (TX + 2) * (S1_TY + 2) * (S1_TZ + 2), red = 4 * 28 * Cout;
    return ((tile > red ? tile : red) + 3) & ~3;
}
struct StemBn { const float* da; int ldda; const float* y; int ldy; const float* mean; const float* rstd; const float* gamma; const float* beta;
                const float* s1; const float* s2; int act; float slope; float invM; };

template <int ACT>
__device__ __forceinline__ void stem1_wgrad_bn_body(const float* __restrict__ x, const StemBn& b, float* __restrict__ part, float* __restrict__ cpart,
                                                    const SmallGeom& g, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Cout = g.Cout, LPV = Cout / 4, TX = 256 / LPV, HX = TX + 2, HY = S1_TY + 2;
    const int cq = threadIdx.x % LPV, xs = threadIdx.x / LPV;
    // per-channel constants in LDS behind the reduction scratch (six quads per thread and tile, live only while d is formed: held in
    // registers across the 27-tap loop they spill)
    float* const ctab = sm + stem_bn_scratch_floats(Cout);
    float* const dtab = ctab + 6 * Cout;              // [8 lines][256 threads] quads of d(conv output)
    for (int c = threadIdx.x; c < Cout; c += 256) {
        ctab[c] = b.mean[c]; ctab[Cout + c] = b.rstd[c]; ctab[2 * Cout + c] = b.gamma ? b.gamma[c] : 1.f; ctab[3 * Cout + c] = b.beta ? b.beta[c] : 0.f;
        ctab[4 * Cout + c] = b.s1[c] * b.invM; ctab[5 * Cout + c] = b.s2[c] * b.invM;
    }
    f32x4 acc[27], col = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntx = g.W / TX, nty = g.H / S1_TY, ntz = g.D / S1_TZ;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int mt = tile;
        const int txi = mt % ntx; mt /= ntx;
        const int tyi = mt % nty; mt /= nty;
        const int tzi = mt % ntz; const int n = mt / ntz;
        const int x0 = txi * TX, y0 = tyi * S1_TY, z0 = tzi * S1_TZ;
        f32x4 dreg[S1_TZ * S1_TY], yreg[S1_TZ * S1_TY];
#pragma unroll
        for (int line = 0; line < S1_TZ * S1_TY; ++line) {
            const int lz = line / S1_TY, ly = line % S1_TY;
            const long long v = (((long long)n * g.D + z0 + lz) * g.H + y0 + ly) * g.W + x0 + xs;
            dreg[line] = ld4(b.da + v * b.ldda + cq * 4);
            yreg[line] = ld4(b.y + v * b.ldy + cq * 4);
        }
        __syncthreads();
        s1_stage(sm, x, g.ldx, n, z0, y0, x0, TX, g.D, g.H, g.W);
        __syncthreads();
        // d(conv output) of the eight lines first, parked in this thread's own LDS slots (the pre-norm values, d(activation) and the
        // per-channel constants are dead before the 27-tap loop starts; the loop then runs two lines per trip as stem1_fwd_kernel does --
        // fully unrolled with everything in registers the kernel spills 50-250 VGPRs)
        {
            const f32x4 m = ld4(ctab + cq * 4), rs = ld4(ctab + Cout + cq * 4), ga = ld4(ctab + 2 * Cout + cq * 4), be = ld4(ctab + 3 * Cout + cq * 4);
            const f32x4 k1 = ld4(ctab + 4 * Cout + cq * 4), k2 = ld4(ctab + 5 * Cout + cq * 4);
#pragma unroll
            for (int line = 0; line < S1_TZ * S1_TY; ++line) {
                f32x4 d;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (yreg[line][j] - m[j]) * rs[j];
                    const float z = fmaf(xh, ga[j], be[j]);
                    const float dz = dreg[line][j] * act_grad(z, ACT >= 0 ? ACT : b.act, b.slope);
                    d[j] = ga[j] * rs[j] * (dz - k1[j] - xh * k2[j]);
                }
                col += d;
                st4(dtab + (line * 256 + threadIdx.x) * 4, d);
            }
        }
#pragma unroll 2
        for (int line = 0; line < S1_TZ * S1_TY; ++line) {
            const int lz = line / S1_TY, ly = line % S1_TY;
            const float* tp = sm + (lz * HY + ly) * HX + xs;
            const f32x4 d = ld4(dtab + (line * 256 + threadIdx.x) * 4);
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int dz = t / 9, dyy = (t / 3) % 3, dx = t % 3;
                acc[t] += tp[(dz * HY + dyy) * HX + dx] * d;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = acc[t][j];
            for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
            acc[t][j] = s;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = col[j];
        for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
        col[j] = s;
    }
    __syncthreads();
    if (lane < LPV) {
#pragma unroll
        for (int t = 0; t < 27; ++t) st4(sm + ((wave * 28 + t) * LPV + lane) * 4, acc[t]);
        st4(sm + ((wave * 28 + 27) * LPV + lane) * 4, col);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 28 * Cout; i += 256) {
        const int t = i / Cout, co = i % Cout;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += sm[((w * 28 + t) * LPV + co / 4) * 4 + co % 4];
        if (t < 27) part[((long long)blockIdx.x * 27 + t) * Cout + co] = s;
        else cpart[(long long)blockIdx.x * Cout + co] = s;
    }
}
// (r6) the U-Net's launches run ReLU: that instantiation has no per-element activation switch
__global__ __launch_bounds__(256, 2) void stem1_wgrad_bn_kernel(const float* __restrict__ x, StemBn b, float* __restrict__ part, float* __restrict__ cpart,
                                                               SmallGeom g, int ntiles) {
    if (b.act == MI355SEG_ACT_RELU) stem1_wgrad_bn_body<MI355SEG_ACT_RELU>(x, b, part, cpart, g, ntiles);
    else stem1_wgrad_bn_body<-1>(x, b, part, cpart, g, ntiles);
}

__global__ __launch_bounds__(64) void colpart_finalize_kernel(const float* __restrict__ cpart, int nblk, int C, float* __restrict__ out) {
    const int c = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) s += (double)cpart[(long long)k * C + c];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[c] = (float)s;
}

// ---------------------------------------------------------------- Cin == 1, k5 p2 stem wgrad (V-Net InputTransition, vnet3d.py:47)
// grid = (blocks, 5 dz planes).  As stem1_wgrad_kernel, with the 25 (dy, dx) taps of one dz plane per block: the LDS tile is
// the z-shifted slab S1_TZ x (S1_TY+4) x (TX+4) of the one-channel input.  part[blk][tap][0][co], tap = dz*25 + dy*5 + dx.
template <typename T>
__global__ __launch_bounds__(256) void stem1k5_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
        float* __restrict__ part, SmallGeom g, int lddy, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Cout = g.Cout, LPV = Cout / 4, TX = 256 / LPV, HX = TX + 4, HY = S1_TY + 4;
    const int cq = threadIdx.x % LPV, xs = threadIdx.x / LPV;
    const int dzp = blockIdx.y;
    f32x4 acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntx = g.W / TX, nty = g.H / S1_TY, ntz = g.D / S1_TZ;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int mt = tile;
        const int txi = mt % ntx; mt /= ntx;
        const int tyi = mt % nty; mt /= nty;
        const int tzi = mt % ntz; const int n = mt / ntz;
        const int x0 = txi * TX, y0 = tyi * S1_TY, z0 = tzi * S1_TZ;
        f32x4 dreg[S1_TZ * S1_TY];
#pragma unroll
        for (int line = 0; line < S1_TZ * S1_TY; ++line) {
            const int lz = line / S1_TY, ly = line % S1_TY;
            const long long v = (((long long)n * g.D + z0 + lz) * g.H + y0 + ly) * g.W + x0 + xs;
            dreg[line] = ld4(dy + v * lddy + cq * 4);
        }
        __syncthreads();
        for (int p = threadIdx.x; p < HX * HY * S1_TZ; p += 256) {
            const int hx = p % HX, r = p / HX, hy = r % HY, hz = r / HY;
            const int gz = z0 + hz + dzp - 2, gy = y0 - 2 + hy, gx = x0 - 2 + hx;
            float v = 0.f;
            if ((unsigned)gz < (unsigned)g.D && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W)
                v = ld1(x + ((((long long)n * g.D + gz) * g.H + gy) * g.W + gx) * g.ldx);
            sm[p] = v;
        }
        __syncthreads();
#pragma unroll
        for (int line = 0; line < S1_TZ * S1_TY; ++line) {
            const int lz = line / S1_TY, ly = line % S1_TY;
            const float* tp = sm + (lz * HY + ly) * HX + xs;
            const f32x4 d = dreg[line];
#pragma unroll
            for (int t = 0; t < 25; ++t) acc[t] += tp[(t / 5) * HX + t % 5] * d;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 25; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = acc[t][j];
            for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
            acc[t][j] = s;
        }
    __syncthreads();
    if (lane < LPV) {
#pragma unroll
        for (int t = 0; t < 25; ++t) st4(sm + ((wave * 25 + t) * LPV + lane) * 4, acc[t]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 25 * Cout; i += 256) {
        const int t = i / Cout, co = i % Cout;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += sm[((w * 25 + t) * LPV + co / 4) * 4 + co % 4];
        part[((long long)blockIdx.x * 125 + dzp * 25 + t) * Cout + co] = s;
    }
}

// ---------------------------------------------------------------- tiny pointwise convolutions (k1, Cin <= 4, Cout <= 4)
// V-Net's classes -> classes output convolution (vnet3d.py:113, 2 -> 2): thread = voxel, weights in registers.
struct TinyPw { int Cin, Cout, ldx, ldy; long long nvox; };
template <typename T>
__global__ __launch_bounds__(256) void tinypw_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ y, TinyPw g) {
    float wr[4][4], br[4];
    for (int co = 0; co < 4; ++co) {
        br[co] = (bias && co < g.Cout) ? bias[co] : 0.f;
        for (int ci = 0; ci < 4; ++ci) wr[co][ci] = (co < g.Cout && ci < g.Cin) ? w[co * g.Cin + ci] : 0.f;
    }
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < g.nvox; v += (long long)gridDim.x * 256) {
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ci = 0; ci < g.Cin; ++ci) xv[ci] = ld1(x + v * g.ldx + ci);
        for (int co = 0; co < g.Cout; ++co)
            st1(y + v * g.ldy + co, br[co] + xv[0] * wr[co][0] + xv[1] * wr[co][1] + xv[2] * wr[co][2] + xv[3] * wr[co][3]);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void tinypw_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, TinyPw g) {
    float wr[4][4];
    for (int co = 0; co < 4; ++co)
        for (int ci = 0; ci < 4; ++ci) wr[co][ci] = (co < g.Cout && ci < g.Cin) ? w[co * g.Cin + ci] : 0.f;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < g.nvox; v += (long long)gridDim.x * 256) {
        float d[4] = {0.f, 0.f, 0.f, 0.f};
        for (int co = 0; co < g.Cout; ++co) d[co] = ld1(dy + v * g.ldy + co);
        for (int ci = 0; ci < g.Cin; ++ci)
            st1(dx + v * g.ldx + ci, d[0] * wr[0][ci] + d[1] * wr[1][ci] + d[2] * wr[2][ci] + d[3] * wr[3][ci]);
    }
}
// part[blk][0][ci][co]: per-block sums in fixed order (wave shuffle, then the four waves through LDS)
template <typename T>
__global__ __launch_bounds__(256) void tinypw_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, TinyPw g) {
    __shared__ float sh[4][16];
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < g.nvox; v += (long long)gridDim.x * 256) {
        float xv[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ci = 0; ci < g.Cin; ++ci) xv[ci] = ld1(x + v * g.ldx + ci);
        for (int co = 0; co < g.Cout; ++co) d[co] = ld1(dy + v * g.ldy + co);
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int co = 0; co < 4; ++co) acc[ci * 4 + co] += xv[ci] * d[co];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float s = wave_sum(acc[i]); if (lane == 0) sh[wave][i] = s; }
    __syncthreads();
    if (threadIdx.x < g.Cin * g.Cout) {
        const int ci = threadIdx.x / g.Cout, co = threadIdx.x % g.Cout;
        part[((long long)blockIdx.x * g.Cin + ci) * g.Cout + co] = sh[0][ci * 4 + co] + sh[1][ci * 4 + co] + sh[2][ci * 4 + co] + sh[3][ci * 4 + co];
    }
}

static bool stem1_tiled_ok(const SmallGeom& g) {
    if (g.Cin != 1) return false;
    const int TX = 256 / (g.Cout / 4);
    return g.W % TX == 0 && g.H % S1_TY == 0 && g.D % S1_TZ == 0;
}
static size_t stem1_lds(int Cout) {
    const int TX = 256 / (Cout / 4);
    size_t a = (size_t)(TX + 2) * (S1_TY + 2) * (S1_TZ + 2) * 4, b = 256 * 9 * 4, c = (size_t)4 * 27 * Cout * 4;
    size_t m = a > b ? a : b;
    return m > c ? m : c;
}

bool stem_wgrad_bn_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    if (!(k == 3 && stride == 1 && pad == 1 && Cin == 1 && Cout % 4 == 0 && Cout >= 4 && Cout <= 64 && ((Cout / 4) & (Cout / 4 - 1)) == 0)) return false;
    SmallGeom g{N, D, H, W, Cin, Cout, 1, Cout};
    return stem1_tiled_ok(g) && ((size_t)stem_bn_scratch_floats(Cout) + 6 * Cout) * 4 + (size_t)S1_TZ * S1_TY * 256 * 16 <= 80 * 1024;
}

int stem_wgrad_bn(const float* da, int ldda, const float* y, int ldy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  int act, float slope, const float* s1, const float* s2, const float* x, int ldx, float* dw, float* db,
                  int N, int D, int H, int W, int Cout, void* ws, size_t ws_bytes, hipStream_t st) {
    SmallGeom g{N, D, H, W, 1, Cout, ldx, 0};
    const long long nvox = (long long)N * D * H * W;
    const int ntiles = (int)(nvox / ((long long)S1_TZ * S1_TY * (256 / (Cout / 4))));
    const int nb = ntiles < 512 ? ntiles : 512;                  // two workgroups per CU
    Carver cv(ws);
    float* part = cv.take<float>((size_t)nb * 27 * Cout);
    float* cpart = cv.take<float>((size_t)nb * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    StemBn b{da, ldda, y, ldy, mean, rstd, gamma, beta, s1, s2, act, slope, 1.f / (float)nvox};
    const size_t lds = ((size_t)stem_bn_scratch_floats(Cout) + 6 * Cout) * 4 + (size_t)S1_TZ * S1_TY * 256 * 16;   // tile / reduction scratch | constants | parked d (32 KB)
    {
        ProfScope ps(PF_DIRECT, 2.0 * nvox * 27.0 * Cout, 4.0 * nvox * (1 + 2.0 * Cout), st);
        hipLaunchKernelGGL(stem1_wgrad_bn_kernel, dim3(nb), dim3(256), lds, st, x, b, part, cpart, g, ntiles);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, nb, 27, 1, Cout, 0, st);
    SEG_CHECK_LAUNCH();
    if (db) {
        hipLaunchKernelGGL(colpart_finalize_kernel, dim3(Cout), dim3(64), 0, st, cpart, nb, Cout, db);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

// ---------------------------------------------------------------- pointwise (k1) small-Cout head
// thread = (voxel, input-channel quad); LPV = Cin/4 lanes per voxel (<= 64)
template <typename T, int COUT>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
        const float* __restrict__ bias, T* __restrict__ y, SmallGeom g) {
    const int LPV = g.Cin / 4, VPB = 256 / LPV;
    const int c4 = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    const long long nvox = (long long)g.N * g.D * g.H * g.W;
    f32x4 wr[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) wr[co] = ld4(w + (long long)co * g.Cin + c4 * 4);
    for (long long v = (long long)blockIdx.x * VPB + vl; v < nvox; v += (long long)gridDim.x * VPB) {
        const f32x4 xv = ld4(x + v * g.ldx + c4 * 4);
        float o[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float s = xv[0] * wr[co][0] + xv[1] * wr[co][1] + xv[2] * wr[co][2] + xv[3] * wr[co][3];
            for (int off = LPV >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            o[co] = s + (bias ? bias[co] : 0.f);
        }
        if (c4 == 0) {
#pragma unroll
            for (int co = 0; co < COUT; ++co) y[v * g.ldy + co] = o[co];
        }
    }
}

template <typename T, int COUT>
__global__ __launch_bounds__(256) void head_dgrad_kernel(const T* __restrict__ dy, int lddy, const float* __restrict__ w,
        T* __restrict__ dx, SmallGeom g) {
    const int LPV = g.Cin / 4, VPB = 256 / LPV;
    const int c4 = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    const long long nvox = (long long)g.N * g.D * g.H * g.W;
    f32x4 wr[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) wr[co] = ld4(w + (long long)co * g.Cin + c4 * 4);
    for (long long v = (long long)blockIdx.x * VPB + vl; v < nvox; v += (long long)gridDim.x * VPB) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc += dy[v * lddy + co] * wr[co];
        st4(dx + v * g.ldx + c4 * 4, acc);
    }
}

// part[blk][0][ci][co] = sum over the block's voxels of x[v][ci] * dy[v][co]
template <typename T, int COUT>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, int lddy,
        float* __restrict__ part, SmallGeom g) {
    __shared__ float sred[4 * 64 * COUT * 4];
    const int LPV = g.Cin / 4, VPB = 256 / LPV;
    const int c4 = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    const long long nvox = (long long)g.N * g.D * g.H * g.W;
    f32x4 acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long v = (long long)blockIdx.x * VPB + vl; v < nvox; v += (long long)gridDim.x * VPB) {
        const f32x4 xv = ld4(x + v * g.ldx + c4 * 4);
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] += dy[v * lddy + co] * xv;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = acc[co][j];
            for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
            acc[co][j] = s;
        }
    // lanes [0, min(LPV,64)) of each wave hold that wave's totals (if LPV == 64 every lane is its own c4)
    const int nl = LPV < 64 ? LPV : 64;
    if (lane < nl) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) st4(sred + ((wave * 64 + lane) * COUT + co) * 4, acc[co]);
    }
    __syncthreads();
    // when LPV == 64 a wave covers exactly one voxel row of quads; waves then hold different voxels of the same quads
    for (int i = threadIdx.x; i < g.Cin * COUT; i += 256) {
        const int ci = i / COUT, co = i % COUT;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += sred[((w * 64 + (ci / 4) % 64) * COUT + co) * 4 + ci % 4];
        part[((long long)blockIdx.x * g.Cin + ci) * COUT + co] = s;
    }
}

static int small_grid(long long nvox, int vpb) {
    long long b = (nvox + vpb - 1) / vpb;
    b = (b + 7) / 8;                       // >= 8 voxel rounds per block
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

// ---------------------------------------------------------------- small-channel wgrad, any k / stride / pad
struct SmallConv { int N, D, H, W, Do, Ho, Wo, Cin, Cout, k, stride, pad, T, ldx, lddy; };

// (A) few input channels: thread = (output voxel, VW output channels); grid = (blocks, Cin, tap groups);
//     acc[tap of the group][VW] += x[in(v, tap)][ci] * dy[v][co..co+VW).  part[blk][tap][ci][co]
template <typename T, int TPG, int VW>
__global__ __launch_bounds__(256) void smallcin_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
        float* __restrict__ part, SmallConv g) {
    extern __shared__ __attribute__((aligned(16))) float sred[];
    const int LPV = g.Cout / VW, VPB = 256 / LPV;
    const int cq = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    const int ci = blockIdx.y, tg = blockIdx.z;
    const long long nvox = (long long)g.N * g.Do * g.Ho * g.Wo;
    float acc[TPG][VW];
#pragma unroll
    for (int t = 0; t < TPG; ++t)
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[t][j] = 0.f;
    for (long long v = (long long)blockIdx.x * VPB + vl; v < nvox; v += (long long)gridDim.x * VPB) {
        int ow = (int)(v % g.Wo); long long r = v / g.Wo;
        int oh = (int)(r % g.Ho); r /= g.Ho;
        int od = (int)(r % g.Do); int n = (int)(r / g.Do);
        float d[VW];
#pragma unroll
        for (int j = 0; j < VW; ++j) d[j] = dy[v * g.lddy + cq * VW + j];
#pragma unroll
        for (int t = 0; t < TPG; ++t) {
            const int tap = tg * TPG + t;
            const int kw = tap % g.k, kh = (tap / g.k) % g.k, kd = tap / (g.k * g.k);
            const int iz = od * g.stride - g.pad + kd, iy = oh * g.stride - g.pad + kh, ix = ow * g.stride - g.pad + kw;
            float xv = 0.f;
            if (tap < g.T && (unsigned)iz < (unsigned)g.D && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
                xv = x[((((long long)n * g.D + iz) * g.H + iy) * g.W + ix) * g.ldx + ci];
#pragma unroll
            for (int j = 0; j < VW; ++j) acc[t][j] += xv * d[j];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < TPG; ++t)
#pragma unroll
        for (int j = 0; j < VW; ++j) {
            float s = acc[t][j];
            for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
            acc[t][j] = s;
        }
    if (lane < LPV) {
#pragma unroll
        for (int t = 0; t < TPG; ++t)
#pragma unroll
            for (int j = 0; j < VW; ++j) sred[((wave * TPG + t) * LPV + lane) * VW + j] = acc[t][j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TPG * g.Cout; i += 256) {
        const int t = i / g.Cout, co = i % g.Cout, tap = tg * TPG + t;
        if (tap >= g.T) continue;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += sred[((w * TPG + t) * LPV + co / VW) * VW + co % VW];
        part[(((long long)blockIdx.x * g.T + tap) * g.Cin + ci) * g.Cout + co] = s;
    }
}

// (B) few output channels: thread = (output voxel, input-channel quad); grid = (blocks, tap groups);
//     acc[tap][co] (f32x4 over the quad) += dy[v][co] * x[in(v, tap)][c4*4 .. +3].  part[blk][tap][ci][co]
template <typename T, int COUT, int TPG>
__global__ __launch_bounds__(256) void smallcout_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
        float* __restrict__ part, SmallConv g) {
    extern __shared__ __attribute__((aligned(16))) float sred[];
    const int LPV = g.Cin / 4, VPB = 256 / LPV;
    const int c4 = threadIdx.x % LPV, vl = threadIdx.x / LPV;
    const int tg = blockIdx.y;
    const long long nvox = (long long)g.N * g.Do * g.Ho * g.Wo;
    f32x4 acc[TPG][COUT];
#pragma unroll
    for (int t = 0; t < TPG; ++t)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[t][co] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long v = (long long)blockIdx.x * VPB + vl; v < nvox; v += (long long)gridDim.x * VPB) {
        int ow = (int)(v % g.Wo); long long r = v / g.Wo;
        int oh = (int)(r % g.Ho); r /= g.Ho;
        int od = (int)(r % g.Do); int n = (int)(r / g.Do);
        float d[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) d[co] = dy[v * g.lddy + co];
#pragma unroll
        for (int t = 0; t < TPG; ++t) {
            const int tap = tg * TPG + t;
            const int kw = tap % g.k, kh = (tap / g.k) % g.k, kd = tap / (g.k * g.k);
            const int iz = od * g.stride - g.pad + kd, iy = oh * g.stride - g.pad + kh, ix = ow * g.stride - g.pad + kw;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (tap < g.T && (unsigned)iz < (unsigned)g.D && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
                xv = ld4(x + ((((long long)n * g.D + iz) * g.H + iy) * g.W + ix) * g.ldx + c4 * 4);
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[t][co] += d[co] * xv;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < TPG; ++t)
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = acc[t][co][j];
                for (int o = 32; o >= LPV; o >>= 1) s += __shfl_xor(s, o, 64);
                acc[t][co][j] = s;
            }
    if (lane < LPV) {
#pragma unroll
        for (int t = 0; t < TPG; ++t)
#pragma unroll
            for (int co = 0; co < COUT; ++co) *reinterpret_cast<f32x4*>(sred + (((wave * TPG + t) * COUT + co) * LPV + lane) * 4) = acc[t][co];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TPG * g.Cin * COUT; i += 256) {
        const int co = i % COUT, ci = (i / COUT) % g.Cin, t = i / (COUT * g.Cin), tap = tg * TPG + t;
        if (tap >= g.T) continue;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += sred[(((w * TPG + t) * COUT + co) * LPV + (ci / 4)) * 4 + ci % 4];
        part[(((long long)blockIdx.x * g.T + tap) * g.Cin + ci) * COUT + co] = s;
    }
}

static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool smallcin_wgrad_supported(int Cin, int Cout, int k) {
    if (Cin > 4 || k > 7) return false;
    const int vw = (Cout % 4 == 0) ? 4 : 1, lpv = Cout / vw;
    return pow2(lpv) && lpv <= 64;
}
bool smallcout_wgrad_supported(int Cin, int Cout, int k, int ldx) {
    const int lpv = Cin / 4;
    return (Cout == 2 || Cout == 4) && Cin % 4 == 0 && pow2(lpv) && lpv <= 16 && ldx % 4 == 0 && k <= 7;   // LDS: 4*TPG*COUT*lpv*16 B
}

size_t small_wgrad_ws_bytes(int Cin, int Cout, int k) {
    return align_up((size_t)512 * k * k * k * Cin * Cout * sizeof(float), 256) + 1024;
}

template <typename T>
int smallcin_wgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout,
                   int k, int stride, int pad, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    SmallConv g{N, D, H, W, (D + 2 * pad - k) / stride + 1, (H + 2 * pad - k) / stride + 1, (W + 2 * pad - k) / stride + 1,
                Cin, Cout, k, stride, pad, k * k * k, ldx, lddy};
    const long long nvox = (long long)N * g.Do * g.Ho * g.Wo;
    if (Cin == 1 && k == 5 && stride == 1 && pad == 2 && Cout % 4 == 0 && Cout <= 64 && (256 % (Cout / 4)) == 0 && lddy % 4 == 0 &&
        ((uintptr_t)dy % (4 * sizeof(T))) == 0) {
        SmallGeom sg{N, D, H, W, 1, Cout, ldx, 0};
        const int TX = 256 / (Cout / 4);
        if (W % TX == 0 && H % S1_TY == 0 && D % S1_TZ == 0) {      // LDS-tiled k5 stem wgrad: one block per (tile strip, dz plane)
            const int ntiles = (int)(nvox / ((long long)S1_TZ * S1_TY * TX));
            int nb = ntiles < 128 ? ntiles : 128;
            SEG_CHECK_WS((size_t)nb * 125 * Cout * sizeof(float), ws_bytes);
            float* part5 = (float*)ws;
            size_t lds5 = (size_t)(TX + 4) * (S1_TY + 4) * S1_TZ * 4, red5 = (size_t)4 * 25 * Cout * 4;
            if (red5 > lds5) lds5 = red5;
            {
                ProfScope ps(PF_DIRECT, 2.0 * nvox * 125.0 * Cout, (double)sizeof(T) * nvox * (1 + Cout), st);
                hipLaunchKernelGGL((stem1k5_wgrad_kernel<T>), dim3(nb, 5), dim3(256), lds5, st, x, dy, part5, sg, lddy, ntiles);
                SEG_CHECK_LAUNCH();
            }
            wgrad_reduce(part5, dw, nb, 125, 1, Cout, accumulate, st);
            SEG_CHECK_LAUNCH();
            return MI355SEG_OK;
        }
    }
    const int vw = (Cout % 4 == 0) ? 4 : 1, lpv = Cout / vw;
    int nblk = small_grid(nvox, 256 / lpv);
    if (nblk > 512) nblk = 512;
    SEG_CHECK_WS((size_t)nblk * g.T * Cin * Cout * sizeof(float), ws_bytes);
    float* part = (float*)ws;
    const int TPG = 25;
    dim3 grid(nblk, Cin, (g.T + TPG - 1) / TPG);
    size_t lds = (size_t)4 * TPG * Cout * 4;
    {
        ProfScope ps(PF_DIRECT, 2.0 * nvox * g.T * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
        if (vw == 4) hipLaunchKernelGGL((smallcin_wgrad_kernel<T, 25, 4>), grid, dim3(256), lds, st, x, dy, part, g);
        else hipLaunchKernelGGL((smallcin_wgrad_kernel<T, 25, 1>), grid, dim3(256), lds, st, x, dy, part, g);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, nblk, g.T, Cin, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

template <typename T>
int smallcout_wgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout,
                    int k, int stride, int pad, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    SmallConv g{N, D, H, W, (D + 2 * pad - k) / stride + 1, (H + 2 * pad - k) / stride + 1, (W + 2 * pad - k) / stride + 1,
                Cin, Cout, k, stride, pad, k * k * k, ldx, lddy};
    const long long nvox = (long long)N * g.Do * g.Ho * g.Wo;
    int nblk = small_grid(nvox, 256 / (Cin / 4));
    if (nblk > 512) nblk = 512;
    SEG_CHECK_WS((size_t)nblk * g.T * Cin * Cout * sizeof(float), ws_bytes);
    float* part = (float*)ws;
    {
        ProfScope ps(PF_DIRECT, 2.0 * nvox * g.T * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
        if (Cout == 2) {
            constexpr int TPG = 25;
            dim3 grid(nblk, (g.T + TPG - 1) / TPG);
            hipLaunchKernelGGL((smallcout_wgrad_kernel<T, 2, TPG>), grid, dim3(256), (size_t)4 * TPG * 2 * (Cin / 4) * 16, st, x, dy, part, g);
        } else {
            constexpr int TPG = 13;
            dim3 grid(nblk, (g.T + TPG - 1) / TPG);
            hipLaunchKernelGGL((smallcout_wgrad_kernel<T, 4, TPG>), grid, dim3(256), (size_t)4 * TPG * 4 * (Cin / 4) * 16, st, x, dy, part, g);
        }
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, nblk, g.T, Cin, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

bool stem_supported(int Cin, int Cout, int k, int stride, int pad, int ldy) {
    const int lpv = Cout / 4;
    return k == 3 && stride == 1 && pad == 1 && (Cin == 1 || Cin == 2 || Cin == 4) && Cout % 4 == 0 && Cout >= 4 && Cout <= 64 &&
           (lpv & (lpv - 1)) == 0 && ldy % 4 == 0;
}
bool head_supported(int Cin, int Cout, int k, int stride, int pad, int ldx) {
    const int lpv = Cin / 4;
    return k == 1 && stride == 1 && pad == 0 && (Cout == 2 || Cout == 4) && Cin % 4 == 0 && Cin >= 4 && Cin <= 256 &&
           (lpv & (lpv - 1)) == 0 && ldx % 4 == 0;
}

size_t small_ws_bytes(int Cin, int Cout, int k) {
    size_t T = (size_t)k * k * k;
    return align_up((size_t)1024 * T * Cin * Cout * sizeof(float), 256) + colsum_ws_bytes(Cout) + 1024;
}

template <typename T>
int stem_fwd(const T* x, int ldx, const float* w, const float* bias, T* y, int ldy, int N, int D, int H, int W, int Cin,
             int Cout, double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st, float* y_amax, bool* amax_done) {
    SmallGeom g{N, D, H, W, Cin, Cout, ldx, ldy};
    const long long nvox = (long long)N * D * H * W;
    const int vpb = 256 / (Cout / 4);
    const int nblk = small_grid(nvox, vpb);
    float* spart = nullptr;       // in-kernel plain fp32 sums are cancellation-prone: statistics come from the pivoted
                                  // channel reduction over y below instead (0.1 ms on the 128^3 stem)
    size_t lds = (size_t)27 * Cin * Cout * 4;
    if (lds < 256 * 8 * 4) lds = 256 * 8 * 4;
    if (stem1_tiled_ok(g)) {
        const int ntiles = (int)(nvox / ((long long)S1_TZ * S1_TY * (256 / (Cout / 4))));   // tile = 2 x 4 x TX voxels
        int nb = ntiles < 768 ? ntiles : 768;          // three workgroups per CU x 256 CUs: one full round (1024 left a third-full second round)
        // statistics from the kernel (per-block {sum, M2, n} triples, pivoted per thread) when there are enough blocks for the
        // two-stage fp64 finalise and the workspace holds the triples
        const size_t sp_bytes = align_up((size_t)nb * Cout * 3 * sizeof(float), 256);
        const bool in_kernel = ssum && nb > 512 && sp_bytes + part_reduce_ws_bytes(Cout) <= ws_bytes;
        if (in_kernel) spart = (float*)ws;
        {
            ProfScope ps(PF_DIRECT, 2.0 * nvox * 27.0 * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
            if (in_kernel) {
                hipLaunchKernelGGL((stem1_fwd_kernel<T, true>), dim3(nb), dim3(256), stem1_lds(Cout), st, x, w, bias, y, spart, g, ntiles, (unsigned*)y_amax);
                if (y_amax && amax_done) *amax_done = true;
            }
            else hipLaunchKernelGGL((stem1_fwd_kernel<T, false>), dim3(nb), dim3(256), stem1_lds(Cout), st, x, w, bias, y, spart, g, ntiles, (unsigned*)nullptr);
            SEG_CHECK_LAUNCH();
        }
        if (in_kernel) {
            if (!tile_stats_finalize2(spart, nb, Cout, ssum, ssq, reinterpret_cast<double*>((char*)ws + sp_bytes), st)) { set_error("stem_fwd: statistics finalise refused"); return MI355SEG_EINVAL; }
            SEG_CHECK_LAUNCH();
            return MI355SEG_OK;
        }
        if (ssum) return channel_sums(y, ldy, nvox, Cout, ssum, ssq, nullptr, 0, ws, ws_bytes, st);
        return MI355SEG_OK;
    }
    {
        ProfScope ps(PF_DIRECT, 2.0 * nvox * 27.0 * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
        if (Cin == 1) hipLaunchKernelGGL((stem_fwd_kernel<T, 1>), dim3(nblk), dim3(256), lds, st, x, w, bias, y, spart, g);
        else if (Cin == 2) hipLaunchKernelGGL((stem_fwd_kernel<T, 2>), dim3(nblk), dim3(256), lds, st, x, w, bias, y, spart, g);
        else hipLaunchKernelGGL((stem_fwd_kernel<T, 4>), dim3(nblk), dim3(256), lds, st, x, w, bias, y, spart, g);
        SEG_CHECK_LAUNCH();
    }
    if (ssum) return channel_sums(y, ldy, nvox, Cout, ssum, ssq, nullptr, 0, ws, ws_bytes, st);
    return MI355SEG_OK;
}

template <typename T>
int stem_wgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout,
               int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    SmallGeom g{N, D, H, W, Cin, Cout, ldx, 0};
    const long long nvox = (long long)N * D * H * W;
    const int vpb = 256 / (Cout / 4);
    int nblk = small_grid(nvox, vpb);
    if (nblk > 512) nblk = 512;
    SEG_CHECK_WS((size_t)nblk * 27 * Cin * Cout * sizeof(float), ws_bytes);
    float* part = (float*)ws;
    size_t lds = (size_t)4 * 27 * Cout * 4;
    if (stem1_tiled_ok(g)) {
        const int ntiles = (int)(nvox / ((long long)S1_TZ * S1_TY * (256 / (Cout / 4))));   // tile = 2 x 4 x TX voxels
        int nb = ntiles < 512 ? ntiles : 512;
        {
            ProfScope ps(PF_DIRECT, 2.0 * nvox * 27.0 * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
            hipLaunchKernelGGL(stem1_wgrad_kernel<T>, dim3(nb), dim3(256), stem1_lds(Cout), st, x, dy, part, g, lddy, ntiles);
            SEG_CHECK_LAUNCH();
        }
        wgrad_reduce(part, dw, nb, 27, Cin, Cout, accumulate, st);
        SEG_CHECK_LAUNCH();
        return MI355SEG_OK;
    }
    if (Cin == 4 && ldx % 4 == 0 && ((uintptr_t)x % (4 * sizeof(T))) == 0 && (size_t)4 * 9 * 4 * Cout * 4 <= 64 * 1024) {
        ProfScope ps(PF_DIRECT, 2.0 * nvox * 27.0 * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
        hipLaunchKernelGGL((stem4_wgrad_kernel<T>), dim3(nblk, 3), dim3(256), (size_t)4 * 9 * 4 * Cout * 4, st, x, dy, part, g, lddy);
        SEG_CHECK_LAUNCH();
    } else {
        ProfScope ps(PF_DIRECT, 2.0 * nvox * 27.0 * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
        hipLaunchKernelGGL((stem_wgrad_kernel<T, 0>), dim3(nblk, Cin), dim3(256), lds, st, x, dy, part, g, lddy);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, nblk, 27, Cin, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

template <typename T>
int head_fwd(const T* x, int ldx, const float* w, const float* bias, T* y, int ldy, int N, int D, int H, int W, int Cin,
             int Cout, hipStream_t st) {
    SmallGeom g{N, D, H, W, Cin, Cout, ldx, ldy};
    const long long nvox = (long long)N * D * H * W;
    const int nblk = small_grid(nvox, 256 / (Cin / 4)) * 2;
    ProfScope ps(PF_DIRECT, 2.0 * nvox * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
    if (Cout == 2) hipLaunchKernelGGL((head_fwd_kernel<T, 2>), dim3(nblk), dim3(256), 0, st, x, w, bias, y, g);
    else hipLaunchKernelGGL((head_fwd_kernel<T, 4>), dim3(nblk), dim3(256), 0, st, x, w, bias, y, g);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

template <typename T>
int head_dgrad(const T* dy, int lddy, const float* w, T* dx, int lddx, int N, int D, int H, int W, int Cin, int Cout,
               hipStream_t st) {
    SmallGeom g{N, D, H, W, Cin, Cout, lddx, 0};
    const long long nvox = (long long)N * D * H * W;
    const int nblk = small_grid(nvox, 256 / (Cin / 4)) * 2;
    ProfScope ps(PF_DIRECT, 2.0 * nvox * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
    if (Cout == 2) hipLaunchKernelGGL((head_dgrad_kernel<T, 2>), dim3(nblk), dim3(256), 0, st, dy, lddy, w, dx, g);
    else hipLaunchKernelGGL((head_dgrad_kernel<T, 4>), dim3(nblk), dim3(256), 0, st, dy, lddy, w, dx, g);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

template <typename T>
int head_wgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout,
               int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    SmallGeom g{N, D, H, W, Cin, Cout, ldx, 0};
    const long long nvox = (long long)N * D * H * W;
    int nblk = small_grid(nvox, 256 / (Cin / 4));
    if (nblk > 512) nblk = 512;
    SEG_CHECK_WS((size_t)nblk * Cin * Cout * sizeof(float), ws_bytes);
    float* part = (float*)ws;
    {
        ProfScope ps(PF_DIRECT, 2.0 * nvox * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
        if (Cout == 2) hipLaunchKernelGGL((head_wgrad_kernel<T, 2>), dim3(nblk), dim3(256), 0, st, x, dy, lddy, part, g);
        else hipLaunchKernelGGL((head_wgrad_kernel<T, 4>), dim3(nblk), dim3(256), 0, st, x, dy, lddy, part, g);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, nblk, 1, Cin, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

bool tinypw_supported(int Cin, int Cout, int k, int stride, int pad) { return k == 1 && stride == 1 && pad == 0 && Cin <= 4 && Cout <= 4; }
template <typename T>
int tinypw_fwd(const T* x, int ldx, const float* w, const float* bias, T* y, int ldy, long long nvox, int Cin, int Cout, hipStream_t st) {
    TinyPw g{Cin, Cout, ldx, ldy, nvox};
    ProfScope ps(PF_DIRECT, 2.0 * nvox * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
    hipLaunchKernelGGL((tinypw_fwd_kernel<T>), dim3(small_grid(nvox, 256) * 4), dim3(256), 0, st, x, w, bias, y, g);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
template <typename T>
int tinypw_dgrad(const T* dy, int lddy, const float* w, T* dx, int lddx, long long nvox, int Cin, int Cout, hipStream_t st) {
    TinyPw g{Cin, Cout, lddx, lddy, nvox};
    ProfScope ps(PF_DIRECT, 2.0 * nvox * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
    hipLaunchKernelGGL((tinypw_dgrad_kernel<T>), dim3(small_grid(nvox, 256) * 4), dim3(256), 0, st, dy, w, dx, g);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
template <typename T>
int tinypw_wgrad(const T* dy, int lddy, const T* x, int ldx, float* dw, long long nvox, int Cin, int Cout, int accumulate, void* ws, size_t ws_bytes,
                 hipStream_t st) {
    TinyPw g{Cin, Cout, ldx, lddy, nvox};
    int nblk = small_grid(nvox, 256);
    if (nblk > 512) nblk = 512;
    SEG_CHECK_WS((size_t)nblk * Cin * Cout * sizeof(float), ws_bytes);
    float* part = (float*)ws;
    {
        ProfScope ps(PF_DIRECT, 2.0 * nvox * Cin * Cout, (double)sizeof(T) * nvox * (Cin + Cout), st);
        hipLaunchKernelGGL((tinypw_wgrad_kernel<T>), dim3(nblk), dim3(256), 0, st, x, dy, part, g);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, nblk, 1, Cin, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// explicit instantiations for the two storage types (declared as templates in internal.h)
#define SEG_INST(T) \
    template int smallcin_wgrad<T>(const T*, int, const T*, int, float*, int, int, int, int, int, int, int, int, int, int, void*, size_t, hipStream_t); \
    template int smallcout_wgrad<T>(const T*, int, const T*, int, float*, int, int, int, int, int, int, int, int, int, int, void*, size_t, hipStream_t); \
    template int stem_fwd<T>(const T*, int, const float*, const float*, T*, int, int, int, int, int, int, int, double*, double*, void*, size_t, hipStream_t, float*, bool*); \
    template int stem_wgrad<T>(const T*, int, const T*, int, float*, int, int, int, int, int, int, int, void*, size_t, hipStream_t); \
    template int head_fwd<T>(const T*, int, const float*, const float*, T*, int, int, int, int, int, int, int, hipStream_t); \
    template int head_dgrad<T>(const T*, int, const float*, T*, int, int, int, int, int, int, int, hipStream_t); \
    template int head_wgrad<T>(const T*, int, const T*, int, float*, int, int, int, int, int, int, int, void*, size_t, hipStream_t); \
    template int tinypw_fwd<T>(const T*, int, const float*, const float*, T*, int, long long, int, int, hipStream_t); \
    template int tinypw_dgrad<T>(const T*, int, const float*, T*, int, long long, int, int, hipStream_t); \
    template int tinypw_wgrad<T>(const T*, int, const T*, int, float*, long long, int, int, int, void*, size_t, hipStream_t);
SEG_INST(float)
SEG_INST(bf16)
#undef SEG_INST

}  // namespace seg
