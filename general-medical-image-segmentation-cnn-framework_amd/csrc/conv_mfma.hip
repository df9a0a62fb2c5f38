// conv_mfma.hip -- Conv3d k3 s1 p1 as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD), forward and dgrad.
//
//   M = output voxels, N = output channels, K = 27 taps x Cin.
//
// Workgroup = 4 waves (256 threads); each wave owns MB M-blocks of 32 voxels and all
// NT = 32*NBW output channels of the tile, i.e. the tile is (128*MB voxels) x NT.
// The input halo tile ((TZ+2) x (TY+2) x (BX+2) voxels x 16 input channels, NDHWC) is
// staged in LDS once per 16-channel chunk and re-read by all 27 taps as shifted
// ds_read_b128 (the shift is an immediate offset, the pitch 20 floats keeps the reads
// bank-conflict free).  Weights are pre-packed so that the B operand of four
// consecutive MFMA k-steps is ONE coalesced 16-byte global load per lane (L2-resident,
// software-prefetched one step ahead); the A operand of the same four k-steps is ONE
// ds_read_b128.  The next chunk's halo is prefetched into registers before the 864
// MFMAs of the current chunk and written to LDS after them (issue-early / write-late).
// Epilogue: bias add, NDHWC store (128 B per half-wave), optional per-channel
// sum / sum-of-squares partials for the BatchNorm that follows (deterministic two-stage).
//
// dgrad of a k3 s1 p1 conv is the same convolution with the 27 taps reversed and the
// channel roles swapped; only the weight packing differs.
#include "common.h"
#include "internal.h"
#include <stdlib.h>
#include <initializer_list>

namespace seg {

// Timing-experiment switches (ablations of the main loop, tile-shape overrides) exist only in a -DMI355SEG_TUNE build
// (make TUNE=1); the shipped kernels carry none of them.
#ifdef MI355SEG_TUNE
#define SEG_DBG(a, bit) ((a).dbg & (bit))
#else
#define SEG_DBG(a, bit) 0
#endif

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// CK = input channels per LDS chunk: 16 for k3 (65 KB halo tile, two workgroups per CU), 8 for k5 (the 5^3 halo
// is 4x larger), 64 for k1 / ConvTranspose where there is no halo and few MFMAs per chunk otherwise.
// PITCH = CK + 4 floats per voxel keeps an odd number of 16-byte slots -> conflict-free ds_read_b128.
template <int KS, int BX, int MB, int CK>
struct Tile {
    static constexpr int PITCH = CK + 4;
    static constexpr int HALO = KS / 2;
    static constexpr int NTAP = KS * KS * KS;
    static constexpr int LPB = 32 / BX;           // x-lines per 32-voxel M-block
    static constexpr int LINES = 4 * MB * LPB;    // x-lines per workgroup tile
    static constexpr int TY = 4;
    static constexpr int TZ = LINES / TY;
    static constexpr int HX = BX + 2 * HALO, HY = TY + 2 * HALO, HZ = TZ + 2 * HALO;
    static constexpr int NVOX = HX * HY * HZ;
    static constexpr int NPIECE = NVOX * (CK / 4);            // 16-byte pieces per chunk
    static constexpr int NITER = (NPIECE + 255) / 256;
    static constexpr int LDS_BYTES = NVOX * PITCH * 4;
    static_assert(LINES % TY == 0, "tile lines must fill whole y-rows");
};

// The M space is always the "base grid" (N, D, H, W).  in_mul / out_mul = 2 turn the same kernel into
// ConvTranspose3d k2 s2: forward scatters N-tile (tap, cout-tile) to child voxel 2*v + tap of the
// (2D,2H,2W) output; dgrad gathers K-chunk (tap, cout-chunk) from child voxel 2*v + tap of the input.
struct IgemmArgs {
    const float* x; const float* wq; const float* bias; float* y; float* spart;
    int ldx, ldy, N, D, H, W, Cout;
    int ntx, nty, ntz, nN;
    int nchunks;        // total K chunks of 16 channels (taps of a ConvT dgrad included)
    int cpt;            // chunks per input tap  (== nchunks when in_mul == 1)
    int nNpt;           // N-tiles per output tap (== nN when out_mul == 1)
    int in_mul, out_mul;
    int nM;             // M-tiles
    int ksplit, cps;    // K-splits and chunks per split (nchunks == ksplit * cps)
    long long split_stride;   // floats between the output slabs of consecutive K-splits
    int dbg;            // -DMI355SEG_TUNE builds only (MI355SEG_DBG): 1 no re-staging, 2 B loaded once per chunk, 4 no stores; else 0 and unread
    // ---- gather / scatter generalisation (strided Conv3d fwd + per-phase dgrad, ConvT with narrow Cout)
    int Di, Hi, Wi;     // extents of the volume x points at   (input voxel = base * in_mul + toff[tap])
    int Do, Ho, Wo;     // extents of the volume y points at   (output voxel = base * out_mul + child + c{z,y,x})
    int cz, cy, cx;     // fixed child offset of a strided-dgrad phase launch
    int flatn;          // != 0: N-tiles cut the flat (child tap, cout) axis, so one 32-column block may span two children
    int by, bz;         // (y, z) tile-block shape of the M-tile walk (divisors of nty, ntz)
    signed char toff[64][4];   // per K-tap input offset (z, y, x); all zero for the stride-1 halo modes
};

// ---------------------------------------------------------------- weight packing
// wq[nt][chunk][tap][kk][h][j][s]  =  B[k = (chunk, tap, kk, h, s)][n = nt*NT + j]  with
//   mode 0 (conv fwd):     B = W[co = n][ci = chunk*16 + kk*8 + h*4 + s][tap]            W: (Cout, Cin, T)
//   mode 1 (conv dgrad):   B = W[ci_f = n .. swapped roles, taps reversed]              W: (Cin_k, Cout_k, T)
//   mode 2 (convT fwd):    n = (tapn, co): B = Wt[ci][co][tapn]                          Wt: (Cin, Cout, 8), T = 1
//   mode 3 (convT dgrad):  chunk = (tapk, cc): B = Wt[ci = n][co = cc*16 + ..][tapk]     Wt: (Cin_f, Cout_f, 8), T = 1
//   mode 5 (gather fwd):   chunk = (tap, cc):  B = W[co = n][ci = cc*CK + ..][tap]      W: (Cout, Cin, TW), aux = Cin, T = 1
//   mode 6 (gather dgrad): chunk = (slot, cc): B = W[co = cc*CK + ..][ci = n][taps.t[slot]]   aux = Cout, T = 1
struct TapList { unsigned char t[64]; };
__global__ void pack_wq_kernel(const float* __restrict__ w, float* __restrict__ wq, int K, int Nn, int T, int NT, int mode, int aux, int CK,
                               int TW, TapList taps) {
    const long long total = (long long)K * Nn * T;        // K = channels in the GEMM K dim (taps of mode 3 included)
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        int s = (int)(r % 4); r /= 4;
        int j = (int)(r % NT); r /= NT;
        int h = (int)(r % 2); r /= 2;
        int kk = (int)(r % (CK / 8)); r /= (CK / 8);
        int tap = (int)(r % T); r /= T;
        int chunk = (int)(r % (K / CK)); r /= (K / CK);
        int nt = (int)r;
        int n = nt * NT + j, k = chunk * CK + kk * 8 + h * 4 + s;
        float v;
        if (mode == 0) v = w[((long long)n * K + k) * T + tap];
        else if (mode == 1) v = w[((long long)k * Nn + n) * T + (T - 1 - tap)];
        else if (mode == 2) { int cout = aux; int tapn = n / cout, co = n % cout; v = w[((long long)k * cout + co) * 8 + tapn]; }
        else if (mode == 3 || mode == 5) { int cout = aux; int tapk = k / cout, co = k % cout; v = w[((long long)n * cout + co) * TW + tapk]; }
        else { int cout = aux; int slot = k / cout, co = k % cout; v = w[((long long)co * Nn + n) * TW + taps.t[slot]]; }
        wq[idx] = v;
    }
}

// ---------------------------------------------------------------- the kernel
// One virtual tile = (M-tile, N-tile, K-split) per workgroup; up to two workgroups share a CU and the
// hardware dispatcher staggers them, so one stages its halo while the other issues MFMAs (a persistent
// variant was measured 8-10 % slower on the large layers: co-resident workgroups fall into lockstep).
// With ksplit > 1 (few-tile deep layers) every split writes raw partial sums to its own slab and a
// tiny second kernel adds them in fixed order.
template <int KS, int BX, int MB, int NBW, int CK>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(IgemmArgs a) {
    using T = Tile<KS, BX, MB, CK>;
    constexpr int PITCH = T::PITCH;
    constexpr int PPV = CK / 4;                            // 16-byte pieces per voxel
    constexpr int NT = 32 * NBW;
    constexpr int NTAP = T::NTAP;
    constexpr int STEP_FLOATS = 2 * NT * 4;                 // packed weights consumed per (tap, kk) step
    constexpr int CHUNK_FLOATS = NTAP * (CK / 8) * STEP_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;

    // XCD-aware block -> tile map: blocks dealt round-robin to the 8 XCDs get contiguous tile ranges,
    // so halo-sharing neighbours and the N-tiles / K-splits of one M-tile share an L2 (bijective).
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ks = t % a.ksplit; t /= a.ksplit;
    const int ntile = t % a.nN;
    const int mtile = t / a.nN;
    // M-tiles are walked in (y, z) blocks of by x bz tiles (x fastest inside a block) so that the ~64 tiles an XCD works
    // on at any moment form a compact brick whose shared halo planes stay in that XCD's L2 (by divides nty; the last
    // z-row of bricks may be shorter than bz)
    int mt = mtile;
    const int per_n = a.ntx * a.nty * a.ntz;
    const int n = mt / per_n; mt -= n * per_n;
    const int nby = a.nty / a.by;
    const int zfull = a.ntz / a.bz;                           // full z-rows of bricks; a ragged last row holds the remaining slabs
    const int rowtiles = a.ntx * a.nty * a.bz;                // tiles per full z-row
    int zrow = mt / rowtiles, bzz = a.bz;
    if (zrow >= zfull) { zrow = zfull; bzz = a.ntz - zfull * a.bz; }
    mt -= zrow * rowtiles;
    const int blk = a.ntx * a.by * bzz;
    const int b = mt / blk; mt -= b * blk;
    const int txi = mt % a.ntx; mt /= a.ntx;
    const int tyi = b * a.by + mt % a.by;
    const int tzi = zrow * a.bz + mt / a.by;
    const int x0 = txi * BX, y0 = tyi * T::TY, z0 = tzi * T::TZ;
    const int tapn = ntile / a.nNpt;                          // output child (ConvT fwd), else 0
    const int n0 = (ntile % a.nNpt) * NT;
    const int Di = a.Di, Hi = a.Hi, Wi = a.Wi;

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mb][nb][v] = 0.f;

    // per-lane LDS base (floats) of the A fragment for each M-block
    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = wave * MB + mb;
        const int line = m * T::LPB + i / BX, xx = i % BX;
        abase[mb] = (((line / T::TY) * T::HY + (line % T::TY)) * T::HX + xx) * PITCH + 4 * h;
    }

    const int c0 = ks * a.cps, c1 = c0 + a.cps;             // this split's chunk range
    const float* wlane = a.wq + (long long)ntile * a.nchunks * CHUNK_FLOATS + (h * NT + i) * 4;

    // ---- halo staging: global -> registers (issue early) -> LDS (write late)
    f32x4 stage[T::NITER];
    auto load_stage = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < T::NITER; ++it) {
            const int p = it * 256 + tid;
            const int vox = p / PPV, part = p % PPV;
            const int hz = vox / (T::HY * T::HX), rem = vox % (T::HY * T::HX);
            const int hy = rem / T::HX, hx = rem % T::HX;
            const int tapk = chunk / a.cpt, cch = chunk - tapk * a.cpt;     // input child (ConvT dgrad), else 0
            const int gz = (z0 - T::HALO + hz) * a.in_mul + a.toff[tapk][0];
            const int gy = (y0 - T::HALO + hy) * a.in_mul + a.toff[tapk][1];
            const int gx = (x0 - T::HALO + hx) * a.in_mul + a.toff[tapk][2];
            const bool ok = (p < T::NPIECE) && (unsigned)gz < (unsigned)Di && (unsigned)gy < (unsigned)Hi && (unsigned)gx < (unsigned)Wi;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const long long off = ((((long long)n * Di + gz) * Hi + gy) * Wi + gx) * a.ldx + cch * CK + part * 4;
                v = *reinterpret_cast<const f32x4*>(a.x + off);
            }
            stage[it] = v;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < T::NITER; ++it) {
            const int p = it * 256 + tid;
            if (p < T::NPIECE) *reinterpret_cast<f32x4*>(lds + (p / PPV) * PITCH + (p % PPV) * 4) = stage[it];
        }
    };

    load_stage(c0);
    for (int chunk = c0; chunk < c1; ++chunk) {
        const float* wp = wlane + (long long)chunk * CHUNK_FLOATS;
        // B fragments run PFD steps ahead of the MFMAs that consume them (register ring, static indices).
        // The ring's first PFD loads are issued BEFORE the next chunk's halo prefetch: vmcnt retires in
        // order, so a B load queued behind 13 halo loads (possible HBM misses) would stall the first MFMAs.
        constexpr int NSTEP = NTAP * (CK / 8);
        constexpr int PFD = NSTEP > 4 ? 4 : 1;
        f32x4 bq[PFD + 1][NBW];
#pragma unroll
        for (int d = 0; d < PFD; ++d)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) bq[d][nb] = *reinterpret_cast<const f32x4*>(wp + d * STEP_FLOATS + nb * 128);
        if (!SEG_DBG(a, 1) || chunk == c0) {
        __syncthreads();                 // every wave is done reading the previous chunk
        write_stage();
        __syncthreads();
        }
        if (chunk + 1 < c1 && !SEG_DBG(a, 1)) load_stage(chunk + 1);
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
            const int tapoff = ((dz * T::HY + dy) * T::HX + dx) * PITCH;
#pragma unroll
            for (int kk = 0; kk < CK / 8; ++kk) {
                const int step = tap * (CK / 8) + kk;
                const int cur = step % (PFD + 1), fill = (step + PFD) % (PFD + 1);
                if (step + PFD < NSTEP && !SEG_DBG(a, 2)) {
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb)
                        bq[fill][nb] = *reinterpret_cast<const f32x4*>(wp + (step + PFD) * STEP_FLOATS + nb * 128);
                }
                f32x4 av[MB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) av[mb] = *reinterpret_cast<const f32x4*>(lds + abase[mb] + tapoff + kk * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                        for (int nb = 0; nb < NBW; ++nb)
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb][s], bq[cur][nb][s], acc[mb][nb], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias, store, optional BatchNorm partial statistics
    float* yout = a.y + (long long)ks * a.split_stride;
    float ssum[NBW], ssq[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        int col = n0 + nb * 32 + i, child = tapn;
        if (a.flatn) { const int nf = ntile * NT + nb * 32 + i; child = nf / a.Cout; col = nf - child * a.Cout; }   // per-lane child
        const int oz = ((child >> 2) & 1) + a.cz, oy = ((child >> 1) & 1) + a.cy, ox = (child & 1) + a.cx;
        const float bv = a.bias ? a.bias[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = wave * MB + mb;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = (v & 3) + 8 * (v >> 2) + 4 * h;      // row of the 32x32 tile held in register v
                const int line = m * T::LPB + r / BX, xx = r % BX;
                const int gz = (z0 + line / T::TY) * a.out_mul + oz;
                const int gy = (y0 + line % T::TY) * a.out_mul + oy;
                const int gx = (x0 + xx) * a.out_mul + ox;
                const float val = acc[mb][nb][v] + bv;
                // partial tiles (extents that are not tile multiples): rows outside the volume are dropped
                const bool inside = (z0 + line / T::TY) < a.D && (y0 + line % T::TY) < a.H && (x0 + xx) < a.W &&
                                    gz < a.Do && gy < a.Ho && gx < a.Wo;
#ifdef MI355SEG_TUNE
                if (inside && (!(a.dbg & 4) || val == 12345.678f))
#else
                if (inside)
#endif
                    yout[((((long long)n * a.Do + gz) * a.Ho + gy) * a.Wo + gx) * a.ldy + col] = val;
                if (inside) { s1 += val; s2 += val * val; }
            }
        }
        ssum[nb] = s1; ssq[nb] = s2;
    }
    if (a.spart) {
        // BatchNorm batch statistics of this tile, cancellation-free: per channel the tile sum, then the tile
        // mean, then M2 = sum (y - tile_mean)^2 from the accumulators still in registers; the second stage
        // combines (n, sum, M2) of all tiles in fp64 (Chan et al.).  spart[mtile][c] = {sum, M2, n}.
        __syncthreads();                 // LDS halo no longer needed
        float cnt = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            float s1 = ssum[nb] + __shfl_xor(ssum[nb], 32, 64);
            if (h == 0) lds[wave * NT + nb * 32 + i] = s1;
        }
        {   // valid rows of this tile (same for every channel)
            const int vz = min(T::TZ, a.D - z0), vy = min(T::TY, a.H - y0), vx = min(BX, a.W - x0);
            cnt = (float)(vz * vy * vx);
        }
        __syncthreads();
        float tmean[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int c = nb * 32 + i;
            tmean[nb] = (lds[c] + lds[NT + c] + lds[2 * NT + c] + lds[3 * NT + c]) / cnt;
        }
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const float bv = a.bias ? a.bias[n0 + nb * 32 + i] : 0.f;
            float m2 = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = wave * MB + mb;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
                    const int line = m * T::LPB + r / BX, xx = r % BX;
                    const bool inside = (z0 + line / T::TY) < a.D && (y0 + line % T::TY) < a.H && (x0 + xx) < a.W;
                    const float d = acc[mb][nb][v] + bv - tmean[nb];
                    if (inside) m2 += d * d;
                }
            }
            m2 += __shfl_xor(m2, 32, 64);
            if (h == 0) lds[4 * NT + wave * NT + nb * 32 + i] = m2;
        }
        __syncthreads();
        if (tid < NT) {
            const float s1 = lds[tid] + lds[NT + tid] + lds[2 * NT + tid] + lds[3 * NT + tid];
            const float m2 = lds[4 * NT + tid] + lds[5 * NT + tid] + lds[6 * NT + tid] + lds[7 * NT + tid];
            float* dst = a.spart + ((long long)mtile * a.Cout + n0 + tid) * 3;
            dst[0] = s1; dst[1] = m2; dst[2] = cnt;
        }
    }
}

// y[r][c] = bias[c] + sum_k part[k][r][c]   (split-K second stage; fixed order)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ksplit, long long stride,
        const float* __restrict__ bias, float* __restrict__ y, int ldy, long long rows, int C) {
    const int cw = C / 4;
    const long long total = rows * cw;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / cw;
        const int c = (int)(idx % cw) * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (bias) s = *reinterpret_cast<const f32x4*>(bias + c);
        for (int k = 0; k < ksplit; ++k) s += *reinterpret_cast<const f32x4*>(part + k * stride + r * C + c);
        *reinterpret_cast<f32x4*>(y + r * ldy + c) = s;
    }
}

// per-channel fp64 combine of the per-tile (sum, M2, n) triples: block per channel, fixed order.
// Emits sum(y) and sum(y^2) as doubles (sum^2 = M2_total + N * mean^2 is exact enough in fp64 for the
// var = E[y^2] - mean^2 that mi355seg_norm_stats_from_sums_f32 forms afterwards).
__global__ __launch_bounds__(256) void igemm_stats_finalize_kernel(const float* __restrict__ spart, int nM, int Cout,
                                                                    double* __restrict__ sum, double* __restrict__ sq) {
    __shared__ double sh[8];
    __shared__ double gmean;
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double s1 = 0.0, n = 0.0;
    for (int m = threadIdx.x; m < nM; m += 256) {
        const float* p = spart + ((long long)m * Cout + c) * 3;
        s1 += (double)p[0]; n += (double)p[2];
    }
    s1 = wave_sum(s1); n = wave_sum(n);
    if (lane == 0) { sh[wv * 2] = s1; sh[wv * 2 + 1] = n; }
    __syncthreads();
    const double S = sh[0] + sh[2] + sh[4] + sh[6], Nn = sh[1] + sh[3] + sh[5] + sh[7];
    if (threadIdx.x == 0) gmean = S / Nn;
    __syncthreads();
    const double mean = gmean;
    double m2 = 0.0;
    for (int m = threadIdx.x; m < nM; m += 256) {
        const float* p = spart + ((long long)m * Cout + c) * 3;
        const double nt = (double)p[2], d = (double)p[0] / nt - mean;
        m2 += (double)p[1] + nt * d * d;
    }
    m2 = wave_sum(m2);
    __syncthreads();
    if (lane == 0) sh[wv] = m2;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double M2 = sh[0] + sh[1] + sh[2] + sh[3];
        sum[c] = S;
        sq[c] = M2 + Nn * mean * mean;
    }
}

// ---------------------------------------------------------------- host side
struct IgemmPlan { int KS, CK, BX, MB, NBW, TZ, nM, nN, ntx, nty, ntz, flat; };

static int pick_ck(int KS, int Kc) {
    if (KS == 5) return 8;
    if (KS == 1 && Kc % 64 == 0) return 64;
    return 16;
}

// Kc = GEMM K channels per input tap (multiple of CK), Nc = GEMM N per output tap (multiple of 32)
static bool igemm_plan(int KS, int N, int D, int H, int W, int Kc, int Nc, int ntaps_out, IgemmPlan* p) {
    if (KS != 1 && KS != 3 && KS != 5) return false;
    const int CK = pick_ck(KS, Kc);
    // ConvT with a narrow Cout: tile the flat (child, cout) axis instead of each child's channels
    const bool flat = ntaps_out > 1 && (Nc % 32) != 0 && ((long long)Nc * ntaps_out) % 32 == 0;
    if (Kc % CK || (Nc % 32 && !flat) || W < 4) return false;          // degenerate volumes stay on the generic path
    // x-extent of an M-block: the candidate with the least padding (ties -> the wider one)
    int BX = 0; long long best = -1;
    for (int bx : {32, 16, 8}) {
        long long padded = (long long)((W + bx - 1) / bx) * bx;
        if (best < 0 || padded < best) { best = padded; BX = bx; }
    }
    const int NBW = ((flat ? Nc * ntaps_out : Nc) % 64 == 0) ? 2 : 1;
    const int nN = flat ? Nc * ntaps_out / (32 * NBW) : Nc / (32 * NBW) * ntaps_out;
    p->flat = flat ? 1 : 0;
    auto tiles = [&](int MB, int* tz) {
        int lines = 4 * MB * (32 / BX);
        *tz = lines / 4;
        return (long long)N * ((D + *tz - 1) / *tz) * ((H + 3) / 4) * ((W + BX - 1) / BX);
    };
    auto waste = [&](int tz) { return (double)(((D + tz - 1) / tz) * tz) / D; };
    int tz2, tz1;
    long long m2 = tiles(2, &tz2), m1 = tiles(1, &tz1);
    (void)m1;
    int MB;
#ifdef MI355SEG_TUNE
    static const char* force = getenv("MI355SEG_IGEMM_MB");          // tuning knob: 1 / 2 force that M-block count
#else
    constexpr const char* force = nullptr;
#endif
    int tz3;
    const long long m3 = tiles(3, &tz3);
    if (KS == 5) MB = 1;                                             // the 5^3 halo of a 2-block tile does not fit twice per CU
    else if (force && force[0] == '1') MB = 1;
    // one 32-channel N-block per tile (Cout = 32 at full resolution): a 384-voxel tile still fits twice per CU (2 x 81.6 KB),
    // cuts the halo overfetch from 3.2x to 2.7x and spreads the per-tile prologue / epilogue over 1.5x the MFMAs (+3 %)
    else if (!(force && force[0] == '2') && KS == 3 && BX == 32 && NBW == 1 && m3 * nN >= 2048 && waste(tz3) <= 1.05) MB = 3;
    else if (m2 * nN >= 512 && waste(tz2) <= waste(tz1) * 1.2) MB = 2;
    else MB = 1;
    p->KS = KS; p->CK = CK; p->BX = BX; p->MB = MB; p->NBW = NBW; p->TZ = MB == 3 ? tz3 : (MB == 2 ? tz2 : tz1);
    p->ntx = (W + BX - 1) / BX; p->nty = (H + 3) / 4; p->ntz = (D + p->TZ - 1) / p->TZ;
    p->nM = N * p->ntz * p->nty * p->ntx; p->nN = nN;
    return true;
}

static bool igemm_shape_ok(int k, int stride, int pad) {
    return stride == 1 && ((k == 1 && pad == 0) || (k == 3 && pad == 1) || (k == 5 && pad == 2));
}

bool conv_mfma_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    if (!igemm_shape_ok(k, stride, pad) || (ldx % 4)) return false;
    IgemmPlan p;
    return igemm_plan(k, N, D, H, W, Cin, Cout, 1, &p);
}

static int pick_ksplit(int tiles, int nchunks);
size_t conv_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    if (!igemm_shape_ok(k, stride, pad)) return 0;
    size_t best = 0;
    const size_t T = (size_t)k * k * k;
    // the same workspace must serve fwd (Cin->Cout) and dgrad (Cout->Cin)
    for (int pass = 0; pass < 2; ++pass) {
        int ci = pass ? Cout : Cin, co = pass ? Cin : Cout;
        IgemmPlan p;
        if (!igemm_plan(k, N, D, H, W, ci, co, 1, &p)) continue;
        const int ks = pick_ksplit(p.nM * p.nN, ci / p.CK);
        size_t need = align_up(T * Cin * Cout * sizeof(float), 256) + align_up((size_t)p.nM * co * 3 * sizeof(float), 256) +
                      (ks > 1 ? align_up((size_t)ks * N * D * H * W * co * sizeof(float), 256) + colsum_ws_bytes(co) : 0) + 1024;
        if (need > best) best = need;
    }
    size_t wg = (k == 3 || k == 5) ? wgrad_mfma_ws_bytes(N, D, H, W, Cin, Cout, k) : (k == 1 ? pw_wgrad_ws_bytes((long long)N * D * H * W, Cin, Cout, 1) : 0);
    return best > wg ? best : wg;
}

template <int KS, int BX, int MB, int NBW, int CK>
static void launch_igemm(const IgemmArgs& a, int nwg, hipStream_t st) {
    using T = Tile<KS, BX, MB, CK>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<KS, BX, MB, NBW, CK>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<KS, BX, MB, NBW, CK>), dim3(nwg), dim3(256), T::LDS_BYTES, st, a);
}

template <int KS, int CK, bool ALLOW_MB2>
static void dispatch_igemm_ck(const IgemmPlan& p, const IgemmArgs& a, int nwg, hipStream_t st) {
#define IGEMM_CASE(bx, mb, nbw) \
    if (p.BX == bx && p.MB == mb && p.NBW == nbw) launch_igemm<KS, bx, mb, nbw, CK>(a, nwg, st)
    if constexpr (ALLOW_MB2 && KS == 3) { IGEMM_CASE(32, 3, 1); }
    if (ALLOW_MB2) {
        IGEMM_CASE(32, 2, 2); else IGEMM_CASE(32, 2, 1); else IGEMM_CASE(16, 2, 2); else IGEMM_CASE(16, 2, 1);
        else IGEMM_CASE(8, 2, 2); else IGEMM_CASE(8, 2, 1);
    }
    IGEMM_CASE(32, 1, 2); else IGEMM_CASE(32, 1, 1); else IGEMM_CASE(16, 1, 2); else IGEMM_CASE(16, 1, 1);
    else IGEMM_CASE(8, 1, 2); else IGEMM_CASE(8, 1, 1);
#undef IGEMM_CASE
}

static int tile_block(int nt) { return nt % 4 == 0 ? 4 : (nt % 2 == 0 ? 2 : 1); }

static void dispatch_igemm(const IgemmPlan& p, const IgemmArgs& a_in, int nwg, hipStream_t st) {
    IgemmArgs a = a_in;
#ifdef MI355SEG_TUNE
    static const char* flat_walk = getenv("MI355SEG_IGEMM_LINEAR_WALK");      // A/B knob: 1 = plain x, y, z tile order
#else
    constexpr const char* flat_walk = nullptr;
#endif
    a.by = flat_walk ? 1 : tile_block(a.nty);
    a.bz = flat_walk ? 1 : (a.ntz >= 4 ? 4 : tile_block(a.ntz));      // z may be ragged (last brick row shorter), y must divide
    if (p.KS == 3) dispatch_igemm_ck<3, 16, true>(p, a, nwg, st);
    else if (p.KS == 5) dispatch_igemm_ck<5, 8, false>(p, a, nwg, st);
    else if (p.CK == 64) dispatch_igemm_ck<1, 64, true>(p, a, nwg, st);
    else dispatch_igemm_ck<1, 16, true>(p, a, nwg, st);
}

#ifdef MI355SEG_TUNE
static int dbg_flags() { static const char* e = getenv("MI355SEG_DBG"); return e ? atoi(e) : 0; }
#else
static constexpr int dbg_flags() { return 0; }
#endif
static int pack_grid(long long total) { return (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256); }

// K-split factor for layers with too few tiles to fill 2 x 256 workgroup slots
static int pick_ksplit(int tiles, int nchunks) {
    int best = 1;
    for (int k = 2; k <= 16; k *= 2) {
        if (nchunks % k || nchunks / k < 2) break;
        if (tiles * (k / 2) >= 512) break;
        best = k;
    }
    return best;
}

// k in {1, 3, 5}, pad = k/2, stride 1
int conv_fwd_mfma(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int N, int D, int H, int W,
                  int Cin, int Cout, int k, int dgrad, double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st) {
    IgemmPlan p;
    SEG_CHECK_ARG(igemm_plan(k, N, D, H, W, Cin, Cout, 1, &p), "conv_fwd_mfma: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0, "conv_fwd_mfma: input pointer must be 16-byte aligned");
    const int T = k * k * k;
    const int nchunks = Cin / p.CK;
    const long long nvox = (long long)N * D * H * W;
    const int ksplit = (ldy % 4 == 0) ? pick_ksplit(p.nM * p.nN, nchunks) : 1;
    Carver cv(ws);
    float* wq = cv.take<float>((size_t)T * Cin * Cout);
    float* spart = (ssum && ksplit == 1) ? cv.take<float>((size_t)p.nM * Cout * 3) : nullptr;
    float* slabs = ksplit > 1 ? cv.take<float>((size_t)ksplit * nvox * Cout) : nullptr;
    size_t tail = cv.used();
    SEG_CHECK_WS(tail + ((ssum && ksplit > 1) ? colsum_ws_bytes(Cout) : 0), ws_bytes);
    hipLaunchKernelGGL(pack_wq_kernel, dim3(pack_grid((long long)T * Cin * Cout)), dim3(256), 0, st, w, wq, Cin, Cout, T, 32 * p.NBW, dgrad ? 1 : 0, 0, p.CK, T, TapList{});
    SEG_CHECK_LAUNCH();
    IgemmArgs a{x, wq, ksplit > 1 ? nullptr : bias, ksplit > 1 ? slabs : y, spart, ldx, ksplit > 1 ? Cout : ldy, N, D, H, W, Cout,
                p.ntx, p.nty, p.ntz, p.nN, nchunks, nchunks, p.nN, 1, 1, p.nM, ksplit, nchunks / ksplit, nvox * Cout, dbg_flags()};
    a.Di = a.Do = D; a.Hi = a.Ho = H; a.Wi = a.Wo = W;
    const int nwg = p.nM * p.nN * ksplit;
    const double vox = (double)nvox;
    {
        ProfScope ps(PF_IGEMM, 2.0 * vox * T * Cin * Cout, 4.0 * (vox * (Cin + Cout) + (double)T * Cin * Cout), st);
        dispatch_igemm(p, a, nwg, st);
        SEG_CHECK_LAUNCH();
        if (ksplit > 1) {
            long long tot = nvox * (Cout / 4);
            int grid = (int)((tot + 255) / 256 > 2048 ? 2048 : (tot + 255) / 256);
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, slabs, ksplit, nvox * Cout, bias, y, ldy, nvox, Cout);
            SEG_CHECK_LAUNCH();
        }
    }
    if (ssum) {
        if (ksplit > 1) return channel_sums(y, ldy, nvox, Cout, ssum, ssq, nullptr, 0, (char*)ws + tail, ws_bytes - tail, st);
        hipLaunchKernelGGL(igemm_stats_finalize_kernel, dim3(Cout), dim3(256), 0, st, spart, p.nM, Cout, ssum, ssq);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

// ---- ConvTranspose3d k2 s2 on the same kernel (KS = 1)
bool convt_mfma_supported(int N, int D, int H, int W, int Cin, int Cout, int ldx, int ldy) {
    IgemmPlan p, q;
    return (ldx % 4) == 0 && (ldy % 4) == 0 && igemm_plan(1, N, D, H, W, Cin, Cout, 8, &p) && igemm_plan(1, N, D, H, W, Cout, Cin, 1, &q) &&
           Cin % 32 == 0;
}

int convt_fwd_mfma(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int N, int D, int H, int W,
                   int Cin, int Cout, void* ws, size_t ws_bytes, hipStream_t st) {
    IgemmPlan p;
    SEG_CHECK_ARG(igemm_plan(1, N, D, H, W, Cin, Cout, 8, &p), "convt_fwd_mfma: unsupported shape");
    Carver cv(ws);
    float* wq = cv.take<float>((size_t)8 * Cin * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    hipLaunchKernelGGL(pack_wq_kernel, dim3(pack_grid((long long)8 * Cin * Cout)), dim3(256), 0, st, w, wq, Cin, 8 * Cout, 1, 32 * p.NBW, 2, Cout, p.CK, 8, TapList{});
    SEG_CHECK_LAUNCH();
    IgemmArgs a{x, wq, bias, y, nullptr, ldx, ldy, N, D, H, W, Cout, p.ntx, p.nty, p.ntz, p.nN, Cin / p.CK, Cin / p.CK,
                p.flat ? p.nN : p.nN / 8, 1, 2, p.nM, 1, Cin / p.CK, 0, 0};
    a.Di = D; a.Hi = H; a.Wi = W; a.Do = 2 * D; a.Ho = 2 * H; a.Wo = 2 * W; a.flatn = p.flat;
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_CONVT, 2.0 * vox * 8 * Cin * Cout, 4.0 * (vox * (Cin + 8.0 * Cout) + 8.0 * Cin * Cout), st);
    dispatch_igemm(p, a, p.nM * p.nN, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// dx[N,D,H,W,Cin] from dy[N,2D,2H,2W,Cout]
int convt_dgrad_mfma(const float* dy, int lddy, const float* w, float* dx, int lddx, int N, int D, int H, int W,
                     int Cin, int Cout, void* ws, size_t ws_bytes, hipStream_t st) {
    IgemmPlan p;
    SEG_CHECK_ARG(igemm_plan(1, N, D, H, W, Cout, Cin, 1, &p), "convt_dgrad_mfma: unsupported shape");
    Carver cv(ws);
    float* wq = cv.take<float>((size_t)8 * Cin * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    hipLaunchKernelGGL(pack_wq_kernel, dim3(pack_grid((long long)8 * Cin * Cout)), dim3(256), 0, st, w, wq, 8 * Cout, Cin, 1, 32 * p.NBW, 3, Cout, p.CK, 8, TapList{});
    SEG_CHECK_LAUNCH();
    IgemmArgs a{dy, wq, nullptr, dx, nullptr, lddy, lddx, N, D, H, W, Cin, p.ntx, p.nty, p.ntz, p.nN, 8 * Cout / p.CK, Cout / p.CK, p.nN, 2, 1,
                p.nM, 1, 8 * Cout / p.CK, 0, 0};
    a.Di = 2 * D; a.Hi = 2 * H; a.Wi = 2 * W; a.Do = D; a.Ho = H; a.Wo = W;
    for (int t = 0; t < 8; ++t) { a.toff[t][0] = (t >> 2) & 1; a.toff[t][1] = (t >> 1) & 1; a.toff[t][2] = t & 1; }
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_CONVT, 2.0 * vox * 8 * Cin * Cout, 4.0 * (vox * (Cin + 8.0 * Cout) + 8.0 * Cin * Cout), st);
    dispatch_igemm(p, a, p.nM * p.nN, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// ---- strided / even-kernel Conv3d on the same kernel (KS = 1, no halo): every K-chunk is one (tap, channel chunk) and
// gathers its own input voxel  base * stride + tap - pad  (zero outside the volume).  Used for the k3 s2 p1 convs of the
// Residual U-Net (residual_unet3d.py:24-60) and V-Net's k2 s2 down-convolutions (vnet3d.py:66).
static bool gather_geom_ok(int k, int stride, int pad) {
    return k >= 1 && k <= 4 && stride >= 1 && stride <= k && pad >= 0 && pad < k && !(stride == 1 && (k == 1 || k == 3) && pad == k / 2);
}
static int out_extent(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

bool conv_gather_fwd_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    if (!gather_geom_ok(k, stride, pad) || (ldx % 4) || D + 2 * pad < k || H + 2 * pad < k || W + 2 * pad < k) return false;
    IgemmPlan p;
    return igemm_plan(1, N, out_extent(D, k, stride, pad), out_extent(H, k, stride, pad), out_extent(W, k, stride, pad), Cin, Cout, 1, &p);
}

// dgrad runs one launch per output phase (u mod stride): the taps that reach a phase are a fixed subset with fixed offsets
bool conv_gather_dgrad_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int lddy, int lddx) {
    if (!gather_geom_ok(k, stride, pad) || (lddy % 4) || D + 2 * pad < k || H + 2 * pad < k || W + 2 * pad < k) return false;
    if (D < stride || H < stride || W / stride < 4) return false;
    IgemmPlan p;
    return igemm_plan(1, N, D / stride, H / stride, W / stride, Cout, Cin, 1, &p);
}

size_t conv_gather_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    if (!gather_geom_ok(k, stride, pad) || D + 2 * pad < k || H + 2 * pad < k || W + 2 * pad < k) return 0;
    const size_t T = (size_t)k * k * k;
    size_t need = align_up(T * Cin * Cout * sizeof(float), 256) + 2048;
    IgemmPlan p;
    if (igemm_plan(1, N, out_extent(D, k, stride, pad), out_extent(H, k, stride, pad), out_extent(W, k, stride, pad), Cin, Cout, 1, &p))
        need += align_up((size_t)p.nM * Cout * 3 * sizeof(float), 256);
    return need;
}

int conv_gather_fwd_mfma(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int N, int D, int H, int W,
                         int Cin, int Cout, int k, int stride, int pad, double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st) {
    const int Do = out_extent(D, k, stride, pad), Ho = out_extent(H, k, stride, pad), Wo = out_extent(W, k, stride, pad);
    IgemmPlan p;
    SEG_CHECK_ARG(igemm_plan(1, N, Do, Ho, Wo, Cin, Cout, 1, &p), "conv_gather_fwd_mfma: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0, "conv_gather_fwd_mfma: input pointer must be 16-byte aligned");
    const int T = k * k * k, cpt = Cin / p.CK, nchunks = T * cpt;
    Carver cv(ws);
    float* wq = cv.take<float>((size_t)T * Cin * Cout);
    float* spart = ssum ? cv.take<float>((size_t)p.nM * Cout * 3) : nullptr;
    SEG_CHECK_WS(cv.used(), ws_bytes);
    hipLaunchKernelGGL(pack_wq_kernel, dim3(pack_grid((long long)T * Cin * Cout)), dim3(256), 0, st, w, wq, T * Cin, Cout, 1, 32 * p.NBW, 5, Cin, p.CK, T, TapList{});
    SEG_CHECK_LAUNCH();
    IgemmArgs a{x, wq, bias, y, spart, ldx, ldy, N, Do, Ho, Wo, Cout, p.ntx, p.nty, p.ntz, p.nN, nchunks, cpt, p.nN, stride, 1,
                p.nM, 1, nchunks, 0, 0};
    a.Di = D; a.Hi = H; a.Wi = W; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    for (int t = 0; t < T; ++t) { a.toff[t][0] = (signed char)(t / (k * k) - pad); a.toff[t][1] = (signed char)((t / k) % k - pad); a.toff[t][2] = (signed char)(t % k - pad); }
    const double vox = (double)N * Do * Ho * Wo;
    {
        ProfScope ps(PF_IGEMM, 2.0 * vox * T * Cin * Cout, 4.0 * ((double)N * D * H * W * Cin + vox * Cout + (double)T * Cin * Cout), st);
        dispatch_igemm(p, a, p.nM * p.nN, st);
        SEG_CHECK_LAUNCH();
    }
    if (ssum) {
        hipLaunchKernelGGL(igemm_stats_finalize_kernel, dim3(Cout), dim3(256), 0, st, spart, p.nM, Cout, ssum, ssq);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

// dx[N,D,H,W,Cin] from dy[N,Do,Ho,Wo,Cout]:  dx[u] = sum over taps t with (u + pad - t) % stride == 0 of dy[(u + pad - t) / stride] W[.,.,t]
int conv_gather_dgrad_mfma(const float* dy, int lddy, const float* w, float* dx, int lddx, int N, int D, int H, int W,
                           int Cin, int Cout, int k, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t st) {
    const int Do = out_extent(D, k, stride, pad), Ho = out_extent(H, k, stride, pad), Wo = out_extent(W, k, stride, pad);
    SEG_CHECK_ARG(((uintptr_t)dy % 16) == 0, "conv_gather_dgrad_mfma: gradient pointer must be 16-byte aligned");
    const int T = k * k * k;
    Carver cv(ws);
    float* wq_all = cv.take<float>((size_t)T * Cin * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    size_t wq_used = 0;
    ProfScope ps(PF_IGEMM, 2.0 * (double)N * Do * Ho * Wo * T * Cin * Cout,
                 4.0 * ((double)N * D * H * W * Cin + (double)N * Do * Ho * Wo * Cout + (double)T * Cin * Cout), st);
    for (int ph = 0; ph < stride * stride * stride; ++ph) {
        const int pz = ph / (stride * stride), py = (ph / stride) % stride, px = ph % stride;
        const int Bz = (D - pz + stride - 1) / stride, By = (H - py + stride - 1) / stride, Bx = (W - px + stride - 1) / stride;
        if (Bz <= 0 || By <= 0 || Bx <= 0) continue;
        int lz[4], ly[4], lx[4], dz[4], dyo[4], dxo[4], nz = 0, ny = 0, nx = 0;
        for (int t = 0; t < k; ++t) {
            if ((pz + pad - t) % stride == 0) { lz[nz] = t; dz[nz++] = (pz + pad - t) / stride; }
            if ((py + pad - t) % stride == 0) { ly[ny] = t; dyo[ny++] = (py + pad - t) / stride; }
            if ((px + pad - t) % stride == 0) { lx[nx] = t; dxo[nx++] = (px + pad - t) / stride; }
        }
        const int nt = nz * ny * nx;
        SEG_CHECK_ARG(nt > 0, "conv_gather_dgrad_mfma: a phase without taps (k < stride)");
        IgemmPlan p;
        SEG_CHECK_ARG(igemm_plan(1, N, Bz, By, Bx, Cout, Cin, 1, &p), "conv_gather_dgrad_mfma: unsupported shape");
        const int cpt = Cout / p.CK, nchunks = nt * cpt;
        float* wq = wq_all + wq_used;
        wq_used += (size_t)nt * Cin * Cout;
        IgemmArgs a{dy, wq, nullptr, dx, nullptr, lddy, lddx, N, Bz, By, Bx, Cin, p.ntx, p.nty, p.ntz, p.nN, nchunks, cpt, p.nN, 1, stride,
                    p.nM, 1, nchunks, 0, 0};
        a.Di = Do; a.Hi = Ho; a.Wi = Wo; a.Do = D; a.Ho = H; a.Wo = W; a.cz = pz; a.cy = py; a.cx = px;
        TapList tl{};
        int slot = 0;
        for (int iz = 0; iz < nz; ++iz)
            for (int iy = 0; iy < ny; ++iy)
                for (int ix = 0; ix < nx; ++ix, ++slot) {
                    tl.t[slot] = (unsigned char)((lz[iz] * k + ly[iy]) * k + lx[ix]);
                    a.toff[slot][0] = (signed char)dz[iz]; a.toff[slot][1] = (signed char)dyo[iy]; a.toff[slot][2] = (signed char)dxo[ix];
                }
        hipLaunchKernelGGL(pack_wq_kernel, dim3(pack_grid((long long)nt * Cin * Cout)), dim3(256), 0, st, w, wq, nt * Cout, Cin, 1, 32 * p.NBW, 6, Cout, p.CK, T, tl);
        SEG_CHECK_LAUNCH();
        dispatch_igemm(p, a, p.nM * p.nN, st);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

}  // namespace seg
