// conv_wgrad_mfma.hip -- weight gradient of Conv3d k3 s1 p1 on the fp32 matrix cores.
//
//   dW[tap][ci][co] = sum over voxels v of  x[v + tap][ci] * dy[v][co]
//
// i.e. 27 GEMMs (one per tap) with M = Cin, N = Cout and K = all output voxels.  A
// workgroup (8 waves, two per SIMD so each covers the other's LDS / barrier stalls) owns one
// 32(ci) x 32(co) block pair and a strip of 256-voxel spatial tiles; its 27 tap-tiles are dealt to
// the eight waves (4/4/4/3/3/3/3/3 = 7/7/7/6 per SIMD), so the whole 27 x 32 x 32 slab stays in accumulators while the
// workgroup walks its strip.  Per tile the x halo (32 channels) and the dy tile live in
// LDS; both MFMA operands are conflict-free ds_read_b32 (lanes = 32 consecutive channels
// of one voxel) and the dy fragment of a k-step is shared by the wave's 7 MFMAs.  The
// next tile is prefetched into registers during the MFMAs (issue early / write late).
// Each workgroup writes its slab once; a fixed-order second stage sums the strips and
// emits the PyTorch (Cout,Cin,3,3,3) layout -> bitwise reproducible, no atomics.
//
// k5 p2 (V-Net's LUConv, vnet3d.py:21-31) runs on the same kernel one tap PLANE at a time: a workgroup owns the
// 25 taps (dy, dx) of one dz, so its x halo is the z-shifted slab TZ x (TY+4) x (BX+4) and the 25 tap-tiles are
// dealt 4/3/3/3/3/3/3/3; the five planes of a (block pair, strip) are five workgroups.
#include "common.h"
#include "internal.h"
#include <initializer_list>

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WG_WAVES = 8;                 // 2 waves per SIMD cover each other's LDS / barrier stalls
constexpr int WG_THREADS = WG_WAVES * 64;
constexpr int TPW = (27 + WG_WAVES - 1) / WG_WAVES;   // tap-tiles per wave (4 for 27 and for 25 taps; the last waves own one fewer)

template <int BX, int KS>
struct WTile {
    static constexpr int HALO = KS / 2;
    static constexpr int NTAPS = KS == 3 ? 27 : KS * KS;          // taps per workgroup (k5: one dz plane)
    static constexpr int PLANES = KS == 3 ? 1 : KS;
    static constexpr int TY = BX == 8 ? 8 : 4;
    static constexpr int LINES = 256 / BX;
    static constexpr int TZ = LINES / TY;
    static constexpr int HX = BX + 2 * HALO, HY = TY + 2 * HALO, HZ = KS == 3 ? TZ + 2 : TZ;
    static_assert((NTAPS + WG_WAVES - 1) / WG_WAVES == TPW, "tap dealing assumes 4 tap-tiles per wave");
    static constexpr int NVOX = HX * HY * HZ;
    static constexpr int XPIECES = NVOX * 8;              // 16-byte pieces of the x halo (32 ch)
    static constexpr int XITER = (XPIECES + WG_THREADS - 1) / WG_THREADS;
    static constexpr int DYITER = 256 * 8 / WG_THREADS;   // dy tile: 256 voxels x 32 ch
    static constexpr int X_FLOATS = NVOX * 32;
    static constexpr int LDS_BYTES = (X_FLOATS + 256 * 32) * 4;
};

struct WgradArgs {
    const float* x; const float* dy; float* part;
    int ldx, lddy, N, D, H, W, Cin, Cout;
    int ntx, nty, ntz, ntiles, nstrips, npairs, ncob, ntaps_total;
};

// MFMAs of one staged tile for a wave that owns NT tap-tiles: 128 k-steps (2 voxels each), fully
// unrolled so every LDS offset is an immediate, with the operands of step k+1 read before the MFMAs
// of step k are issued (explicit one-step software pipeline; the compiler does not build it itself).
template <int BX, int KS, int NT>
__device__ __forceinline__ void wgrad_tile_mfma(f32x16 (&acc)[TPW], const float* __restrict__ xs, const float* __restrict__ ds,
                                                const int (&abase)[TPW], int bbase) {
    using T = WTile<BX, KS>;
    constexpr int KSTEPS = 128;
    auto xoff = [](int ks) { const int line = ks / (BX / 2), xp = ks % (BX / 2);
                             return (((line / T::TY) * T::HY + (line % T::TY)) * T::HX) * 32 + xp * 64; };
    auto doff = [](int ks) { const int line = ks / (BX / 2), xp = ks % (BX / 2); return line * BX * 32 + xp * 64; };
    float ac[NT], an[NT], bc, bn = 0.f;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) { ac[tt] = xs[abase[tt] + xoff(0)]; an[tt] = 0.f; }
    bc = ds[bbase + doff(0)];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        if (ks + 1 < KSTEPS) {
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) an[tt] = xs[abase[tt] + xoff(ks + 1)];
            bn = ds[bbase + doff(ks + 1)];
        }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[tt], bc, acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) ac[tt] = an[tt];
        bc = bn;
    }
}

template <int BX, int KS>
__global__ __launch_bounds__(WG_THREADS, 2) void conv_wgrad_kernel(WgradArgs a) {
    using T = WTile<BX, KS>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* ds = lds + T::X_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, i = lane & 31;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int plane = t % T::PLANES, tp = t / T::PLANES;          // k5: which dz plane of taps this workgroup owns
    const int pair = tp % a.npairs, strip = tp / a.npairs;
    const int cib = pair / a.ncob, cob = pair % a.ncob;
    const int ci0 = cib * 32, co0 = cob * 32;

    // the taps of this wave: wave, wave + 8, wave + 16 (, wave + 24 for the first waves)
    const bool has_last = wave + WG_WAVES * (TPW - 1) <= T::NTAPS - 1;       // wave-uniform
    int abase[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        int tap = wave + WG_WAVES * tt;
        if (tap > T::NTAPS - 1) tap = T::NTAPS - 1;
        const int dz = KS == 3 ? tap / 9 : 0, dy = KS == 3 ? (tap / 3) % 3 : tap / KS, dx = tap % KS;
        abase[tt] = (((dz * T::HY + dy) * T::HX + dx) + h) * 32 + i;
    }
    const int bbase = h * 32 + i;

    f32x16 acc[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[tt][v] = 0.f;

    f32x4 sx[T::XITER], sd[T::DYITER];
    auto load_stage = [&](int tile) {
        int mt = tile;
        const int txi = mt % a.ntx; mt /= a.ntx;
        const int tyi = mt % a.nty; mt /= a.nty;
        const int tzi = mt % a.ntz;
        const int n = mt / a.ntz;
        const int x0 = txi * BX, y0 = tyi * T::TY, z0 = tzi * T::TZ;
#pragma unroll
        for (int it = 0; it < T::XITER; ++it) {
            const int p = it * WG_THREADS + tid;
            const int vox = p >> 3, part = p & 7;
            const int hz = vox / (T::HY * T::HX), rem = vox % (T::HY * T::HX);
            const int hy = rem / T::HX, hx = rem % T::HX;
            const int gz = z0 + hz + (KS == 3 ? -1 : plane - T::HALO), gy = y0 - T::HALO + hy, gx = x0 - T::HALO + hx;
            const bool ok = (p < T::XPIECES) && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(a.x + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldx + ci0 + part * 4);
            sx[it] = v;
        }
#pragma unroll
        for (int it = 0; it < T::DYITER; ++it) {
            const int p = it * WG_THREADS + tid;
            const int vox = p >> 3, part = p & 7;
            const int line = vox / BX, xx = vox % BX;
            const int gz = z0 + line / T::TY, gy = y0 + line % T::TY, gx = x0 + xx;
            f32x4 dv = {0.f, 0.f, 0.f, 0.f};          // partial tiles: voxels outside the volume contribute nothing
            if (gz < a.D && gy < a.H && gx < a.W)
                dv = *reinterpret_cast<const f32x4*>(a.dy + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.lddy + co0 + part * 4);
            sd[it] = dv;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < T::XITER; ++it) {
            const int p = it * WG_THREADS + tid;
            if (p < T::XPIECES) *reinterpret_cast<f32x4*>(xs + p * 4) = sx[it];
        }
#pragma unroll
        for (int it = 0; it < T::DYITER; ++it) *reinterpret_cast<f32x4*>(ds + (it * WG_THREADS + tid) * 4) = sd[it];
    };

    int tile = strip;
    if (tile < a.ntiles) load_stage(tile);
    for (; tile < a.ntiles; tile += a.nstrips) {
        __syncthreads();
        write_stage();
        __syncthreads();
        if (tile + a.nstrips < a.ntiles) load_stage(tile + a.nstrips);
        if (has_last) wgrad_tile_mfma<BX, KS, TPW>(acc, xs, ds, abase, bbase);
        else wgrad_tile_mfma<BX, KS, TPW - 1>(acc, xs, ds, abase, bbase);
    }

    // slab store: part[strip][tap][ci][co]; rows of the 32x32 tile = ci, lanes (cols) = co
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int tap = wave + WG_WAVES * tt;
        if (tap > T::NTAPS - 1) break;
        float* dst = a.part + (((long long)strip * a.ntaps_total + plane * T::NTAPS + tap) * a.Cin + ci0) * a.Cout + co0 + i;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
            dst[(long long)r * a.Cout] = acc[tt][v];
        }
    }
}

struct WgradPlan { int KS, BX, ntx, nty, ntz, ntiles, nstrips, npairs, taps, planes; };

static bool wgrad_plan(int KS, int N, int D, int H, int W, int Cin, int Cout, WgradPlan* p) {
    if ((KS != 3 && KS != 5) || Cin % 32 || Cout % 32 || W < 4) return false;
    int BX = 0; long long best = -1;
    for (int bx : {32, 16, 8}) {
        long long padded = (long long)((W + bx - 1) / bx) * bx;
        if (best < 0 || padded < best) { best = padded; BX = bx; }
    }
    const int TY = BX == 8 ? 8 : 4, TZ = (256 / BX) / TY;
    p->KS = KS; p->BX = BX; p->ntx = (W + BX - 1) / BX; p->nty = (H + TY - 1) / TY; p->ntz = (D + TZ - 1) / TZ;
    p->ntiles = N * p->ntz * p->nty * p->ntx;
    p->npairs = (Cin / 32) * (Cout / 32);
    p->taps = KS * KS * KS; p->planes = KS == 3 ? 1 : KS;
    const int per_strip = p->npairs * p->planes;
    int want = 256 / per_strip;                        // one workgroup per CU (106-136 KB of LDS each): never more than 256 in all
    long long cap = (long long)(160u << 20) / ((long long)p->taps * Cin * Cout * 4);   // keep the slab workspace <= 160 MB
    if (cap < 1) cap = 1;
    if (want > cap) want = (int)cap;
    if (want > p->ntiles) want = p->ntiles;
    if (want < 1) want = 1;
    p->nstrips = want;
    return true;
}

bool wgrad_mfma_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int lddy) {
    if (!((k == 3 && pad == 1) || (k == 5 && pad == 2)) || stride != 1 || (ldx % 4) || (lddy % 4)) return false;
    WgradPlan p;
    return wgrad_plan(k, N, D, H, W, Cin, Cout, &p);
}

size_t wgrad_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k) {
    WgradPlan p;
    if (!wgrad_plan(k, N, D, H, W, Cin, Cout, &p)) return 0;
    return align_up((size_t)p.nstrips * p.taps * Cin * Cout * sizeof(float), 256) + 1024;
}

template <int BX, int KS>
static void launch_wgrad(const WgradArgs& a, int nwg, hipStream_t st) {
    using T = WTile<BX, KS>;
    SEG_SET_LDS((conv_wgrad_kernel<BX, KS>), T::LDS_BYTES);
    hipLaunchKernelGGL((conv_wgrad_kernel<BX, KS>), dim3(nwg), dim3(WG_THREADS), T::LDS_BYTES, st, a);
}

int conv_wgrad_mfma(const float* dy, int lddy, const float* x, int ldx, float* dw, int N, int D, int H, int W, int Cin,
                    int Cout, int k, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    WgradPlan p;
    SEG_CHECK_ARG(wgrad_plan(k, N, D, H, W, Cin, Cout, &p), "conv_wgrad_mfma: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "conv_wgrad_mfma: pointers must be 16-byte aligned");
    Carver cv(ws);
    float* part = cv.take<float>((size_t)p.nstrips * p.taps * Cin * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    WgradArgs a{x, dy, part, ldx, lddy, N, D, H, W, Cin, Cout, p.ntx, p.nty, p.ntz, p.ntiles, p.nstrips, p.npairs, Cout / 32, p.taps};
    const int nwg = p.nstrips * p.npairs * p.planes;
    const double vox = (double)N * D * H * W;
    {
        ProfScope ps(PF_WGRAD, 2.0 * vox * p.taps * Cin * Cout, 4.0 * (vox * (Cin + Cout) + (double)p.taps * Cin * Cout), st);
        if (k == 3) {
            if (p.BX == 32) launch_wgrad<32, 3>(a, nwg, st);
            else if (p.BX == 16) launch_wgrad<16, 3>(a, nwg, st);
            else launch_wgrad<8, 3>(a, nwg, st);
        } else {
            if (p.BX == 32) launch_wgrad<32, 5>(a, nwg, st);
            else if (p.BX == 16) launch_wgrad<16, 5>(a, nwg, st);
            else launch_wgrad<8, 5>(a, nwg, st);
        }
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, p.nstrips, p.taps, Cin, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // namespace seg
