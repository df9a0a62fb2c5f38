// conv_gwgrad.hip -- weight gradient of a Conv3d with ANY cubic kernel / stride / padding on the fp32 MFMA
// (k5 p2 and k2 s2 of the V-Net, k3 s2 of the residual U-Net, ragged channel counts):
//
//   dW[tap][ci][co] = sum over output voxels v of  x[in(v, tap)][ci] * dy[v][co]
//
// One GEMM per tap with K = output voxels.  A workgroup (4 waves) owns a 32(ci) x 32(co) block pair, a group of
// up to 8 taps (two per wave) and a strip of 64-voxel tiles: the dy tile (shared by the taps) and the 8
// tap-shifted gathers of x are staged in LDS (72 KB, two workgroups per CU), the next tile is prefetched into
// registers during the MFMAs, both operands are conflict-free ds_read_b32.  Channel blocks are zero-padded, so
// any Cin / Cout that is a multiple of 4 works.  Slabs part[strip][tap][ci][co] -> fixed-order second stage.
#include "common.h"
#include "internal.h"

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct GwArgs {
    const void* x; const void* dy; float* part;      // x / dy: fp32 or bf16 (the kernel's IN_T), staged into LDS as fp32
    int ldx, lddy, N, D, H, W, Do, Ho, Wo, Cin, Cout, k, stride, pad, T;
    int ntiles, nstrips, npairs, ncob, ngroups;
};

constexpr int GW_V = 64, GW_TG = 8;
constexpr int GW_XIT = GW_TG * GW_V * 8 / 256, GW_DIT = GW_V * 8 / 256;
constexpr int GW_LDS = (GW_V * 32 + GW_TG * GW_V * 32) * 4;

template <typename IN_T>
__global__ __launch_bounds__(256, 2) void conv_gwgrad_kernel(GwArgs a) {
    const IN_T* __restrict__ xin = reinterpret_cast<const IN_T*>(a.x);
    const IN_T* __restrict__ dyin = reinterpret_cast<const IN_T*>(a.dy);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ds = lds;                       // [V][32]
    float* xs = lds + GW_V * 32;           // [TG][V][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, i = lane & 31;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tg = t % a.ngroups; t /= a.ngroups;
    const int pair = t % a.npairs, strip = t / a.npairs;
    const int ci0 = (pair / a.ncob) * 32, co0 = (pair % a.ncob) * 32;
    const long long nvox = (long long)a.N * a.Do * a.Ho * a.Wo;

    f32x16 acc[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[tt][v] = 0.f;

    f32x4 sx[GW_XIT], sd[GW_DIT];
    auto load_stage = [&](int tile) {
        const long long v0 = (long long)tile * GW_V;
#pragma unroll
        for (int it = 0; it < GW_DIT; ++it) {
            const int p = it * 256 + tid;
            const int part = p & 7, vl = p >> 3;
            f32x4 dv = {0.f, 0.f, 0.f, 0.f};
            if (v0 + vl < nvox && co0 + part * 4 < a.Cout) dv = ld4(dyin + (v0 + vl) * a.lddy + co0 + part * 4);
            sd[it] = dv;
        }
#pragma unroll
        for (int it = 0; it < GW_XIT; ++it) {
            const int p = it * 256 + tid;
            const int part = p & 7, vl = (p >> 3) % GW_V, tl = (p >> 3) / GW_V;
            const int tap = tg * GW_TG + tl;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            long long v = v0 + vl;
            if (tap < a.T && v < nvox && ci0 + part * 4 < a.Cin) {
                const int ow = (int)(v % a.Wo); v /= a.Wo;
                const int oh = (int)(v % a.Ho); v /= a.Ho;
                const int od = (int)(v % a.Do); const int n = (int)(v / a.Do);
                const int kw = tap % a.k, kh = (tap / a.k) % a.k, kd = tap / (a.k * a.k);
                const int iz = od * a.stride - a.pad + kd, iy = oh * a.stride - a.pad + kh, ix = ow * a.stride - a.pad + kw;
                if ((unsigned)iz < (unsigned)a.D && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                    xv = ld4(xin + ((((long long)n * a.D + iz) * a.H + iy) * a.W + ix) * a.ldx + ci0 + part * 4);
            }
            sx[it] = xv;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < GW_DIT; ++it) *reinterpret_cast<f32x4*>(ds + (it * 256 + tid) * 4) = sd[it];
#pragma unroll
        for (int it = 0; it < GW_XIT; ++it) *reinterpret_cast<f32x4*>(xs + (it * 256 + tid) * 4) = sx[it];
    };

    int tile = strip;
    if (tile < a.ntiles) load_stage(tile);
    for (; tile < a.ntiles; tile += a.nstrips) {
        __syncthreads();
        write_stage();
        __syncthreads();
        if (tile + a.nstrips < a.ntiles) load_stage(tile + a.nstrips);
        const float* xa = xs + (wave * 2 * GW_V + h) * 32 + i;
        const float* db = ds + h * 32 + i;
        float a0 = xa[0], a1 = xa[GW_V * 32], b = db[0];
#pragma unroll
        for (int ks = 0; ks < GW_V / 2; ++ks) {
            float na0 = 0.f, na1 = 0.f, nb = 0.f;
            if (ks + 1 < GW_V / 2) { na0 = xa[(ks + 1) * 64]; na1 = xa[GW_V * 32 + (ks + 1) * 64]; nb = db[(ks + 1) * 64]; }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
            a0 = na0; a1 = na1; b = nb;
        }
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int tap = tg * GW_TG + wave * 2 + tt;
        if (tap >= a.T || co0 + i >= a.Cout) continue;
        float* dst = a.part + (((long long)strip * a.T + tap) * a.Cin + ci0) * a.Cout + co0 + i;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
            if (ci0 + r < a.Cin) dst[(long long)r * a.Cout] = acc[tt][v];
        }
    }
}

struct GwPlan { int ntiles, nstrips, npairs, ngroups, T; };

static bool gw_plan(int N, int Do, int Ho, int Wo, int Cin, int Cout, int k, GwPlan* p) {
    if (Cin % 4 || Cout % 4 || Cin < 8 || Cout < 8) return false;
    const long long nvox = (long long)N * Do * Ho * Wo;
    p->T = k * k * k;
    p->ntiles = (int)((nvox + GW_V - 1) / GW_V);
    p->npairs = ((Cin + 31) / 32) * ((Cout + 31) / 32);
    p->ngroups = (p->T + GW_TG - 1) / GW_TG;
    long long units = (long long)p->npairs * p->ngroups;
    long long want = (768 + units - 1) / units;
    long long cap = (long long)(96u << 20) / ((long long)p->T * Cin * Cout * 4);      // slab workspace <= 96 MB
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want > p->ntiles) want = p->ntiles;
    if (want < 1) want = 1;
    p->nstrips = (int)want;
    return units * want < (1ll << 30);
}

size_t gwgrad_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    GwPlan p;
    const int Do = (D + 2 * pad - k) / stride + 1, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (!gw_plan(N, Do, Ho, Wo, Cin, Cout, k, &p)) return 0;
    return align_up((size_t)p.nstrips * p.T * Cin * Cout * sizeof(float), 256) + 1024;
}

bool gwgrad_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int lddy) {
    GwPlan p;
    const int Do = (D + 2 * pad - k) / stride + 1, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    return (ldx % 4) == 0 && (lddy % 4) == 0 && k <= 7 && gw_plan(N, Do, Ho, Wo, Cin, Cout, k, &p);
}

template <typename IN_T>
int conv_gwgrad(const IN_T* dy, int lddy, const IN_T* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout,
                int k, int stride, int pad, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    GwPlan p;
    const int Do = (D + 2 * pad - k) / stride + 1, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    SEG_CHECK_ARG(gw_plan(N, Do, Ho, Wo, Cin, Cout, k, &p), "conv_gwgrad: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % (4 * sizeof(IN_T))) == 0 && ((uintptr_t)dy % (4 * sizeof(IN_T))) == 0, "conv_gwgrad: pointers must be aligned to four elements");
    Carver cv(ws);
    float* part = cv.take<float>((size_t)p.nstrips * p.T * Cin * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    GwArgs a{x, dy, part, ldx, lddy, N, D, H, W, Do, Ho, Wo, Cin, Cout, k, stride, pad, p.T,
             p.ntiles, p.nstrips, p.npairs, (Cout + 31) / 32, p.ngroups};
    const int nwg = p.nstrips * p.npairs * p.ngroups;
    const double vox = (double)N * Do * Ho * Wo;
    {
        ProfScope ps(PF_WGRAD, 2.0 * vox * p.T * Cin * Cout, (double)sizeof(IN_T) * vox * ((double)p.T * Cin + Cout) + 4.0 * p.T * Cin * Cout, st);
        SEG_SET_LDS((conv_gwgrad_kernel<IN_T>), GW_LDS);
        hipLaunchKernelGGL(conv_gwgrad_kernel<IN_T>, dim3(nwg), dim3(256), GW_LDS, st, a);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, p.nstrips, p.T, Cin, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

template int conv_gwgrad<float>(const float*, int, const float*, int, float*, int, int, int, int, int, int, int, int, int, int, void*, size_t, hipStream_t);
template int conv_gwgrad<bf16>(const bf16*, int, const bf16*, int, float*, int, int, int, int, int, int, int, int, int, int, void*, size_t, hipStream_t);

}  // namespace seg
