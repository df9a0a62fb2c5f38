// conv_pw_wgrad.hip -- weight gradients that are plain "K = voxels" GEMMs, on the fp32 MFMA:
//   T = 1: Conv3d k1                dW[ci][co]    = sum_v x[v][ci] * dy[v][co]
//   T = 8: ConvTranspose3d k2 s2    dW[ci][co][t] = sum_v x[v][ci] * dy[child(v, t)][co]
// A workgroup (4 waves) owns one 32(ci) x 32(co) block pair and a strip of V-voxel tiles.
// x (A operand, shared by all taps) and the dy children (B operand) are staged in LDS with the
// next tile prefetched into registers during the MFMAs; both operands are conflict-free
// ds_read_b32 (32 consecutive channels of a voxel).  T = 8: every wave owns two taps; T = 1:
// the four waves split the tile's voxels and their accumulators are summed through LDS.
// Slabs part[strip][t][ci][co] are summed in fixed order by a second stage (deterministic).
#include "common.h"
#include "internal.h"

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct PwArgs {
    const void* x; const void* dy; float* part;      // x / dy: fp32 or bf16 (the kernel's IN_T), staged into LDS as fp32
    int ldx, lddy, N, D, H, W, Cin, Cout;
    int ntiles, nstrips, npairs, ncob;
};

template <int T>
struct PwCfg {
    static constexpr int V = (T == 8) ? 64 : 256;         // voxels per tile
    static constexpr int TPW = (T == 8) ? 2 : 1;          // taps per wave
    static constexpr int KSPLIT = (T == 8) ? 1 : 4;       // waves splitting the tile's voxels
    static constexpr int XP = V * 8, DP = T * V * 8;      // 16-byte pieces
    static constexpr int XIT = XP / 256, DIT = DP / 256;
    static constexpr int X_FLOATS = V * 32;
    static constexpr int LDS_BYTES = (V * 32 + T * V * 32) * 4;
};

template <int T, typename IN_T>
__global__ __launch_bounds__(256, 2) void pw_wgrad_kernel(PwArgs a) {
    using C = PwCfg<T>;
    const IN_T* __restrict__ xin = reinterpret_cast<const IN_T*>(a.x);
    const IN_T* __restrict__ dyin = reinterpret_cast<const IN_T*>(a.dy);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;
    float* ds = lds + C::X_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, i = lane & 31;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int pair = t % a.npairs, strip = t / a.npairs;
    const int ci0 = (pair / a.ncob) * 32, co0 = (pair % a.ncob) * 32;

    f32x16 acc[C::TPW];
#pragma unroll
    for (int tt = 0; tt < C::TPW; ++tt)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[tt][v] = 0.f;

    const long long nvox = (long long)a.N * a.D * a.H * a.W;
    f32x4 sx[C::XIT], sd[C::DIT];
    auto load_stage = [&](int tile) {
        const long long v0 = (long long)tile * C::V;
#pragma unroll
        for (int it = 0; it < C::XIT; ++it) {
            const int p = it * 256 + tid;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (v0 + (p >> 3) < nvox && ci0 + (p & 7) * 4 < a.Cin) xv = ld4(xin + (v0 + (p >> 3)) * a.ldx + ci0 + (p & 7) * 4);
            sx[it] = xv;
        }
#pragma unroll
        for (int it = 0; it < C::DIT; ++it) {
            const int p = it * 256 + tid;
            const int part = p & 7, vl = (p >> 3) % C::V, tap = (p >> 3) / C::V;
            long long ov;
            if (T == 8) {
                long long v = v0 + vl;
                const int xw = (int)(v % a.W); v /= a.W;
                const int yh = (int)(v % a.H); v /= a.H;
                const int zd = (int)(v % a.D); const int n = (int)(v / a.D);
                ov = (((long long)n * (2 * a.D) + 2 * zd + (tap >> 2)) * (2 * a.H) + 2 * yh + ((tap >> 1) & 1)) * (2 * a.W) + 2 * xw + (tap & 1);
            } else {
                ov = v0 + vl;
            }
            f32x4 dv = {0.f, 0.f, 0.f, 0.f};
            if (v0 + vl < nvox && co0 + part * 4 < a.Cout) dv = ld4(dyin + ov * a.lddy + co0 + part * 4);
            sd[it] = dv;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < C::XIT; ++it) *reinterpret_cast<f32x4*>(xs + (it * 256 + tid) * 4) = sx[it];
#pragma unroll
        for (int it = 0; it < C::DIT; ++it) *reinterpret_cast<f32x4*>(ds + (it * 256 + tid) * 4) = sd[it];
    };

    int tile = strip;
    if (tile < a.ntiles) load_stage(tile);
    for (; tile < a.ntiles; tile += a.nstrips) {
        __syncthreads();
        write_stage();
        __syncthreads();
        if (tile + a.nstrips < a.ntiles) load_stage(tile + a.nstrips);
        constexpr int KSTEPS = C::V / 2 / C::KSPLIT;
        const int kbase = (C::KSPLIT > 1 ? wave : 0) * KSTEPS;
        const float* xa = xs + (2 * kbase + h) * 32 + i;
        const float* db = ds + ((C::KSPLIT > 1 ? 0 : wave * C::TPW) * C::V + 2 * kbase + h) * 32 + i;
#pragma unroll 8
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const float av = xa[ks * 64];
#pragma unroll
            for (int tt = 0; tt < C::TPW; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, db[tt * C::V * 32 + ks * 64], acc[tt], 0, 0, 0);
        }
    }

    if (C::KSPLIT > 1) {                 // sum the four waves' accumulators through LDS
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 16; ++v) lds[(wave * 16 + v) * 64 + lane] = acc[0][v];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v)
                acc[0][v] = lds[v * 64 + lane] + lds[(16 + v) * 64 + lane] + lds[(32 + v) * 64 + lane] + lds[(48 + v) * 64 + lane];
        }
    }
    if (C::KSPLIT == 1 || wave == 0) {
#pragma unroll
        for (int tt = 0; tt < C::TPW; ++tt) {
            const int tap = C::KSPLIT > 1 ? 0 : wave * C::TPW + tt;
            float* dst = a.part + (((long long)strip * T + tap) * a.Cin + ci0) * a.Cout + co0 + i;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
                if (ci0 + r < a.Cin && co0 + i < a.Cout) dst[(long long)r * a.Cout] = acc[tt][v];      // ragged last blocks
            }
        }
    }
}

struct PwPlan { int ntiles, nstrips, npairs; };

static bool pw_plan(long long nvox, int Cin, int Cout, int T, PwPlan* p) {
    if (Cin % 4 || Cout % 4 || Cin < 16 || Cout < 16 || (T != 1 && T != 8)) return false;    // 16-byte channel pieces; blocks of 32 zero-padded
    const int V = T == 8 ? 64 : 256;
    p->ntiles = (int)((nvox + V - 1) / V);
    p->npairs = ((Cin + 31) / 32) * ((Cout + 31) / 32);
    int want = 512 / p->npairs;                        // two workgroups per CU: never more than 512 in all
    long long cap = (long long)(64u << 20) / ((long long)T * Cin * Cout * 4);
    if (cap < 1) cap = 1;
    if (want > cap) want = (int)cap;
    if (want > p->ntiles) want = p->ntiles;
    if (want < 1) want = 1;
    p->nstrips = want;
    return true;
}

size_t pw_wgrad_ws_bytes(long long nvox, int Cin, int Cout, int T) {
    PwPlan p;
    if (!pw_plan(nvox, Cin, Cout, T, &p)) return 0;
    return align_up((size_t)p.nstrips * T * Cin * Cout * sizeof(float), 256) + 1024;
}

bool pw_wgrad_supported(long long nvox, int Cin, int Cout, int T, int ldx, int lddy) {
    PwPlan p;
    return (ldx % 4) == 0 && (lddy % 4) == 0 && pw_plan(nvox, Cin, Cout, T, &p);
}

// returns the slab pointer/strip count through *part_out / *nstrips_out; the caller runs the layout-specific reduce
template <typename IN_T>
int pw_wgrad_mfma(const IN_T* dy, int lddy, const IN_T* x, int ldx, int N, int D, int H, int W, int Cin, int Cout, int T,
                  float** part_out, int* nstrips_out, void* ws, size_t ws_bytes, hipStream_t st) {
    PwPlan p;
    const long long nvox = (long long)N * D * H * W;
    SEG_CHECK_ARG(pw_plan(nvox, Cin, Cout, T, &p), "pw_wgrad_mfma: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % (4 * sizeof(IN_T))) == 0 && ((uintptr_t)dy % (4 * sizeof(IN_T))) == 0, "pw_wgrad_mfma: pointers must be aligned to four elements");
    Carver cv(ws);
    float* part = cv.take<float>((size_t)p.nstrips * T * Cin * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    PwArgs a{x, dy, part, ldx, lddy, N, D, H, W, Cin, Cout, p.ntiles, p.nstrips, p.npairs, (Cout + 31) / 32};
    const int nwg = p.nstrips * p.npairs;
    ProfScope ps(T == 8 ? PF_CONVT : PF_WGRAD, 2.0 * nvox * T * Cin * Cout, (double)sizeof(IN_T) * nvox * (Cin + (double)T * Cout) + 4.0 * T * Cin * Cout, st);
    if (T == 8) {
        SEG_SET_LDS((pw_wgrad_kernel<8, IN_T>), PwCfg<8>::LDS_BYTES);
        hipLaunchKernelGGL((pw_wgrad_kernel<8, IN_T>), dim3(nwg), dim3(256), PwCfg<8>::LDS_BYTES, st, a);
    } else {
        SEG_SET_LDS((pw_wgrad_kernel<1, IN_T>), PwCfg<1>::LDS_BYTES);
        hipLaunchKernelGGL((pw_wgrad_kernel<1, IN_T>), dim3(nwg), dim3(256), PwCfg<1>::LDS_BYTES, st, a);
    }
    SEG_CHECK_LAUNCH();
    *part_out = part; *nstrips_out = p.nstrips;
    return MI355SEG_OK;
}
template int pw_wgrad_mfma<float>(const float*, int, const float*, int, int, int, int, int, int, int, int, float**, int*, void*, size_t, hipStream_t);
template int pw_wgrad_mfma<bf16>(const bf16*, int, const bf16*, int, int, int, int, int, int, int, int, float**, int*, void*, size_t, hipStream_t);

}  // namespace seg
