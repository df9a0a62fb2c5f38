// convt_wgrad_lowp.hip -- weight gradient of ConvTranspose3d k2 s2 on the bf16 matrix cores.
//
//   dW[ci][co][tap] = sum over base voxels v of  x[v][ci] * dy[child(v, tap)][co]          (eight taps = the 2x2x2 children)
//
// Eight "K = voxels" GEMMs that share their A operand.  Same operand path as conv_wgrad_lowp.hip: tiles stay in LDS as they
// arrive ([voxel][32 channels], 64 * NP byte rows, NP = 3 bf16 planes h | m | l of an fp32 tensor -- the bf16x6 policy -- or
// one plane of a bf16 tensor) and both MFMA operands come from the transposing read ds_read_b64_tr_b16.  A workgroup (four
// waves) owns a 64 (ci) x 32 (co) block pair and a strip of V-voxel tiles; wave w owns taps 2w, 2w + 1 for both 32-channel
// halves of ci: four accumulators, and every fragment it reads feeds two MFMA groups (8 NP reads per 4 MFMA groups; one
// tap and one ci block per wave would read 4 NP per group and leave the matrix core waiting on LDS).  The fp32 MFMA kernel
// this replaces (pw_wgrad_kernel<8>) ran the U-Net's four ConvT weight gradients in 0.75 ms at cfg 2 and UNETR's ten in
// 1.45 ms.  Slabs part[strip][tap][ci][co] -> convt_wgrad_reduce_kernel (fixed order: deterministic).
#include "common.h"
#include "internal.h"
#include <type_traits>

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int NP>
struct CwCfg {
    static constexpr int V = NP == 3 ? 32 : 64;               // base voxels per tile (three planes: half the tile)
    static constexpr int ROW = 64 * NP;                       // bytes per voxel row of a 32-channel sub-tile
    static constexpr int X_BYTES = 2 * V * ROW, D_BYTES = 8 * V * ROW;
    static constexpr int LDS_BYTES = X_BYTES + D_BYTES;       // 60 KB (bf16x6) / 40 KB (bf16)
    static constexpr int KSTEPS = V / 16;
};

struct CwArgs {
    const void* x; const void* dy; float* part;
    int ldx, lddy, N, D, H, W, Cin, Cout;
    int ntiles, nstrips, npairs, ncob;
};

__device__ __forceinline__ bf16x8_t cw_frag(const unsigned char* p, int row_bytes) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * row_bytes));
    return __builtin_bit_cast(bf16x8_t, (s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int NP, typename IN_T>
__global__ __launch_bounds__(256, 2) void convt_wgrad_lowp_kernel(CwArgs a) {
    using C = CwCfg<NP>;
    constexpr int EPP = std::is_same<IN_T, float>::value ? 4 : 8;            // elements per staged 16-byte piece
    constexpr int PPV = 32 / EPP;                                             // pieces per voxel per 32 channels
    constexpr int XIT = C::V * 2 * PPV / 256, DIT = 8 * C::V * PPV / 256;
    static_assert(C::V * 2 * PPV % 256 == 0 && 8 * C::V * PPV % 256 == 0, "tiles must split evenly over the workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xs = lds;
    unsigned char* ds = lds + C::X_BYTES;
    const IN_T* __restrict__ xin = reinterpret_cast<const IN_T*>(a.x);
    const IN_T* __restrict__ din = reinterpret_cast<const IN_T*>(a.dy);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, i = lane & 31;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int pair = t % a.npairs, strip = t / a.npairs;
    const int ci0 = (pair / a.ncob) * 64, co0 = (pair % a.ncob) * 32;

    // transposing-read lane geometry (conv_wgrad_lowp.hip): lane 4q + p of a 16-lane group addresses voxel row q, channels
    // 4p .. 4p+3 of the group's 16-channel half; group (h, cg) covers k = 8h .. 8h+7 and channels 16cg .. 16cg+15
    const int li = lane & 15, q = li >> 2, p = li & 3, cg = (lane >> 4) & 1;
    const int lane_off = (8 * h + q) * C::ROW + (16 * cg + 4 * p) * 2;

    f32x16 acc[2][2];                                            // [ci half][tap of this wave]
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[cb][tt][v] = 0.f;

    const long long nvox = (long long)a.N * a.D * a.H * a.W;
    using stage_t = typename std::conditional<EPP == 4, f32x4, bf16x8_t>::type;
    stage_t sx[XIT], sd[DIT];
    // r5: when a tile's V consecutive base voxels lie in ONE x-row (W % V == 0: every level but the deepest) the children's addresses are
    // a per-tile base (wave-uniform: one decode per tile) + a per-piece offset that never changes -- the three integer divisions per
    // piece of the general path (24 per thread and tile, more VALU time than the tile's MFMAs) are gone
    // (W < V: a tile is V / W whole x-rows of one z-plane when V % W == 0 and H % (V / W) == 0 -- the deep levels)
    const bool rowtile = (a.W % C::V) == 0 || (C::V % a.W == 0 && a.H % (C::V / a.W) == 0);
    int reld[DIT];
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
        const int pc = it * 256 + tid, part = pc % PPV, vl = (pc / PPV) % C::V, tap = pc / (PPV * C::V);
        const int r = a.W >= C::V ? 0 : vl / a.W, xo = a.W >= C::V ? vl : vl % a.W;
        reld[it] = (((tap >> 2) * (2 * a.H) + 2 * r + ((tap >> 1) & 1)) * (2 * a.W) + 2 * xo + (tap & 1)) * a.lddy + part * EPP;
    }
    auto load_stage = [&](int tile) {
        const long long v0 = (long long)tile * C::V;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int pc = it * 256 + tid, vl = pc / (2 * PPV), part = pc % (2 * PPV);
            stage_t xv = {};
            if (v0 + vl < nvox && ci0 + part * EPP < a.Cin) xv = *reinterpret_cast<const stage_t*>(xin + (v0 + vl) * a.ldx + ci0 + part * EPP);
            sx[it] = xv;
        }
        if (rowtile) {
            long long v = v0;
            const int xw = (int)(v % a.W); v /= a.W;
            const int yh = (int)(v % a.H); v /= a.H;
            const int zd = (int)(v % a.D); const int n = (int)(v / a.D);
            const IN_T* base = din + ((((long long)n * (2 * a.D) + 2 * zd) * (2 * a.H) + 2 * yh) * (2 * a.W) + 2 * xw) * a.lddy + co0;
#pragma unroll
            for (int it = 0; it < DIT; ++it) {
                const int pc = it * 256 + tid, part = pc % PPV;
                stage_t dv = {};
                if (v0 < nvox && co0 + part * EPP < a.Cout) dv = *reinterpret_cast<const stage_t*>(base + reld[it]);
                sd[it] = dv;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int pc = it * 256 + tid, part = pc % PPV, vl = (pc / PPV) % C::V, tap = pc / (PPV * C::V);
            stage_t dv = {};
            if (v0 + vl < nvox && co0 + part * EPP < a.Cout) {       // a ragged last co block (Cout % 32 == 16) stages zeros
                long long v = v0 + vl;
                const int xw = (int)(v % a.W); v /= a.W;
                const int yh = (int)(v % a.H); v /= a.H;
                const int zd = (int)(v % a.D); const int n = (int)(v / a.D);
                const long long ov = (((long long)n * (2 * a.D) + 2 * zd + (tap >> 2)) * (2 * a.H) + 2 * yh + ((tap >> 1) & 1)) * (2 * a.W) + 2 * xw + (tap & 1);
                dv = *reinterpret_cast<const stage_t*>(din + ov * a.lddy + co0 + part * EPP);
            }
            sd[it] = dv;
        }
    };
    auto put = [&](unsigned char* dst, const stage_t& v) {
        if constexpr (EPP == 4) {                                // fp32 tensors: split once per staged value, planes h | m | l
            bf16x4_t qh, qm, ql;
#pragma unroll
            for (int e = 0; e < 4; ++e) { bf16 bh, bm, bl; split3(v[e], bh, bm, bl); qh[e] = bh; qm[e] = bm; ql[e] = bl; }
            *reinterpret_cast<bf16x4_t*>(dst) = qh;
            *reinterpret_cast<bf16x4_t*>(dst + 64) = qm;
            *reinterpret_cast<bf16x4_t*>(dst + 128) = ql;
        } else {
            *reinterpret_cast<bf16x8_t*>(dst) = v;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int pc = it * 256 + tid, vl = pc / (2 * PPV), ch = (pc % (2 * PPV)) * EPP;
            put(xs + ((ch >> 5) * C::V + vl) * C::ROW + (ch & 31) * 2, sx[it]);
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int pc = it * 256 + tid, part = pc % PPV, vl = (pc / PPV) % C::V, tap = pc / (PPV * C::V);
            put(ds + (tap * C::V + vl) * C::ROW + part * EPP * 2, sd[it]);
        }
    };

    int tile = strip;
    if (tile < a.ntiles) load_stage(tile);
    for (; tile < a.ntiles; tile += a.nstrips) {
        __syncthreads();
        write_stage();
        __syncthreads();
        if (tile + a.nstrips < a.ntiles) load_stage(tile + a.nstrips);
#pragma unroll
        for (int ks = 0; ks < C::KSTEPS; ++ks) {
            bf16x8_t ac[2][NP], bc[2][NP];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) ac[cb][pl] = cw_frag(xs + (cb * C::V + ks * 16) * C::ROW + lane_off + pl * 64, C::ROW);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) bc[tt][pl] = cw_frag(ds + ((2 * wave + tt) * C::V + ks * 16) * C::ROW + lane_off + pl * 64, C::ROW);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    f32x16 c = acc[cb][tt];
                    if constexpr (NP == 3) {                    // planes 0 / 1 / 2 = h / m / l; the small cross terms go in first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][2], bc[tt][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][0], bc[tt][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][1], bc[tt][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][1], bc[tt][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][0], bc[tt][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][0], bc[tt][0], c, 0, 0, 0);
                    } else {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][0], bc[tt][0], c, 0, 0, 0);
                    }
                    acc[cb][tt] = c;
                }
        }
    }

    // slab store: part[strip][tap][ci][co]; rows of the 32x32 tile = ci, lanes (columns) = co
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float* dst = a.part + (((long long)strip * 8 + 2 * wave + tt) * a.Cin + ci0 + cb * 32) * a.Cout + co0 + i;
            if (co0 + i < a.Cout && ci0 + cb * 32 < a.Cin) {
#pragma unroll
                for (int v = 0; v < 16; ++v) dst[(long long)((v & 3) + 8 * (v >> 2) + 4 * h) * a.Cout] = acc[cb][tt][v];
            }
        }
}

// ---------------------------------------------------------------- pointwise (k1) weight gradient: dW[ci][co] = sum_v x[v][ci] * dy[v][co]
// The same operand path with ONE tap: the four waves split a tile's k-steps instead of its taps and add their accumulators
// through LDS at the end (fixed order).  part[strip][ci][co].
template <int NP>
struct PwlCfg {
    static constexpr int V = NP == 3 ? 128 : 256;             // voxels per tile
    static constexpr int ROW = 64 * NP;
    static constexpr int X_BYTES = 2 * V * ROW, D_BYTES = V * ROW;
    static constexpr int LDS_BYTES = X_BYTES + D_BYTES;       // 72 KB (bf16x6) / 48 KB (bf16); the final reduce needs 32 KB of it
    static constexpr int KSTEPS = V / 16;
};

template <int NP, typename IN_T>
__global__ __launch_bounds__(256, 2) void pw_wgrad_lowp_kernel(CwArgs a) {
    using C = PwlCfg<NP>;
    constexpr int EPP = std::is_same<IN_T, float>::value ? 4 : 8;
    constexpr int PPV = 32 / EPP;
    constexpr int XIT = C::V * 2 * PPV / 256, DIT = C::V * PPV / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xs = lds;
    unsigned char* ds = lds + C::X_BYTES;
    const IN_T* __restrict__ xin = reinterpret_cast<const IN_T*>(a.x);
    const IN_T* __restrict__ din = reinterpret_cast<const IN_T*>(a.dy);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, i = lane & 31;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int pair = t % a.npairs, strip = t / a.npairs;
    const int ci0 = (pair / a.ncob) * 64, co0 = (pair % a.ncob) * 32;
    const int li = lane & 15, q = li >> 2, p = li & 3, cg = (lane >> 4) & 1;
    const int lane_off = (8 * h + q) * C::ROW + (16 * cg + 4 * p) * 2;
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[cb][v] = 0.f;
    const long long nvox = (long long)a.N * a.D * a.H * a.W;
    using stage_t = typename std::conditional<EPP == 4, f32x4, bf16x8_t>::type;
    stage_t sx[XIT], sd[DIT];
    auto load_stage = [&](int tile) {
        const long long v0 = (long long)tile * C::V;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int pc = it * 256 + tid, vl = pc / (2 * PPV), part = pc % (2 * PPV);
            stage_t xv = {};
            if (v0 + vl < nvox && ci0 + part * EPP < a.Cin) xv = *reinterpret_cast<const stage_t*>(xin + (v0 + vl) * a.ldx + ci0 + part * EPP);
            sx[it] = xv;
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int pc = it * 256 + tid, vl = pc / PPV, part = pc % PPV;
            stage_t dv = {};
            if (v0 + vl < nvox && co0 + part * EPP < a.Cout) dv = *reinterpret_cast<const stage_t*>(din + (v0 + vl) * a.lddy + co0 + part * EPP);
            sd[it] = dv;
        }
    };
    auto put = [&](unsigned char* dst, const stage_t& v) {
        if constexpr (EPP == 4) {
            bf16x4_t qh, qm, ql;
#pragma unroll
            for (int e = 0; e < 4; ++e) { bf16 bh, bm, bl; split3(v[e], bh, bm, bl); qh[e] = bh; qm[e] = bm; ql[e] = bl; }
            *reinterpret_cast<bf16x4_t*>(dst) = qh;
            *reinterpret_cast<bf16x4_t*>(dst + 64) = qm;
            *reinterpret_cast<bf16x4_t*>(dst + 128) = ql;
        } else {
            *reinterpret_cast<bf16x8_t*>(dst) = v;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int pc = it * 256 + tid, vl = pc / (2 * PPV), ch = (pc % (2 * PPV)) * EPP;
            put(xs + ((ch >> 5) * C::V + vl) * C::ROW + (ch & 31) * 2, sx[it]);
        }
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int pc = it * 256 + tid;
            put(ds + (pc / PPV) * C::ROW + (pc % PPV) * EPP * 2, sd[it]);
        }
    };
    int tile = strip;
    if (tile < a.ntiles) load_stage(tile);
    for (; tile < a.ntiles; tile += a.nstrips) {
        __syncthreads();
        write_stage();
        __syncthreads();
        if (tile + a.nstrips < a.ntiles) load_stage(tile + a.nstrips);
#pragma unroll
        for (int s = 0; s < C::KSTEPS / 4; ++s) {
            const int ks = wave + 4 * s;
            bf16x8_t ac[2][NP], bc[NP];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) ac[cb][pl] = cw_frag(xs + (cb * C::V + ks * 16) * C::ROW + lane_off + pl * 64, C::ROW);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) bc[pl] = cw_frag(ds + (ks * 16) * C::ROW + lane_off + pl * 64, C::ROW);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x16 c = acc[cb];
                if constexpr (NP == 3) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][2], bc[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][0], bc[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][1], bc[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][1], bc[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][0], bc[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][0], bc[0], c, 0, 0, 0);
                } else {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[cb][0], bc[0], c, 0, 0, 0);
                }
                acc[cb] = c;
            }
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);                 // [4 waves][2 cb][32 rows][32 cols]
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int v = 0; v < 16; ++v) red[((wave * 2 + cb) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h) * 32 + i] = acc[cb][v];
    __syncthreads();
    for (int e = tid; e < 2048; e += 256) {
        const int row = e >> 5, col = e & 31;                    // row = cb * 32 + ci
        if (ci0 + row < a.Cin && co0 + col < a.Cout) {
            const float sum = ((red[e] + red[2048 + e]) + red[4096 + e]) + red[6144 + e];
            a.part[((long long)strip * a.Cin + ci0 + row) * a.Cout + co0 + col] = sum;
        }
    }
}

struct CwPlan { int ntiles, nstrips, npairs; };

static bool pwl_plan(int NP, long long nvox, int Cin, int Cout, CwPlan* p) {
    if (Cin % 32 || Cout % 16 || nvox < 1) return false;
    const int V = NP == 3 ? 128 : 256;
    p->ntiles = (int)((nvox + V - 1) / V);
    p->npairs = ((Cin + 63) / 64) * ((Cout + 31) / 32);
    int want = 512 / p->npairs;
    long long cap = (long long)(64u << 20) / ((long long)Cin * Cout * 4);
    if (cap < 1) cap = 1;
    if (want > cap) want = (int)cap;
    if (want > p->ntiles) want = p->ntiles;
    if (want < 1) want = 1;
    p->nstrips = want;
    return true;
}

size_t pw_wgrad_lowp_ws_bytes(long long nvox, int Cin, int Cout) {
    size_t best = 0;
    for (int np : {1, 3}) {
        CwPlan p;
        if (!pwl_plan(np, nvox, Cin, Cout, &p)) continue;
        const size_t b = align_up((size_t)p.nstrips * Cin * Cout * sizeof(float), 256) + 1024;
        if (b > best) best = b;
    }
    return best;
}

bool pw_wgrad_lowp_supported(long long nvox, int Cin, int Cout, int ldx, int lddy, int elem_bytes) {
    CwPlan p;
    const int al = 16 / elem_bytes;
    return (ldx % al) == 0 && (lddy % al) == 0 && pwl_plan(elem_bytes == 4 ? 3 : 1, nvox, Cin, Cout, &p);
}

template <typename IN_T>
int pw_wgrad_lowp(const IN_T* dy, int lddy, const IN_T* x, int ldx, int N, int D, int H, int W, int Cin, int Cout,
                  float** part_out, int* nstrips_out, void* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int NP = std::is_same<IN_T, float>::value ? 3 : 1;
    CwPlan p;
    const long long nvox = (long long)N * D * H * W;
    SEG_CHECK_ARG(pwl_plan(NP, nvox, Cin, Cout, &p), "pw_wgrad_lowp: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "pw_wgrad_lowp: pointers must be 16-byte aligned");
    Carver cv(ws);
    float* part = cv.take<float>((size_t)p.nstrips * Cin * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    CwArgs a{x, dy, part, ldx, lddy, N, D, H, W, Cin, Cout, p.ntiles, p.nstrips, p.npairs, (Cout + 31) / 32};
    SEG_SET_LDS((pw_wgrad_lowp_kernel<NP, IN_T>), PwlCfg<NP>::LDS_BYTES);
    ProfScope ps(PF_WGRAD, 2.0 * nvox * Cin * Cout, (double)sizeof(IN_T) * nvox * (Cin + (double)Cout) + 4.0 * Cin * Cout, st);
    hipLaunchKernelGGL((pw_wgrad_lowp_kernel<NP, IN_T>), dim3(p.nstrips * p.npairs), dim3(256), PwlCfg<NP>::LDS_BYTES, st, a);
    SEG_CHECK_LAUNCH();
    *part_out = part; *nstrips_out = p.nstrips;
    return MI355SEG_OK;
}
template int pw_wgrad_lowp<float>(const float*, int, const float*, int, int, int, int, int, int, int, float**, int*, void*, size_t, hipStream_t);
template int pw_wgrad_lowp<bf16>(const bf16*, int, const bf16*, int, int, int, int, int, int, int, float**, int*, void*, size_t, hipStream_t);


static bool cw_plan(int NP, long long nvox, int Cin, int Cout, CwPlan* p) {
    if (Cin % 32 || Cout % 16 || nvox < 1) return false;          // 16-byte staged pieces; half-empty last ci / co blocks are allowed
    const int V = NP == 3 ? 32 : 64;
    p->ntiles = (int)((nvox + V - 1) / V);
    p->npairs = ((Cin + 63) / 64) * ((Cout + 31) / 32);
    int want = 512 / p->npairs;                        // two workgroups per CU
    long long cap = (long long)(64u << 20) / ((long long)8 * Cin * Cout * 4);
    if (cap < 1) cap = 1;
    if (want > cap) want = (int)cap;
    if (want > p->ntiles) want = p->ntiles;
    if (want < 1) want = 1;
    p->nstrips = want;
    return true;
}

size_t convt_wgrad_lowp_ws_bytes(long long nvox, int Cin, int Cout) {
    size_t best = 0;
    for (int np : {1, 3}) {
        CwPlan p;
        if (!cw_plan(np, nvox, Cin, Cout, &p)) continue;
        const size_t b = align_up((size_t)p.nstrips * 8 * Cin * Cout * sizeof(float), 256) + 1024;
        if (b > best) best = b;
    }
    return best;
}

bool convt_wgrad_lowp_supported(long long nvox, int Cin, int Cout, int ldx, int lddy, int elem_bytes) {
    CwPlan p;
    const int al = 16 / elem_bytes;                                     // 16-byte staged pieces
    return (ldx % al) == 0 && (lddy % al) == 0 && cw_plan(elem_bytes == 4 ? 3 : 1, nvox, Cin, Cout, &p);
}

template <typename IN_T>
int convt_wgrad_lowp(const IN_T* dy, int lddy, const IN_T* x, int ldx, int N, int D, int H, int W, int Cin, int Cout,
                     float** part_out, int* nstrips_out, void* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int NP = std::is_same<IN_T, float>::value ? 3 : 1;
    CwPlan p;
    const long long nvox = (long long)N * D * H * W;
    SEG_CHECK_ARG(cw_plan(NP, nvox, Cin, Cout, &p), "convt_wgrad_lowp: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "convt_wgrad_lowp: pointers must be 16-byte aligned");
    Carver cv(ws);
    float* part = cv.take<float>((size_t)p.nstrips * 8 * Cin * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    CwArgs a{x, dy, part, ldx, lddy, N, D, H, W, Cin, Cout, p.ntiles, p.nstrips, p.npairs, (Cout + 31) / 32};
    SEG_SET_LDS((convt_wgrad_lowp_kernel<NP, IN_T>), CwCfg<NP>::LDS_BYTES);
    ProfScope ps(PF_CONVT, 2.0 * nvox * 8 * Cin * Cout, (double)sizeof(IN_T) * nvox * (Cin + 8.0 * Cout) + 32.0 * Cin * Cout, st);
    hipLaunchKernelGGL((convt_wgrad_lowp_kernel<NP, IN_T>), dim3(p.nstrips * p.npairs), dim3(256), CwCfg<NP>::LDS_BYTES, st, a);
    SEG_CHECK_LAUNCH();
    *part_out = part; *nstrips_out = p.nstrips;
    return MI355SEG_OK;
}
template int convt_wgrad_lowp<float>(const float*, int, const float*, int, int, int, int, int, int, int, float**, int*, void*, size_t, hipStream_t);
template int convt_wgrad_lowp<bf16>(const bf16*, int, const bf16*, int, int, int, int, int, int, int, float**, int*, void*, size_t, hipStream_t);

}  // namespace seg
