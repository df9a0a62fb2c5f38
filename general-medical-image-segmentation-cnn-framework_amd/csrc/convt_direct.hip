// convt_direct.hip -- ConvTranspose3d(k2, s2) forward / input gradient as ONE plain GEMM on the bf16 matrix cores, its operand
// read straight from the tensor and its result written straight to the 2x2x2 children (no halo, no per-tap restaging).
//
//   forward (SCATTER):  Y[child(v, t), co] = b[co] + sum_ci X[v, ci] W[ci, co, t]        M = coarse voxels, K = Cin,      N = 8 Cout
//   dgrad   (GATHER):   dX[v, ci]          =         sum_{t, co} dY[child(v, t), co] W[ci, co, t]             K = 8 Cout,  N = Cin
//
// Reference: nn.ConvTranspose3d unet3d.py:29-43 (upconv4..1), vnet3d.py:86 (UpTransition.up_conv), unetr.py:11; the same two
// GEMMs are the input gradient / forward of a k2 s2 Conv3d (vnet3d.py:66 DownTransition.down_conv).
//
// fp32 tensors: every operand is split x = h + m + l into three bf16 planes while it is staged (six v_mfma_f32_16x16x32_bf16 per
// product, fp32 accumulate -- the bf16x6 conv math of conv_x3s.hip); bf16 tensors: one plane, one MFMA.  The layers this serves are
// HBM-bound (64 -> 32 @ 64^3: 17 GFLOP over 671 MB), so the kernel is built around full-width 16-byte loads / stores and two
// resident workgroups per CU, not around the MFMA rate.
//
// Tile: BM voxels x BN columns per workgroup (4 waves as 2 x 2), K in chunks of 64.  D^T orientation -- A = weights (rows = GEMM
// columns n), B = voxels -- so a lane ends with 4 consecutive n of one voxel: one 16-byte store.  LDS holds both operands as
// 16-byte slots [plane][k-group of 8][row], row stride = rows + 8 slots (== 8 mod 16: the 8 k-groups x 8 rows a wave writes per
// instruction spread over all 64 banks; the 16 rows a fragment read touches are consecutive).
#include "common.h"
#include "internal.h"
#include "pack.h"

namespace seg {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

struct CtArgs {
    const void* x;          // voxel operand (coarse x for SCATTER, fine dy for GATHER)
    const void* wq;         // packed weight planes, [ntile][chunk][plane][kgroup 8][row BN] slots of 8 bf16
    const float* bias;      // SCATTER only, may be null
    void* y;                // result (fine y for SCATTER, coarse dx for GATHER)
    int ldx, ldy;           // row strides (elements) of the voxel operand / the result
    int D, H, W;            // coarse extents
    int Cf;                 // channels of the fine tensor (Cout of the ConvT)
    long long nvox;         // N * D * H * W coarse voxels
    int nchunk;             // K / 64
    int ntn;                // N-tiles
    unsigned* amax_out;     // SCATTER, optional: max |y| of what this launch writes, max-combined (the f16x3 scale of the convolution that reads the concat buffer)
    int ksplit, cps;        // GEMM form, GATHER on fp32 tensors (r5): K-splits and chunks per split; ksplit > 1: y is an fp32 slab array
    long long slab_stride;  // floats between the slabs of consecutive splits
};

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h2, unsigned& m2, unsigned& l2) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2_t));
    const float r0 = x0 - __builtin_bit_cast(float, h2 << 16), r1 = x1 - __builtin_bit_cast(float, h2 & 0xFFFF0000u);
    m2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2_t));
    const float q0 = r0 - __builtin_bit_cast(float, m2 << 16), q1 = r1 - __builtin_bit_cast(float, m2 & 0xFFFF0000u);
    l2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{q0, q1}, bf16x2_t));
}

// row index (fine tensor) of child (0, 0, 0) of coarse voxel v
__device__ __forceinline__ long long fine_base(long long v, int D, int H, int W) {
    const int x = (int)(v % W); long long r = v / W;
    const int y = (int)(r % H); r /= H;
    const int z = (int)(r % D); const long long n = r / D;
    return ((n * 2 * D + 2 * z) * 2 * H + 2 * y) * 2 * W + 2 * x;
}

template <typename TT, bool GATHER, int BM, int BN>
__global__ __launch_bounds__(256, 2) void convt_gemm_kernel(CtArgs a) {
    constexpr bool F32 = sizeof(TT) == 4;
    constexpr int NP = F32 ? 3 : 1;
    // slots per k-group.  The voxel tile is WRITTEN by lanes (k-group tid & 7, row tid >> 3): the 16 lanes of a write phase hold 8 k-groups x 2
    // rows, so the k-group pitch must be == 2 mod 16 slots for them to fall on 16 distinct slot positions (r6; with BM + 8 == 8 mod 16 the even
    // k-groups shared one position and the odd ones another: four-way conflicts, LDS conflict share 0.625 in r05_pmc_sq_unetr.csv); the weight
    // tile is written row-contiguously, any pitch does
#ifndef CT_XPAD
#define CT_XPAD 2
#endif
    constexpr int XS = BM + CT_XPAD, WS = BN + 8;
    constexpr int XPL = 8 * XS, WPL = 8 * WS;                      // slots per plane
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int XIT = BM * 8 / 256, WIT = NP * 8 * BN / 256;
    extern __shared__ unsigned char lds_raw[];
    unsigned char* const xl = lds_raw;
    unsigned char* const wl = lds_raw + NP * XPL * 16;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int wm = wave & 1, wn = wave >> 1;
    // (r5) K-splits of a tile are adjacent workgroups: the deep levels' input gradients cut 64-256 tiles of a long K (8 Cout = 1024-2048)
    const int ks = a.ksplit > 1 ? (int)(blockIdx.x % a.ksplit) : 0;
    const unsigned bid = a.ksplit > 1 ? blockIdx.x / a.ksplit : blockIdx.x;
    const int ntile = bid % a.ntn;
    const long long vox0 = (long long)(bid / a.ntn) * BM;
    const int c0 = a.ksplit > 1 ? ks * a.cps : 0, c1 = a.ksplit > 1 ? c0 + a.cps : a.nchunk;
    const TT* const xg = static_cast<const TT*>(a.x);
    const int fH = 2 * a.H, fW = 2 * a.W;

    // ---- staging assignment: slot (voxel row v, k-group kg) per thread and iteration
    const int kg = tid & 7;
    long long rowb[XIT]; bool ok[XIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const long long v = vox0 + (tid >> 3) + 32 * it;
        ok[it] = v < a.nvox;
        rowb[it] = GATHER ? fine_base(ok[it] ? v : 0, a.D, a.H, a.W) : (ok[it] ? v : 0);
    }
    using stage_t = typename std::conditional<F32, f32x4, u32x4>::type;
    stage_t xs[XIT][F32 ? 2 : 1];
    u32x4 wsr[WIT];
    auto load_global = [&](int chunk) {
        // fp32: the 8 k of a slot are channels {4g .. 4g+3} and {16 + 4g .. 16 + 4g+3} of its 32-wide k-step (kperm below)
        const int k0 = F32 ? chunk * 64 + 32 * (kg >> 2) + 4 * (kg & 3) : chunk * 64 + kg * 8;
        auto src = [&](int it, int k) -> const TT* {
            if (GATHER) {
                const int t = k / a.Cf, co = k - t * a.Cf;
                return xg + (rowb[it] + ((long long)(t >> 2) * fH + ((t >> 1) & 1)) * fW + (t & 1)) * a.ldx + co;
            }
            return xg + rowb[it] * a.ldx + k;
        };
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            // rows past the end read voxel 0 (their columns are never stored): unconditional loads keep the vmcnt bookkeeping exact
            xs[it][0] = *reinterpret_cast<const stage_t*>(src(it, k0));
            if (F32) xs[it][F32 ? 1 : 0] = *reinterpret_cast<const stage_t*>(src(it, k0 + 16));
        }
        const u32x4* wsrc = static_cast<const u32x4*>(a.wq) + ((long long)ntile * a.nchunk + chunk) * (NP * 8 * BN);
#pragma unroll
        for (int j = 0; j < WIT; ++j) wsr[j] = wsrc[tid + 256 * j];
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            unsigned char* dst = xl + (kg * XS + (tid >> 3) + 32 * it) * 16;
            if constexpr (F32) {
                unsigned h[4], m[4], l[4];
                split_pair(xs[it][0][0], xs[it][0][1], h[0], m[0], l[0]);
                split_pair(xs[it][0][2], xs[it][0][3], h[1], m[1], l[1]);
                split_pair(xs[it][1][0], xs[it][1][1], h[2], m[2], l[2]);
                split_pair(xs[it][1][2], xs[it][1][3], h[3], m[3], l[3]);
                const u32x4 qh = {h[0], h[1], h[2], h[3]}, qm = {m[0], m[1], m[2], m[3]}, ql = {l[0], l[1], l[2], l[3]};
                *reinterpret_cast<u32x4*>(dst) = qh;
                *reinterpret_cast<u32x4*>(dst + XPL * 16) = qm;
                *reinterpret_cast<u32x4*>(dst + 2 * XPL * 16) = ql;
            } else {
                *reinterpret_cast<u32x4*>(dst) = xs[it][0];
            }
        }
#pragma unroll
        for (int j = 0; j < WIT; ++j) {
            const int i = tid + 256 * j;                           // slot (plane * 8 + kgroup, row)
            *reinterpret_cast<u32x4*>(wl + ((i / BN) * WS + (i % BN)) * 16) = wsr[j];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned char* const xrd = xl + (g * XS + wm * WM + r) * 16;
    const unsigned char* const wrd = wl + (g * WS + wn * WN + r) * 16;

    load_global(c0);
    for (int chunk = c0; chunk < c1; ++chunk) {
        __syncthreads();
        write_lds();
        __syncthreads();
        if (chunk + 1 < c1) load_global(chunk + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t wf[TN][NP], xf[TM][NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[j][pl] = *reinterpret_cast<const bf16x8_t*>(wrd + (pl * WPL + 4 * s * WS + j * 16) * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i) xf[i][pl] = *reinterpret_cast<const bf16x8_t*>(xrd + (pl * XPL + 4 * s * XS + i * 16) * 16);
            }
            if constexpr (F32) {
                // planes 0 / 1 / 2 = h / m / l; the small cross terms go in first
                constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][PW[pr]], xf[i][PX[pr]], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], xf[i][0], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane holds rows n = 4g .. 4g+3 of column (voxel) r of every 16 x 16 tile
    TT* const yg = static_cast<TT*>(a.y) + (GATHER && F32 ? (long long)ks * a.slab_stride : 0);      // (ksplit > 1: this split's slab, pitch ldy = the GEMM width)
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long long v = vox0 + wm * WM + i * 16 + r;
        if (v >= a.nvox) continue;
        const long long fb = GATHER ? 0 : fine_base(v, a.D, a.H, a.W);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = ntile * BN + wn * WN + j * 16 + 4 * g;
            f32x4 o = acc[i][j];
            TT* p;
            if (GATHER) {
                p = yg + v * a.ldy + n;
            } else {
                const int t = n / a.Cf, co = n - t * a.Cf;
                if (a.bias) o += *reinterpret_cast<const f32x4*>(a.bias + co);
                p = yg + (fb + ((long long)(t >> 2) * fH + ((t >> 1) & 1)) * fW + (t & 1)) * a.ldy + co;
            }
            if constexpr (F32 && !GATHER) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
            if constexpr (F32) {
                *reinterpret_cast<f32x4*>(p) = o;
            } else {
                bf16x4_t q;
                q[0] = (bf16)o[0]; q[1] = (bf16)o[1]; q[2] = (bf16)o[2]; q[3] = (bf16)o[3];
                *reinterpret_cast<bf16x4_t*>(p) = q;
            }
        }
    }
    if constexpr (F32 && !GATHER) {
        if (a.amax_out) block_amax_commit(amax, a.amax_out);       // (uniform: every thread of the workgroup gets here)
    }
}

// ---- streaming form for K <= 256 (the full-resolution layers, where the op is HBM-bound): the weight planes of one N-tile stay in
// LDS for the whole launch, every wave walks its own 16 * TM-voxel tiles with a fixed stride, reads its voxel fragments STRAIGHT from
// the tensor (lane (r, g) owns channels 8g .. 8g+7 of voxel r of each 32-wide k-step: 32 contiguous bytes), splits them in registers
// and prefetches the next tile while the MFMAs of this one run.  No barrier after the prologue, no LDS traffic for the voxel operand.
template <typename TT, bool GATHER, int KK, int BN, int TM, int NW, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void convt_stream_kernel(CtArgs a) {
    constexpr bool F32 = sizeof(TT) == 4;
    constexpr int NP = F32 ? 3 : 1;
    constexpr int KG = KK / 8, KS = KK / 32, TN = BN / 16, NCH = KK / 64;
    extern __shared__ unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int ntile = blockIdx.x % a.ntn;
    const int P = gridDim.x / a.ntn;                               // workgroups per N-tile
    {   // weight planes of this N-tile: [ntile][chunk][plane][kgroup 8][row] in memory -> [plane][kgroup KG][row] in LDS
        const u32x4* wsrc = static_cast<const u32x4*>(a.wq) + (long long)ntile * (NCH * NP * 8 * BN);
        for (int i = tid; i < NCH * NP * 8 * BN; i += NW * 64) {
            const int row = i % BN, q = i / BN, kgp = q % 8, pl = (q / 8) % NP, ch = q / (8 * NP);
            *reinterpret_cast<u32x4*>(lds_raw + ((pl * KG + ch * 8 + kgp) * BN + row) * 16) = wsrc[i];
        }
        if (!GATHER)
            for (int i = tid; i < BN; i += NW * 64)
                reinterpret_cast<float*>(lds_raw + NP * KG * BN * 16)[i] = a.bias ? a.bias[(ntile * BN + i) % a.Cf] : 0.f;
    }
    const unsigned char* const bl = lds_raw + NP * KG * BN * 16;
    const TT* const xg = static_cast<const TT*>(a.x);
    TT* const yg = static_cast<TT*>(a.y);
    const int fH = 2 * a.H, fW = 2 * a.W;
    // per-lane element offsets that do not depend on the voxel: operand k-steps (GATHER: tap row + channel), result n-tiles
    constexpr int SW = F32 ? 2 : 1;
    int koff[KS][SW], noff[TN];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int h = 0; h < SW; ++h) {
            // fp32: lane g holds channels {4g .. 4g+3} and {16 + 4g ..} of the k-step, so each load instruction reads 64 contiguous bytes per voxel
            const int k = F32 ? 32 * s + 16 * h + 4 * g : 32 * s + 8 * g;
            if (GATHER) { const int t = k / a.Cf, co = k - t * a.Cf; koff[s][h] = (((t >> 2) * fH + ((t >> 1) & 1)) * fW + (t & 1)) * a.ldx + co; }
            else koff[s][h] = k;
        }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ntile * BN + j * 16 + 4 * g;
        if (GATHER) noff[j] = n;
        else { const int t = n / a.Cf, co = n - t * a.Cf; noff[j] = (((t >> 2) * fH + ((t >> 1) & 1)) * fW + (t & 1)) * a.ldy + co; }
    }
    const unsigned nvox = (unsigned)a.nvox;
    const unsigned ntiles = nvox / (16 * TM);
    const unsigned stride = (unsigned)P * NW;
    auto fine_row = [&](unsigned v) -> long long {
        const unsigned x = v % (unsigned)a.W; unsigned q = v / (unsigned)a.W;
        const unsigned y = q % (unsigned)a.H; q /= (unsigned)a.H;
        const unsigned z = q % (unsigned)a.D; const unsigned n = q / (unsigned)a.D;
        return (((long long)n * 2 * a.D + 2 * z) * fH + 2 * y) * fW + 2 * x;
    };
    using stage_t = typename std::conditional<F32, f32x4, u32x4>::type;
    stage_t xa[TM][KS][SW], xb[TM][KS][SW];
    // every load and store is issued unconditionally (nvox is a multiple of the tile, the prefetch past the last tile re-reads it): a
    // memory instruction under a branch makes the compiler drain vmcnt to 0 at the join, i.e. wait for the prefetch it just issued
    // and for the previous tile's stores
    auto load_x = [&](unsigned t, stage_t (&xs)[TM][KS][SW]) {
        t = t < ntiles ? t : ntiles - 1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned v = t * (16 * TM) + i * 16 + r;
            const TT* base = xg + (GATHER ? fine_row(v) : (long long)v) * a.ldx;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int h = 0; h < SW; ++h) xs[i][s][h] = *reinterpret_cast<const stage_t*>(base + koff[s][h]);
        }
    };
    const unsigned char* const wrd = lds_raw + (g * BN + r) * 16;
    float amax = 0.f;
    auto compute_store = [&](unsigned t, stage_t (&xs)[TM][KS][SW]) {
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8_t xf[TM][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (F32) {
                    unsigned h[4], m[4], l[4];
                    split_pair(xs[i][s][0][0], xs[i][s][0][1], h[0], m[0], l[0]);
                    split_pair(xs[i][s][0][2], xs[i][s][0][3], h[1], m[1], l[1]);
                    split_pair(xs[i][s][1][0], xs[i][s][1][1], h[2], m[2], l[2]);
                    split_pair(xs[i][s][1][2], xs[i][s][1][3], h[3], m[3], l[3]);
                    xf[i][0] = __builtin_bit_cast(bf16x8_t, u32x4{h[0], h[1], h[2], h[3]});
                    xf[i][1] = __builtin_bit_cast(bf16x8_t, u32x4{m[0], m[1], m[2], m[3]});
                    xf[i][2] = __builtin_bit_cast(bf16x8_t, u32x4{l[0], l[1], l[2], l[3]});
                } else {
                    xf[i][0] = __builtin_bit_cast(bf16x8_t, xs[i][s][0]);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bf16x8_t wf[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) wf[pl] = *reinterpret_cast<const bf16x8_t*>(wrd + ((pl * KG + 4 * s) * BN + j * 16) * 16);
                if constexpr (F32) {
                    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[PW[pr]], xf[i][PX[pr]], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], xf[i][0], acc[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                     // keep the weight-fragment reads of later k-steps where they are
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned v = t * (16 * TM) + i * 16 + r;                // the host sends only whole tiles here: no store under a branch
            TT* base = yg + (GATHER ? (long long)v : fine_row(v)) * a.ldy;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 o = acc[i][j];
                if (!GATHER) o += *reinterpret_cast<const f32x4*>(bl + (j * 16 + 4 * g) * 4);      // LDS: its own counter, no vmcnt drain
                if constexpr (F32 && !GATHER) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
                if constexpr (F32) {
                    *reinterpret_cast<f32x4*>(base + noff[j]) = o;
                } else {
                    bf16x4_t q;
                    q[0] = (bf16)o[0]; q[1] = (bf16)o[1]; q[2] = (bf16)o[2]; q[3] = (bf16)o[3];
                    *reinterpret_cast<bf16x4_t*>(base + noff[j]) = q;
                }
            }
        }
    };
    // Every wave runs the same trip count (host: P * NW <= ntiles); a trip past the end redoes the wave's own last tile (same values
    // to the same addresses).  vmcnt retires in order and the compiler takes the minimum over all paths into a join, so the first trip
    // is peeled: on both ways into the loop exactly this tile's 8 stores are younger than the loads being waited for, and the wait
    // becomes vmcnt(stores) instead of a drain of the store queue.
    const unsigned t0 = (unsigned)(blockIdx.x / a.ntn) * NW + wave;
    const unsigned trips = (ntiles + stride - 1) / stride;
    auto tile_of = [&](unsigned it) { const unsigned t = t0 + it * stride; return t < ntiles ? t : t - stride; };
    auto copy_x = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int h = 0; h < SW; ++h) xa[i][s][h] = xb[i][s][h];
    };
    load_x(t0, xb);
    __syncthreads();                                               // weight planes (and bias row) in place
    copy_x();
    load_x(tile_of(1), xb);
    compute_store(t0, xa);
    for (unsigned it = 1; it < trips; ++it) {
        copy_x();
        load_x(tile_of(it + 1), xb);
        compute_store(tile_of(it), xa);
    }
    if constexpr (F32 && !GATHER) {
        if (a.amax_out) block_amax_commit(amax, a.amax_out);
    }
}

// (the packed planes of w (Cin, Cout, 8) -- slot (ntile, chunk, plane, kgroup, row) holds k = chunk * 64 + kgroup * 8 .. + 7 of GEMM column
// n = ntile * BN + row; SCATTER: n = (t, co), k = ci; GATHER: n = ci, k = (t, co) -- are formed by convt_pack_planes_body, prepack.hip)

template <typename TT, bool GATHER, int KK, int BN, int TM, int NW, int OCC>
void launch_stream(const CtArgs& a, hipStream_t st) {
    constexpr int NP = sizeof(TT) == 4 ? 3 : 1;
    constexpr size_t lds = (size_t)NP * (KK / 8) * BN * 16 + BN * 4;
    SEG_SET_LDS((convt_stream_kernel<TT, GATHER, KK, BN, TM, NW, OCC>), lds);
    const long long ntiles = a.nvox / (16 * TM);
    const int per_cu = (int)(163840 / lds) < (OCC * 4 / NW) ? (int)(163840 / lds) : OCC * 4 / NW;
    long long P = (256ll * per_cu + a.ntn - 1) / a.ntn;            // one resident wave of workgroups, split over the N-tiles
    if (P > ntiles / NW) P = ntiles / NW;                            // every wave owns at least one tile
    if (P < 1) P = 1;
    hipLaunchKernelGGL((convt_stream_kernel<TT, GATHER, KK, BN, TM, NW, OCC>), dim3((unsigned)(P * a.ntn)), dim3(NW * 64), lds, st, a);
}

template <typename TT, bool GATHER, int BM, int BN>
void launch_ct(const CtArgs& a, int mtiles, hipStream_t st) {
    constexpr int NP = sizeof(TT) == 4 ? 3 : 1;
    constexpr size_t lds = (size_t)NP * 8 * ((BM + 8) + (BN + 8)) * 16;      // (the voxel pitch is BM + 2 now: within this)
    SEG_SET_LDS((convt_gemm_kernel<TT, GATHER, BM, BN>), lds);
    hipLaunchKernelGGL((convt_gemm_kernel<TT, GATHER, BM, BN>), dim3((unsigned)(mtiles * a.ntn * (a.ksplit > 1 ? a.ksplit : 1))), dim3(256), lds, st, a);
}

// dx[v][c] = sum over the K-splits' slabs (fixed order)
__global__ __launch_bounds__(256) void convt_splitk_reduce_kernel(const float* __restrict__ slabs, int ksplit, long long stride, float* __restrict__ dx, int lddx,
                                                                  long long nvox, int C) {
    const int cw = C / 4;
    const long long total = nvox * cw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long v = i / cw;
        const int c = (int)(i - v * cw) * 4;
        f32x4 s = *reinterpret_cast<const f32x4*>(slabs + v * C + c);
        for (int k = 1; k < ksplit; ++k) s += *reinterpret_cast<const f32x4*>(slabs + k * stride + v * C + c);
        *reinterpret_cast<f32x4*>(dx + v * lddx + c) = s;
    }
}

}  // namespace

// K and the GEMM width of the two directions; the tile width and the kernel form (streaming for K <= 256)
struct CtPlan { int K, Nc, BN; bool stream; };
static bool ct_plan(int elem_bytes, bool gather, long long nvox, int Cin, int Cout, CtPlan* p) {
    p->K = gather ? 8 * Cout : Cin;
    p->Nc = gather ? Cin : 8 * Cout;
    if (p->K % 64 || p->Nc % 64 || Cout % 8) return false;
    p->stream = (p->K == 64 || p->K == 128 || p->K == 256) && nvox % 32 == 0 && nvox >= 512;   // whole wave tiles only, a tile for every wave
    const bool wide = p->Nc % 128 == 0;
    if (!p->stream) p->BN = wide ? 128 : 64;
    else if (elem_bytes == 2) p->BN = wide ? 128 : 64;
    else p->BN = (p->K == 64 && wide) ? 128 : 64;                  // fp32: three planes, 48 KB (96 KB at K = 256) of LDS per N-tile
    return true;
}

bool convt_direct_supported(int elem_bytes, bool gather, int N, int D, int H, int W, int Cin, int Cout, int ld_coarse, int ld_fine) {
    CtPlan p;
    const int al = elem_bytes == 2 ? 8 : 4;
    if (!ct_plan(elem_bytes, gather, (long long)N * D * H * W, Cin, Cout, &p)) return false;
    if (ld_coarse % al || ld_fine % al) return false;
    const long long nvox = (long long)N * D * H * W;
    // 32-bit voxel indices and per-lane element offsets inside the kernels
    return nvox < (1ll << 31) - 4096 && 8ll * H * W * (ld_fine > ld_coarse ? ld_fine : ld_coarse) < (1ll << 31) && nvox / 64 * (p.Nc / p.BN) < (1ll << 31) - 65536;
}

// K-splits of the GEMM-form input gradient on fp32 tensors: few tiles (the deep levels) and a long K
static int ct_ksplit(bool gather, int elem_bytes, const CtPlan& p, long long nvox) {
    if (!gather || elem_bytes != 4 || p.stream) return 1;
    const long long tiles = ((nvox + (p.BN == 128 ? 63 : 127)) / (p.BN == 128 ? 64 : 128)) * (p.Nc / p.BN);
    const int nchunk = p.K / 64;
    int ks = 1;
    while (tiles * ks < 512 && ks < 8 && nchunk % (ks * 2) == 0 && nchunk / (ks * 2) >= 2) ks *= 2;
    return ks;
}
size_t convt_direct_ws_bytes(int Cin, int Cout) { return align_up((size_t)8 * Cin * Cout * 6, 256); }
// (+ the split-K slabs of the deep levels' input gradient: at most 8 x 512 tiles' worth of fp32 results)
size_t convt_direct_slab_bytes(long long nvox, int Cin, int Cout) {
    CtPlan p;
    if (!ct_plan(4, true, nvox, Cin, Cout, &p)) return 0;
    const int ks = ct_ksplit(true, 4, p, nvox);
    return ks > 1 ? align_up((size_t)ks * nvox * Cin * sizeof(float), 256) : 0;
}

template <typename TT, bool GATHER>
static void launch_any(const CtPlan& p, const CtArgs& a, hipStream_t st) {
    constexpr bool F32 = sizeof(TT) == 4;
    if (p.stream) {
        // (K, BN, voxel tiles per wave, waves per workgroup, waves per SIMD the register budget is cut for)
        if (p.K == 64) { if (p.BN == 128) launch_stream<TT, GATHER, 64, 128, 1, 4, 3>(a, st); else launch_stream<TT, GATHER, 64, 64, 2, 4, 3>(a, st); }
        else if (p.K == 128) {
            if (p.BN == 128) { if constexpr (!F32) launch_stream<TT, GATHER, 128, 128, 1, 4, 3>(a, st); }
            else launch_stream<TT, GATHER, 128, 64, F32 ? 1 : 2, 4, 3>(a, st);
        } else {
            if (p.BN == 128) { if constexpr (!F32) launch_stream<TT, GATHER, 256, 128, 1, 4, 2>(a, st); }
            else launch_stream<TT, GATHER, 256, 64, 1, F32 ? 8 : 4, 2>(a, st);
        }
        return;
    }
    // the wide direction gets the wide tile: 64 x 128 when the GEMM has >= 128 columns, else 128 x 64
    if (p.BN == 128) launch_ct<TT, GATHER, 64, 128>(a, (int)((a.nvox + 63) / 64), st);
    else launch_ct<TT, GATHER, 128, 64>(a, (int)((a.nvox + 127) / 128), st);
}

template <typename TT>
int convt_direct(bool gather, const TT* x, int ldx, const float* w, const float* bias, TT* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                 void* ws, size_t ws_bytes, hipStream_t st, float* y_amax) {
    constexpr int NP = sizeof(TT) == 4 ? 3 : 1;
    CtPlan p;
    SEG_CHECK_ARG(ct_plan((int)sizeof(TT), gather, (long long)N * D * H * W, Cin, Cout, &p), "convt_direct: unsupported shape");
    Carver cv(ws);
    bf16x8_t* wq = reinterpret_cast<bf16x8_t*>(cv.take<char>((size_t)p.K * p.Nc * 2 * NP));
    SEG_CHECK_WS(cv.used(), ws_bytes);
    const long long slots = (long long)p.K * p.Nc / 8;
    {
        const PackKey pkey = make_pack_key(w, PK_CONVT_DIRECT, NP, Cin, Cout, gather ? 1 : 0, p.BN, p.K, p.Nc);
        void* hit = nullptr;
        if (prepack_find(pkey, &hit, nullptr)) wq = reinterpret_cast<bf16x8_t*>(hit);
        else {
            PackDesc pd{};
            pd.kind = PD_CONVT; pd.np = NP; pd.w = w; pd.dst = wq; pd.K = Cin; pd.Nn = Cout; pd.mode = gather ? 1 : 0; pd.P = p.BN; pd.aux = p.K; pd.TW = p.Nc;
            pack_launch(pd, st);
            SEG_CHECK_LAUNCH();
            prepack_note(pkey, (size_t)p.K * p.Nc * 2 * NP, pd);
        }
    }
    CtArgs a{x, wq, gather ? nullptr : bias, y, ldx, ldy, D, H, W, Cout, (long long)N * D * H * W, p.K / 64, p.Nc / p.BN, gather ? nullptr : reinterpret_cast<unsigned*>(y_amax),
             1, p.K / 64, 0};
    const double vox = (double)a.nvox;
    ProfScope ps(PF_CONVT, 2.0 * vox * 8 * Cin * Cout, sizeof(TT) * vox * (Cin + 8.0 * Cout) + 4.0 * 8 * Cin * Cout, st);
    const int ks = ct_ksplit(gather, (int)sizeof(TT), p, a.nvox);
    float* slabs = nullptr;
    if (ks > 1) {
        slabs = cv.take<float>((size_t)ks * a.nvox * Cin);
        if (cv.used() <= ws_bytes) { a.ksplit = ks; a.cps = a.nchunk / ks; a.slab_stride = a.nvox * Cin; a.y = slabs; a.ldy = Cin; }     // (no room: unsplit)
    }
    if (gather) launch_any<TT, true>(p, a, st); else launch_any<TT, false>(p, a, st);
    SEG_CHECK_LAUNCH();
    if (a.ksplit > 1) {
        const long long tot = a.nvox * (Cin / 4);
        hipLaunchKernelGGL(convt_splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256 > 2048 ? 2048 : (tot + 255) / 256)), dim3(256), 0, st, slabs, a.ksplit, a.slab_stride,
                           reinterpret_cast<float*>(y), ldy, a.nvox, Cin);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

template int convt_direct<float>(bool, const float*, int, const float*, const float*, float*, int, int, int, int, int, int, int, void*, size_t, hipStream_t, float*);
template int convt_direct<bf16>(bool, const bf16*, int, const float*, const float*, bf16*, int, int, int, int, int, int, int, void*, size_t, hipStream_t, float*);

}  // namespace seg
