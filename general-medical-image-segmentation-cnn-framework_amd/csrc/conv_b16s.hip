// conv_b16s.hip -- Conv3d k3 s1 p1 / k5 s1 p2 forward and input gradient for bf16 tensors (the bf16 configurations: V-Net
// vnet3d.py:21-31, Residual U-Net residual_unet3d.py:82-107, UNETR decoder unetr.py:19-42) on v_mfma_f32_16x16x32_bf16.
//
// What bounded the generic kernel's bf16 tiles (igemm_kernel.h, MATH_B16; r3 PMC, profiles/r03_pmc_b16_layers_before.csv):
// with ONE MFMA per (voxel fragment, weight fragment) pair every MFMA needs a kilobyte of each operand.  The four waves of a
// workgroup load the SAME weight fragments through the CU's L1 (64 B/clk), and with two M-blocks per wave (k5) that path is
// saturated (k5 64->64: waves parked in s_waitcnt 69 % of their cycles, MFMA busy 0.34); the thin k3 layers idle in tile
// prologues / epilogues (MFMA busy 0.37).  Here:
//   * a wave owns EIGHT x-lines of 16 voxels and 32 output channels (two 16-channel MFMA tiles): a weight fragment feeds 8 MFMAs
//     (L1 traffic 32 B/clk/CU) and a voxel fragment 2 (LDS 128 B/clk/CU), 16 MFMAs per 10 fragment loads;
//   * the four waves form a 4 x 1 grid (512 voxels x 32 channels) or a 2 x 2 grid (256 voxels x 64 channels);
//   * K = 32 per MFMA = two taps x 16 input channels (consecutive taps paired; the odd last tap pairs with zero weights:
//     1/28 resp. 1/126 of the MFMAs), the halo tile in LDS is piece-major ([channel half][voxel], 16-byte slots) exactly as in
//     conv_x3s.hip -- 32 bytes per voxel, so even the 5^3 halo of a 512-voxel tile (61 KB) fits twice per CU;
//   * D^T orientation with the output channels permuted inside a wave's 32 so that a lane holds EIGHT consecutive channels of
//     one voxel: one 16-byte bf16 store per voxel line;
//   * halo loads are buffer loads at per-tile precomputed offsets (zero fill by the range check), weights run two K-steps
//     ahead in a register ring, the voxel fragments of the next half-step are requested behind the current MFMAs.
#include "common.h"
#include "internal.h"
#include "igemm_kernel.h"
#include <type_traits>

namespace seg {

namespace {

constexpr int SBX = 16, STY = 4, SRV = 8;                        // x-line, y-lines per z-slab, lines per wave

template <int KS, int WMG>
struct SGeo {
    static constexpr int HALO = KS / 2, NTAP = KS * KS * KS, NSTEP = (NTAP + 1) / 2;
    static constexpr int WNG = 4 / WMG, NT = 32 * WNG;
    static constexpr int LINES = WMG * SRV, TZ = LINES / STY;
    static constexpr int HX = SBX + 2 * HALO, HY = STY + 2 * HALO, HZ = TZ + 2 * HALO;
    static constexpr int NVOX = HX * HY * HZ;
    static constexpr int PS = (NVOX + 15) / 16 * 16;            // slots per piece, a multiple of 256 bytes
    static constexpr int LDS_BYTES = 2 * PS * 16;
    static constexpr int NPIECE = NVOX * 2;                     // staged 16-byte pieces (8 bf16 channels) per chunk
    static constexpr int NITER = (NPIECE + 255) / 256;
    static constexpr int slot(int t) { return ((t / (KS * KS)) * HY + (t / KS) % KS) * HX + t % KS; }
    // second tap of K-step s relative to the first: next voxel in x (0), first voxel of the next row (1) / plane (2), none (3)
    static constexpr int kind(int s) {
        const int t = 2 * s;
        return t + 1 >= NTAP ? 3 : (t % KS != KS - 1 ? 0 : ((t / KS) % KS != KS - 1 ? 1 : 2));
    }
    static constexpr int D1 = HX - (KS - 1), D2 = HY * HX - (KS - 1) * HX - (KS - 1);
    static_assert(LDS_BYTES >= 8 * NT * 4, "the statistics epilogue needs 8 x NT floats");
};

__device__ __forceinline__ float row16_sum_b(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

#ifndef B16S_K3_OCC
#define B16S_K3_OCC 2
#endif
#ifndef B16S_WD
#define B16S_WD 2
#endif
#ifndef B16S_XD
#define B16S_XD 1               // regions (half K-steps) the voxel fragments are requested ahead of their MFMAs
#endif
template <int KS, int WMG>
__global__ __launch_bounds__(256, KS == 3 ? B16S_K3_OCC : 2) void conv_b16s_kernel(IgemmArgs a) {
    using G = SGeo<KS, WMG>;
    constexpr int NT = G::NT, WNG = G::WNG, NSTEP = G::NSTEP, PS = G::PS;
    constexpr int UNIT = (NT / 16) * 512;                        // bf16 elements of one K-step of packed weights: [tile][lane][8]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wm = wave / WNG, wn = wave % WNG;

    // ---- block -> tile map: as conv_igemm_kernel (XCD-contiguous ranges, (y, z) bricks of M-tiles, K-splits of a tile adjacent)
    int ks, ntile, mtile, n, x0, y0, z0;
    {
        const int total = (int)gridDim.x, bid = blockIdx.x;
        const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
        const int xstart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, xcnt = q8 + (xcd < r8 ? 1 : 0);
        const int local = bid >> 3;
        if (local >= xcnt) return;
        int t = xstart + local;
        ks = t % a.ksplit; t /= a.ksplit;
        ntile = t % a.nN;
        mtile = t / a.nN;
        int mt = mtile;
        const int per_n = a.ntx * a.nty * a.ntz;
        n = mt / per_n; mt -= n * per_n;
        const int zfull = a.ntz / a.bz;
        const int rowtiles = a.ntx * a.nty * a.bz;
        int zrow = mt / rowtiles, bzz = a.bz;
        if (zrow >= zfull) { zrow = zfull; bzz = a.ntz - zfull * a.bz; }
        mt -= zrow * rowtiles;
        const int blk = a.ntx * a.by * bzz;
        const int b = mt / blk; mt -= b * blk;
        const int txi = mt % a.ntx; mt /= a.ntx;
        const int tyi = b * a.by + mt % a.by;
        const int tzi = zrow * a.bz + mt / a.by;
        x0 = txi * SBX; y0 = tyi * STY; z0 = tzi * G::TZ;
    }
    const int n0 = ntile * NT;
    const int c0 = ks * a.cps, c1 = c0 + a.cps;                   // this split's chunk range
    const bf16* __restrict__ xin = reinterpret_cast<const bf16*>(a.x);

    // ---- halo staging: buffer loads at per-tile offsets (zero fill by the range check) -> registers -> LDS
    // (r6: formed incrementally -- additions and selects, no per-piece divisions or multiplies: halo_piece_offsets, igemm_kernel.h)
    int voff[G::NITER];
    halo_piece_offsets<G::NITER, G::NPIECE, 2, G::HX, G::HY, G::HZ>(voff, tid, x0 - G::HALO, y0 - G::HALO, z0 - G::HALO, a.D, a.H, a.W, a.ldx * 2);
    const bf16* xsample = xin + (long long)n * a.D * a.H * a.W * a.ldx;
    const int sample_bytes = a.D * a.H * a.W * a.ldx * 2;        // < 2^31: checked on the host
    bf16x8_t stage[G::NITER];
    auto load_stage = [&](int chunk) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(xsample + chunk * 16), 0, sample_bytes, 0x00020000);
#pragma unroll
        for (int it = 0; it < G::NITER; ++it)
            stage[it] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[it], 0, 0));
    };
    // piece p = it * 256 + tid sits at LDS byte  ((p & 1) * PS + p / 2) * 16  =  wdst + it * 2048
    unsigned char* const wdst = lds_raw + ((tid & 1) * PS + (tid >> 1)) * 16;
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < G::NITER; ++it)
            if (it * 256 + tid < G::NPIECE) *reinterpret_cast<bf16x8_t*>(wdst + it * 2048) = stage[it];
    };

    // per-lane LDS byte bases of the voxel fragments: first line of this wave, tap (0, 0, 0); one per pair kind
    const int line0 = wm * SRV;
    const int lane_slot = ((line0 / STY) * G::HY) * G::HX + r + (g & 1) * PS;
    const int hi = g >> 1;
    const int xb0 = (lane_slot + hi) * 16, xb1 = (lane_slot + hi * G::D1) * 16, xb2 = (lane_slot + hi * G::D2) * 16, xb3 = lane_slot * 16;
    auto xaddr = [&](int s, int j) {                             // s, j compile-time after unrolling: everything but the base folds into the offset field
        const int base = G::kind(s) == 0 ? xb0 : (G::kind(s) == 1 ? xb1 : (G::kind(s) == 2 ? xb2 : xb3));
        return base + (((j / STY) * G::HY + j % STY) * G::HX + G::slot(2 * s)) * 16;
    };

    f32x4 acc[SRV][2];
#pragma unroll
    for (int j = 0; j < SRV; ++j) { acc[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const bf16* wlane = reinterpret_cast<const bf16*>(a.wq) + (long long)ntile * a.nchunks * (NSTEP * UNIT) + wn * 1024 + lane * 8;
    // (-DMI355SEG_TUNE timing probes, MI355SEG_DBG; same launch, garbage results: 1 = the halo is staged for a tile's first chunk only,
    //  64 = no halo loads at all (the first chunk stages stale registers), 128 = no MFMAs, 32 = no output stores, 256 = no K loop)
    if (!SEG_DBG(a, 64)) load_stage(c0);
    for (int chunk = c0; chunk < (SEG_DBG(a, 256) ? c0 : c1); ++chunk) {     // (probe 256: no K loop at all -- a tile's prologue and epilogue alone)
        const bf16* wp = wlane + (long long)chunk * (NSTEP * UNIT);
        constexpr int WD = B16S_WD;                              // K-steps of weights in flight ahead of the MFMAs
        constexpr int XD = B16S_XD;
        bf16x8_t wf[WD + 1][2], xf[XD + 1][4];
        auto load_w = [&](int s) {
            wf[s % (WD + 1)][0] = *reinterpret_cast<const bf16x8_t*>(wp + s * UNIT);
            wf[s % (WD + 1)][1] = *reinterpret_cast<const bf16x8_t*>(wp + s * UNIT + 512);
        };
        // the first weight fragments are requested BEFORE the next chunk's halo: vmcnt retires in order
#pragma unroll
        for (int s = 0; s < WD; ++s) load_w(s);
        if (!SEG_DBG(a, 1) || chunk == c0) {
        __syncthreads();                                         // every wave is done reading the previous chunk
        write_stage();
        __syncthreads();
        }
        if (chunk + 1 < c1 && !SEG_DBG(a, 1) && !SEG_DBG(a, 64)) load_stage(chunk + 1);
#pragma unroll
        for (int q0 = 0; q0 < XD; ++q0)
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[q0][j] = *reinterpret_cast<const bf16x8_t*>(lds_raw + xaddr(q0 >> 1, (q0 & 1) * 4 + j));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2 * NSTEP; ++q) {                    // region = (K-step, half of the wave's lines)
            const int s = q >> 1, hf = q & 1, cur = q % (XD + 1), nxt = (q + XD) % (XD + 1);
            if (hf == 0 && s + WD < NSTEP) load_w(s + WD);
            if (q + XD < 2 * NSTEP) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[nxt][j] = *reinterpret_cast<const bf16x8_t*>(lds_raw + xaddr((q + XD) >> 1, ((q + XD) & 1) * 4 + j));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (!SEG_DBG(a, 128)) acc[hf * 4 + j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s % (WD + 1)][t], xf[cur][j], acc[hf * 4 + j][t], 0, 0, 0);
                    else acc[hf * 4 + j][t][0] += (float)wf[s % (WD + 1)][t][0] + (float)xf[cur][j][0];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            if (hf == 0) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: bias, one 16-byte bf16 store per line, optional BatchNorm partial statistics
    // acc[j][t][e] = y[voxel (line j, x = r)][channel n0 + 32 wn + 8 g + 4 t + e]
    bf16* yout = reinterpret_cast<bf16*>(a.y);
    float* yslab = reinterpret_cast<float*>(a.y) + (long long)ks * a.split_stride;     // ksplit > 1: raw fp32 partial sums of this split
    const bool slab = a.ksplit > 1;
    const int gx = x0 + r, cbase = n0 + wn * 32 + 8 * g;
    float bv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) bv[c] = a.bias ? a.bias[cbase + c] : 0.f;
    float ssum[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) ssum[c] = 0.f;
    // r6: a line's address = this lane's voxel of the tile's first line + a wave-uniform line offset (line0 is wave-uniform: scalar
    // arithmetic instead of three 64-bit vector multiplies per line), and the residual's eight fragments are requested up front,
    // unconditionally at clamped coordinates -- under the per-line `if (inside)` each was a load -> s_waitcnt vmcnt(0) -> store round
    // trip that also waited for the previous line's store (the conv_x3s epilogue's r5 finding)
    const long long tile_row = (((long long)n * a.D + z0) * a.H + y0) * a.W;
    const long long ylane = (tile_row + gx) * a.ldy + cbase;
    bf16x8_t rs[SRV] = {};
    if (a.res && !slab) {
        const bf16* rlane = reinterpret_cast<const bf16*>(a.res) + (long long)min(gx, a.W - 1) * a.ldres + cbase;
#pragma unroll
        for (int j = 0; j < SRV; ++j) {
            const int line = line0 + j;
            const int cz = min(z0 + line / STY, a.D - 1), cy = min(y0 + line % STY, a.H - 1);
            rs[j] = *reinterpret_cast<const bf16x8_t*>(rlane + (((long long)n * a.D + cz) * a.H + cy) * a.W * a.ldres);
        }
    }
    // (r6) the store loop in two instantiations -- ACT: the fused inference forward's activation (a per-element switch with exp / division
    // branches: 64 inlined copies of it were 60 KB of the training launches' 70-KB instruction stream, all of it jumped over) -- so that the
    // training path's epilogue is a few hundred contiguous instructions
    auto store_lines = [&](auto ACTc) {
        constexpr bool ACT = decltype(ACTc)::value;
#pragma unroll
        for (int j = 0; j < SRV; ++j) {
            const int line = line0 + j;
            const int gz = z0 + line / STY, gy = y0 + line % STY;
            const bool inside = gz < a.D && gy < a.H && gx < a.W;
            const long long off = ylane + (long long)(((line / STY) * a.H + line % STY) * a.W) * a.ldy;
            if (slab) {                                              // bias and statistics belong to the reduce pass
                if (inside) {
                    *reinterpret_cast<f32x4*>(yslab + off) = acc[j][0];
                    *reinterpret_cast<f32x4*>(yslab + off + 4) = acc[j][1];
                }
            } else {
                bf16x8_t o;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float v = acc[j][c >> 2][c & 3] + bv[c];
                    if constexpr (ACT) v = act_apply(v, a.act, a.slope);
                    if (a.res) v = (float)(bf16)v + (float)rs[j][c];       // the convolution's own bf16 rounding, then the sum's
                    o[c] = (bf16)v;
                    if (inside) ssum[c] += v;
                }
                if (inside && (!SEG_DBG(a, 32) || o[0] == (bf16)12345.f)) *reinterpret_cast<bf16x8_t*>(yout + off) = o;
            }
        }
    };
    if (a.act) store_lines(std::true_type{}); else store_lines(std::false_type{});
    if (a.spart) {
        // per channel (sum, M2 about the tile mean, n) of this tile, as conv_igemm_kernel: spart[mtile][c] = {sum, M2, n}
        float* lds = reinterpret_cast<float*>(lds_raw);
        const int cl = wn * 32 + 8 * g;                          // this lane's first channel inside the tile
        __syncthreads();                 // LDS halo no longer needed
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float s1 = row16_sum_b(ssum[c]);
            if (r == 0) lds[wm * NT + cl + c] = s1;
        }
        const int vz = min(G::TZ, a.D - z0), vy = min(STY, a.H - y0), vx = min(SBX, a.W - x0);
        const float cnt = (float)(vz * vy * vx);
        __syncthreads();
        float tmean[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float ts = 0.f;
#pragma unroll
            for (int w = 0; w < WMG; ++w) ts += lds[w * NT + cl + c];
            tmean[c] = ts / cnt;
        }
        __syncthreads();
        float m2[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) m2[c] = 0.f;
#pragma unroll
        for (int j = 0; j < SRV; ++j) {
            const int line = line0 + j;
            if ((z0 + line / STY) < a.D && (y0 + line % STY) < a.H && gx < a.W) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { const float d = acc[j][c >> 2][c & 3] + bv[c] - tmean[c]; m2[c] += d * d; }
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = row16_sum_b(m2[c]);
            if (r == 0) lds[(4 + wm) * NT + cl + c] = v;
        }
        __syncthreads();
        if (tid < NT) {
            float s1 = 0.f, mm = 0.f;
#pragma unroll
            for (int w = 0; w < WMG; ++w) { s1 += lds[w * NT + tid]; mm += lds[(4 + w) * NT + tid]; }
            float* dst = a.spart + ((long long)mtile * a.Cout + n0 + tid) * 3;
            dst[0] = s1; dst[1] = mm; dst[2] = cnt;
        }
    }
}

template <int KS, int WMG>
void launch_b16s(const IgemmArgs& a, int nwg, hipStream_t st) {
    constexpr int LDSB = SGeo<KS, WMG>::LDS_BYTES;
    SEG_SET_LDS((conv_b16s_kernel<KS, WMG>), LDSB);
    hipLaunchKernelGGL((conv_b16s_kernel<KS, WMG>), dim3(nwg), dim3(256), LDSB, st, a);
}

int g_b16_tiles = 0;      // 0 auto (enough tiles to fill the chip), 1 wherever the geometry allows, 2 never

}  // namespace

void set_b16_tiles(int mode) { g_b16_tiles = mode; }
int get_b16_tiles() { return g_b16_tiles; }

// 512-voxel x 32-channel tiles (4 x 1 waves) or 256-voxel x 64-channel tiles (2 x 2 waves), x-lines of 16 voxels
bool b16s_geom(int KS, int N, int D, int H, int W, int Cin, int Cout, B16sPlan* p) {
    if ((KS != 3 && KS != 5) || Cin % 16 || Cout % 32 || W < 4) return false;
    p->KS = KS;
    p->WMG = Cout % 64 == 0 ? 2 : 4;
    p->NT = p->WMG == 2 ? 64 : 32;
    p->TZ = p->WMG * SRV / STY;
    p->ntx = (W + SBX - 1) / SBX; p->nty = (H + STY - 1) / STY; p->ntz = (D + p->TZ - 1) / p->TZ;
    p->nM = N * p->ntz * p->nty * p->ntx; p->nN = Cout / p->NT;
    p->nsteps = (KS * KS * KS + 1) / 2;
    p->ksplit = pick_ksplit(p->nM * p->nN, Cin / 16);
    return true;
}
bool b16s_plan(int KS, int N, int D, int H, int W, int Cin, int Cout, const void* x, int ldx, const void* y, int ldy, B16sPlan* p) {
    if (g_b16_tiles == 2 || !b16s_geom(KS, N, D, H, W, Cin, Cout, p)) return false;
    if ((ldx % 8) || (ldy % 8) || ((uintptr_t)x % 16) || ((uintptr_t)y % 16)) return false;
    if ((long long)D * H * W * ldx * 2 >= 0x7FFFFFF0LL) return false;       // one sample is addressed through a 32-bit buffer offset
    if (g_b16_tiles == 1) return true;
    const double waste = (double)p->ntx * SBX * p->nty * STY * p->ntz * p->TZ / ((double)W * H * D);
    if (waste <= 1.35 && (long long)p->nM * p->nN * p->ksplit >= 192) return true;
    // r5: the deep levels (20^3 ... 4^3 voxels, 64-512 channels) pad these tiles by up to 4.7x and cut few of them -- and still run 1.2-2.8x
    // faster here than on the generic 32x32x16 tiles (profiles/r05_b16s_small_volumes_ab.log: every shape with >= 1,728 products per output
    // value, i.e. Cin >= 64 at k3 or any k5 layer; the two shapes that lost have Cin = 32 at k3: two K chunks per tile and no tap split)
    return (long long)Cin * KS * KS * KS >= 1728;
}
size_t b16s_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k) {
    B16sPlan p;
    if (!b16s_geom(k, N, D, H, W, Cin, Cout, &p)) return 0;
    // what conv_fwd_mfma carves on this path, in its own expressions: the packed weights as wq_bytes(MATH_B16, T Cin Cout) sizes them
    // (the 28-tap-style padding + 64 elements, larger than the 2 * nsteps taps for k5), the per-tile statistics triples with their
    // two-stage reduce scratch, the split-K slabs with the column-sum scratch of the statistics that then follow the reduce
    const size_t T = (size_t)k * k * k, nelem = T * Cin * Cout;
    return align_up((nelem + nelem / 27 + 64) * 2, 256) + align_up((size_t)p.nM * Cout * 3 * sizeof(float), 256) + align_up(part_reduce_ws_bytes(Cout), 256) +
           (p.ksplit > 1 ? align_up((size_t)p.ksplit * N * D * H * W * Cout * sizeof(float), 256) + colsum_ws_bytes(Cout) : 0) + 1024;
}

void dispatch_b16s(const B16sPlan& p, const IgemmArgs& a, int nwg, hipStream_t st) {
    if (p.KS == 3) { if (p.WMG == 2) launch_b16s<3, 2>(a, nwg, st); else launch_b16s<3, 4>(a, nwg, st); }
    else { if (p.WMG == 2) launch_b16s<5, 2>(a, nwg, st); else launch_b16s<5, 4>(a, nwg, st); }
}

}  // namespace seg
