// conv_x3s.hip -- Conv3d k3 s1 p1 forward / input gradient for fp32 tensors on the bf16 matrix cores with the "bf16x6"
// split (x = h + m + l, six bf16 products per fp32 product, fp32 accumulate: see igemm_kernel.h, MATH_X3), built on
// v_mfma_f32_16x16x32_bf16 instead of the 32x32x16 shape of the generic kernel.
//
// Why a second kernel: the bf16x6 loop is power-limited (r2 PMC: MFMA busy 0.65-0.69 at an effective 1.85 GHz).  A timing
// probe that replaced every 32x32x16 MFMA of the generic loop by two 16x16x32 MFMAs on the same operands (same FLOPs, same
// LDS and L2 traffic) ran 10-21 % faster on the cfg-2 layers (profiles/r03_mfma_shape_probe.log): the chip holds a higher
// clock on the smaller shape (MI355X_MICROARCH.md, DVFS give-back (7)).
//
//   D^T orientation: A operand = weights (16 output channels x K), B operand = voxels (K x 16 voxels of one x-line), so a
//   lane ends up with FOUR CONSECUTIVE CHANNELS of one voxel per accumulator: the epilogue stores 16 bytes per lane.
//   K = 32 per MFMA = two taps x 16 input channels (CK = 16 keeps the 3-plane halo tile at 63 KB: two workgroups per CU);
//   the 27 taps form 13 pairs + one tap paired with zero weights (14 K-steps per 16-channel chunk, 3.7 % padding).
//   LDS tile: piece-major -- six arrays (plane h/m/l x channel half) of 16-byte voxel slots -- so that the four k-groups of a
//   fragment read (lane = (voxel r, k-group g): g&1 = channel half, g>>1 = first / second tap of the pair) are one
//   conflict-free ds_read_b128: piece stride = 0 mod 256 B puts the two halves of a 32-lane group on the 16 distinct
//   slots r, and a tap shift only rotates them.
//   Tile = the generic kernel's BX = 16 tile (4 x 4 x 16 or 2 x 4 x 16 voxels, 32 or 64 channels, same walk, same split-K,
//   same BatchNorm-statistics epilogue), so the host-side plan is shared (conv_mfma.hip).
//
// F16 = true ("f16x3", round 4): the same kernel with a TWO-piece split on v_mfma_f32_16x16x32_f16.  v * 2^s = h + l with fp16 h, l
// (11 + 11 mantissa bits and the sign of l: |v 2^s - h - l| <= 2^-23 |v 2^s|), the per-tensor power of two 2^s placing the
// tensor's largest magnitude in [2^14, 2^15) so that h never overflows and l (>= 2^-11 of its value) stays above the fp16
// underflow for every value within 2^-18 of the maximum (smaller values keep an absolute error of 2^-40 of the maximum: the fp16
// subnormals are exact on the matrix cores).  Three MFMAs per product (l h, h l, h h; the dropped l l term is <= 2^-22 of the
// product, 2^-24.6 rms) instead of six, 8 conversion VALU per four values instead of 22, four LDS pieces instead of six.  The two
// maxima are read from device memory (a.amax_x, a.amax_w: upper bounds of max |x|, max |w|); the accumulators are scaled back
// by 2^-(sx + sw) before anything in the epilogue looks at them.
#include "common.h"
#include "internal.h"
#include "igemm_kernel.h"
#include <type_traits>

namespace seg {

namespace {

#ifndef X3S_WN2
#define X3S_WN2 1
#endif
#ifndef X3S_WD
#define X3S_WD 3
#endif
#ifndef X3S_XD
#define X3S_XD 2
#endif
#ifndef X3S_HALO_AUX
#define X3S_HALO_AUX 0          // cache policy of the halo loads (A/B knob: 2 = nt, 16 = sc1)
#endif
#ifndef X3S_SPLIT_MIX
#define X3S_SPLIT_MIX 0          // 1: h / l of the scaled split as one v_fma_mix{lo,hi}_f16 each (10 instead of 12 VALU per staged quad; measured 0.99x: kept off)
#endif
#ifndef X3S_BNX_EARLY
#define X3S_BNX_EARLY 1
#endif
constexpr int XBX = 16, XTY = 4, XHX = XBX + 2, XHY = XTY + 2;
constexpr int X3S_PRO_MAX_CIN = 512;             // the prologue table (8 bytes per input channel) must fit beside two halo tiles per CU

template <int LW, bool F16 = false, int WN = 1>
struct Geo {
    static constexpr int NPL = F16 ? 2 : 3;                     // planes of the split
    static constexpr int LINES = (4 / WN) * LW, TZ = LINES / XTY, HZ = TZ + 2;
    static constexpr int NVOX = XHX * XHY * HZ;
    static constexpr int PS = (NVOX + 15) / 16 * 16;            // 16-byte slots per piece; a multiple of 16 slots (256 B)
    static constexpr int LDS_BYTES = 2 * NPL * PS * 16;
    static constexpr int NPIECE = NVOX * 4;                     // staged 16-byte pieces (4 fp32 channels) per chunk
    static constexpr int NITER = (NPIECE + 255) / 256;
};


__device__ __forceinline__ constexpr int tap_slot(int t) { return ((t / 9) * XHY + (t / 3) % 3) * XHX + t % 3; }
// which per-lane base a K-step uses: the second tap of the pair is +1 slot (x), +HX (y), +HY*HX (z) or the same voxel (zero weights)
__device__ __forceinline__ constexpr int pair_kind(int s) { return s < 9 ? 0 : (s < 12 ? 1 : (s == 12 ? 2 : 3)); }

// sum over the 16 lanes of a DPP row (lanes with equal lane >> 4); result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// ---- epilogue shared by the kernels of this file: a wave holds LW x-lines (line0 .. line0 + LW - 1 of a TZ x 4 x 16 tile) x all
// NT = 32 NBW channels of the tile; scale_exp != 0: the accumulators are first multiplied by 2^scale_exp (f16x3)
// WN = 2: the four waves form a 2 (lines) x 2 (channels) grid -- wave (wm, wn) = (wave / 2, wave % 2) holds its LW lines x the 32 NBW
// channels n0 + 32 NBW wn .. of a tile of 64 NBW channels; the per-tile statistics then sum the two waves that share a channel block.
template <int LW, int NBW, int TZ, int WN = 1>
__device__ __forceinline__ void x3_epilogue(const IgemmArgs& a, f32x4 (&acc)[LW][2 * NBW], int scale_exp, int ks, int mtile, int n, int x0, int y0, int z0,
                                            int n0_tile, int line0, int r, int g, int wave, int tid, unsigned char* lds_raw) {
    constexpr int NT = 32 * NBW, NTW = 2 * NBW;              // channels / 16-channel MFMA tiles of ONE wave
    constexpr int NTT = NT * WN, WM = 4 / WN;                // channels of the tile, waves along the lines
    const int wn = WN == 1 ? 0 : wave % WN;
    const int n0 = n0_tile + wn * NT;
    // ---- epilogue: bias, 16-byte stores, optional BatchNorm partial statistics
    // acc[j][t][e] = y[voxel (line j, x = r)][channel n0 + 16 t + 4 g + e]
    const int gx = x0 + r;
    // BatchNorm-backward column sums of the layer in front (input-gradient launch, no bias / activation / split): the tile about to be
    // stored is d(activation); that layer's pre-norm tensor is read at the same voxels.  r5: ALL of these loads are issued here, ahead of
    // the scaling and the stores, unconditionally and at clamped coordinates (a voxel outside the volume contributes through a zero
    // mask) -- under the per-line `if (inside)` they were 2 LW load -> s_waitcnt vmcnt(0) -> reduce round trips per wave, each also
    // waiting for the tile's own stores: 14 us per tile (0.756 ms against the plain kernel's 0.53 on 32 -> 32 @ 2 x 128^3)
    // (groups of eight fragments: the first is requested here, the next behind the stores once the previous one is reduced -- all
    // LW x NTW fragments at once next to the accumulators spill)
    constexpr int LH = (8 / NTW) < LW ? (8 / NTW) : LW;
    // two buffers: group k + 1 is requested before group k is reduced (a group's loads behind the previous group's reduction were an
    // exposed HBM round trip per group: +20 % on the 32 -> 32 layers)
    constexpr int NG = (LW + LH - 1) / LH;
    constexpr bool DB = NG > 1 && NTW <= 2;          // (the 64-channel-per-wave forms have no registers for a second group)
    f32x4 bxv[DB ? 2 : 1][LH][NTW];
    const int bnx_lane = min(gx, a.W - 1) * a.ldbnx + n0 + 4 * g;
    auto load_bnx = [&](int j0, int buf) {
#pragma unroll
        for (int j = 0; j < LH; ++j) {
            const int line = line0 + min(j0 + j, LW - 1);
            const int cz = min(z0 + line / XTY, a.D - 1), cy = min(y0 + line % XTY, a.H - 1);
            // (line0 is wave-uniform: the row's first voxel is scalar arithmetic, the lane adds its clamped x)
            const float* src = a.bnx + (((long long)n * a.D + cz) * a.H + cy) * a.W * a.ldbnx + bnx_lane;
#pragma unroll
            for (int t = 0; t < NTW; ++t) bxv[buf][j][t] = *reinterpret_cast<const f32x4*>(src + 16 * t);
        }
    };
    // (r6) BOTH buffers are requested ahead of the stores: vmcnt retires in order, so a group requested behind the tile's sixteen stores per
    // lane is only reduced once every one of those stores has been acknowledged
    if (a.bnpart) { load_bnx(0, 0); if (DB && X3S_BNX_EARLY) load_bnx(LH, 1); }
    if (scale_exp != 0) {
#pragma unroll
        for (int j = 0; j < LW; ++j)
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][t][e] = __builtin_ldexpf(acc[j][t][e], scale_exp);
    }
    float* yslab = reinterpret_cast<float*>(a.y) + (long long)ks * a.split_stride;     // ksplit > 1: raw partial sums of this split (split_stride 0 otherwise)
    float* const ylane = yslab + ((((long long)n * a.D + z0) * a.H + y0) * a.W + gx) * a.ldy + n0 + 4 * g;      // line 0 of the TILE, this lane's voxel and channels
    float ssum[NTW][4];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) ssum[t][e] = 0.f;
    float ymax = 0.f;                    // max |y| of this lane's stored values (a.amax_y)
    f32x4 bv[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) bv[t] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n0 + 16 * t + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    // (r6) the store loop in two instantiations -- ACT: the fused inference forward's activation, a per-element switch with exp / division
    // branches whose 4 LW NTW inlined copies were most of a training launch's instruction stream (108 of 130 KB on the <8, 1, F16> tile), all
    // of it jumped over -- so that the training path's epilogue is short and contiguous
    auto store_lines = [&](auto ACTc) {
        constexpr bool ACT = decltype(ACTc)::value;
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            const int line = line0 + j;
            const int gz = z0 + line / XTY, gy = y0 + line % XTY;
            const bool inside = gz < a.D && gy < a.H && gx < a.W;
            float* dst = ylane + (long long)(((line / XTY) * a.H + line % XTY) * a.W) * a.ldy;     // (wave-uniform offset of the line)
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                f32x4 v = acc[j][t] + bv[t];
                if constexpr (ACT) { v[0] = act_apply(v[0], a.act, a.slope); v[1] = act_apply(v[1], a.act, a.slope); v[2] = act_apply(v[2], a.act, a.slope); v[3] = act_apply(v[3], a.act, a.slope); }
                if (inside) {
                    if (!SEG_DBG(a, 2048) || v[0] == 12345.678f) *reinterpret_cast<f32x4*>(dst + 16 * t) = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ssum[t][e] += v[e];
                    if (a.amax_y) ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                }
            }
        }
    };
    if (a.act) store_lines(std::true_type{}); else store_lines(std::false_type{});
    if (a.amax_y) { __syncthreads(); block_amax_commit(ymax, a.amax_y); }
    if (a.bnpart) {
        // reduce dz and dz * xhat per channel over the tile (dz = d(activation) * act'(gamma xhat + beta))
        float sa[NTW][4], sb[NTW][4];
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) { sa[t][e] = 0.f; sb[t][e] = 0.f; }
        auto reduce = [&](int j0, int buf, auto grad) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int c = n0 + 16 * t + 4 * g;
                const f32x4 mu = *reinterpret_cast<const f32x4*>(a.bn_mean + c), rs = *reinterpret_cast<const f32x4*>(a.bn_rstd + c);
                const f32x4 ga = *reinterpret_cast<const f32x4*>(a.bn_gamma + c), be = *reinterpret_cast<const f32x4*>(a.bn_beta + c);
#pragma unroll
                for (int j = 0; j < LH; ++j) {
                    if (j0 + j >= LW) continue;
                    const int line = line0 + j0 + j;
                    const float m = ((z0 + line / XTY) < a.D && (y0 + line % XTY) < a.H && gx < a.W) ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (bxv[buf][j][t][e] - mu[e]) * rs[e];
                        const float dz = acc[j0 + j][t][e] * grad(fmaf(xh, ga[e], be[e])) * m;
                        sa[t][e] += dz; sb[t][e] = fmaf(dz, xh, sb[t][e]);
                    }
                }
            }
        };
        const int bact = a.bn_act;
        const float bslope = a.bn_slope;
        auto relu_grad = [](float z) { return z > 0.f ? 1.f : 0.f; };                       // (the U-Net's case: no per-element switch)
        auto any_grad = [&](float z) { return act_grad(z, bact, bslope); };
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            if (DB && !X3S_BNX_EARLY && k + 1 < NG) load_bnx((k + 1) * LH, (k + 1) & 1);
            if (!DB && k > 0) load_bnx(k * LH, 0);
            if (bact == MI355SEG_ACT_RELU) reduce(k * LH, DB ? (k & 1) : 0, relu_grad); else reduce(k * LH, DB ? (k & 1) : 0, any_grad);
            if (DB && X3S_BNX_EARLY && k + 2 < NG) load_bnx((k + 2) * LH, k & 1);
        }
        float* lds = reinterpret_cast<float*>(lds_raw);
        __syncthreads();                 // LDS halo no longer needed
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v1 = row16_sum(sa[t][e]), v2 = row16_sum(sb[t][e]);
                if (r == 0) { lds[wave * NT + 16 * t + 4 * g + e] = v1; lds[(4 + wave) * NT + 16 * t + 4 * g + e] = v2; }
            }
        __syncthreads();
        if (tid < NTT) {
            const int cw = tid / NT, cl = tid % NT;          // which channel block of the tile, channel inside it
            float v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { v1 += lds[(w * WN + cw) * NT + cl]; v2 += lds[(4 + w * WN + cw) * NT + cl]; }
            float* dst = a.bnpart + ((long long)mtile * a.Cout + n0_tile + tid) * 2;
            dst[0] = v1; dst[1] = v2;
        }
    }
    if (a.spart) {
        // per channel (sum, M2 about the tile mean, n) of this tile, as conv_igemm_kernel: spart[mtile][c] = {sum, M2, n}
        float* lds = reinterpret_cast<float*>(lds_raw);
        __syncthreads();                 // LDS halo no longer needed
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s1 = row16_sum(ssum[t][e]);
                if (r == 0) lds[wave * NT + 16 * t + 4 * g + e] = s1;
            }
        const int vz = min(TZ, a.D - z0), vy = min(XTY, a.H - y0), vx = min(XBX, a.W - x0);
        const float cnt = (float)(vz * vy * vx);
        __syncthreads();
        float tmean[NTW][4];
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = 16 * t + 4 * g + e;
                float sm = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) sm += lds[(w * WN + wn) * NT + c];
                tmean[t][e] = sm / cnt;
            }
        __syncthreads();
        float m2[NTW][4];
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) m2[t][e] = 0.f;
#pragma unroll
        for (int j = 0; j < LW; ++j) {
            const int line = line0 + j;
            const bool inside = (z0 + line / XTY) < a.D && (y0 + line % XTY) < a.H && gx < a.W;
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = acc[j][t][e] + bv[t][e] - tmean[t][e];
                    if (inside) m2[t][e] += d * d;
                }
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = row16_sum(m2[t][e]);
                if (r == 0) lds[(4 + wave) * NT + 16 * t + 4 * g + e] = v;
            }
        __syncthreads();
        if (tid < NTT) {
            const int cw = tid / NT, cl = tid % NT;
            float s1 = 0.f, mm = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { s1 += lds[(w * WN + cw) * NT + cl]; mm += lds[(4 + w * WN + cw) * NT + cl]; }
            float* dst = a.spart + ((long long)mtile * a.Cout + n0_tile + tid) * 3;
            dst[0] = s1; dst[1] = mm; dst[2] = cnt;
        }
    }
}

// LW = x-lines (16 voxels each) per wave: 4 -> tile 4 (z) x 4 (y) x 16, wave w owns z-slab w; 2 -> tile 2 x 4 x 16.
// NBW = 32-channel blocks of the tile (NT = 32 * NBW output channels).
// WN = 2 (r4, f16x3): 2 x 2 wave grid on the 64 NBW-channel tile -- a wave owns LW lines and ONE of the two channel blocks, so it loads
// half of the tile's weight fragments and each feeds twice the MFMAs (the r4 probes: weight-fragment loads are the first bound of
// the three-MFMA loop: 43 B/clk/CU of L1 traffic on the <4, 2> tile)
// PRO (r5, f16x3): the norm + activation prologue of IgemmArgs::pro_al -- a separate instantiation, so that the plain launches keep their
// register allocation (the <8, 1> tile sits at 256)
template <int LW, int NBW, bool F16, int WN = 1, bool PRO = false>
__global__ __launch_bounds__(256, 2) void conv_x3s_kernel(IgemmArgs a) {
    using G = Geo<LW, F16, WN>;
    constexpr int NPL = G::NPL;
    constexpr int NT = 32 * NBW * WN;                            // channels of the tile
    constexpr int NTW = 2 * NBW;                                 // 16-channel MFMA tiles per wave
    constexpr int NU = X3S_NPAIR * NBW;                          // (K-step, 32-channel half) units per chunk that THIS wave runs
    constexpr int NUP = X3S_NPAIR * NBW * WN;                    // ... that the packed weights of the tile hold per chunk
    constexpr int UNIT = 2 * NPL * 512;                          // 16-bit elements of one unit of packed weights: [plane][t2][lane][8]
    constexpr int PS = G::PS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;

    // ---- block -> tile map: identical to conv_igemm_kernel (XCD-contiguous ranges, (y, z) bricks of M-tiles)
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
        x0 = txi * XBX; y0 = tyi * XTY; z0 = tzi * G::TZ;
    }
    const int n0 = ntile * NT;
    const int c0 = ks * a.cps, c1 = c0 + a.cps;
    const float* __restrict__ xin = reinterpret_cast<const float*>(a.x);
    // f16x3: power-of-two scales of the two operands (the weights were scaled while they were packed)
    int sx = 0, sw = 0;
    if constexpr (F16) { sx = f16x_scale_exp(*a.amax_x); sw = f16x_scale_exp(*a.amax_w); }
    const float xscale = pow2f(sx);
    // norm + activation prologue (f16x3): per input channel al * 2^sx | be * 2^sx behind the halo tile (act is positively homogeneous:
    // act(z) 2^sx = act(z 2^sx), so the scale rides in the table); visible after the first barrier of the chunk loop
    float* const ptab = reinterpret_cast<float*>(lds_raw + G::LDS_BYTES);
    const int pcin = a.nchunks * 16;
    if constexpr (PRO) {
        for (int i = tid; i < pcin; i += 256) { ptab[i] = a.pro_al[i] * xscale; ptab[pcin + i] = a.pro_be[i] * xscale; }
    }

    // ---- halo staging: global -> registers (issue early) -> split3 -> LDS (write late).
    // Piece p = it * 256 + tid is (halo voxel p / 4, four channels p % 4) of the chunk.  Its byte offset inside sample n is the
    // same for every chunk, so it is computed once per tile; the loads are buffer loads on a per-chunk descriptor
    // (base = sample + 16 * chunk channels) whose range check returns zeros for the pieces outside the volume (offset
    // 0x7FFFFFF0 >= num_records): no address arithmetic and no select in the K loop.
    // (r6: formed incrementally -- additions and selects, no per-piece divisions or multiplies: halo_piece_offsets, igemm_kernel.h)
    int voff[G::NITER];
    halo_piece_offsets<G::NITER, G::NPIECE, 4, XHX, XHY, G::HZ>(voff, tid, x0 - 1, y0 - 1, z0 - 1, a.D, a.H, a.W, a.ldx * 4);
    const float* xsample = xin + (long long)n * a.D * a.H * a.W * a.ldx;
    const int sample_bytes = a.D * a.H * a.W * a.ldx * 4;        // < 2^31: checked on the host (x3s_plan_ok)
    f32x4 stage[G::NITER];
    auto load_stage = [&](int chunk) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsample + chunk * 16), 0, sample_bytes, 0x00020000);
#pragma unroll
        for (int it = 0; it < G::NITER; ++it)
            stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[it], 0, X3S_HALO_AUX));
    };
    // one piece of the NEXT chunk's halo (the main loop requests one per scheduling region: a burst of all of them in front
    // of the loop would sit in front of every weight fragment requested after it -- vmcnt retires in order)
    auto load_piece = [&](int chunk, int it) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsample + chunk * 16), 0, sample_bytes, 0x00020000);
        stage[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[it], 0, X3S_HALO_AUX));
    };
    // x = h + m + l for a pair of values: three packed conversions, the remainders formed from the packed words
    auto split_pair = [](float x0_, float x1_, unsigned& h2, unsigned& m2, unsigned& l2) {
        using f32x2 = f32x2_t;
        h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0_, x1_}, bf16x2_t));
        const float r0 = x0_ - __builtin_bit_cast(float, h2 << 16), r1 = x1_ - __builtin_bit_cast(float, h2 & 0xFFFF0000u);
        m2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16x2_t));
        const float q0 = r0 - __builtin_bit_cast(float, m2 << 16), q1 = r1 - __builtin_bit_cast(float, m2 & 0xFFFF0000u);
        l2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{q0, q1}, bf16x2_t));
    };
    // f16x3: v 2^s = h + l, two packed conversions; the remainder is one v_fma_mix per value (f16 operand read in place)
    // (r6) scaled form: h = fp16(x sc) and l = fp16(x sc - h) are ONE v_fma_mix{lo,hi}_f16 each (the product with the power of two is exact, the
    // difference is exact in fp32: the same bits as scale -> convert -> subtract -> convert), four VALU per pair instead of six and no packing
#if X3S_SPLIT_MIX
    auto split_pair_h = [&](float x0_, float x1_, unsigned& h2, unsigned& l2, float sc) {
        f16x2_t hh, ll;
        hh[0] = (_Float16)__builtin_fmaf(x0_, sc, 0.f); hh[1] = (_Float16)__builtin_fmaf(x1_, sc, 0.f);
        ll[0] = (_Float16)__builtin_fmaf(x0_, sc, -(float)hh[0]); ll[1] = (_Float16)__builtin_fmaf(x1_, sc, -(float)hh[1]);
        h2 = __builtin_bit_cast(unsigned, hh); l2 = __builtin_bit_cast(unsigned, ll);
    };
#else
    auto split_pair_h = [&](float x0_, float x1_, unsigned& h2, unsigned& l2, float sc) {
        const float s0 = x0_ * sc, s1 = x1_ * sc;
        const f16x2_t hh = __builtin_convertvector(f32x2_t{s0, s1}, f16x2_t);
        h2 = __builtin_bit_cast(unsigned, hh);
        const float r0 = __builtin_fmaf((float)hh[0], -1.f, s0), r1 = __builtin_fmaf((float)hh[1], -1.f, s1);
        l2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t));
    };
#endif
    // piece p = it * 256 + tid sits at LDS byte  ((part >> 1) * PS + tid / 4) * 16 + (part & 1) * 8  +  it * 1024
    unsigned char* const wdst = lds_raw + (((tid & 3) >> 1) * PS + (tid >> 2)) * 16 + (tid & 1) * 8;
    // PRO: the norm + activation prologue on the staged values (this thread's four channels of the chunk: part = tid & 3 for every
    // piece); MASK: the tile touches the volume's border -- pieces that were requested past the descriptor read as zeros and must
    // STAY zero (the padding pads the activation, not its pre-norm tensor); RELU: the U-Net's activation without the generic switch
    auto write_stage = [&](int chunk, auto WPRO, auto MASK, auto RELU) {
        f32x4 pal = {0.f, 0.f, 0.f, 0.f}, pbe = pal;
        if constexpr (decltype(WPRO)::value) {
            pal = *reinterpret_cast<const f32x4*>(ptab + chunk * 16 + (tid & 3) * 4);
            pbe = *reinterpret_cast<const f32x4*>(ptab + pcin + chunk * 16 + (tid & 3) * 4);
        }
#pragma unroll
        for (int it = 0; it < G::NITER; ++it) {
            if (it * 256 + tid < G::NPIECE) {
                using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
                unsigned char* dst = wdst + it * 1024;
                if constexpr (F16) {
                    unsigned h0, l0, h1, l1;
                    if constexpr (decltype(WPRO)::value) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float z = fmaf(stage[it][e], pal[e], pbe[e]);
                            v[e] = decltype(RELU)::value ? fmaxf(z, 0.f) : fmaxf(z, z * a.pro_slope);       // LeakyReLU, slope in [0, 1)
                        }
                        if constexpr (decltype(MASK)::value) { if (voff[it] == 0x7FFFFFF0) v = f32x4{0.f, 0.f, 0.f, 0.f}; }
                        split_pair_h(v[0], v[1], h0, l0, 1.f);
                        split_pair_h(v[2], v[3], h1, l1, 1.f);
                    } else if (SEG_DBG(a, 256)) {
                        // timing probe: the operand arrives ALREADY split (a piece = four fp16 h | four fp16 l): staging is a copy
                        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
                        const u32x4 u = __builtin_bit_cast(u32x4, stage[it]);
                        h0 = u[0]; h1 = u[1]; l0 = u[2]; l1 = u[3];
                    } else {
                        split_pair_h(stage[it][0], stage[it][1], h0, l0, xscale);
                        split_pair_h(stage[it][2], stage[it][3], h1, l1, xscale);
                    }
                    const u32x2 qh = {h0, h1}, ql = {l0, l1};
                    *reinterpret_cast<u32x2*>(dst) = qh;
                    *reinterpret_cast<u32x2*>(dst + 2 * PS * 16) = ql;
                } else {
                    unsigned h0, m0, l0, h1, m1, l1;
                    split_pair(stage[it][0], stage[it][1], h0, m0, l0);
                    split_pair(stage[it][2], stage[it][3], h1, m1, l1);
                    const u32x2 qh = {h0, h1}, qm = {m0, m1}, ql = {l0, l1};
                    *reinterpret_cast<u32x2*>(dst) = qh;
                    *reinterpret_cast<u32x2*>(dst + 2 * PS * 16) = qm;
                    *reinterpret_cast<u32x2*>(dst + 4 * PS * 16) = ql;
                }
            }
        }
    };

    // per-lane LDS byte bases of the voxel fragments: line 0 of this wave, tap (0, 0, 0), plane h; one per pair kind
    const int wm = WN == 1 ? wave : wave / WN, wn = WN == 1 ? 0 : wave % WN;     // position in the wave grid
    const int line0 = wm * LW;
    const int lane_slot = ((line0 / XTY) * XHY + (line0 % XTY)) * XHX + r + (g & 1) * PS;
    const int hi = g >> 1;
    const int xb0 = (lane_slot + hi) * 16, xb1 = (lane_slot + hi * XHX) * 16, xb2 = (lane_slot + hi * XHY * XHX) * 16, xb3 = lane_slot * 16;
    auto xaddr = [&](int s, int j, int pl) {                     // s, j, pl are compile-time after unrolling: the rest folds into the offset field
        const int base = pair_kind(s) == 0 ? xb0 : (pair_kind(s) == 1 ? xb1 : (pair_kind(s) == 2 ? xb2 : xb3));
        return base + (((j >> 2) * XHY + (j & 3)) * XHX + tap_slot(x3s_pair_tap(s, 0)) + 2 * pl * PS) * 16;      // (LW = 8: lines 4 .. 7 are the next z-slab)
    };

    f32x4 acc[LW][NTW];
#pragma unroll
    for (int j = 0; j < LW; ++j)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (wave-uniform) the whole halo box lies inside the volume: no piece of this tile was zero-filled
    const bool interior = x0 >= 1 && y0 >= 1 && z0 >= 1 && x0 + XBX + 1 <= a.W && y0 + XTY + 1 <= a.H && z0 + G::TZ + 1 <= a.D;

    const bf16* wlane = reinterpret_cast<const bf16*>(a.wq) + (long long)ntile * a.nchunks * (NUP * UNIT) + lane * 8;
    // (-DMI355SEG_TUNE probes of a tile's fixed part: MI355SEG_DBG 512 = no K loop, 1024 = no first halo request, 2048 = no output stores)
    if (!SEG_DBG(a, 1024)) load_stage(c0);
    for (int chunk = c0; chunk < (SEG_DBG(a, 512) ? c0 : c1); ++chunk) {
        const bf16* wp = wlane + (long long)chunk * (NUP * UNIT) + wn * (NBW * UNIT);       // unit u of this wave = packed unit (u / NBW) * NBW * WN + wn * NBW + u % NBW
        // weight units in flight ahead of the MFMAs (bf16x6: the 64-channel 4-line tile is register-bound) and voxel fragments
        // requested XD regions ahead.  f16x3 (r4 ablation: weights loaded once per chunk +28-41 % on <4, 2> at one unit = 384 cycles of
        // lead, +13 % on <4, 1> at two; voxel fragments read once +14-17 % at one region = 96 cycles of lead): the two-plane fragments
        // leave the registers for three units and two regions
        // (the prologue variant of the <8, 1> tile runs two units ahead: its table registers would otherwise push the per-tile offsets into scratch)
        constexpr int WD = F16 ? ((PRO && LW == 8 && WN == 1) ? 2 : X3S_WD) : (NBW == 2 && LW == 4 ? 1 : 2);
        constexpr int XD = F16 ? X3S_XD : 1;
        bf16x8_t wf[WD + 1][2][NPL], xf[XD + 1][NPL];
        auto load_w = [&](int u) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) wf[u % (WD + 1)][t2][pl] = *reinterpret_cast<const bf16x8_t*>(wp + ((u / NBW) * (NBW * WN) + u % NBW) * UNIT + (pl * 2 + t2) * 512);
        };
        // the first weight units are requested BEFORE the next chunk's halo: vmcnt retires in order
#pragma unroll
        for (int u = 0; u < WD; ++u) load_w(u);
        // (-DMI355SEG_TUNE timing probes, MI355SEG_DBG: 1 = stage the halo for the first chunk only, 2 = weight fragments loaded for the first
        //  unit of a chunk only, 8 = voxel fragments read for the first (unit, line) only; same MFMAs, garbage results)
        if (!SEG_DBG(a, 1) || chunk == c0) {
        __syncthreads();                                         // every wave is done reading the previous chunk
        if (!SEG_DBG(a, 4) || chunk == c0) {
            using TT = std::true_type; using FF = std::false_type;
            if constexpr (!PRO) write_stage(chunk, FF{}, FF{}, FF{});
            else if (a.pro_act == MI355SEG_ACT_RELU) { if (interior) write_stage(chunk, TT{}, FF{}, TT{}); else write_stage(chunk, TT{}, TT{}, TT{}); }
            else write_stage(chunk, TT{}, TT{}, FF{});
        }
        __syncthreads();
        }
        const bool more = chunk + 1 < c1;
#pragma unroll
        for (int q0 = 0; q0 < XD; ++q0)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) xf[q0][pl] = *reinterpret_cast<const bf16x8_t*>(lds_raw + xaddr((q0 / LW) / NBW, q0 % LW, pl));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int s = u / NBW, nh = u % NBW;
#pragma unroll
            for (int j = 0; j < LW; ++j) {
                const int q = u * LW + j, cur = q % (XD + 1), nxt = (q + XD) % (XD + 1);
                if (j == 0 && u + WD < NU && !SEG_DBG(a, 2)) load_w(u + WD);
                if (q < G::NITER && more && !SEG_DBG(a, 1) && !SEG_DBG(a, 16)) load_piece(chunk + 1, q);
                if (q + XD < NU * LW && !SEG_DBG(a, 8)) {
                    const int u2 = (q + XD) / LW, j2 = (q + XD) % LW;
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) xf[nxt][pl] = *reinterpret_cast<const bf16x8_t*>(lds_raw + xaddr(u2 / NBW, j2, pl));
                }
                if constexpr (F16) {
                    // planes 0 / 1 = h / l; the two cross terms go in first; the two channel tiles alternate
                    constexpr int PW[3] = {1, 0, 0}, PX[3] = {0, 1, 0};
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2)
                            acc[j][nh * 2 + t2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wf[u % (WD + 1)][t2][PW[pr]]),
                                                                                      __builtin_bit_cast(f16x8_t, xf[cur][PX[pr]]), acc[j][nh * 2 + t2], 0, 0, 0);
                } else {
                    // planes 0 / 1 / 2 = h / m / l; the small cross terms go in first; the two channel tiles alternate
                    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int t2 = 0; t2 < 2; ++t2)
                            acc[j][nh * 2 + t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u % (WD + 1)][t2][PW[pr]], xf[cur][PX[pr]], acc[j][nh * 2 + t2], 0, 0, 0);
                }
                // interleave: the next line's voxels (DS, needed first) behind the first MFMAs, then the weights two units ahead
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                if (j == 0 || q < G::NITER) {
#pragma unroll
                    for (int k = 0; k < 2 * NPL + 1; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);               // one scheduling region per (unit, line)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    x3_epilogue<LW, NBW, G::TZ, WN>(a, acc, F16 ? -(sx + sw) : 0, ks, mtile, n, x0, y0, z0, n0, line0, r, g, wave, tid, lds_raw);
}

template <int LW, int NBW, bool F16, int WN = 1>
void launch_x3s(const IgemmArgs& a, int nwg, hipStream_t st) {
    constexpr int LDSB = Geo<LW, F16, WN>::LDS_BYTES;
    static_assert(LDSB >= 8 * 64 * 4, "the statistics epilogue needs 8 x NT floats");
    if constexpr (F16) {
        if (a.pro_al) {                       // the prologue's per-channel table sits behind the halo tile
            SEG_SET_LDS((conv_x3s_kernel<LW, NBW, true, WN, true>), LDSB + X3S_PRO_MAX_CIN * 8);
            hipLaunchKernelGGL((conv_x3s_kernel<LW, NBW, true, WN, true>), dim3(nwg), dim3(256), LDSB + a.nchunks * 16 * 8, st, a);
            return;
        }
    }
    SEG_SET_LDS((conv_x3s_kernel<LW, NBW, F16, WN>), LDSB);
    hipLaunchKernelGGL((conv_x3s_kernel<LW, NBW, F16, WN>), dim3(nwg), dim3(256), LDSB, st, a);
}


// ---------------------------------------------------------------- conv_x3w: the f16x3 form for the layers that fill the chip
// r4 ablation of conv_x3s<.., F16> (profiles/r04_x3s_f16_ablation_probe.log: same MFMAs, one input stream removed at a time): weight
// fragments loaded once per chunk +21-45 %, no re-staging +8-13 %, voxel fragments read once +10-18 %, all three 1.55-1.65x.  With
// three MFMAs per product instead of six the loop is bound by the four waves' re-loads of the same weight fragments through the
// L1 (a fragment fed 6 MFMAs: 43 B/clk/CU of the 64) and by the staging phase between two barriers.  This kernel trades occupancy
// for registers: ONE workgroup per CU, one wave per SIMD (up to 512 registers), tile = 8 (z) x 4 (y) x 16 (x) voxels x 32 NBW channels,
// a wave owns EIGHT x-lines and all channels of the tile -- a weight fragment feeds 12 MFMAs (21 B/clk/CU), NBW = 2: a voxel fragment
// feeds 6 -- and the halo tile is double-buffered in LDS (2 x 68 KB): the next chunk's pieces are requested, split and written
// into the other buffer a few at a time between the MFMAs of the current chunk (a ring of RING pieces instead of the whole
// chunk in registers), one barrier per chunk.  Weight fragments run WL K-steps ahead in a register ring of WR slots; NCH chunks
// per loop iteration make the ring's phase static (NCH * 14 steps = 0 mod WR).  ksplit == 1 only; same weight packing (LAYOUT 2),
// same tile walk, same epilogue as conv_x3s.
template <int NBW>
struct GeoW {
    static constexpr int LW = 8, TZ = 8, HZ = TZ + 2;
    static constexpr int NVOX = XHX * XHY * HZ;                  // 1080
    static constexpr int PS = (NVOX + 15) / 16 * 16;            // 1088
    static constexpr int BUF = 4 * PS * 16;                     // one buffer: planes h / l x channel half
    static constexpr int LDS_BYTES = 2 * BUF;
    static constexpr int NPIECE = NVOX * 4;
    static constexpr int NITER = (NPIECE + 255) / 256;          // 17
    static constexpr int NCH = NBW == 2 ? 1 : 2;                // chunks per loop iteration
    static constexpr int WR = NBW == 2 ? 2 : 4, WL = NBW == 2 ? 1 : 2;      // weight ring slots / K-steps of lead
    static constexpr int RING = 8;                              // halo pieces in flight
    static_assert((NCH * X3S_NPAIR) % WR == 0 && WL < WR, "the weight ring's phase must be static");
};

template <int NBW>
__global__ __launch_bounds__(256, 1) void conv_x3w_kernel(IgemmArgs a) {
    using G = GeoW<NBW>;
    constexpr int LW = G::LW, NT = 32 * NBW, NTW = 2 * NBW, PS = G::PS, NITER = G::NITER, RING = G::RING;
    constexpr int NS = X3S_NPAIR;                                // K-steps per chunk
    constexpr int STEP = NBW * 2048;                             // fp16 elements of one K-step of packed weights: [nh][plane][t2][lane][8]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    int ntile, mtile, n, x0, y0, z0;
    {
        const int total = (int)gridDim.x, bid = blockIdx.x;
        const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
        const int xstart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, xcnt = q8 + (xcd < r8 ? 1 : 0);
        const int local = bid >> 3;
        if (local >= xcnt) return;
        int t = xstart + local;
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
        x0 = txi * XBX; y0 = tyi * XTY; z0 = tzi * G::TZ;
    }
    const int n0 = ntile * NT;
    const int c1 = a.cps;                                        // ksplit == 1: all chunks
    const float* __restrict__ xin = reinterpret_cast<const float*>(a.x);
    const int sx = f16x_scale_exp(*a.amax_x), sw = f16x_scale_exp(*a.amax_w);
    const float xscale = pow2f(sx);

    // ---- halo pieces: piece p = it * 256 + tid = (halo voxel p / 4, four channels p % 4); per-tile byte offsets, buffer loads with
    // a range check that zero-fills the pieces outside the volume (as conv_x3s)
    int voff[NITER];
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
        const int p = it * 256 + tid;
        const int vox = p >> 2, part = p & 3;
        const int hz = vox / (XHY * XHX), rem = vox % (XHY * XHX);
        const int hy = rem / XHX, hx = rem % XHX;
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool ok = (p < G::NPIECE) && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        voff[it] = ok ? (((gz * a.H + gy) * a.W + gx) * a.ldx + part * 4) * 4 : 0x7FFFFFF0;
    }
    const float* xsample = xin + (long long)n * a.D * a.H * a.W * a.ldx;
    const int sample_bytes = a.D * a.H * a.W * a.ldx * 4;
    f32x4 ring[RING];
    auto load_piece = [&](int chunk, int it) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsample + chunk * 16), 0, sample_bytes, 0x00020000);
        ring[it % RING] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[it], 0, X3S_HALO_AUX));
    };
    auto split_pair_h = [&](float x0_, float x1_, unsigned& h2, unsigned& l2) {
        const float s0 = x0_ * xscale, s1 = x1_ * xscale;
        const f16x2_t hh = __builtin_convertvector(f32x2_t{s0, s1}, f16x2_t);
        h2 = __builtin_bit_cast(unsigned, hh);
        const float r0 = __builtin_fmaf((float)hh[0], -1.f, s0), r1 = __builtin_fmaf((float)hh[1], -1.f, s1);
        l2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t));
    };
    unsigned char* const wdst = lds_raw + (((tid & 3) >> 1) * PS + (tid >> 2)) * 16 + (tid & 1) * 8;
    auto write_piece = [&](int buf, int it) {
        if (it * 256 + tid < G::NPIECE) {
            using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
            const f32x4 v = ring[it % RING];
            unsigned h0, l0, h1, l1;
            split_pair_h(v[0], v[1], h0, l0);
            split_pair_h(v[2], v[3], h1, l1);
            unsigned char* dst = wdst + buf * G::BUF + it * 1024;
            *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(dst + 2 * PS * 16) = u32x2{l0, l1};
        }
    };

    // per-lane LDS byte bases of the voxel fragments (line 0 of this wave, plane h, buffer 0), one per pair kind
    const int line0 = wave * LW;
    const int lane_slot = ((line0 / XTY) * XHY + (line0 % XTY)) * XHX + r + (g & 1) * PS;
    const int hi = g >> 1;
    const int xb0 = (lane_slot + hi) * 16, xb1 = (lane_slot + hi * XHX) * 16, xb2 = (lane_slot + hi * XHY * XHX) * 16, xb3 = lane_slot * 16;
    auto xoffs = [](int buf, int s, int j, int pl) {             // compile-time part of a fragment address: folds into the offset field
        return buf * G::BUF + ((((j >> 2) * XHY + (j & 3)) * XHX + tap_slot(x3s_pair_tap(s, 0))) + 2 * pl * PS) * 16;
    };

    f32x4 acc[LW][NTW];
#pragma unroll
    for (int j = 0; j < LW; ++j)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight fragments of linear K-step `lin` (= chunk * 14 + s): fragment (tile t = 2 nh + t2, plane pl) at nh * 2048 + pl * 1024 + t2 * 512
    const _Float16* wlane = reinterpret_cast<const _Float16*>(a.wq) + (long long)ntile * a.nchunks * (NS * STEP) + lane * 8;
    const int nlin = c1 * NS;
    bf16x8_t wf[G::WR][NTW][2];
    auto load_w_frag = [&](int slot, int lin, int f) {         // f = 0 .. 2 NTW - 1: (tile, plane)
        const int t = f >> 1, pl = f & 1;
        wf[slot][t][pl] = *reinterpret_cast<const bf16x8_t*>(wlane + (long long)lin * STEP + (t >> 1) * 2048 + pl * 1024 + (t & 1) * 512);
    };

    // ---- prologue: the first chunk's halo (through the ring, RING pieces at a time), the first WL steps of weights
#pragma unroll
    for (int sl = 0; sl < G::WL; ++sl)
#pragma unroll
        for (int f = 0; f < 2 * NTW; ++f) load_w_frag(sl, sl < nlin ? sl : 0, f);
#pragma unroll
    for (int it0 = 0; it0 < NITER; it0 += RING) {
#pragma unroll
        for (int it = it0; it < it0 + RING && it < NITER; ++it) load_piece(0, it);
#pragma unroll
        for (int it = it0; it < it0 + RING && it < NITER; ++it) write_piece(0, it);
    }
    __syncthreads();

    bf16x8_t xf[2][2];
    for (int chunk = 0; chunk < c1; chunk += G::NCH) {
#pragma unroll
        for (int h = 0; h < G::NCH; ++h) {
            const int cur = G::NCH == 2 ? h : (chunk & 1);      // NCH == 1: the buffer alternates with the (runtime) chunk parity
            const int cbuf = G::NCH == 2 ? h : 0;               // compile-time part of the buffer choice (NCH == 1: added at run time below)
            const int rt = G::NCH == 2 ? 0 : cur * G::BUF;      // run-time buffer offset (bytes)
            const int nbuf = 1 - cur;                            // where the next chunk's halo goes
            const bool more = chunk + h + 1 < c1;
            const int b0 = xb0 + rt, b1 = xb1 + rt, b2 = xb2 + rt, b3 = xb3 + rt;
            const unsigned char* xl = lds_raw;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) xf[0][pl] = *reinterpret_cast<const bf16x8_t*>(xl + b0 + xoffs(cbuf, 0, 0, pl));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int hs = h * NS + s;                       // step index inside the loop iteration: the ring phase
                const int lin = (chunk + h) * NS + s;
#pragma unroll
                for (int j = 0; j < LW; ++j) {
                    const int q = s * LW + j, cb = q & 1, nb = cb ^ 1;
                    // weights WL steps ahead, spread over the lines of the step
                    {
                        constexpr int FPL = (2 * NTW + LW - 1) / LW;               // fragments requested per line
#pragma unroll
                        for (int f = j * FPL; f < (j + 1) * FPL && f < 2 * NTW; ++f) {
                            const int l2 = lin + G::WL;
                            load_w_frag((hs + G::WL) % G::WR, l2 < nlin ? l2 : nlin - 1, f);
                        }
                    }
                    // the next chunk's halo: piece q requested in region q, split and written RING regions later
                    if (q >= RING && q - RING < NITER && more) write_piece(nbuf, q - RING);
                    if (q < NITER && more) load_piece(chunk + h + 1, q);
                    if (q + 1 < NS * LW) {
                        const int s2 = (q + 1) / LW, j2 = (q + 1) % LW;
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
                            xf[nb][pl] = *reinterpret_cast<const bf16x8_t*>(xl + (pair_kind(s2) == 0 ? b0 : (pair_kind(s2) == 1 ? b1 : (pair_kind(s2) == 2 ? b2 : b3))) + xoffs(cbuf, s2, j2, pl));
                    }
                    constexpr int PW[3] = {1, 0, 0}, PX[3] = {0, 1, 0};            // planes 0 / 1 = h / l; the cross terms first
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int t = 0; t < NTW; ++t)
                            acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wf[hs % G::WR][t][PW[pr]]),
                                                                            __builtin_bit_cast(f16x8_t, xf[cb][PX[pr]]), acc[j][t], 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);           // one scheduling region per (K-step, line)
                }
            }
            __syncthreads();                                     // the other buffer is complete, this one is free
        }
    }

    x3_epilogue<LW, NBW, G::TZ>(a, acc, -(sx + sw), 0, mtile, n, x0, y0, z0, n0, line0, r, g, wave, tid, lds_raw);
}

template <int NBW>
void launch_x3w(const IgemmArgs& a, int nwg, hipStream_t st) {
    constexpr int LDSB = GeoW<NBW>::LDS_BYTES;
    SEG_SET_LDS((conv_x3w_kernel<NBW>), LDSB);
    hipLaunchKernelGGL((conv_x3w_kernel<NBW>), dim3(nwg), dim3(256), LDSB, st, a);
}

}  // namespace

// the generic plan's BX = 16 tiles (MB = 2 <-> four lines per wave, MB = 1 <-> two) map one to one onto this kernel
bool x3s_plan_ok(const IgemmPlan& p, const void* x, int ldx, const void* y, int ldy, long long sample_voxels) {
    if (sample_voxels * ldx * 4 >= 0x7FFFFFF0LL) return false;        // the halo loads address one sample through a 32-bit buffer offset
    return p.KS == 3 && p.CK == 16 && p.BX == 16 && (p.MB == 1 || p.MB == 2) && (p.NBW == 1 || p.NBW == 2) && p.WN == 1 && p.TY == 4 &&
           (ldx % 4) == 0 && (ldy % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0;
}

// conv_x3w (f16x3, one 8 x 4 x 16-voxel tile per CU at a time): for layers that cut enough of its tiles to fill the chip a few times
// over with little z padding; NBW = 1 runs two chunks per loop iteration.  Rewrites the M-tile geometry of the plan.
bool x3w_plan(IgemmPlan& p, int N, int D, int H, int W, int Cin, int Cout, int ksplit) {
#ifdef MI355SEG_TUNE
    static const char* off = getenv("MI355SEG_NO_X3W");
    if (off && off[0] == '1') return false;
#endif
    if (ksplit != 1 || p.WN != 1 || p.NBW != 1) return false;      // (NBW = 2: conv_x3w<2> measured at parity with conv_x3s<4, 2>: kept off)
    const int ntz = (D + 7) / 8, nty = (H + 3) / 4, ntx = (W + 15) / 16;
    const long long tiles = (long long)N * ntz * nty * ntx * p.nN;
    if (tiles < 512 || (double)(ntz * 8) / D > 1.15) return false;
    p.MB = 4; p.TZ = 8; p.ntx = ntx; p.nty = nty; p.ntz = ntz; p.nM = N * ntz * nty * ntx;
    return true;
}

// a.amax_x != nullptr selects the f16x3 form (the weights at a.wq are then its two-plane fp16 packing)
void dispatch_x3s(const IgemmPlan& p, const IgemmArgs& a, int nwg, hipStream_t st) {
    if (a.amax_x && p.MB == 4) {
#ifdef X3W_NBW1
        if (p.NBW == 2) launch_x3w<2>(a, nwg, st); else launch_x3w<1>(a, nwg, st);
#else
        if (p.NBW == 2) launch_x3w<2>(a, nwg, st); else launch_x3s<8, 1, true>(a, nwg, st);
#endif
        return;
    }
    if (a.amax_x) {
        // 64-channel tile of four lines per wave-row: the 2 x 2 wave grid (eight lines x 32 channels per wave)
        if (p.MB == 2) { if (p.NBW == 2) launch_x3s<X3S_WN2 ? 8 : 4, X3S_WN2 ? 1 : 2, true, X3S_WN2 ? 2 : 1>(a, nwg, st); else launch_x3s<4, 1, true>(a, nwg, st); }
        else { if (p.NBW == 2) launch_x3s<2, 2, true>(a, nwg, st); else launch_x3s<2, 1, true>(a, nwg, st); }
        return;
    }
    if (p.MB == 2) { if (p.NBW == 2) launch_x3s<4, 2, false>(a, nwg, st); else launch_x3s<4, 1, false>(a, nwg, st); }
    else { if (p.NBW == 2) launch_x3s<2, 2, false>(a, nwg, st); else launch_x3s<2, 1, false>(a, nwg, st); }
}

}  // namespace seg
