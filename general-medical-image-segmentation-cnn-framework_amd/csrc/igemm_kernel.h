// igemm_kernel.h -- the implicit-GEMM convolution kernel (Conv3d k1 / k3 / k5 forward + input gradient, gather mode for
// strided / even kernels, ConvTranspose3d k2 s2) with three arithmetic policies behind ONE kernel body:
//
//   MATH_F32  fp32 tensors, fp32 halo tile in LDS, v_mfma_f32_32x32x2_f32 (exact fp32 products);
//   MATH_X3   fp32 tensors, "bf16x6": every staged value is split once into three bf16 parts (x = h + m + l, 24 mantissa
//             bits) kept as three planes of the LDS tile, the weights are split by the pack kernel, and six
//             v_mfma_f32_32x32x16_bf16 (hh, hm, mh, mm, hl, lh; fp32 accumulate; small terms first) form each product:
//             fp32-level accuracy at 2.7x the fp32 matrix rate;
//   MATH_B16  bf16 tensors in HBM (activations in, activations out), one bf16 plane in LDS, one bf16 MFMA per
//             16-channel k-step, fp32 accumulate, fp32 bias / BatchNorm statistics in the epilogue.
//
//   M = output voxels, N = output channels, K = taps x Cin.
//
// Workgroup = 4 waves (256 threads); each wave owns MB M-blocks of 32 voxels and all NT = 32*NBW output channels of
// the tile, i.e. the tile is (128*MB voxels) x NT.  The input halo tile ((TZ+2h) x (TY+2h) x (BX+2h) voxels x CK input
// channels, NDHWC) is staged in LDS once per CK-channel chunk and re-read by all taps as shifted ds_read_b128 (the shift
// is an immediate offset; the voxel pitch is an odd number of 16-byte slots, so the reads are bank-conflict free).
// Weights are pre-packed so that the B operand of a k-step is ONE coalesced 16-byte global load per lane (L2-resident,
// software-prefetched through a register ring); the next chunk's halo is prefetched into registers before the MFMAs of the
// current chunk and written to LDS after them (issue-early / write-late).  Epilogue: bias add, NDHWC store, optional
// per-channel (sum, M2, n) partials for the BatchNorm that follows (deterministic two-stage).
#pragma once
#include "common.h"
#include <type_traits>

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// Timing-experiment switches (ablations of the main loop, tile-shape overrides) exist only in a -DMI355SEG_TUNE build
// (make TUNE=1); the shipped kernels carry none of them.
#ifdef MI355SEG_TUNE
#define SEG_DBG(a, bit) ((a).dbg & (bit))
#else
#define SEG_DBG(a, bit) 0
#endif

template <int MATH> struct MathTraits;
template <> struct MathTraits<MATH_F32> { using in_t = float; using out_t = float; static constexpr int NP = 1, EPP = 4, LDS_ELEM = 4, KGRAN = 8; };
template <> struct MathTraits<MATH_X3> { using in_t = float; using out_t = float; static constexpr int NP = 3, EPP = 4, LDS_ELEM = 2, KGRAN = 16; };
template <> struct MathTraits<MATH_B16> { using in_t = bf16; using out_t = bf16; static constexpr int NP = 1, EPP = 8, LDS_ELEM = 2, KGRAN = 16; };

// CK = input channels per LDS chunk: 16 for k3 (two workgroups per CU), 8 for fp32 k5 (the 5^3 halo is 4x larger), 64 for
// k1 / ConvTranspose where there is no halo and few MFMAs per chunk otherwise.
// PITCH (in LDS elements) = NP planes of CK channels + 16 bytes of padding -> an odd number of 16-byte slots per voxel.
template <int MATH, int KS, int BX, int MB, int CK, int WN = 1>
struct Tile {
    using MT = MathTraits<MATH>;
    static constexpr int PITCH = MT::NP * CK + 16 / MT::LDS_ELEM;
    static constexpr int HALO = KS / 2;
    static constexpr int NTAP = KS * KS * KS;
    static constexpr int LPB = 32 / BX;           // x-lines per 32-voxel M-block
    static constexpr int LINES = (4 / WN) * MB * LPB;    // x-lines per workgroup tile (WN waves share an M-block row, each on its own N-blocks)
    // bf16 k3 tiles of three or more full-width M-blocks per wave: the wave's blocks are the consecutive y-lines of ONE z-slab
    // (TY = MB), so that the halo line (z + dz, y') it reads once from LDS feeds every (block, dy) pair with block + dy == y'
    // -- MB + 2 fragment reads per (dz, dx) instead of 3 * MB (see the SLIDE main loop)
    static constexpr bool SLIDE = MATH == MATH_B16 && KS == 3 && BX == 32 && MB >= 3;
    // persistent tile walk with the next tile's first halo chunk prefetched behind the last MFMAs of the current one: measured
    // SLOWER on every bf16 layer (32->32 @160x192x160 619 -> 589 TFLOP/s, 64->64 886 -> 779): vmcnt retires in order, so the
    // weight fragments requested after the prefetch wait for it either way, and the tile loop's scalar state spills.  Kept off.
    static constexpr bool PERSIST = false;
    static constexpr int TY = SLIDE ? MB : 4;
    static constexpr int TZ = LINES / TY;
    static constexpr int HX = BX + 2 * HALO, HY = TY + 2 * HALO, HZ = TZ + 2 * HALO;
    static constexpr int NVOX = HX * HY * HZ;
    static constexpr int PPV = CK / MT::EPP;                  // staged pieces (one global load each) per voxel
    static constexpr int NPIECE = NVOX * PPV;
    static constexpr int NITER = (NPIECE + 255) / 256;
    // x-row stride in LDS elements.  A 32-wide M-block reads ONE x-row per ds_read_b128 (32 lanes = 32 consecutive voxels)
    // and the odd voxel pitch alone keeps it conflict-free; the 16- / 8-wide tiles read 2 / 4 rows per instruction, and
    // the lanes of different rows share 16-byte slots unless the next row continues the slot sequence of the previous
    // one, i.e. row stride == BX * voxel pitch (mod 16 slots): the rows of those tiles are padded up to that (at most 15
    // slots per row; r1 PMC: 0.50 / 0.65 conflict cycles per active LDS cycle on the unpadded BX = 16 / 8 tiles, none by
    // construction now -- tools/lds_conflicts.py enumerates the lane groups).
    static constexpr int SLOT = 16 / MT::LDS_ELEM;              // elements per 16-byte slot
    static constexpr int ROW_RAW = HX * PITCH;
    static constexpr int ROW_SLOTS = ROW_RAW / SLOT + ((BX < 32 && HALO > 0) ? ((BX * (PITCH / SLOT) - ROW_RAW / SLOT) % 16 + 16) % 16 : 0);
    static constexpr int ROW = ROW_SLOTS * SLOT;
    static constexpr int LDS_BYTES = HY * HZ * ROW * MT::LDS_ELEM;
    static_assert(LINES % TY == 0, "tile lines must fill whole y-rows");
    static_assert(CK % MT::KGRAN == 0, "chunk must hold whole MFMA k-steps");
    static_assert(((PITCH * MT::LDS_ELEM / 16) & 1) == 1, "voxel pitch must be an odd number of 16-byte slots");
};

// The M space is always the "base grid" (N, D, H, W).  in_mul / out_mul = 2 turn the same kernel into
// ConvTranspose3d k2 s2: forward scatters N-tile (tap, cout-tile) to child voxel 2*v + tap of the
// (2D,2H,2W) output; dgrad gathers K-chunk (tap, cout-chunk) from child voxel 2*v + tap of the input.
// x / wq / y are float or bf16 arrays as the arithmetic policy of the launch says; ldx / ldy count elements.
struct IgemmArgs {
    const void* x; const void* wq; const float* bias; void* y; float* spart;
    int ldx, ldy, N, D, H, W, Cout;
    int ntx, nty, ntz, nN;
    int nchunks;        // total K chunks of CK channels (taps of a ConvT dgrad included)
    int cpt;            // chunks per input tap  (== nchunks when in_mul == 1)
    int nNpt;           // N-tiles per output tap (== nN when out_mul == 1)
    int in_mul, out_mul;
    int nM;             // M-tiles
    int ksplit, cps;    // K-splits and chunks per split (nchunks == ksplit * cps); ksplit > 1: y is an fp32 slab array
    long long split_stride;   // floats between the output slabs of consecutive K-splits
    int dbg;            // -DMI355SEG_TUNE builds only (MI355SEG_DBG): 1 no re-staging, 2 B loaded once per chunk, 4 no stores; else 0 and unread
    // ---- gather / scatter generalisation (strided Conv3d fwd + per-phase dgrad, ConvT with narrow Cout)
    int Di, Hi, Wi;     // extents of the volume x points at   (input voxel = base * in_mul + toff[tap])
    int Do, Ho, Wo;     // extents of the volume y points at   (output voxel = base * out_mul + child + c{z,y,x})
    int cz, cy, cx;     // fixed child offset of a strided-dgrad phase launch
    int flatn;          // != 0: N-tiles cut the flat (child tap, cout) axis, so one 32-column block may span two children
    int by, bz;         // (y, z) tile-block shape of the M-tile walk (divisors of nty, ntz)
    int total;          // virtual tiles = nM * nN * ksplit (the persistent variants walk them with a smaller grid); set by dispatch
    signed char toff[64][4];   // per K-tap input offset (z, y, x); all zero for the stride-1 halo modes
    int act; float slope;      // activation applied after the bias in the epilogue (0 = none): the fused inference forward
    // ---- BatchNorm-backward column sums in the epilogue of an input-gradient launch (conv_x3s.hip): y is then d(activation) of the
    // norm + activation that produced this convolution's input; bnx = that norm's input (same voxels, pitch ldbnx), per channel
    // bn_mean / bn_rstd / bn_gamma / bn_beta; bnpart[mtile][channel] = {sum dz, sum dz * xhat} with dz = y * act'(z)
    const float* bnx; int ldbnx;
    const float* bn_mean; const float* bn_rstd; const float* bn_gamma; const float* bn_beta;
    int bn_act; float bn_slope;
    float* bnpart;
    // ---- f16x3 (conv_x3s.hip): device pointers to upper bounds of max |x| and max |w|; amax_x != nullptr selects that form
    const float* amax_x; const float* amax_w;
    // ---- residual sum in the epilogue (conv_b16s.hip, whole-K launches): y = bf16(bf16(conv + bias) + res), the value of the reference's
    // separate `conv(x) + res` on bf16 tensors (residual_unet3d.py:121,140-168); res has y's geometry at pitch ldres
    const void* res; int ldres;
    // ---- norm + activation PROLOGUE (conv_x3s.hip, f16x3; r5): x is the PRE-NORM tensor of the layer in front and every staged value
    // becomes act(pro_al[c] * x + pro_be[c]) on its way into LDS (the zero padding stays zero) -- conv2 of a double-conv block reads
    // conv1's raw output and the activation between them is never written (unet3d.py:80-101).  amax_x then bounds max |act(...)|.
    const float* pro_al; const float* pro_be; int pro_act; float pro_slope;
    // ---- max |y| (bias included) max-combined into this device scalar from the epilogue (whole-K launches): what bounds the next
    // layer's prologue output
    unsigned* amax_y;
};


// One launch over the (up to 8) output PHASES of a strided input gradient (r6; conv_gather_dgrad_mfma): phase p = the dx voxels with
// (u mod stride) = (cz, cy, cx) is a GEMM of its own -- base grid (D, H, W), its taps (toff rows 8 p .. of the launch's IgemmArgs), its
// K range -- that used to be a launch of its own (eight launches of 60 workgroups each on the deep levels).  Blocks first .. of the
// grid belong to the phase (first is a multiple of 8: the XCD map of a phase starts on XCD 0); wq = the phase's first chunk inside the
// ONE packing of all taps, wstride = chunks per N-tile of that packing.
struct IgemmPhase { const void* wq; int D, H, W, ntx, nty, ntz, by, bz, nchunks, wstride, cz, cy, cx, first; };
struct IgemmPhases { IgemmPhase ph[8]; int n; };

struct TapList { unsigned char t[64]; };
struct IgemmPlan { int KS, CK, BX, MB, NBW, TZ, nM, nN, ntx, nty, ntz, flat, WN, NT, TY; };     // NT = 32 * NBW * WN: tile width in channels

// ---------------------------------------------------------------- the kernel
// One virtual tile = (M-tile, N-tile, K-split) per workgroup; up to two workgroups share a CU and the
// hardware dispatcher staggers them, so one stages its halo while the other issues MFMAs (a persistent
// variant was measured 8-10 % slower on the large layers: co-resident workgroups fall into lockstep).
// With ksplit > 1 (few-tile deep layers) every split writes raw fp32 partial sums to its own slab and a
// tiny second kernel adds them in fixed order.
// Wave grid: WN = 1 -- every wave owns MB M-blocks and all NBW N-blocks of the tile; WN = 2 -- the four waves form a
// 2 (M) x 2 (N) grid, a wave owns MB M-blocks and its own NBW N-blocks (tile = 64*MB voxels x 64*NBW channels): each weight
// fragment a wave loads then feeds MB MFMAs and is loaded by two waves instead of four -- the layout of the bf16 tiles,
// whose single MFMA per k-step would otherwise leave the L1 path saturated by the four waves' re-loads of the same weights.
// MULTI: the launch runs several phases (IgemmPhases); `ph` then overrides the per-phase fields of `a`, `bid` / `total` are the block's
// index inside its phase and the phase's tile count, `tap0` its first row of a.toff
template <int MATH, int KS, int BX, int MB, int NBW, int CK, int WN, bool MULTI>
__device__ __forceinline__ void conv_igemm_body(const IgemmArgs& a, const IgemmPhase& ph, const int bid_in, const int total_in, const int tap0_in) {
    using T = Tile<MATH, KS, BX, MB, CK, WN>;
    using MT = MathTraits<MATH>;
    using in_t = typename MT::in_t;
    using out_t = typename MT::out_t;
    constexpr int PITCH = T::PITCH;
    constexpr int PPV = T::PPV;
    constexpr int NT = 32 * NBW * WN;                         // output channels of the tile
    constexpr int WM = 4 / WN;                                // waves along M
    constexpr int NTAP = T::NTAP;
    constexpr int NP = MT::NP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;
    const int wave_m = wave / WN, wave_n = wave % WN;          // position in the wave grid
    const int nbase = wave_n * NBW;                           // first N-block of this wave

    // XCD-aware block -> tile map: blocks dealt round-robin to the 8 XCDs get contiguous tile ranges,
    // so halo-sharing neighbours and the N-tiles / K-splits of one M-tile share an L2 (bijective).
    // M-tiles are walked in (y, z) blocks of by x bz tiles (x fastest inside a block) so that the ~64 tiles an XCD works
    // on at any moment form a compact brick whose shared halo planes stay in that XCD's L2 (by divides nty; the last
    // z-row of bricks may be shorter than bz)
    // the fields a phase of a MULTI launch overrides (all wave-uniform)
    const void* const awq = MULTI ? ph.wq : a.wq;
    const int aD = MULTI ? ph.D : a.D, aH = MULTI ? ph.H : a.H, aW = MULTI ? ph.W : a.W;
    const int antx = MULTI ? ph.ntx : a.ntx, anty = MULTI ? ph.nty : a.nty, antz = MULTI ? ph.ntz : a.ntz, aby = MULTI ? ph.by : a.by, abz = MULTI ? ph.bz : a.bz;
    const int anchunks = MULTI ? ph.wstride : a.nchunks, acps = MULTI ? ph.nchunks : a.cps;     // (N-tile pitch of the packing | chunks this block runs)
    const int acz = MULTI ? ph.cz : a.cz, acy = MULTI ? ph.cy : a.cy, acx = MULTI ? ph.cx : a.cx;
    const int tap0 = MULTI ? tap0_in : 0;
    struct TC { int ks, ntile, mtile, n, x0, y0, z0; };
    auto decode = [&](int t) {
        TC c;
        c.ks = t % a.ksplit; t /= a.ksplit;
        c.ntile = t % a.nN;
        c.mtile = t / a.nN;
        int mt = c.mtile;
        const int per_n = antx * anty * antz;
        c.n = mt / per_n; mt -= c.n * per_n;
        const int zfull = antz / abz;                         // full z-rows of bricks; a ragged last row holds the remaining slabs
        const int rowtiles = antx * anty * abz;               // tiles per full z-row
        int zrow = mt / rowtiles, bzz = abz;
        if (zrow >= zfull) { zrow = zfull; bzz = antz - zfull * abz; }
        mt -= zrow * rowtiles;
        const int blk = antx * aby * bzz;
        const int b = mt / blk; mt -= b * blk;
        const int txi = mt % antx; mt /= antx;
        const int tyi = b * aby + mt % aby;
        const int tzi = zrow * abz + mt / aby;
        c.x0 = txi * BX; c.y0 = tyi * T::TY; c.z0 = tzi * T::TZ;
        return c;
    };
    // PERSIST (the bf16 tiles): the grid is at most two workgroups per CU and a workgroup walks every S-th tile of its XCD's
    // range, requesting the first halo chunk of its next tile behind the MFMAs of the current tile's last chunk -- a thin
    // layer (Cin = 32: two chunks, 3.5 us of MFMAs per tile) otherwise spends most of a tile's life waiting for the first
    // loads of a freshly launched workgroup.  Not PERSIST: one tile per workgroup, grid = tiles.
    constexpr bool PERSIST = T::PERSIST;
    const int total = PERSIST ? a.total : total_in, bid = bid_in;
    const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7;
    const int xstart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, xcnt = q8 + (xcd < r8 ? 1 : 0);
    const int xslots = ((int)gridDim.x + 7) >> 3;             // workgroups per XCD
    int local = bid >> 3;
    if (local >= xcnt) return;
    TC tc = decode(xstart + local), tn = tc;
    // the k3 / k5 kernels only ever run the plain stride-1 convolution (the gather / scatter / ConvT generalisations are k1
    // plans): their multipliers, tap offsets and child logic fold away at compile time
    constexpr bool PLAIN = KS != 1;
    const int in_mul = PLAIN ? 1 : a.in_mul, out_mul = PLAIN ? 1 : a.out_mul;
    const int Di = PLAIN ? aD : a.Di, Hi = PLAIN ? aH : a.Hi, Wi = PLAIN ? aW : a.Wi;
    const int Do = PLAIN ? aD : a.Do, Ho = PLAIN ? aH : a.Ho, Wo = PLAIN ? aW : a.Wo;
    const in_t* __restrict__ xin = reinterpret_cast<const in_t*>(a.x);

    // per-lane LDS base (LDS elements) of the A fragment for each M-block: lane (i, h) reads voxel i of the block and the
    // h-th half of a k-step's channels (4 floats / 8 bf16 = 16 bytes)
    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = wave_m * MB + mb;
        const int line = m * T::LPB + i / BX, xx = i % BX;
        abase[mb] = ((line / T::TY) * T::HY + (line % T::TY)) * T::ROW + xx * PITCH + (16 / MT::LDS_ELEM) * h;
    }


    // ---- halo staging: global -> registers (issue early) -> LDS (write late)
    using stage_t = typename std::conditional<MATH == MATH_B16, bf16x8_t, f32x4>::type;
    stage_t stage[T::NITER];
    auto load_stage = [&](const TC& c, int chunk) {
#pragma unroll
        for (int it = 0; it < T::NITER; ++it) {
            const int p = it * 256 + tid;
            const int vox = p / PPV, part = p % PPV;
            const int hz = vox / (T::HY * T::HX), rem = vox % (T::HY * T::HX);
            const int hy = rem / T::HX, hx = rem % T::HX;
            const int tapk = PLAIN ? 0 : chunk / a.cpt, cch = PLAIN ? chunk : chunk - tapk * a.cpt;     // input child (ConvT dgrad), else 0
            const int gz = (c.z0 - T::HALO + hz) * in_mul + (PLAIN ? 0 : a.toff[tap0 + tapk][0]);
            const int gy = (c.y0 - T::HALO + hy) * in_mul + (PLAIN ? 0 : a.toff[tap0 + tapk][1]);
            const int gx = (c.x0 - T::HALO + hx) * in_mul + (PLAIN ? 0 : a.toff[tap0 + tapk][2]);
            const bool ok = (p < T::NPIECE) && (unsigned)gz < (unsigned)Di && (unsigned)gy < (unsigned)Hi && (unsigned)gx < (unsigned)Wi;
            stage_t v = {};
            if (ok) {
                const long long off = ((((long long)c.n * Di + gz) * Hi + gy) * Wi + gx) * a.ldx + cch * CK + part * MT::EPP;
                v = *reinterpret_cast<const stage_t*>(xin + off);
            }
            stage[it] = v;
        }
    };
    auto vox_off = [](int vox) { return (vox / T::HX) * T::ROW + (vox % T::HX) * PITCH; };     // halo voxel -> LDS element offset
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < T::NITER; ++it) {
            const int p = it * 256 + tid;
            if (p < T::NPIECE) {
                if constexpr (MATH == MATH_F32) {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(lds_raw) + vox_off(p / PPV) + (p % PPV) * 4) = stage[it];
                } else if constexpr (MATH == MATH_X3) {          // split once per staged value: planes h | m | l of the voxel
                    bf16x4_t qh, qm, ql;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bf16 bh, bm, bl; split3(stage[it][e], bh, bm, bl); qh[e] = bh; qm[e] = bm; ql[e] = bl; }
                    bf16* dst = reinterpret_cast<bf16*>(lds_raw) + vox_off(p / PPV) + (p % PPV) * 4;
                    *reinterpret_cast<bf16x4_t*>(dst) = qh;
                    *reinterpret_cast<bf16x4_t*>(dst + CK) = qm;
                    *reinterpret_cast<bf16x4_t*>(dst + 2 * CK) = ql;
                } else {
                    *reinterpret_cast<bf16x8_t*>(reinterpret_cast<bf16*>(lds_raw) + vox_off(p / PPV) + (p % PPV) * 8) = stage[it];
                }
            }
        }
    };

    load_stage(tc, tc.ks * acps);
    for (;;) {
    const int ks = tc.ks, ntile = tc.ntile, mtile = tc.mtile, n = tc.n, x0 = tc.x0, y0 = tc.y0, z0 = tc.z0;
    const int tapn = PLAIN ? 0 : ntile / a.nNpt;              // output child (ConvT fwd), else 0
    const int n0 = (PLAIN ? ntile : ntile % a.nNpt) * NT;
    const int c0 = ks * acps, c1 = c0 + acps;               // this split's chunk range
    bool has_next = false;
    if constexpr (PERSIST) {
        local += xslots;
        has_next = local < xcnt;
        if (has_next) tn = decode(xstart + local);
    }
    // the chunk after `chunk`: the next one of this tile, else the first one of this workgroup's next tile
    auto prefetch = [&](int chunk) {
        if (chunk + 1 < c1) load_stage(tc, chunk + 1);
        else if (PERSIST && has_next) load_stage(tn, tn.ks * acps);
    };
    f32x16 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mb][nb][v] = 0.f;
    if constexpr (MATH == MATH_F32) {
        constexpr int STEP_FLOATS = 2 * NT * 4;                 // packed weights consumed per (tap, kk) step
        constexpr int CHUNK_FLOATS = NTAP * (CK / 8) * STEP_FLOATS;
        const float* lds = reinterpret_cast<const float*>(lds_raw);
        const float* wlane = reinterpret_cast<const float*>(awq) + (long long)ntile * anchunks * CHUNK_FLOATS + (h * NT + i) * 4;
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
                for (int nb = 0; nb < NBW; ++nb) bq[d][nb] = *reinterpret_cast<const f32x4*>(wp + d * STEP_FLOATS + (nbase + nb) * 128);
            if (!SEG_DBG(a, 1) || chunk == c0) {
                __syncthreads();                 // every wave is done reading the previous chunk
                write_stage();
                __syncthreads();
            }
            if (!SEG_DBG(a, 1)) prefetch(chunk);
#pragma unroll
            for (int tap = 0; tap < NTAP; ++tap) {
                const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
                const int tapoff = (dz * T::HY + dy) * T::ROW + dx * PITCH;
#pragma unroll
                for (int kk = 0; kk < CK / 8; ++kk) {
                    const int step = tap * (CK / 8) + kk;
                    const int cur = step % (PFD + 1), fill = (step + PFD) % (PFD + 1);
                    if (step + PFD < NSTEP && !SEG_DBG(a, 2)) {
#pragma unroll
                        for (int nb = 0; nb < NBW; ++nb)
                            bq[fill][nb] = *reinterpret_cast<const f32x4*>(wp + (step + PFD) * STEP_FLOATS + (nbase + nb) * 128);
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
    } else {
        // bf16 matrix cores: one k-step = 16 channels; lane (i, h) holds channels 8h .. 8h+7 of its voxel / output column.
        // wq[nt][chunk][tap][kstep][plane][h][j][8]
        constexpr int KSTEPS = CK / 16;
        constexpr int PLANE = 2 * NT * 8;                        // bf16 elements of one plane of one k-step
        constexpr int STEP = NP * PLANE;
        constexpr int NSTEP = NTAP * KSTEPS;
        constexpr int CHUNK = NSTEP * STEP;
        const bf16* lds = reinterpret_cast<const bf16*>(lds_raw);
        const bf16* wlane = reinterpret_cast<const bf16*>(awq) + (long long)ntile * anchunks * CHUNK + (h * NT + i) * 8;
        if constexpr (T::SLIDE) {
            // bf16 k3, sliding window along y.  At one MFMA per (voxel fragment, weight fragment) pair the plain loop reads a
            // 1 KB fragment from LDS per MFMA -- exactly the LDS bandwidth of a CU at full MFMA rate, so LDS, not the matrix
            // core, set the pace (0.42 MFMA-busy).  Here the wave walks the MB + 2 halo lines of its z-slab once per (dz, dx):
            // line y' feeds the blocks y' - dy, dy = 0..2, against the three weight fragments of that (dz, dx) column.
            constexpr int NG = 9 * KSTEPS;                          // (dz, dx, k-step) groups
            constexpr int NL = MB + 2;                              // halo lines per group
            constexpr int NB3 = 3 * NBW;                            // weight fragments per group
            constexpr int BPL = (NB3 + NL - 1) / NL;                // ... requested per line block
            constexpr int AD = 3;                                   // fragment reads in flight
            constexpr int BD = MB * NBW >= 6 ? 1 : 2;               // weight groups in flight (register-bound on the deep tiles)
            const int aslab = (wave_m * T::HY) * T::ROW + i * PITCH + 8 * h;
            auto a_off = [&](int q) {
                const int g = q / NL, hy = q % NL, gk = g / KSTEPS, kk = g % KSTEPS;
                return aslab + ((gk / 3) * T::HY + hy) * T::ROW + (gk % 3) * PITCH + kk * 16;
            };
            auto b_ptr = [&](const bf16* wp, int g, int f) {        // fragment f = (dy, nb) of group g = (dz, dx, kk)
                const int gk = g / KSTEPS, kk = g % KSTEPS, dy = f / NBW, nb = f % NBW;
                const int tap = (gk / 3) * 9 + dy * 3 + gk % 3;
                return wp + (tap * KSTEPS + kk) * STEP + (nbase + nb) * 256;
            };
            for (int chunk = c0; chunk < c1; ++chunk) {
                const bf16* wp = wlane + (long long)chunk * CHUNK;
                bf16x8_t bq[BD + 1][NB3], av[AD + 1];
#pragma unroll
                for (int d = 0; d < BD; ++d)
#pragma unroll
                    for (int f = 0; f < NB3; ++f) bq[d][f] = *reinterpret_cast<const bf16x8_t*>(b_ptr(wp, d, f));
                __syncthreads();
                write_stage();
                __syncthreads();
                prefetch(chunk);
#pragma unroll
                for (int q = 0; q < AD; ++q) av[q] = *reinterpret_cast<const bf16x8_t*>(lds + a_off(q));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
#pragma unroll
                    for (int hy = 0; hy < NL; ++hy) {
                        const int q = g * NL + hy;
                        if (q + AD < NG * NL) av[(q + AD) % (AD + 1)] = *reinterpret_cast<const bf16x8_t*>(lds + a_off(q + AD));
                        if (g + BD < NG) {
#pragma unroll
                            for (int f = hy * BPL; f < (hy + 1) * BPL && f < NB3; ++f)
                                bq[(g + BD) % (BD + 1)][f] = *reinterpret_cast<const bf16x8_t*>(b_ptr(wp, g + BD, f));
                        }
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int mb = hy - dy;
                            if (mb >= 0 && mb < MB) {
#pragma unroll
                                for (int nb = 0; nb < NBW; ++nb)
                                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[q % (AD + 1)], bq[g % (BD + 1)][dy * NBW + nb], acc[mb][nb], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, BPL, 0);
                        __builtin_amdgcn_sched_barrier(0);          // one scheduling region per halo line
                    }
                }
            }
        } else if constexpr (NP == 3 && NSTEP > 4) {
            // bf16x6: 6 * NBW MFMAs per (k-step, M-block) leave room to software-pipeline by hand -- the weights of step s+1
            // and the voxels of the next M-block are requested while the MFMAs of the current one issue, and the scheduling
            // groups pin that interleave (left alone the compiler sinks the loads next to their first use to save registers,
            // which exposed an LDS or L2 latency every few MFMAs: 0.60 MFMA-busy before).
            for (int chunk = c0; chunk < c1; ++chunk) {
                const bf16* wp = wlane + (long long)chunk * CHUNK;
                bf16x8_t bq[2][NBW][3], av[2][3];
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) bq[0][nb][pl] = *reinterpret_cast<const bf16x8_t*>(wp + pl * PLANE + (nbase + nb) * 256);
                __syncthreads();
                write_stage();
                __syncthreads();
                prefetch(chunk);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) av[0][pl] = *reinterpret_cast<const bf16x8_t*>(lds + abase[0] + pl * CK);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int step = 0; step < NSTEP; ++step) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        const int q = step * MB + mb, cur = q & 1, nxt = cur ^ 1;
                        if (mb == 0 && step + 1 < NSTEP) {
#pragma unroll
                            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                                for (int pl = 0; pl < 3; ++pl)
                                    bq[(step + 1) & 1][nb][pl] = *reinterpret_cast<const bf16x8_t*>(wp + (step + 1) * STEP + pl * PLANE + (nbase + nb) * 256);
                        }
                        if (q + 1 < NSTEP * MB) {
                            const int ns = (q + 1) / MB, nm = (q + 1) % MB, tap = ns / KSTEPS, kk = ns % KSTEPS;
                            const int tapoff = ((tap / (KS * KS)) * T::HY + (tap / KS) % KS) * T::ROW + (tap % KS) * PITCH;
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) av[nxt][pl] = *reinterpret_cast<const bf16x8_t*>(lds + abase[nm] + tapoff + pl * CK + kk * 16);
                        }
                        // planes 0 / 1 / 2 = h / m / l; the small cross terms go in first; N-blocks alternate so that no MFMA
                        // waits on the accumulator of the one before it
                        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                            for (int nb = 0; nb < NBW; ++nb)
                                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[cur][PA[pr]], bq[step & 1][nb][PB[pr]], acc[mb][nb], 0, 0, 0);
                        // interleave: the next block's voxels (DS, needed first) behind the first MFMAs, then the next step's weights
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        }
                        if (mb == 0) {
#pragma unroll
                            for (int g = 0; g < 3 * NBW; ++g) {
                                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);      // one scheduling region per (k-step, M-block)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
        for (int chunk = c0; chunk < c1; ++chunk) {
            const bf16* wp = wlane + (long long)chunk * CHUNK;
            constexpr int PFD = NSTEP > 4 ? ((NP == 3 || MB >= 4) ? 2 : 4) : 1;      // ring depth; the deep-tile variants are register-bound
            bf16x8_t bq[PFD + 1][NBW][NP];
#pragma unroll
            for (int d = 0; d < PFD; ++d)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) bq[d][nb][pl] = *reinterpret_cast<const bf16x8_t*>(wp + d * STEP + pl * PLANE + (nbase + nb) * 256);
            __syncthreads();                 // every wave is done reading the previous chunk
            write_stage();
            __syncthreads();
            prefetch(chunk);
#pragma unroll
            for (int tap = 0; tap < NTAP; ++tap) {
                const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
                const int tapoff = (dz * T::HY + dy) * T::ROW + dx * PITCH;
#pragma unroll
                for (int kk = 0; kk < KSTEPS; ++kk) {
                    const int step = tap * KSTEPS + kk;
                    const int cur = step % (PFD + 1), fill = (step + PFD) % (PFD + 1);
                    if (step + PFD < NSTEP) {
#pragma unroll
                        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                            for (int pl = 0; pl < NP; ++pl)
                                bq[fill][nb][pl] = *reinterpret_cast<const bf16x8_t*>(wp + (step + PFD) * STEP + pl * PLANE + (nbase + nb) * 256);
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) {
                        bf16x8_t av[NP];
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) av[pl] = *reinterpret_cast<const bf16x8_t*>(lds + abase[mb] + tapoff + pl * CK + kk * 16);
#pragma unroll
                        for (int nb = 0; nb < NBW; ++nb) {
                            f32x16 c = acc[mb][nb];
                            if constexpr (NP == 3) {            // planes 0 / 1 / 2 = h / m / l; the small cross terms go in first
                                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2], bq[cur][nb][0], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bq[cur][nb][2], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bq[cur][nb][1], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bq[cur][nb][0], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bq[cur][nb][1], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bq[cur][nb][0], c, 0, 0, 0);
                            } else {
                                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bq[cur][nb][0], c, 0, 0, 0);
                            }
                            acc[mb][nb] = c;
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue: bias, store, optional BatchNorm partial statistics
    float* yslab = reinterpret_cast<float*>(a.y) + (long long)ks * a.split_stride;     // ksplit > 1: raw fp32 partial sums
    out_t* yout = reinterpret_cast<out_t*>(a.y);
    const bool slab = a.ksplit > 1;
    float ssum[NBW];
    // (r6) two instantiations of the store loop -- ACT: the fused inference forward's activation (a per-element switch with exp / division
    // branches, 16 MB NBW inlined copies of it) stays out of the training launches' instruction stream
    auto store_tile = [&](auto ACTc) {
        constexpr bool ACT = decltype(ACTc)::value;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            int col = n0 + (nbase + nb) * 32 + i, child = tapn;
            if (!PLAIN && a.flatn) { const int nf = ntile * NT + (nbase + nb) * 32 + i; child = nf / a.Cout; col = nf - child * a.Cout; }   // per-lane child
            const int oz = PLAIN ? 0 : ((child >> 2) & 1) + acz, oy = PLAIN ? 0 : ((child >> 1) & 1) + acy, ox = PLAIN ? 0 : (child & 1) + acx;
            const float bv = a.bias ? a.bias[col] : 0.f;
            float s1 = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = wave_m * MB + mb;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int r = (v & 3) + 8 * (v >> 2) + 4 * h;      // row of the 32x32 tile held in register v
                    const int line = m * T::LPB + r / BX, xx = r % BX;
                    const int gz = (z0 + line / T::TY) * out_mul + oz;
                    const int gy = (y0 + line % T::TY) * out_mul + oy;
                    const int gx = (x0 + xx) * out_mul + ox;
                    float val = acc[mb][nb][v] + bv;
                    if constexpr (ACT) val = act_apply(val, a.act, a.slope);
                    // partial tiles (extents that are not tile multiples): rows outside the volume are dropped
                    const bool inside = (z0 + line / T::TY) < aD && (y0 + line % T::TY) < aH && (x0 + xx) < aW &&
                                        gz < Do && gy < Ho && gx < Wo;
#ifdef MI355SEG_TUNE
                    if (inside && (!(a.dbg & 4) || val == 12345.678f))
#else
                    if (inside)
#endif
                    {
                        const long long off = ((((long long)n * Do + gz) * Ho + gy) * Wo + gx) * a.ldy + col;
                        if constexpr (MATH == MATH_B16) {
                            if (slab) yslab[off] = val; else yout[off] = (out_t)val;
                        } else {
                            yslab[off] = val;                           // ks == 0 and split_stride == 0 when there is one split
                        }
                    }
                    if (inside) s1 += val;
                }
            }
            ssum[nb] = s1;
        }
    };
    if (a.act) store_tile(std::true_type{}); else store_tile(std::false_type{});
    if (a.spart) {
        // BatchNorm batch statistics of this tile, cancellation-free: per channel the tile sum, then the tile
        // mean, then M2 = sum (y - tile_mean)^2 from the accumulators still in registers; the second stage
        // combines (n, sum, M2) of all tiles in fp64 (Chan et al.).  spart[mtile][c] = {sum, M2, n}.
        float* lds = reinterpret_cast<float*>(lds_raw);
        __syncthreads();                 // LDS halo no longer needed
        float cnt = 0.f;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            float s1 = ssum[nb] + __shfl_xor(ssum[nb], 32, 64);
            if (h == 0) lds[wave_m * NT + (nbase + nb) * 32 + i] = s1;
        }
        {   // valid rows of this tile (same for every channel)
            const int vz = min(T::TZ, aD - z0), vy = min(T::TY, aH - y0), vx = min(BX, aW - x0);
            cnt = (float)(vz * vy * vx);
        }
        __syncthreads();
        float tmean[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const int c = (nbase + nb) * 32 + i;
            float ts = 0.f;
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) ts += lds[wm * NT + c];
            tmean[nb] = ts / cnt;
        }
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const float bv = a.bias ? a.bias[n0 + (nbase + nb) * 32 + i] : 0.f;
            float m2 = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = wave_m * MB + mb;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
                    const int line = m * T::LPB + r / BX, xx = r % BX;
                    const bool inside = (z0 + line / T::TY) < aD && (y0 + line % T::TY) < aH && (x0 + xx) < aW;
                    const float d = acc[mb][nb][v] + bv - tmean[nb];
                    if (inside) m2 += d * d;
                }
            }
            m2 += __shfl_xor(m2, 32, 64);
            if (h == 0) lds[WM * NT + wave_m * NT + (nbase + nb) * 32 + i] = m2;
        }
        __syncthreads();
        if (tid < NT) {
            float s1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int wm = 0; wm < WM; ++wm) { s1 += lds[wm * NT + tid]; m2 += lds[(WM + wm) * NT + tid]; }
            float* dst = a.spart + ((long long)mtile * a.Cout + n0 + tid) * 3;
            dst[0] = s1; dst[1] = m2; dst[2] = cnt;
        }
    }
    if (!has_next) break;
    tc = tn;
    }
}

template <int MATH, int KS, int BX, int MB, int NBW, int CK, int WN = 1>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(IgemmArgs a) {
    conv_igemm_body<MATH, KS, BX, MB, NBW, CK, WN, false>(a, IgemmPhase{}, (int)blockIdx.x, (int)gridDim.x, 0);
}

// the phases of a strided input gradient in one launch (k1 gather plans only): block -> phase by the phases' first blocks
template <int MATH, int KS, int BX, int MB, int NBW, int CK>
__global__ __launch_bounds__(256, 2) void conv_igemm_phases_kernel(IgemmArgs a, IgemmPhases pt) {
    const int bid = (int)blockIdx.x;
    int p = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) if (k < pt.n && bid >= pt.ph[k].first) p = k;
    const int end = p + 1 < pt.n ? pt.ph[p + 1].first : (int)gridDim.x;
    const int tiles = pt.ph[p].ntx * pt.ph[p].nty * pt.ph[p].ntz * a.N * a.nN;      // (the phase's blocks past its tiles leave in the XCD map)
    conv_igemm_body<MATH, KS, BX, MB, NBW, CK, 1, true>(a, pt.ph[p], bid - pt.ph[p].first, min(tiles, end - pt.ph[p].first), 8 * p);
}

template <int MATH, int KS, int BX, int MB, int NBW, int CK>
static void launch_igemm_phases(const IgemmArgs& a, const IgemmPhases& pt, int nwg, hipStream_t st) {
    using T = Tile<MATH, KS, BX, MB, CK, 1>;
    static_assert(KS == 1 && !T::PERSIST, "phase launches are gather plans");
    constexpr int LDSB = T::LDS_BYTES < 8 * 64 * 4 ? 8 * 64 * 4 : T::LDS_BYTES;
    SEG_SET_LDS((conv_igemm_phases_kernel<MATH, KS, BX, MB, NBW, CK>), LDSB);
    hipLaunchKernelGGL((conv_igemm_phases_kernel<MATH, KS, BX, MB, NBW, CK>), dim3(nwg), dim3(256), LDSB, st, a, pt);
}

template <int MATH, int KS, int BX, int MB, int NBW, int CK, int WN = 1>
static void launch_igemm(const IgemmArgs& a, int nwg, hipStream_t st) {
    using T = Tile<MATH, KS, BX, MB, CK, WN>;
    constexpr int LDSB = T::LDS_BYTES < 8 * 64 * 4 ? 8 * 64 * 4 : T::LDS_BYTES;      // the statistics epilogue needs 8 x NT floats
    SEG_SET_LDS((conv_igemm_kernel<MATH, KS, BX, MB, NBW, CK, WN>), LDSB);
    const int grid = (T::PERSIST && nwg > 512) ? 512 : nwg;   // persistent variants: two workgroups per CU walk the tiles
    hipLaunchKernelGGL((conv_igemm_kernel<MATH, KS, BX, MB, NBW, CK, WN>), dim3(grid), dim3(256), LDSB, st, a);
}

// K-split factor for layers with too few tiles to fill 2 x 256 workgroup slots (every split writes an fp32 slab, a second
// kernel adds them in fixed order)
inline int pick_ksplit(int tiles, int nchunks) {
    int best = 1;
    for (int k = 2; k <= 16; k *= 2) {
        if (nchunks % k || nchunks / k < 2) break;
        if (tiles * (k / 2) >= 512) break;
        best = k;
    }
    return best;
}

// conv_x3s.hip: the bf16x6 k3 kernel on v_mfma_f32_16x16x32_bf16 (fp32 tensors, BX = 16 tiles of the plan above).
// ---- per-tile byte offsets of a thread's halo pieces, formed incrementally (r6; conv_x3s.hip, conv_b16s.hip).
// Piece p = it * 256 + tid is (halo voxel p / PPV, 16-byte part p % PPV) of a HZ x HY x HX halo box; consecutive pieces of a thread are
// STEP = 256 / PPV halo voxels apart = (DZ planes + DY rows + DX voxels), so (hx, hy, hz) and the byte offset advance by constants with two
// carries -- additions and selects only.  The direct form (two divisions by constants and a three-level multiply per piece) compiled to
// 34 v_mad_u64_u32 + 17 v_mul_lo_u32 + 80 16-/24-bit multiplies per thread ahead of the tile's first load: ~1.2 us of a 17-us tile.
// voff[it] = byte offset inside the sample (bytes_x per voxel step), or 0x7FFFFFF0 (past the descriptor's range: reads as zeros) for a
// piece outside the volume / past the tile's last piece.  (ox, oy, oz) = the box's first voxel, may be -HALO.
template <int NITER, int NPIECE, int PPV, int HX, int HY, int HZ>
__device__ __forceinline__ void halo_piece_offsets(int (&voff)[NITER], int tid, int ox, int oy, int oz, int D, int H, int W, int bytes_x) {
    constexpr int STEP = 256 / PPV, DZ = STEP / (HY * HX), DY = (STEP % (HY * HX)) / HX, DX = STEP % HX;
    static_assert(256 % PPV == 0 && STEP < 2 * HY * HX, "piece walk: one carry per axis");
    const int sx = bytes_x, sy = W * sx, sz = H * sy;                             // wave-uniform
    const int c1 = DX * sx + DY * sy + DZ * sz, c2 = sy - HX * sx, c3 = sz - HY * sy;
    // valid halo coordinates of this tile: [l, l + n) per axis (wave-uniform)
    const int lz = oz < 0 ? -oz : 0, ly = oy < 0 ? -oy : 0, lx = ox < 0 ? -ox : 0;
    const unsigned nz = (unsigned)(min(HZ, D - oz) - lz), ny = (unsigned)(min(HY, H - oy) - ly), nx = (unsigned)(min(HX, W - ox) - lx);
    const int v0 = tid / PPV;                                                    // < 256 / PPV <= HY * HX * 2
    int hz = v0 / (HY * HX);
    const int rem = v0 - hz * (HY * HX);
    int hy = rem / HX, hx = rem - hy * HX;
    int off = ((oz * H + oy) * W + ox) * sx + hz * sz + hy * sy + hx * sx + (tid % PPV) * 16;
    auto walk = [&](auto BORDER) {
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            bool ok = true;
            if constexpr (decltype(BORDER)::value) ok = (unsigned)(hz - lz) < nz && (unsigned)(hy - ly) < ny && (unsigned)(hx - lx) < nx;
            if ((it + 1) * 256 > NPIECE) ok = ok && (it * 256 + tid < NPIECE);
            voff[it] = ok ? off : 0x7FFFFFF0;
            hx += DX; off += c1;
            const bool cx = hx >= HX;
            hx -= cx ? HX : 0; hy += cx ? DY + 1 : DY; off += cx ? c2 : 0;
            const bool cy = hy >= HY;
            hy -= cy ? HY : 0; off += cy ? c3 : 0;
            if constexpr (decltype(BORDER)::value) hz += cy ? DZ + 1 : DZ;
        }
    };
    // (wave-uniform) the whole halo box lies inside the volume: no piece of this tile is zero-filled, no bounds to test
    if (ox >= 0 && oy >= 0 && oz >= 0 && ox + HX <= W && oy + HY <= H && oz + HZ <= D) walk(std::false_type{});
    else walk(std::true_type{});
}

// K-step s of a 16-channel chunk contracts the tap pair (x3s_pair_tap(s, 0), x3s_pair_tap(s, 1)); tap 27 = zero weights.
constexpr int X3S_NPAIR = 14;
__host__ __device__ constexpr int x3s_pair_tap(int s, int which) {
    if (s < 9) return (s / 3) * 9 + (s % 3) * 3 + which;           // (dz, dy, dx = 0 | 1)
    if (s < 12) return (s - 9) * 9 + which * 3 + 2;                 // (dz, dy = 0 | 1, dx = 2)
    if (s == 12) return which * 9 + 8;                              // (dz = 0 | 1, dy = 2, dx = 2)
    return which == 0 ? 26 : 27;                                    // (2, 2, 2) + zero
}
bool x3s_plan_ok(const IgemmPlan& p, const void* x, int ldx, const void* y, int ldy, long long sample_voxels);
void dispatch_x3s(const IgemmPlan& p, const IgemmArgs& a, int nwg, hipStream_t st);
bool x3w_plan(IgemmPlan& p, int N, int D, int H, int W, int Cin, int Cout, int ksplit);

// conv_b16s.hip: bf16 tensors, k3 / k5 stride 1, on v_mfma_f32_16x16x32_bf16 (eight x-lines of 16 voxels x 32 channels per wave).
// K-step s of a 16-channel chunk contracts taps 2s and 2s + 1 (the odd last tap pairs with zero weights).
struct B16sPlan { int KS, WMG, NT, TZ, ntx, nty, ntz, nM, nN, nsteps, ksplit; };
bool b16s_geom(int KS, int N, int D, int H, int W, int Cin, int Cout, B16sPlan* p);
bool b16s_plan(int KS, int N, int D, int H, int W, int Cin, int Cout, const void* x, int ldx, const void* y, int ldy, B16sPlan* p);
size_t b16s_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k);
void dispatch_b16s(const B16sPlan& p, const IgemmArgs& a, int nwg, hipStream_t st);
void set_wgrad_wide(int mode);
int get_wgrad_wide();
void set_b16_tiles(int mode);
int get_b16_tiles();

// conv_igemm_lowp.hip: the MATH_X3 / MATH_B16 instantiations (their own translation unit: they compile in parallel)
void dispatch_igemm_lowp(int math, const IgemmPlan& p, const IgemmArgs& a, int nwg, hipStream_t st);
bool dispatch_igemm_phases_lowp(int math, const IgemmPlan& p, const IgemmArgs& a, const IgemmPhases& pt, int nwg, hipStream_t st);     // false: no such instantiation
bool igemm_lowp_has(int math, int KS, int CK, int BX, int MB);

}  // namespace seg
