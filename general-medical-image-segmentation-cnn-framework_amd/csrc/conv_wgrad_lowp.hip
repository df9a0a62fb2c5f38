// conv_wgrad_lowp.hip -- weight gradient of Conv3d k3 s1 p1 / k5 s1 p2 on the bf16 matrix cores.
//
//   dW[tap][ci][co] = sum over voxels v of  x[v + tap][ci] * dy[v][co]
//
// i.e. one GEMM per tap with M = Cin, N = Cout and K = all output voxels, on v_mfma_f32_16x16x32_bf16 (2 x 2 MFMA tiles per
// 32 x 32 channel block and 32-voxel k-step; the chip holds a higher clock under this shape than under 32x32x16, see
// conv_x3s.hip) with fp32 accumulation.  Two arithmetic policies (common.h):
//   MATH_X3  fp32 tensors ("bf16x6"): x and dy are split once, while they are staged, into three bf16 planes each
//            (v = h + m + l, 24 mantissa bits) and six MFMAs (lh, hl, mm, mh, hm, hh) form each product -- fp32-level
//            accuracy at 2.7x the fp32 matrix rate;
//            NP = 2 ("f16x3", x3_f16()): the two-piece fp16 split of conv_x3s.hip -- v 2^s = h + l under per-tensor power-of-two
//            scales from the tensors' maxima, three v_mfma_f32_16x16x32_f16 (lh, hl, hh) per product, slab scaled back on the way out;
//   MATH_B16 bf16 tensors: one plane, one MFMA per k-step.
//
// K runs over voxels, but NDHWC keeps the CHANNELS of a voxel contiguous, so both MFMA operands are k-strided in memory.
// The tiles are therefore stored in LDS exactly as they arrive ([voxel][32 channels], 64-byte rows, planes interleaved
// per voxel) and read with gfx950's transposing LDS load ds_read_b64_tr_b16: per 16-lane group it fetches a block of
// 4 voxels x 16 channels and hands lane i the 4 voxels of channel i -- two of them are the 8 consecutive k of one lane's
// MFMA fragment.  Row stride 64 * NP bytes puts the 4 rows of a block on disjoint 16-bank spans: conflict-free.
//
// A workgroup (8 waves, two per SIMD) owns one 32(ci) x 32(co) block pair and a strip of spatial tiles; its 27 (k5: the
// 25 of one dz plane) tap-tiles are dealt to the eight waves, so the whole slab stays in accumulators while the workgroup
// walks its strip; the dy fragment of a k-step is shared by the wave's tap-tiles.  The next tile is prefetched into
// registers during the MFMAs (issue early / write late).  Each workgroup writes its slab once; the fixed-order second
// stage (wgrad_reduce) sums the strips and emits the PyTorch (Cout,Cin,k,k,k) layout -> bitwise reproducible, no atomics.
#include "common.h"
#include "internal.h"
#include <initializer_list>
#include <type_traits>
#include <stdlib.h>

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int LW_WAVES = 8;
constexpr int LW_THREADS = LW_WAVES * 64;
constexpr int LW_TPW = 4;                    // tap-tiles per wave (27 or 25 taps over 8 waves; the last waves own one fewer)

// S = stride of the convolution (1, or 2 for the k3 s2 p1 down-convolutions of the Residual U-Net, residual_unet3d.py:30-60):
// the tile is cut in OUTPUT space, its input halo spans (B - 1) * S + KS voxels per axis, and since every lane of a
// transposing read supplies its own row address the S-strided voxels of a k-step need no gather pass.
template <int BX, int KS, int NP, int S = 1>
struct LTile {
    static constexpr int HALO = KS / 2;
    static constexpr int NTAPS = KS == 3 ? 27 : KS * KS;          // taps per workgroup (k5: one dz plane)
    static constexpr int PLANES = KS == 3 ? 1 : KS;
    static constexpr int VOX = (NP >= 2 || S == 2) ? 128 : 256;   // output voxels per tile (split planes per operand / strided halo: half the tile)
    static constexpr int TY = BX == 8 ? 8 : 4;
    static constexpr int LINES = VOX / BX;
    static constexpr int TZ = LINES / TY;
    static constexpr int HX = (BX - 1) * S + KS, HY = (TY - 1) * S + KS, HZ = KS == 3 ? (TZ - 1) * S + 3 : TZ;
    static_assert(S == 1 || KS == 3, "strided tiles are built for k3");
    static constexpr int NVOX = HX * HY * HZ;
    // bytes per voxel row: NP planes of 32 bf16 channels + 32 bytes of padding.  A transposing read serves 32 lanes per LDS
    // cycle = eight voxel rows x 32 bytes; with the padding, row pitch / 32 is odd, so ANY eight consecutive rows fall on the
    // eight distinct 32-byte bank spans (unpadded, rows four apart share one: 0.44 conflict cycles per LDS cycle, r3 PMC)
    static constexpr int ROW = 64 * NP + (S == 1 ? 32 : 0);     // (the strided halo is too large to pad and reads every second row anyway)
    static constexpr int KSTEPS = VOX / 32;                       // 32-voxel k-steps
    static constexpr int X_BYTES = NVOX * ROW, D_BYTES = VOX * ROW;
    static constexpr int LDS_BYTES = X_BYTES + D_BYTES;
    static_assert(TZ >= 1 && LINES % TY == 0, "tile lines must fill whole y-rows");
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the LDS");
};

struct LWgradArgs {
    const void* x; const void* dy; float* part;
    int ldx, lddy, N, D, H, W, Cin, Cout;      // D, H, W: input (x) extents
    int Do, Ho, Wo;                            // output (dy) extents
    int ntx, nty, ntz, ntiles, nstrips, npairs, ncob, ntaps_total;
    const float* amax_x; const float* amax_dy;     // NP = 2: device scalars >= max |x|, max |dy|
    // norm + activation prologue of the x operand (f16x3, r5): x is the pre-norm tensor of the layer in front, the staged value is
    // act(pro_al[c] x + pro_be[c]) (zero padding stays zero); amax_x then bounds the prologue's output (internal.h, ConvPro)
    const float* pro_al; const float* pro_be; int pro_act; float pro_slope;
    int dbg;                                       // -DMI355SEG_TUNE timing probes (MI355SEG_DBG): 8 no tile loads after the first, 16 no split / LDS writes after the
                                                   // first, 64 every tile loads the strip's FIRST tile again (same instructions, cache-resident data)
};
#ifdef MI355SEG_TUNE
#define LW_DBG(a, bit) ((a).dbg & (bit))
#else
#define LW_DBG(a, bit) 0
#endif

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* lds, int off0, int off1) {
    // two transposing reads = the 8 consecutive k (voxels) of this lane's channel
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off1));
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

// PRO (r5): the norm + activation prologue of the x operand (LWgradArgs::pro_al) -- its own instantiation: these kernels sit at their
// register limits and the plain launches must keep their allocation
template <int BX, int KS, int NP, typename IN_T, int S = 1, bool PRO = false>
__global__ __launch_bounds__(LW_THREADS, 2) void conv_wgrad_lowp_kernel(LWgradArgs a) {
    using T = LTile<BX, KS, NP, S>;
    constexpr int EPP = std::is_same<IN_T, float>::value ? 4 : 8;            // elements per staged 16-byte piece
    constexpr int PPV = 32 / EPP;                                             // pieces per voxel (32 channels)
    constexpr int XPIECES = T::NVOX * PPV, DPIECES = T::VOX * PPV;
    constexpr int XITER = (XPIECES + LW_THREADS - 1) / LW_THREADS, DITER = DPIECES / LW_THREADS;
    static_assert(DPIECES % LW_THREADS == 0, "dy tile must split evenly over the workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xs = lds;
    unsigned char* ds = lds + T::X_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int plane = t % T::PLANES, tp = t / T::PLANES;          // k5: which dz plane of taps this workgroup owns
    const int pair = tp % a.npairs, strip = tp / a.npairs;
    const int cib = pair / a.ncob, cob = pair % a.ncob;
    const int ci0 = cib * 32, co0 = cob * 32;
    const IN_T* __restrict__ xin = reinterpret_cast<const IN_T*>(a.x);
    const IN_T* __restrict__ din = reinterpret_cast<const IN_T*>(a.dy);
    int sx = 0, sd = 0;
    if constexpr (NP == 2) { sx = f16x_scale_exp(*a.amax_x); sd = f16x_scale_exp(*a.amax_dy); }
    const float xscale = pow2f(sx), dscale = pow2f(sd);

    // transposing-read lane geometry: lane 4q+p of a 16-lane group addresses voxel row q, channels 4p..4p+3 of the fragment's
    // 16 channels and receives the four voxels of channel 4q+p.  Group g = lane / 16 holds k = {4g .. 4g+3} (first read) and
    // {16+4g .. 16+4g+3} (second read) of the 32-voxel k-step -- the same assignment for x and dy -- so the two groups that share
    // an LDS cycle (g = 0, 1 / g = 2, 3) read eight CONSECUTIVE voxels of one x-line.  A k-step is one x-line (BX = 32: a
    // read covers 16 of its voxels), two (BX = 16: one line per read) or four (BX = 8: g / 2 picks the line of the pair).
    const int li = lane & 15, q = li >> 2, p = li & 3, g = lane >> 4;
    const int chan_off = 4 * p * 2;
    const int kq_x = BX == 8 ? (g >> 1) * S * T::HX + (4 * (g & 1) + q) * S : (4 * g + q) * S;
    const int lane_x = kq_x * T::ROW + chan_off;
    const int lane_d = (4 * g + q) * T::ROW + chan_off;

    // the taps of this wave: wave, wave + 8, wave + 16 (, wave + 24 for the first waves)
    const bool has_last = wave + LW_WAVES * (LW_TPW - 1) <= T::NTAPS - 1;       // wave-uniform
    int abase[LW_TPW];
#pragma unroll
    for (int tt = 0; tt < LW_TPW; ++tt) {
        int tap = wave + LW_WAVES * tt;
        if (tap > T::NTAPS - 1) tap = T::NTAPS - 1;
        const int dz = KS == 3 ? tap / 9 : 0, dy = KS == 3 ? (tap / 3) % 3 : tap / KS, dx = tap % KS;
        abase[tt] = ((dz * T::HY + dy) * T::HX + dx) * T::ROW + lane_x;
    }

    f32x4 acc[LW_TPW][2][2];                                      // [tap-tile][ci half][co half]
#pragma unroll
    for (int tt = 0; tt < LW_TPW; ++tt)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) acc[tt][ab >> 1][ab & 1] = f32x4{0.f, 0.f, 0.f, 0.f};

    using stage_t = typename std::conditional<EPP == 4, f32x4, bf16x8_t>::type;
    stage_t stx[XITER], std_[DITER];
    // ---- tile loads.  A piece's place inside the tile never changes: its three halo coordinates (one byte each, packed) and its
    // byte offset from the tile's first halo voxel are computed once; per tile only wave-uniform values change -- the descriptor
    // base (the tile's first halo voxel: scalar arithmetic) and the packed bounds the coordinates are tested against, all three
    // axes in two packed additions (field + (128 - lo) sets bit 7 iff h >= lo; (127 + hi) - field sets it iff h < hi; fields stay
    // below 256, so nothing carries).  Pieces outside the volume are requested past the descriptor's range and read as zeros.
    // Six full-rate VALU instructions per piece instead of the ~10 quarter-rate multiplies of the 64-bit address of each piece,
    // and the pipelined kernels request ONE piece per scheduling region, inside the MFMA stream (r4: a burst of address arithmetic
    // in all eight waves ahead of the MFMAs left the MFMA pipes idle for a third of the tile, profiles/r04_wgrad_f16_ablation_probe.log)
    constexpr int ESZ = (int)sizeof(IN_T);
    constexpr unsigned OOB = 0x7FFFFFF0u;
    int crd[XITER], rel[XITER], dcrd[DITER], drel[DITER];
#pragma unroll
    for (int it = 0; it < XITER; ++it) {
        const int pc = it * LW_THREADS + tid;
        const int vox = pc / PPV, part = pc % PPV;
        const int hz = vox / (T::HY * T::HX), rem = vox % (T::HY * T::HX);
        const int hy = rem / T::HX, hx = rem % T::HX;
        crd[it] = pc < XPIECES ? (hx | (hy << 8) | (hz << 16)) : 0x7f7f7f;
        rel[it] = (((hz * a.H + hy) * a.W + hx) * a.ldx + part * EPP) * ESZ;
    }
#pragma unroll
    for (int it = 0; it < DITER; ++it) {
        const int pc = it * LW_THREADS + tid;
        const int vox = pc / PPV, part = pc % PPV;
        const int line = vox / BX, xx = vox % BX;
        dcrd[it] = xx | ((line % T::TY) << 8) | ((line / T::TY) << 16);
        drel[it] = ((((line / T::TY) * a.Ho + line % T::TY) * a.Wo + xx) * a.lddy + part * EPP) * ESZ;
    }
    __amdgpu_buffer_rsrc_t rx, rd;
    int xA = 0, xB = 0, dA = 0;
    auto clamp7 = [](int v) { return v < 0 ? 0 : (v > 127 ? 127 : v); };
    auto tile_geom = [&](int tile, bool valid) {                 // wave-uniform
        int mt = tile;
        const int txi = mt % a.ntx; mt /= a.ntx;
        const int tyi = mt % a.nty; mt /= a.nty;
        const int tzi = mt % a.ntz;
        const int n = mt / a.ntz;
        const int x0 = txi * BX, y0 = tyi * T::TY, z0 = tzi * T::TZ;
        const int ox = x0 * S - T::HALO, oy = y0 * S - T::HALO, oz = z0 * S + (KS == 3 ? -1 : plane - T::HALO);      // first halo voxel
        const int lox = clamp7(-ox), loy = clamp7(-oy), loz = clamp7(-oz);
        const int hix = valid ? clamp7(a.W - ox) : 0, hiy = clamp7(a.H - oy), hiz = clamp7(a.D - oz);
        xB = (128 - lox) | ((128 - loy) << 8) | ((128 - loz) << 16);
        xA = (127 + hix) | ((127 + hiy) << 8) | ((127 + hiz) << 16);
        dA = (127 + (valid ? clamp7(a.Wo - x0) : 0)) | ((127 + clamp7(a.Ho - y0)) << 8) | ((127 + clamp7(a.Do - z0)) << 16);
        const IN_T* xb = xin + ((((long long)n * a.D + oz) * a.H + oy) * a.W + ox) * a.ldx + ci0;
        const IN_T* db = din + ((((long long)n * a.Do + z0) * a.Ho + y0) * a.Wo + x0) * a.lddy + co0;
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<IN_T*>(xb), 0, (int)OOB, 0x00020000);
        rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<IN_T*>(db), 0, (int)OOB, 0x00020000);
    };
    auto load_piece = [&](int j) {                               // j: compile-time after unrolling
        if (j < XITER) {
            const int u = xA - crd[j], v = crd[j] + xB;
            const bool ok = ((u & v) & 0x808080) == 0x808080;
            stx[j] = __builtin_bit_cast(stage_t, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)rel[j] : OOB, 0, 0));
        } else {
            const int k = j - XITER;
            const bool ok = ((dA - dcrd[k]) & 0x808080) == 0x808080;
            std_[k] = __builtin_bit_cast(stage_t, __builtin_amdgcn_raw_buffer_load_b128(rd, ok ? (unsigned)drel[k] : OOB, 0, 0));
        }
    };
    constexpr int NPC = XITER + DITER;
    auto load_stage = [&]() {
#pragma unroll
        for (int j = 0; j < NPC; ++j) load_piece(j);
    };
    auto put = [&](unsigned char* base, int pc, const stage_t& v, float scale) {
        unsigned char* dst = base + (pc / PPV) * T::ROW + (pc % PPV) * (EPP * 2);
        if constexpr (EPP == 4) {
            if constexpr (NP == 2) {                  // two fp16 planes h | l of the scaled voxel row
                f16x4_t qh, ql;
                if (LW_DBG(a, 256)) {                 // timing probe: the operand arrives already split -- staging is a copy
                    using f32x2p = __attribute__((ext_vector_type(2))) float;
                    qh = __builtin_bit_cast(f16x4_t, f32x2p{v[0], v[1]}); ql = __builtin_bit_cast(f16x4_t, f32x2p{v[2], v[3]});
                } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { _Float16 bh, bl; split2h(v[e] * scale, bh, bl); qh[e] = bh; ql[e] = bl; }
                }
                *reinterpret_cast<f16x4_t*>(dst) = qh;
                *reinterpret_cast<f16x4_t*>(dst + 64) = ql;
            } else if constexpr (NP == 3) {                  // split once per staged value: planes h | m | l of the voxel row
                bf16x4_t qh, qm, ql;
#pragma unroll
                for (int e = 0; e < 4; ++e) { bf16 bh, bm, bl; split3(v[e], bh, bm, bl); qh[e] = bh; qm[e] = bm; ql[e] = bl; }
                *reinterpret_cast<bf16x4_t*>(dst) = qh;
                *reinterpret_cast<bf16x4_t*>(dst + 64) = qm;
                *reinterpret_cast<bf16x4_t*>(dst + 128) = ql;
            } else {
                bf16x4_t qh;
#pragma unroll
                for (int e = 0; e < 4; ++e) qh[e] = (bf16)v[e];
                *reinterpret_cast<bf16x4_t*>(dst) = qh;
            }
        } else {
            *reinterpret_cast<bf16x8_t*>(dst) = v;
        }
    };
    // prologue table of this workgroup's 32 input channels behind the tiles: al * 2^sx | be * 2^sx (the activation is positively
    // homogeneous, so the f16x3 scale rides in the table); a piece's four channels are (tid % PPV) * 4 .. for every piece of a thread
    float* const ptab = reinterpret_cast<float*>(lds + T::LDS_BYTES);
    if constexpr (PRO) {
        if (tid < 32) { ptab[tid] = a.pro_al[ci0 + tid] * xscale; ptab[32 + tid] = a.pro_be[ci0 + tid] * xscale; }
    }
    auto write_stage = [&]() {
        if constexpr (PRO) {
            {
                const f32x4 pal = *reinterpret_cast<const f32x4*>(ptab + (tid % PPV) * 4), pbe = *reinterpret_cast<const f32x4*>(ptab + 32 + (tid % PPV) * 4);
#pragma unroll
                for (int it = 0; it < XITER; ++it) {
                    const int pc = it * LW_THREADS + tid;
                    if (pc < XPIECES) {
                        const int u = xA - crd[it], v = crd[it] + xB;
                        const bool ok = ((u & v) & 0x808080) == 0x808080;       // (the bounds of the tile these pieces were loaded for)
                        f32x4 z;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = fmaf(stx[it][e], pal[e], pbe[e]);
                            z[e] = ok ? fmaxf(t, t * a.pro_slope) : 0.f;          // ReLU (slope 0) / LeakyReLU
                        }
                        put(xs, pc, z, 1.f);
                    }
                }
#pragma unroll
                for (int it = 0; it < DITER; ++it) put(ds, it * LW_THREADS + tid, std_[it], dscale);
                return;
            }
        }
#pragma unroll
        for (int it = 0; it < XITER; ++it) {
            const int pc = it * LW_THREADS + tid;
            if (pc < XPIECES) put(xs, pc, stx[it], xscale);
        }
#pragma unroll
        for (int it = 0; it < DITER; ++it) put(ds, it * LW_THREADS + tid, std_[it], dscale);
    };

    // byte offset of 32-voxel k-step ks, read t inside the x halo / the dy tile (lane part excluded)
    auto xoff = [](int ks, int t) {
        const int line = BX == 32 ? ks : (BX == 16 ? 2 * ks + t : 4 * ks + 2 * t);
        return ((((line / T::TY) * S * T::HY + (line % T::TY) * S) * T::HX) + (BX == 32 ? 16 * t * S : 0)) * T::ROW;
    };
    auto doff = [](int ks, int t) { return (ks * 32 + 16 * t) * T::ROW; };

    // bf16 tensors (one MFMA per fragment pair): one scheduling region per (k-step, tap-tile); the x fragments of the NEXT region
    // are requested behind the first MFMAs of the current one (left alone the compiler sinks the transposing reads next to
    // their first use and every tap-tile starts with an exposed LDS latency).  The MFMAs of a region run co-half 0 first: the
    // dy fragments are single-buffered, half 0 of the next k-step is requested behind the last region's half-0 MFMAs and
    // half 1 at the start of the next region.
    auto tile_mfma = [&](auto ntc) {
        constexpr int NTT = decltype(ntc)::value;
        constexpr int NREG = T::KSTEPS * NTT;
        bf16x8_t bc[2][NP], ac[2][2][NP];
        auto load_b = [&](int ks, int b) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) bc[b][pl] = tr_frag(ds, lane_d + doff(ks, 0) + b * 32 + pl * 64, lane_d + doff(ks, 1) + b * 32 + pl * 64);
        };
        auto load_a = [&](int buf, int ks, int tt) {
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) ac[buf][a2][pl] = tr_frag(xs, abase[tt] + xoff(ks, 0) + a2 * 32 + pl * 64, abase[tt] + xoff(ks, 1) + a2 * 32 + pl * 64);
        };
        auto mfma_half = [&](int cur, int tt, int b) {
            if constexpr (NP == 2) {                // planes 0 / 1 = h / l of the fp16 split; the two cross terms go in first
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2)
                        acc[tt][a2][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ac[cur][a2][PA[pr]]), __builtin_bit_cast(f16x8_t, bc[b][PB[pr]]),
                                                                               acc[tt][a2][b], 0, 0, 0);
            } else if constexpr (NP == 3) {                // planes 0 / 1 / 2 = h / m / l; the small cross terms go in first; the two ci tiles alternate
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2)
                        acc[tt][a2][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[cur][a2][PA[pr]], bc[b][PB[pr]], acc[tt][a2][b], 0, 0, 0);
            } else {
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2)
                    acc[tt][a2][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[cur][a2][0], bc[b][0], acc[tt][a2][b], 0, 0, 0);
            }
        };
        if constexpr (NP == 3 || S == 2) {
            // (the strided tiles are register-bound too.)  bf16x6: 24 MFMAs per tap-tile hide the reads of the next one by themselves (two waves per SIMD); pinning the order by
            // hand costs registers the three-plane fragments do not leave (measured: 42 spilled VGPRs, 0.68x) -- compiler order
#pragma unroll
            for (int ks = 0; ks < T::KSTEPS; ++ks) {
                load_b(ks, 0); load_b(ks, 1);
#pragma unroll
                for (int tt = 0; tt < NTT; ++tt) {
                    load_a(0, ks, tt);
                    mfma_half(0, tt, 0);
                    mfma_half(0, tt, 1);
                }
                if (ks == 0) load_stage();                          // the next tile, behind the first k-step's reads
            }
        } else {
            load_b(0, 0); load_b(0, 1);
            load_a(0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NREG; ++u) {
                const int ks = u / NTT, tt = u % NTT, cur = u & 1;
                if (tt == 0 && ks > 0) load_b(ks, 1);
                if (u + 1 < NREG) load_a(cur ^ 1, (u + 1) / NTT, (u + 1) % NTT);
                mfma_half(cur, tt, 0);
                if (tt == NTT - 1 && ks + 1 < T::KSTEPS) load_b(ks + 1, 0);
                {                                                   // the next tile's pieces, spread over the regions
                    constexpr int PPR = (NPC + NREG - 1) / NREG;
#pragma unroll
                    for (int j = u * PPR; j < (u + 1) * PPR && j < NPC; ++j) load_piece(j);
                }
                mfma_half(cur, tt, 1);
                // spread the region's reads over its MFMAs (groups that find no read left are no-ops)
                if constexpr (NP == 1) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) { __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
#pragma unroll
                    for (int k = 0; k < 2; ++k) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
                } else {
                    // f16x3: 12 MFMAs per region, 8 transposing reads for the next region's x fragments (+ up to 8 for the next k-step's dy)
#pragma unroll
                    for (int k = 0; k < 8; ++k) { __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    int tile = strip;
    tile_geom(tile, tile < a.ntiles);
    load_stage();
    for (; tile < a.ntiles; tile += a.nstrips) {
        __syncthreads();
        if (!LW_DBG(a, 16) || tile == strip) write_stage();
        __syncthreads();
        // the loads of the next tile are requested inside tile_mfma; past the strip's last tile (hi bound 0) they all read zeros
        tile_geom(LW_DBG(a, 64) ? strip : tile + a.nstrips, tile + a.nstrips < a.ntiles && !LW_DBG(a, 8));
        // every lane of every wave runs the transposing reads (they need EXEC all ones); a wave without a fourth tap
        // simply issues one tap-tile fewer
        if (has_last) tile_mfma(std::integral_constant<int, LW_TPW>{});
        else tile_mfma(std::integral_constant<int, LW_TPW - 1>{});
    }

    // slab store: part[strip][tap][ci][co]; a 16 x 16 tile holds ci = 4 (lane / 16) + e in its four registers, co on the lanes
#pragma unroll
    for (int tt = 0; tt < LW_TPW; ++tt) {
        const int tap = wave + LW_WAVES * tt;
        if (tap > T::NTAPS - 1) break;
        float* dst = a.part + (((long long)strip * a.ntaps_total + plane * T::NTAPS + tap) * a.Cin + ci0 + 4 * g) * a.Cout + co0 + li;
#pragma unroll
        for (int ab = 0; ab < 4; ++ab)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[tt][ab >> 1][ab & 1][e];
                if constexpr (NP == 2) v = __builtin_ldexpf(v, -(sx + sd));
                dst[(long long)(16 * (ab >> 1) + e) * a.Cout + 16 * (ab & 1)] = v;
            }
    }
}

// ---- the WIDE kernel (stride 1, Cout % 64 == 0; f16x3 on fp32 tensors, or bf16 tensors): four waves, one per SIMD with up to 512
// registers, a 32 (ci) x 64 (co) block pair per workgroup.  The 8-wave kernel above issues 0.83-0.89 (f16x3) / 1.25 (bf16) transposing
// reads per MFMA and runs its staging with the MFMA pipes idle; here a wave owns SEVEN tap-tiles of 2 x 4 accumulators (224 AGPRs):
// an x fragment feeds four co quarters, a dy fragment seven taps -- 0.43 / 0.64 reads per MFMA.  Same tile, slab layout and reduction.
#ifndef FW_SPLIT
#define FW_SPLIT 1
#endif
constexpr int FW_WAVES = 4, FW_THREADS = FW_WAVES * 64, FW_TPW = 7;

template <int BX, int KS, int NP>
struct FTile {
    using X = LTile<BX, KS, NP, 1>;
    static constexpr int DROW = 128 * NP + 32;                   // NP planes of 64 16-bit channels + 32 bytes: pitch / 32 odd
    static constexpr int D_BYTES = X::VOX * DROW;
    static constexpr int LDS_BYTES = X::X_BYTES + D_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the LDS");
};

template <int BX, int KS, typename IN_T, bool PRO = false>
__global__ __launch_bounds__(FW_THREADS, 1) void conv_wgrad_wide_kernel(LWgradArgs a) {
    constexpr bool F32 = std::is_same<IN_T, float>::value;
    constexpr int NP = F32 ? 2 : 1, NPR = F32 ? 3 : 1;           // operand planes, MFMA products per fragment pair
    constexpr int EPP = F32 ? 4 : 8, ESZ = (int)sizeof(IN_T);
    constexpr int PPX = 32 / EPP, PPD = 64 / EPP;                 // 16-byte pieces per voxel: 32 ci / 64 co
    using T = LTile<BX, KS, NP, 1>;
    using F = FTile<BX, KS, NP>;
    constexpr int XPIECES = T::NVOX * PPX, DPIECES = T::VOX * PPD;
    constexpr int XITER = (XPIECES + FW_THREADS - 1) / FW_THREADS, DITER = DPIECES / FW_THREADS, NPC = XITER + DITER;
    constexpr int VPI = FW_THREADS / PPD, LPI = VPI / BX;         // dy voxels / tile lines per piece index
    static_assert(DPIECES % FW_THREADS == 0 && VPI % BX == 0 && T::TY % LPI == 0, "dy pieces must cover whole lines");
    constexpr int NTAPS = T::NTAPS;                               // 27, or the 25 of one dz plane (k5)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xs = lds;
    unsigned char* ds = lds + T::X_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int plane = t % T::PLANES, tp = t / T::PLANES;          // k5: which dz plane of taps this workgroup owns
    const int pair = tp % a.npairs, strip = tp / a.npairs;
    const int cib = pair / a.ncob, cob = pair % a.ncob;
    const int ci0 = cib * 32, co0 = cob * 64;
    const IN_T* __restrict__ xin = reinterpret_cast<const IN_T*>(a.x);
    const IN_T* __restrict__ din = reinterpret_cast<const IN_T*>(a.dy);
    int sx = 0, sd = 0;
    if constexpr (F32) { sx = f16x_scale_exp(*a.amax_x); sd = f16x_scale_exp(*a.amax_dy); }
    const float xscale = pow2f(sx), dscale = pow2f(sd);

    // transposing-read lane geometry: as in conv_wgrad_lowp_kernel
    const int li = lane & 15, q = li >> 2, p = li & 3, g = lane >> 4;
    const int chan_off = 4 * p * 2;
    const int kq_x = BX == 8 ? (g >> 1) * T::HX + (4 * (g & 1) + q) : (4 * g + q);
    const int lane_x = kq_x * T::ROW + chan_off;
    const int lane_d = (4 * g + q) * F::DROW + chan_off;

    int abase[FW_TPW];                                            // wave w owns taps w, w + 4, ...; a slot past the last tap repeats it (dropped at the end)
#pragma unroll
    for (int tt = 0; tt < FW_TPW; ++tt) {
        int tap = wave + FW_WAVES * tt;
        if (tap > NTAPS - 1) tap = NTAPS - 1;
        const int dz = KS == 3 ? tap / 9 : 0, dy = KS == 3 ? (tap / 3) % 3 : tap / KS, dx = tap % KS;
        abase[tt] = ((dz * T::HY + dy) * T::HX + dx) * T::ROW + lane_x;
    }

    f32x4 acc[FW_TPW][2][4];                                      // [tap-tile][ci half][co quarter]
#pragma unroll
    for (int tt = 0; tt < FW_TPW; ++tt)
#pragma unroll
        for (int ab = 0; ab < 8; ++ab) acc[tt][ab >> 2][ab & 3] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- tile loads: x pieces as above (packed halo coordinates + byte offset per piece).  dy piece `it` of a lane is voxel
    // it * VPI + tid / PPD: the same x (and the same line inside the LPI lines of a piece index) for every `it`, so one packed
    // coordinate and one offset per lane serve all of them; the line of piece `it` moves the bounds and the descriptor offset by
    // wave-uniform amounts.
    constexpr unsigned OOB = 0x7FFFFFF0u;
    using stage_t = typename std::conditional<F32, f32x4, bf16x8_t>::type;
    stage_t stx[XITER], std_[DITER];
    int crd[XITER], rel[XITER];
#pragma unroll
    for (int it = 0; it < XITER; ++it) {
        const int pc = it * FW_THREADS + tid;
        const int vox = pc / PPX, part = pc % PPX;
        const int hz = vox / (T::HY * T::HX), rem = vox % (T::HY * T::HX);
        const int hy = rem / T::HX, hx = rem % T::HX;
        crd[it] = pc < XPIECES ? (hx | (hy << 8) | (hz << 16)) : 0x7f7f7f;
        rel[it] = (((hz * a.H + hy) * a.W + hx) * a.ldx + part * EPP) * ESZ;
    }
    const int dvox0 = tid / PPD, dpart = tid % PPD;
    const int dxx = dvox0 % BX, dhb = dvox0 / BX;                 // dhb < LPI
    const int dcrd = dxx | (dhb << 8);
    const int drel = ((dhb * a.Wo + dxx) * a.lddy + dpart * EPP) * ESZ;
    __amdgpu_buffer_rsrc_t rx, rd;
    int xA = 0, xB = 0, dA = 0;
    auto clamp7 = [](int v) { return v < 0 ? 0 : (v > 127 ? 127 : v); };
    auto tile_geom = [&](int tile, bool valid) {                 // wave-uniform
        int mt = tile;
        const int txi = mt % a.ntx; mt /= a.ntx;
        const int tyi = mt % a.nty; mt /= a.nty;
        const int tzi = mt % a.ntz;
        const int n = mt / a.ntz;
        const int x0 = txi * BX, y0 = tyi * T::TY, z0 = tzi * T::TZ;
        const int ox = x0 - T::HALO, oy = y0 - T::HALO, oz = z0 + (KS == 3 ? -1 : plane - T::HALO);      // first halo voxel
        const int lox = clamp7(-ox), loy = clamp7(-oy), loz = clamp7(-oz);
        const int hix = valid ? clamp7(a.W - ox) : 0, hiy = clamp7(a.H - oy), hiz = clamp7(a.D - oz);
        xB = (128 - lox) | ((128 - loy) << 8) | ((128 - loz) << 16);
        xA = (127 + hix) | ((127 + hiy) << 8) | ((127 + hiz) << 16);
        dA = (127 + (valid ? clamp7(a.Wo - x0) : 0)) | ((127 + clamp7(a.Ho - y0)) << 8) | ((127 + clamp7(a.Do - z0)) << 16);
        const IN_T* xb = xin + ((((long long)n * a.D + oz) * a.H + oy) * a.W + ox) * a.ldx + ci0;
        const IN_T* db = din + ((((long long)n * a.Do + z0) * a.Ho + y0) * a.Wo + x0) * a.lddy + co0;
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<IN_T*>(xb), 0, (int)OOB, 0x00020000);
        rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<IN_T*>(db), 0, (int)OOB, 0x00020000);
    };
    auto load_piece = [&](int j) {                               // j: compile-time after unrolling
        if (j < XITER) {
            const int u = xA - crd[j], v = crd[j] + xB;
            const bool ok = ((u & v) & 0x808080) == 0x808080;
            stx[j] = __builtin_bit_cast(stage_t, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)rel[j] : OOB, 0, 0));
        } else {
            const int k = j - XITER, ln = k * LPI;               // first line of the piece index (+ dhb in the lane)
            const int ly = ln % T::TY, lz = ln / T::TY;          // compile-time
            const bool ok = (((dA - ((ly << 8) | (lz << 16))) - dcrd) & 0x808080) == 0x808080;
            const int soff = ((lz * a.Ho + ly) * a.Wo) * a.lddy * ESZ;      // wave-uniform
            std_[k] = __builtin_bit_cast(stage_t, __builtin_amdgcn_raw_buffer_load_b128(rd, ok ? (unsigned)drel : OOB, soff, 0));
        }
    };
    auto lds_dst = [&](int j) {
        const int pc = j * FW_THREADS + tid;
        return j < XITER ? xs + (pc / PPX) * T::ROW + (pc % PPX) * 16 / NP : ds + ((j - XITER) * VPI + dvox0) * F::DROW + dpart * 16 / NP;
    };
    // f16x3: the splits run inside the MFMA stream (FW_SPLIT), in the registers the piece arrived in: between the barriers only the LDS
    // writes remain
    f16x4_t cvh[F32 && FW_SPLIT ? NPC : 1], cvl[F32 && FW_SPLIT ? NPC : 1];
    // norm + activation prologue of the x pieces (f16x3): this thread's four channels are the same for every piece
    // (the table sits in LDS behind the tiles and is read per piece: eight more long-lived registers spill in the MFMA stream)
    float* const ptab = reinterpret_cast<float*>(lds + F::LDS_BYTES);
    constexpr bool pro = PRO;
    if constexpr (PRO) {
        if (tid < 32) { ptab[tid] = a.pro_al[ci0 + tid] * xscale; ptab[32 + tid] = a.pro_be[ci0 + tid] * xscale; }
        __syncthreads();
    }
    auto pro_x = [&](int j, f32x4 v) {                            // (xA / xB: the bounds of the tile piece j was loaded for)
        const int u = xA - crd[j], w = crd[j] + xB;
        const bool ok = ((u & w) & 0x808080) == 0x808080;
        const f32x4 pal = *reinterpret_cast<const f32x4*>(ptab + (tid % PPX) * 4), pbe = *reinterpret_cast<const f32x4*>(ptab + 32 + (tid % PPX) * 4);
        f32x4 z;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = fmaf(v[e], pal[e], pbe[e]);
            z[e] = ok ? fmaxf(t, t * a.pro_slope) : 0.f;                  // ReLU (slope 0) / LeakyReLU
        }
        return z;
    };
    auto split_piece = [&](int j) {
        if constexpr (F32 && FW_SPLIT) {
            float sc = j < XITER ? xscale : dscale;
            f32x4 v;
            if (j < XITER) v = stx[j]; else v = std_[j - XITER];
            if (j < XITER && pro) { v = pro_x(j, v); sc = 1.f; }
            if (LW_DBG(a, 256) && !(j < XITER && pro)) {      // timing probe: the operand arrives already split -- staging is a copy
                using f32x2p = __attribute__((ext_vector_type(2))) float;
                cvh[j] = __builtin_bit_cast(f16x4_t, f32x2p{v[0], v[1]}); cvl[j] = __builtin_bit_cast(f16x4_t, f32x2p{v[2], v[3]});
                return;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { _Float16 bh, bl; split2h(v[e] * sc, bh, bl); cvh[j][e] = bh; cvl[j][e] = bl; }
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            if (j < XITER && j * FW_THREADS + tid >= XPIECES) continue;
            unsigned char* dst = lds_dst(j);
            if constexpr (!F32) {
                *reinterpret_cast<bf16x8_t*>(dst) = j < XITER ? stx[j] : std_[j < XITER ? 0 : j - XITER];
            } else if constexpr (FW_SPLIT) {
                *reinterpret_cast<f16x4_t*>(dst) = cvh[j];
                *reinterpret_cast<f16x4_t*>(dst + (j < XITER ? 64 : 128)) = cvl[j];
            } else {
                float sc = j < XITER ? xscale : dscale;
                f32x4 v;
                if (j < XITER) v = stx[j]; else v = std_[j < XITER ? 0 : j - XITER];
                if (j < XITER && pro) { v = pro_x(j, v); sc = 1.f; }
                f16x4_t qh, ql;
#pragma unroll
                for (int e = 0; e < 4; ++e) { _Float16 bh, bl; split2h(v[e] * sc, bh, bl); qh[e] = bh; ql[e] = bl; }
                *reinterpret_cast<f16x4_t*>(dst) = qh;
                *reinterpret_cast<f16x4_t*>(dst + (j < XITER ? 64 : 128)) = ql;
            }
        }
    };

    auto xoff = [](int ks, int t2) {
        const int line = BX == 32 ? ks : (BX == 16 ? 2 * ks + t2 : 4 * ks + 2 * t2);
        return ((((line / T::TY) * T::HY + (line % T::TY)) * T::HX) + (BX == 32 ? 16 * t2 : 0)) * T::ROW;
    };
    auto doff = [](int ks, int t2) { return (ks * 32 + 16 * t2) * F::DROW; };

    // one scheduling region per (k-step, tap-tile): 8 NPR MFMAs; the x fragments of the next region are requested behind its first
    // MFMAs, the dy fragments of the next k-step during the k-step's last tap-tile, tile pieces in the first regions
    auto tile_mfma = [&]() {
        constexpr int NTT = FW_TPW, NREG = T::KSTEPS * NTT;
        constexpr int PPR = 2, NLR = (NPC + PPR - 1) / PPR;        // tile pieces requested per region, regions that request
        static_assert(NREG >= 2 * NLR, "the requests and the splits each need their regions");
        bf16x8_t bc[4][NP], ac[2][2][NP];
        auto load_b1 = [&](int b, int ks) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                bc[b][pl] = tr_frag(ds, lane_d + doff(ks, 0) + b * 32 + pl * 128, lane_d + doff(ks, 1) + b * 32 + pl * 128);
        };
        auto load_a = [&](int buf, int ks, int tt) {
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    ac[buf][a2][pl] = tr_frag(xs, abase[tt] + xoff(ks, 0) + a2 * 32 + pl * 64, abase[tt] + xoff(ks, 1) + a2 * 32 + pl * 64);
        };
        auto mfma = [&](int cur, int tt, int a2, int b, int pr) {
            if constexpr (F32) {                                  // planes 0 / 1 = h / l; the two cross terms go in first
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
                acc[tt][a2][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, ac[cur][a2][PA[pr]]), __builtin_bit_cast(f16x8_t, bc[b][PB[pr]]),
                                                                       acc[tt][a2][b], 0, 0, 0);
            } else {
                acc[tt][a2][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[cur][a2][0], bc[b][0], acc[tt][a2][b], 0, 0, 0);
            }
        };
#pragma unroll
        for (int b = 0; b < 4; ++b) load_b1(b, 0);
        load_a(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NREG; ++u) {
            const int ks = u / NTT, tt = u % NTT, cur = u & 1;
            if (u + 1 < NREG) load_a(cur ^ 1, (u + 1) / NTT, (u + 1) % NTT);
            // the next tile: two pieces requested per region in the first regions, split (f16x3) in the last ones -- with one wave per
            // SIMD a wait on memory stalls the SIMD, so the distance is as long as the tile allows
#pragma unroll
            for (int k = 0; k < PPR; ++k) {
                const int jl = u * PPR + k, js = (u - (NREG - NLR)) * PPR + k;
                if (jl < NPC) load_piece(jl);
                if (js >= 0 && js < NPC) split_piece(js);
            }
            if (tt == NTT - 1 && ks + 1 < T::KSTEPS) {
                // the k-step's last tap-tile: one co quarter at a time, its dy fragments of the NEXT k-step requested right behind its
                // MFMAs (single-buffered: the fragments' registers are free by then)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
                        for (int a2 = 0; a2 < 2; ++a2) mfma(cur, tt, a2, b, pr);
                    load_b1(b, ks + 1);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 2 * NPR, 0); __builtin_amdgcn_sched_group_barrier(0x100, 3 * NP, 0); }
            } else {
#pragma unroll
                for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                        for (int b = 0; b < 4; ++b) mfma(cur, tt, a2, b, pr);
#pragma unroll
                for (int k = 0; k < 8; ++k) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, NPR, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int tile = strip;
    tile_geom(tile, tile < a.ntiles);
#pragma unroll
    for (int j = 0; j < NPC; ++j) load_piece(j);
#pragma unroll
    for (int j = 0; j < NPC; ++j) split_piece(j);
    for (; tile < a.ntiles; tile += a.nstrips) {
        __syncthreads();
        if (!LW_DBG(a, 16) || tile == strip) write_stage();
        __syncthreads();
        tile_geom(LW_DBG(a, 64) ? strip : tile + a.nstrips, tile + a.nstrips < a.ntiles && !LW_DBG(a, 8));
        // (ONE instantiation of the MFMA stream: with a shorter one for the waves that own a tap fewer in a branch, the register
        // allocator moved accumulators between AGPRs and VGPRs at every merge -- 250-330 spills)
        tile_mfma();
    }

    // slab store: part[strip][tap][ci][co]
#pragma unroll
    for (int tt = 0; tt < FW_TPW; ++tt) {
        const int tap = wave + FW_WAVES * tt;
        float* dst = a.part + (((long long)strip * a.ntaps_total + plane * NTAPS + tap) * a.Cin + ci0 + 4 * g) * a.Cout + co0 + li;
        if (tap <= NTAPS - 1) {
#pragma unroll
            for (int ab = 0; ab < 8; ++ab)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[tt][ab >> 2][ab & 3][e];
                    if constexpr (F32) v = __builtin_ldexpf(v, -(sx + sd));
                    dst[(long long)(16 * (ab >> 2) + e) * a.Cout + 16 * (ab & 3)] = v;
                }
        }
    }
}

// 0: never the wide f16x3 kernel, 1 (default): where it pays, 2: wherever the geometry allows (tests)
static int g_wgrad_wide = [] { const char* e = getenv("MI355SEG_WGRAD_WIDE"); return e ? atoi(e) : 1; }();
void set_wgrad_wide(int mode) { g_wgrad_wide = mode; }
int get_wgrad_wide() { return g_wgrad_wide; }

struct LWgradPlan { int KS, BX, ntx, nty, ntz, ntiles, nstrips, npairs, taps, planes, cob; };      // cob: output channels per workgroup (64: the wide f16x3 kernel)

// D, H, W: OUTPUT extents (the space the tiles are cut in)
static bool lwgrad_plan(int math, int KS, int stride, int N, int D, int H, int W, int Cin, int Cout, LWgradPlan* p, bool wide = false) {
    if ((KS != 3 && KS != 5) || Cin % 32 || Cout % 32 || W < 4) return false;
    if (wide && (stride != 1 || Cout % 64 || (math == MATH_X3 && KS != 3))) return false;
    if (stride != 1 && !(stride == 2 && KS == 3 && math == MATH_B16)) return false;
    const int vox = (math == MATH_X3 || stride == 2) ? 128 : 256;
    int BX = 0; long long best = -1;
    for (int bx : {32, 16, 8}) {
        if (math == MATH_X3 && bx == 32) continue;          // 128-voxel tiles: the 16-wide tile has the smaller halo
        long long padded = (long long)((W + bx - 1) / bx) * bx;
        if (best < 0 || padded < best) { best = padded; BX = bx; }
    }
    const int TY = BX == 8 ? 8 : 4, TZ = (vox / BX) / TY;
    p->KS = KS; p->BX = BX; p->ntx = (W + BX - 1) / BX; p->nty = (H + TY - 1) / TY; p->ntz = (D + TZ - 1) / TZ;
    p->ntiles = N * p->ntz * p->nty * p->ntx;
    p->cob = wide ? 64 : 32;
    p->npairs = (Cin / 32) * (Cout / p->cob);
    p->taps = KS * KS * KS; p->planes = KS == 3 ? 1 : KS;
    const int per_strip = p->npairs * p->planes;
    int want = 256 / per_strip;                             // one 8-wave workgroup per CU: never more than 256 in all
    long long cap = (long long)(160u << 20) / ((long long)p->taps * Cin * Cout * 4);   // keep the slab workspace <= 160 MB
    if (cap < 1) cap = 1;
    if (want > cap) want = (int)cap;
    if (want > p->ntiles) want = p->ntiles;
    if (want < 1) want = 1;
    p->nstrips = want;
    // the wide kernel runs one wave per SIMD: with few tiles per strip its exposed prologue / slab store outweigh the fewer LDS reads
    // (2 x 16^3, 256 -> 256: 8 tiles per strip, 0.89x; 2 x 32^3, 128 -> 128: 16 tiles, 1.06x)
    if (wide && g_wgrad_wide != 2 && p->ntiles / p->nstrips < 12) return false;
    return true;
}

static int lw_out(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

// the tile pieces are addressed by 32-bit byte offsets from the tile's first (halo) voxel on a buffer descriptor: the deepest halo (ten
// planes of the input, eight of the output gradient) must stay below the descriptor's 2 GB range
static bool lw_offsets_fit(int math, int H, int W, int ldx, int Ho, int Wo, int lddy) {
    const long long esz = math == MATH_B16 ? 2 : 4;
    return 10ll * (H + 8) * (W + 8) * ldx * esz < 0x7FFF0000ll && 8ll * (Ho + 8) * (Wo + 8) * lddy * esz < 0x7FFF0000ll;
}

bool wgrad_lowp_supported(int math, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int lddy) {
    const int al = math == MATH_B16 ? 8 : 4;
    if (!((k == 3 && pad == 1) || (k == 5 && pad == 2)) || (ldx % al) || (lddy % al)) return false;
    if (!lw_offsets_fit(math, H, W, ldx, lw_out(H, k, stride, pad), lw_out(W, k, stride, pad), lddy)) return false;
    LWgradPlan p;
    return lwgrad_plan(math, k, stride, N, lw_out(D, k, stride, pad), lw_out(H, k, stride, pad), lw_out(W, k, stride, pad), Cin, Cout, &p);
}

// any supported geometry (D, H, W = input extents)
size_t wgrad_lowp_ws_bytes_geom(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    if (!((k == 3 && pad == 1) || (k == 5 && pad == 2)) || D + 2 * pad < k || H + 2 * pad < k || W + 2 * pad < k) return 0;
    size_t best = 0;
    for (int math : {MATH_X3, MATH_B16}) {
        for (bool wide : {false, true}) {
            LWgradPlan p;
            if (!lwgrad_plan(math, k, stride, N, lw_out(D, k, stride, pad), lw_out(H, k, stride, pad), lw_out(W, k, stride, pad), Cin, Cout, &p, wide)) continue;
            const size_t need = align_up((size_t)p.nstrips * p.taps * Cin * Cout * sizeof(float), 256) + 1024;
            if (need > best) best = need;
        }
    }
    // the swapped-role wide form (conv_wgrad_lowp: Cin % 64 == 0, Cout % 64 != 0) plans its strips with the channel counts exchanged
    if (k == 3 && stride == 1 && Cin % 64 == 0 && Cout % 64 != 0) {
        LWgradPlan p;
        if (lwgrad_plan(MATH_X3, k, stride, N, lw_out(D, k, stride, pad), lw_out(H, k, stride, pad), lw_out(W, k, stride, pad), Cout, Cin, &p, true)) {
            const size_t need = align_up((size_t)p.nstrips * p.taps * Cin * Cout * sizeof(float), 256) + 1024;
            if (need > best) best = need;
        }
    }
    return best;
}

size_t wgrad_lowp_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k) {
    size_t best = 0;
    for (int math : {MATH_X3, MATH_B16}) {
        for (bool wide : {false, true}) {
            LWgradPlan p;
            if (!lwgrad_plan(math, k, 1, N, D, H, W, Cin, Cout, &p, wide)) continue;
            const size_t need = align_up((size_t)p.nstrips * p.taps * Cin * Cout * sizeof(float), 256) + 1024;
            if (need > best) best = need;
        }
    }
    if (k == 3 && Cin % 64 == 0 && Cout % 64 != 0) {              // the swapped-role wide form
        LWgradPlan p;
        if (lwgrad_plan(MATH_X3, k, 1, N, D, H, W, Cout, Cin, &p, true)) {
            const size_t need = align_up((size_t)p.nstrips * p.taps * Cin * Cout * sizeof(float), 256) + 1024;
            if (need > best) best = need;
        }
    }
    return best;
}

template <int BX, int KS, int NP, typename IN_T, int S = 1>
static void launch_lwgrad(const LWgradArgs& a, int nwg, hipStream_t st) {
    using T = LTile<BX, KS, NP, S>;
    if constexpr (NP == 2 && std::is_same<IN_T, float>::value && S == 1 && KS == 3) {
        if (a.pro_al) {                       // + the prologue table
            SEG_SET_LDS((conv_wgrad_lowp_kernel<BX, KS, NP, IN_T, S, true>), T::LDS_BYTES + 256);
            hipLaunchKernelGGL((conv_wgrad_lowp_kernel<BX, KS, NP, IN_T, S, true>), dim3(nwg), dim3(LW_THREADS), T::LDS_BYTES + 256, st, a);
            return;
        }
    }
    SEG_SET_LDS((conv_wgrad_lowp_kernel<BX, KS, NP, IN_T, S>), T::LDS_BYTES);
    hipLaunchKernelGGL((conv_wgrad_lowp_kernel<BX, KS, NP, IN_T, S>), dim3(nwg), dim3(LW_THREADS), T::LDS_BYTES, st, a);
}

template <int BX, int KS, typename IN_T>
static void launch_wide(const LWgradArgs& a, int nwg, hipStream_t st) {
    using F = FTile<BX, KS, std::is_same<IN_T, float>::value ? 2 : 1>;
    if constexpr (std::is_same<IN_T, float>::value && KS == 3) {
        if (a.pro_al) {
            SEG_SET_LDS((conv_wgrad_wide_kernel<BX, KS, IN_T, true>), F::LDS_BYTES + 256);
            hipLaunchKernelGGL((conv_wgrad_wide_kernel<BX, KS, IN_T, true>), dim3(nwg), dim3(FW_THREADS), F::LDS_BYTES + 256, st, a);
            return;
        }
    }
    SEG_SET_LDS((conv_wgrad_wide_kernel<BX, KS, IN_T>), F::LDS_BYTES);
    hipLaunchKernelGGL((conv_wgrad_wide_kernel<BX, KS, IN_T>), dim3(nwg), dim3(FW_THREADS), F::LDS_BYTES, st, a);
}

template <int KS>
static void dispatch_lwgrad(int math, const LWgradPlan& p, const LWgradArgs& a, int nwg, hipStream_t st) {
    if (math == MATH_X3 && a.amax_x) {
        if (p.BX == 16) launch_lwgrad<16, KS, 2, float>(a, nwg, st);
        else launch_lwgrad<8, KS, 2, float>(a, nwg, st);
    } else if (math == MATH_X3) {
        if (p.BX == 16) launch_lwgrad<16, KS, 3, float>(a, nwg, st);
        else launch_lwgrad<8, KS, 3, float>(a, nwg, st);
    } else {
        if (p.BX == 32) launch_lwgrad<32, KS, 1, bf16>(a, nwg, st);
        else if (p.BX == 16) launch_lwgrad<16, KS, 1, bf16>(a, nwg, st);
        else launch_lwgrad<8, KS, 1, bf16>(a, nwg, st);
    }
}

int conv_wgrad_lowp(int math, const void* dy, int lddy, const void* x, int ldx, float* dw, int N, int D, int H, int W, int Cin,
                    int Cout, int k, int stride, int accumulate, void* ws, size_t ws_bytes, hipStream_t st, const float* x_amax, const float* dy_amax,
                    const ConvPro* pro) {
    LWgradPlan p;
    const int pad = k / 2, Do = lw_out(D, k, stride, pad), Ho = lw_out(H, k, stride, pad), Wo = lw_out(W, k, stride, pad);
    // f16x3, k3 s1, Cout % 64 == 0: the wide kernel (mi355seg_set_wgrad_wide)
    const bool wide = g_wgrad_wide != 0 && (math == MATH_B16 || (math == MATH_X3 && x3_f16())) && lwgrad_plan(math, k, stride, N, Do, Ho, Wo, Cin, Cout, &p, true);
    // r5: a k3 s1 layer with Cin % 64 == 0 but Cout = 32 (dec1conv1, 64 -> 32 @ 128^3: the largest weight gradient of cfg 2) has no 64-wide co
    // block for the wide kernel -- but dW[tap][ci][co] = sum_v x[v + tap][ci] dy[v][co] = sum_u dy[u - tap][co] x[u][ci] is the same
    // weight gradient with the operands' roles swapped and the taps mirrored (both tensors are zero outside the same volume): x becomes the
    // centred 64-channel operand, dy the haloed 32-channel one, and the slabs come out as [27 - 1 - tap][co][ci] (wgrad_reduce_swapped)
    LWgradPlan psw;
    const bool swap = !wide && !pro && g_wgrad_wide != 0 && math == MATH_X3 && x3_f16() && k == 3 && stride == 1 && Cin % 64 == 0 && Cout % 64 != 0 &&
                      lwgrad_plan(math, k, stride, N, Do, Ho, Wo, Cout, Cin, &psw, true) && psw.nstrips >= 8 && lw_offsets_fit(math, Ho, Wo, lddy, H, W, ldx);
    if (swap) {
        SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "conv_wgrad_lowp: pointers must be 16-byte aligned");
        Carver cv(ws);
        float* part = cv.take<float>((size_t)psw.nstrips * psw.taps * Cin * Cout);
        float* amax = cv.take<float>(2);
        SEG_CHECK_WS(cv.used(), ws_bytes);
        LWgradArgs a{dy, x, part, lddy, ldx, N, D, H, W, Cout, Cin, Do, Ho, Wo, psw.ntx, psw.nty, psw.ntz, psw.ntiles, psw.nstrips, psw.npairs, Cin / psw.cob, psw.taps,
                     nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0};
        if (!x_amax || !dy_amax) {
            if (hipMemsetAsync(amax, 0, 2 * sizeof(float), st) != hipSuccess) { set_error("conv_wgrad_lowp: hipMemsetAsync failed"); return MI355SEG_EHIP; }
            if (!x_amax) { tensor_amax((const float*)x, ldx, (long long)N * D * H * W, Cin, nullptr, amax, st); x_amax = amax; }
            if (!dy_amax) { tensor_amax((const float*)dy, lddy, (long long)N * Do * Ho * Wo, Cout, nullptr, amax + 1, st); dy_amax = amax + 1; }
            SEG_CHECK_LAUNCH();
        }
        a.amax_x = dy_amax; a.amax_dy = x_amax;                    // (roles swapped)
#ifdef MI355SEG_TUNE
        { static const char* e = getenv("MI355SEG_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
        const int nwg = psw.nstrips * psw.npairs * psw.planes;
        const double vox = (double)N * Do * Ho * Wo;
        {
            ProfScope ps(PF_WGRAD, 2.0 * vox * psw.taps * Cin * Cout, 4.0 * vox * (Cin + Cout) + 4.0 * psw.taps * Cin * Cout, st);
            if (psw.BX == 16) launch_wide<16, 3, float>(a, nwg, st); else launch_wide<8, 3, float>(a, nwg, st);
            SEG_CHECK_LAUNCH();
        }
        wgrad_reduce_swapped(part, dw, psw.nstrips, psw.taps, Cin, Cout, accumulate, st);
        SEG_CHECK_LAUNCH();
        return MI355SEG_OK;
    }
    SEG_CHECK_ARG(wide || lwgrad_plan(math, k, stride, N, Do, Ho, Wo, Cin, Cout, &p), "conv_wgrad_lowp: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "conv_wgrad_lowp: pointers must be 16-byte aligned");
    SEG_CHECK_ARG(lw_offsets_fit(math, H, W, ldx, Ho, Wo, lddy), "conv_wgrad_lowp: a tile's halo spans more than 2 GB (wgrad_lowp_supported says so)");
    Carver cv(ws);
    float* part = cv.take<float>((size_t)p.nstrips * p.taps * Cin * Cout);
    const bool f16 = math == MATH_X3 && x3_f16();
    float* amax = f16 ? cv.take<float>(2) : nullptr;
    SEG_CHECK_WS(cv.used(), ws_bytes);
    LWgradArgs a{x, dy, part, ldx, lddy, N, D, H, W, Cin, Cout, Do, Ho, Wo, p.ntx, p.nty, p.ntz, p.ntiles, p.nstrips, p.npairs, Cout / p.cob, p.taps, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0};
    if (f16) {
        if (!x_amax || !dy_amax) {
            if (hipMemsetAsync(amax, 0, 2 * sizeof(float), st) != hipSuccess) { set_error("conv_wgrad_lowp: hipMemsetAsync failed"); return MI355SEG_EHIP; }
            if (!x_amax) { tensor_amax((const float*)x, ldx, (long long)N * D * H * W, Cin, nullptr, amax, st); x_amax = amax; }
            if (!dy_amax) { tensor_amax((const float*)dy, lddy, (long long)N * Do * Ho * Wo, Cout, nullptr, amax + 1, st); dy_amax = amax + 1; }
            SEG_CHECK_LAUNCH();
        }
        a.amax_x = x_amax; a.amax_dy = dy_amax;
    }
    if (pro) {
        SEG_CHECK_ARG(f16 && stride == 1 && k == 3 && pro->al && pro->be && conv_pro_act_ok(pro->act) && x_amax, "conv_wgrad_lowp: the norm + activation prologue needs the f16x3 k3 kernels and the caller's bound on the prologue's output");
        a.pro_al = pro->al; a.pro_be = pro->be; a.pro_act = pro->act; a.pro_slope = pro->act == MI355SEG_ACT_RELU ? 0.f : pro->slope;
        SEG_CHECK_ARG(a.pro_slope >= 0.f && a.pro_slope < 1.f, "conv_wgrad_lowp: prologue slope must lie in [0, 1)");
    }
#ifdef MI355SEG_TUNE
    { static const char* e = getenv("MI355SEG_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
    const int nwg = p.nstrips * p.npairs * p.planes;
    const double vox = (double)N * Do * Ho * Wo;
    {
        ProfScope ps(PF_WGRAD, 2.0 * vox * p.taps * Cin * Cout, (math == MATH_B16 ? 2.0 : 4.0) * vox * (Cin + Cout) + 4.0 * p.taps * Cin * Cout, st);
        if (wide) {
            if (math == MATH_X3) { if (p.BX == 16) launch_wide<16, 3, float>(a, nwg, st); else launch_wide<8, 3, float>(a, nwg, st); }
            else if (k == 3) { if (p.BX == 32) launch_wide<32, 3, bf16>(a, nwg, st); else if (p.BX == 16) launch_wide<16, 3, bf16>(a, nwg, st); else launch_wide<8, 3, bf16>(a, nwg, st); }
            else { if (p.BX == 32) launch_wide<32, 5, bf16>(a, nwg, st); else if (p.BX == 16) launch_wide<16, 5, bf16>(a, nwg, st); else launch_wide<8, 5, bf16>(a, nwg, st); }
        }
        else if (k == 3 && stride == 2) {
            if (p.BX == 32) launch_lwgrad<32, 3, 1, bf16, 2>(a, nwg, st);
            else if (p.BX == 16) launch_lwgrad<16, 3, 1, bf16, 2>(a, nwg, st);
            else launch_lwgrad<8, 3, 1, bf16, 2>(a, nwg, st);
        }
        else if (k == 3) dispatch_lwgrad<3>(math, p, a, nwg, st);
        else dispatch_lwgrad<5>(math, p, a, nwg, st);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, p.nstrips, p.taps, Cin, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // namespace seg
