// conv_wgrad_b16d.hip -- weight gradient of Conv3d k3 s1 p1 / k5 s1 p2 on bf16 tensors, TWO 32-channel co blocks per workgroup,
// tiles staged by LDS-DMA (round 4).
//
//   dW[tap][ci][co] = sum over voxels v of  x[v + tap][ci] * dy[v][co]
//
// The one-block kernel (conv_wgrad_lowp.hip, NP = 1) sits at 0.33-0.36 of the bf16 matrix peak in all three bf16 legs: 1.25
// transposing LDS reads per MFMA (its x fragments feed 2 MFMAs each), MFMA busy 0.38-0.41 (r3 PMC).  Two co blocks per workgroup
// make an x fragment feed four MFMAs (0.75-0.83 reads per MFMA) and halve the x bytes staged per MFMA, but r3's attempts spilled:
// 128 accumulators + the next tile held in 36 staging registers do not fit 256.  Here the tiles never touch the register file:
// `buffer_load_dwordx4 ... lds` writes them into LDS directly (one 1 KB piece per wave-instruction, destination = wave-uniform base +
// 16 x lane, so the tile rows are UNPADDED -- r3 measured the padding at 1-3 % --; a piece outside the volume is an out-of-range
// buffer offset and lands as zeros: checked on the device, ab/dma_oob.hip), into the OTHER of two buffers while the MFMAs of the
// current tile run, one `s_waitcnt vmcnt(0)` + barrier per tile (the guide's two-buffer glds recipe).  No conversion, no staging
// VALU beyond one offset per piece (interior tiles: tile base + a per-lane constant).
//
// Workgroup = 8 waves (two per SIMD); tile = 256 output voxels (BX = 16: 4 x 4 x 16, BX = 8: 4 x 8 x 8); LDS per buffer = halo rows
// x 64 B (32 ci) + 256 rows x 128 B (64 co): 74 KB, two buffers 148 KB.  The 27 (k5: the 25 of one dz plane) tap-tiles are dealt to
// the eight waves; a wave holds its tap-tiles x 32 ci x 64 co in 128 accumulator registers for the whole strip of tiles.
#include "common.h"
#include "internal.h"
#include <type_traits>

namespace seg {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int DW_WAVES = 8, DW_THREADS = 512, DW_TPW = 4;

template <int BX, int KS>
struct DTile {
    static constexpr int HALO = KS / 2;
    static constexpr int NTAPS = KS == 3 ? 27 : KS * KS;          // taps per workgroup (k5: one dz plane)
    static constexpr int PLANES = KS == 3 ? 1 : KS;
    static constexpr int VOX = 256;
    static constexpr int TY = BX == 8 ? 8 : 4;
    static constexpr int LINES = VOX / BX, TZ = LINES / TY;
    static constexpr int HX = BX + KS - 1, HY = TY + KS - 1, HZ = KS == 3 ? TZ + 2 : TZ;
    static constexpr int NVOX = HX * HY * HZ;
    static constexpr int XROW = 64, DROW = 128;                   // bytes per voxel row: 32 ci / 64 co bf16
    static constexpr int XBLK = (NVOX * XROW + 1023) / 1024;      // 1 KB DMA pieces of the x halo (16 rows each; the last one may run past NVOX: padded)
    static constexpr int DBLK = VOX * DROW / 1024;                // ... of the dy tile (8 rows each)
    static constexpr int X_BYTES = XBLK * 1024, D_BYTES = DBLK * 1024;
    static constexpr int BUF_BYTES = X_BYTES + D_BYTES;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static constexpr int KSTEPS = VOX / 32;
    static constexpr int NBLK = XBLK + DBLK, BPW = (NBLK + DW_WAVES - 1) / DW_WAVES;     // DMA pieces per tile / per wave
    static_assert(LDS_BYTES <= 160 * 1024, "two tile buffers must fit the LDS");
};

struct DWArgs {
    const bf16* x; const bf16* dy; float* part;
    int ldx, lddy, N, D, H, W, Cin, Cout;
    int ntx, nty, ntz, ntiles, nstrips, npairs, ncob, ntaps_total;
};

__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* lds, int off0, int off1) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off1));
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
}

template <int BX, int KS>
__global__ __launch_bounds__(DW_THREADS, 2) void conv_wgrad_b16d_kernel(DWArgs a) {
    using T = DTile<BX, KS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int plane = t % T::PLANES, tp = t / T::PLANES;
    const int pair = tp % a.npairs, strip = tp / a.npairs;
    const int cib = pair / a.ncob, cob = pair % a.ncob;
    const int ci0 = cib * 32, co0 = cob * 64;

    // ---- transposing-read lane geometry (as conv_wgrad_lowp.hip): lane 4q+p of a 16-lane group addresses voxel row q, channels
    // 4p .. 4p+3 of the fragment's 16; group g holds k = {4g..4g+3} (first read) and {16+4g..} (second read) of the 32-voxel k-step
    const int li = lane & 15, q = li >> 2, p = li & 3, g = lane >> 4;
    const int chan_off = 4 * p * 2;
    const int kq_x = BX == 8 ? (g >> 1) * T::HX + (4 * (g & 1) + q) : (4 * g + q);
    const int lane_x = kq_x * T::XROW + chan_off;
    const int lane_d = T::X_BYTES + (4 * g + q) * T::DROW + chan_off;

    const bool has_last = wave + DW_WAVES * (DW_TPW - 1) <= T::NTAPS - 1;       // wave-uniform
    int abase[DW_TPW];
#pragma unroll
    for (int tt = 0; tt < DW_TPW; ++tt) {
        int tap = wave + DW_WAVES * tt;
        if (tap > T::NTAPS - 1) tap = T::NTAPS - 1;
        const int dz = KS == 3 ? tap / 9 : 0, dy = KS == 3 ? (tap / 3) % 3 : tap / KS, dx = tap % KS;
        abase[tt] = ((dz * T::HY + dy) * T::HX + dx) * T::XROW + lane_x;
    }

    f32x4 acc[DW_TPW][2][4];                                      // [tap-tile][ci half][co quarter]
#pragma unroll
    for (int tt = 0; tt < DW_TPW; ++tt)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[tt][a2][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- LDS-DMA pieces of this wave: piece index blk = wave + 8 * i; x pieces first (16 halo rows x 4 sixteen-byte parts), then dy
    // pieces (8 rows x 8 parts).  Per lane and piece: the byte offset from the tile's halo origin (x) / tile origin (dy), constant
    // over tiles; a piece row past the halo's end gets an out-of-range offset (zeros land in the padding rows).
    constexpr unsigned OOB = 0x7FFFFFF0u;
    unsigned rel[T::BPW];
#pragma unroll
    for (int i = 0; i < T::BPW; ++i) {
        const int blk = wave + DW_WAVES * i;
        if (blk < T::XBLK) {
            const int row = blk * 16 + (lane >> 2), part = lane & 3;
            const int hz = row / (T::HY * T::HX), rem = row % (T::HY * T::HX), hy = rem / T::HX, hx = rem % T::HX;
            rel[i] = row < T::NVOX ? (unsigned)((((hz * a.H + hy) * a.W + hx) * a.ldx + part * 8) * 2) : OOB;
        } else {
            const int row = (blk - T::XBLK) * 8 + (lane >> 3), part = lane & 7;
            const int line = row / BX, xx = row % BX;
            rel[i] = (unsigned)(((((line / T::TY) * a.H + line % T::TY) * a.W + xx) * a.lddy + part * 8) * 2);
        }
        asm volatile("" : "+v"(rel[i]));                     // keep it in a register: re-deriving it costs quarter-rate multiplies per tile
    }
    const long long xsample = (long long)a.D * a.H * a.W * a.ldx, dsample = (long long)a.D * a.H * a.W * a.lddy;    // elements per sample (stride 1: dy has x's extents)

    auto issue_dma = [&](int tile, int buf) {
        int mt = tile;
        const int txi = mt % a.ntx; mt /= a.ntx;
        const int tyi = mt % a.nty; mt /= a.nty;
        const int tzi = mt % a.ntz;
        const int n = mt / a.ntz;
        const int x0 = txi * BX, y0 = tyi * T::TY, z0 = tzi * T::TZ;
        const int oz = z0 + (KS == 3 ? -1 : plane - T::HALO), oy = y0 - T::HALO, ox = x0 - T::HALO;     // halo origin
        const bool interior = oz >= 0 && oy >= 0 && ox >= 0 && oz + T::HZ <= a.D && oy + T::HY <= a.H && ox + T::HX <= a.W &&
                              z0 + T::TZ <= a.D && y0 + T::TY <= a.H && x0 + BX <= a.W;
        // buffer descriptors over one sample's channel slice: offsets are relative to the sample, a piece outside the volume is
        // given an out-of-range offset and arrives as zeros
        const auto rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x + n * xsample + ci0), 0, (int)(xsample * 2), 0x00020000);
        const auto rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.dy + n * dsample + co0), 0, (int)(dsample * 2), 0x00020000);
        const int xbase = ((oz * a.H + oy) * a.W + ox) * a.ldx * 2;            // may be negative for boundary tiles (then the slow path)
        const int dbase = ((z0 * a.H + y0) * a.W + x0) * a.lddy * 2;
        unsigned char* dst = lds + buf * T::BUF_BYTES;
        // (boundary tiles: the per-piece halo coordinates are re-derived from an opaque copy of the lane index -- left visible as
        // tile-invariant they are hoisted out of the tile loop, ~60 registers that the accumulators need)
        int lane_o = lane;
        if (!interior) asm volatile("" : "+v"(lane_o));
#pragma unroll
        for (int i = 0; i < T::BPW; ++i) {
            const int blk = wave + DW_WAVES * i;              // wave-uniform
            if (blk >= T::NBLK) break;
            unsigned off;
            if (interior) {
                off = rel[i] == OOB ? OOB : (unsigned)((blk < T::XBLK ? xbase : dbase) + (int)rel[i]);
            } else if (blk < T::XBLK) {
                const int row = blk * 16 + (lane_o >> 2), part = lane_o & 3;
                const int hz = row / (T::HY * T::HX), rem = row % (T::HY * T::HX), hy = rem / T::HX, hx = rem % T::HX;
                const int gz = oz + hz, gy = oy + hy, gx = ox + hx;
                const bool ok = row < T::NVOX && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
                off = ok ? (unsigned)((((gz * a.H + gy) * a.W + gx) * a.ldx + part * 8) * 2) : OOB;
            } else {
                const int row = (blk - T::XBLK) * 8 + (lane_o >> 3), part = lane_o & 7;
                const int line = row / BX, xx = row % BX;
                const int gz = z0 + line / T::TY, gy = y0 + line % T::TY, gx = x0 + xx;
                const bool ok = gz < a.D && gy < a.H && gx < a.W;           // partial tiles: voxels outside the volume contribute nothing
                off = ok ? (unsigned)((((gz * a.H + gy) * a.W + gx) * a.lddy + part * 8) * 2) : OOB;
            }
            if (blk < T::XBLK) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(dst + blk * 1024), 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void*)(dst + blk * 1024), 16, off, 0, 0, 0);
        }
    };

    // byte offset of 32-voxel k-step ks, read t inside the x halo / the dy tile (lane part excluded)
    auto xoff = [](int ks, int t) {
        const int line = BX == 16 ? 2 * ks + t : 4 * ks + 2 * t;
        return (((line / T::TY) * T::HY + (line % T::TY)) * T::HX) * T::XROW;
    };
    auto doff = [](int ks, int t) { return (ks * 32 + 16 * t) * T::DROW; };

    // One scheduling region per (k-step, tap-tile): 8 MFMAs (2 ci halves x 4 co quarters); the x fragments of the NEXT region are
    // requested behind the first MFMAs of the current one; the four dy fragments of a k-step are single-buffered: quarter b of the
    // next k-step is requested right behind the last region's MFMAs on quarter b.
    auto tile_mfma = [&](auto ntc, int boff) {
        constexpr int NTT = decltype(ntc)::value;
        constexpr int NREG = T::KSTEPS * NTT;
        const unsigned char* base = lds + boff;
        bf16x8_t bc[4], ac[2][2];
        auto load_b = [&](int ks, int b) { bc[b] = tr_frag(base, lane_d + doff(ks, 0) + b * 32, lane_d + doff(ks, 1) + b * 32); };
        auto load_a = [&](int buf, int ks, int tt) {
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2) ac[buf][a2] = tr_frag(base, abase[tt] + xoff(ks, 0) + a2 * 32, abase[tt] + xoff(ks, 1) + a2 * 32);
        };
#pragma unroll
        for (int b = 0; b < 4; ++b) load_b(0, b);
        load_a(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NREG; ++u) {
            const int ks = u / NTT, tt = u % NTT, cur = u & 1;
            if (u + 1 < NREG) load_a(cur ^ 1, (u + 1) / NTT, (u + 1) % NTT);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2)
                    acc[tt][a2][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[cur][a2], bc[b], acc[tt][a2][b], 0, 0, 0);
                if (tt == NTT - 1 && ks + 1 < T::KSTEPS) load_b(ks + 1, b);
            }
            // spread the region's reads over its MFMAs (groups that find no read left are no-ops)
#pragma unroll
            for (int k = 0; k < 6; ++k) { __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int tile = strip, cur = 0;
    if (tile < a.ntiles) issue_dma(tile, 0);
    for (; tile < a.ntiles; tile += a.nstrips) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0) alone: this wave's pieces of the tile have landed
        __syncthreads();                                     // ... everybody's have, and everybody is done reading the other buffer
        if (tile + a.nstrips < a.ntiles) issue_dma(tile + a.nstrips, cur ^ 1);
        // every lane of every wave runs the transposing reads (they need EXEC all ones); a wave without a fourth tap issues one tap-tile fewer
        if (has_last) tile_mfma(std::integral_constant<int, DW_TPW>{}, cur * T::BUF_BYTES);
        else tile_mfma(std::integral_constant<int, DW_TPW - 1>{}, cur * T::BUF_BYTES);
        cur ^= 1;
    }

    // ---- slab store: part[strip][tap][ci][co]; a 16 x 16 tile holds ci = 4 (lane / 16) + e in its four registers, co on the lanes
#pragma unroll
    for (int tt = 0; tt < DW_TPW; ++tt) {
        const int tap = wave + DW_WAVES * tt;
        if (tap > T::NTAPS - 1) break;
        float* dst = a.part + (((long long)strip * a.ntaps_total + plane * T::NTAPS + tap) * a.Cin + ci0 + 4 * g) * a.Cout + co0 + li;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(long long)(16 * a2 + e) * a.Cout + 16 * b] = acc[tt][a2][b][e];
    }
}

struct DWPlan { int KS, BX, ntx, nty, ntz, ntiles, nstrips, npairs, taps, planes; };

bool dw_plan(int KS, int N, int D, int H, int W, int Cin, int Cout, DWPlan* p) {
    if ((KS != 3 && KS != 5) || Cin % 32 || Cout % 64 || W < 4) return false;
    // x-extent of the tile: 16 unless 8 pads less (two 85 KB buffers of the 32-wide tile do not fit the LDS)
    const long long pad16 = (long long)((W + 15) / 16) * 16, pad8 = (long long)((W + 7) / 8) * 8;
    const int BX = pad8 < pad16 ? 8 : 16;
    const int TY = BX == 8 ? 8 : 4, TZ = (256 / BX) / TY;
    p->KS = KS; p->BX = BX; p->ntx = (W + BX - 1) / BX; p->nty = (H + TY - 1) / TY; p->ntz = (D + TZ - 1) / TZ;
    p->ntiles = N * p->ntz * p->nty * p->ntx;
    p->npairs = (Cin / 32) * (Cout / 64);
    p->taps = KS * KS * KS; p->planes = KS == 3 ? 1 : KS;
    const int per_strip = p->npairs * p->planes;
    int want = 256 / per_strip;                             // one 8-wave workgroup per CU: never more than 256 in all
    long long cap = (long long)(160u << 20) / ((long long)p->taps * Cin * Cout * 4);   // keep the slab workspace <= 160 MB
    if (cap < 1) cap = 1;
    if (want > cap) want = (int)cap;
    if (want > p->ntiles) want = p->ntiles;
    if (want < 1) want = 1;
    p->nstrips = want;
    return true;
}

template <int BX, int KS>
void launch_dw(const DWArgs& a, int nwg, hipStream_t st) {
    using T = DTile<BX, KS>;
    SEG_SET_LDS((conv_wgrad_b16d_kernel<BX, KS>), T::LDS_BYTES);
    hipLaunchKernelGGL((conv_wgrad_b16d_kernel<BX, KS>), dim3(nwg), dim3(DW_THREADS), T::LDS_BYTES, st, a);
}

}  // namespace

// bf16 tensors, stride 1, Cout a multiple of 64 (two co blocks per workgroup); samples addressed through 31-bit byte offsets
bool wgrad_b16d_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int lddy) {
    if (stride != 1 || !((k == 3 && pad == 1) || (k == 5 && pad == 2)) || (ldx % 8) || (lddy % 8)) return false;
    if ((long long)D * H * W * ldx * 2 >= 0x7FFFFFF0LL || (long long)D * H * W * lddy * 2 >= 0x7FFFFFF0LL) return false;
    DWPlan p;
    return dw_plan(k, N, D, H, W, Cin, Cout, &p);
}

size_t wgrad_b16d_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k) {
    DWPlan p;
    if (!dw_plan(k, N, D, H, W, Cin, Cout, &p)) return 0;
    return align_up((size_t)p.nstrips * p.taps * Cin * Cout * sizeof(float), 256) + 1024;
}

int conv_wgrad_b16d(const bf16* dy, int lddy, const bf16* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int Cout, int k,
                    int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    DWPlan p;
    SEG_CHECK_ARG(dw_plan(k, N, D, H, W, Cin, Cout, &p), "conv_wgrad_b16d: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "conv_wgrad_b16d: pointers must be 16-byte aligned");
    Carver cv(ws);
    float* part = cv.take<float>((size_t)p.nstrips * p.taps * Cin * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    DWArgs a{x, dy, part, ldx, lddy, N, D, H, W, Cin, Cout, p.ntx, p.nty, p.ntz, p.ntiles, p.nstrips, p.npairs, Cout / 64, p.taps};
    const int nwg = p.nstrips * p.npairs * p.planes;
    const double vox = (double)N * D * H * W;
    {
        ProfScope ps(PF_WGRAD, 2.0 * vox * p.taps * Cin * Cout, 2.0 * vox * (Cin + Cout) + 4.0 * p.taps * Cin * Cout, st);
        if (k == 3) { if (p.BX == 16) launch_dw<16, 3>(a, nwg, st); else launch_dw<8, 3>(a, nwg, st); }
        else { if (p.BX == 16) launch_dw<16, 5>(a, nwg, st); else launch_dw<8, 5>(a, nwg, st); }
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, p.nstrips, p.taps, Cin, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // namespace seg
