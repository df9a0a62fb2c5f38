// conv_mfma.hip -- Conv3d k3 s1 p1 as an implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD), forward and dgrad.
//
//   M = output voxels, N = output channels, K = 27 taps x Cin.
//
// Workgroup = 4 waves (256 threads); each wave owns MB M-blocks of 32 voxels and all
// NT = 32*NBW output channels of the tile, i.e. the tile is (128*MB voxels) x NT.
// The input halo tile ((TZ+2) x (TY+2) x (BX+2) voxels x 16 input channels, NDHWC) is
// staged in LDS once per 16-channel chunk and re-read by all 27 taps as shifted
// ds_read_b128 (the shift is an immediate offset, the pitch 20 floats keeps the reads
// bank-conflict free).  Weights are pre-packed so that the B operand of four
// consecutive MFMA k-steps is ONE coalesced 16-byte global load per lane (L2-resident,
// software-prefetched one step ahead); the A operand of the same four k-steps is ONE
// ds_read_b128.  The next chunk's halo is prefetched into registers before the 864
// MFMAs of the current chunk and written to LDS after them (issue-early / write-late).
// Epilogue: bias add, NDHWC store (128 B per half-wave), optional per-channel
// sum / sum-of-squares partials for the BatchNorm that follows (deterministic two-stage).
//
// dgrad of a k3 s1 p1 conv is the same convolution with the 27 taps reversed and the
// channel roles swapped; only the weight packing differs.
#include "common.h"
#include "internal.h"

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int CK = 16;          // input channels per LDS chunk
constexpr int PITCH = CK + 4;   // floats per halo voxel in LDS (odd number of 16-B slots)

template <int BX, int MB>
struct Tile {
    static constexpr int LPB = 32 / BX;           // x-lines per 32-voxel M-block
    static constexpr int LINES = 4 * MB * LPB;    // x-lines per workgroup tile
    static constexpr int TY = 4;
    static constexpr int TZ = LINES / TY;
    static constexpr int HX = BX + 2, HY = TY + 2, HZ = TZ + 2;
    static constexpr int NVOX = HX * HY * HZ;
    static constexpr int NPIECE = NVOX * (CK / 4);            // 16-byte pieces per chunk
    static constexpr int NITER = (NPIECE + 255) / 256;
    static constexpr int LDS_BYTES = NVOX * PITCH * 4;
    static_assert(LINES % TY == 0, "tile lines must fill whole y-rows");
};

struct IgemmArgs {
    const float* x; const float* wq; const float* bias; float* y; float* spart;
    int ldx, ldy, N, D, H, W, Cin, Cout;
    int ntx, nty, ntz, nN;
};

// ---------------------------------------------------------------- weight packing
// wq[nt][chunk][tap][kk][h][j][s]  =  W[co = nt*NT + j][ci = chunk*16 + kk*8 + h*4 + s][tap]
__global__ void pack_wq_kernel(const float* __restrict__ w, float* __restrict__ wq, int Cin, int Cout, int NT, int dgrad) {
    const long long total = (long long)Cin * Cout * 27;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        int s = (int)(r % 4); r /= 4;
        int j = (int)(r % NT); r /= NT;
        int h = (int)(r % 2); r /= 2;
        int kk = (int)(r % (CK / 8)); r /= (CK / 8);
        int tap = (int)(r % 27); r /= 27;
        int chunk = (int)(r % (Cin / CK)); r /= (Cin / CK);
        int nt = (int)r;
        int co = nt * NT + j, ci = chunk * CK + kk * 8 + h * 4 + s;
        float v = dgrad ? w[((long long)ci * Cout + co) * 27 + (26 - tap)] : w[((long long)co * Cin + ci) * 27 + tap];
        wq[idx] = v;
    }
}

// ---------------------------------------------------------------- the kernel
template <int BX, int MB, int NBW>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(IgemmArgs a) {
    using T = Tile<BX, MB>;
    constexpr int NT = 32 * NBW;
    constexpr int STEP_FLOATS = 2 * NT * 4;                 // packed weights consumed per (tap, kk) step
    constexpr int CHUNK_FLOATS = 27 * (CK / 8) * STEP_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;

    // XCD-aware block -> tile map: blocks dealt round-robin to the 8 XCDs get contiguous tile ranges,
    // so halo-sharing neighbours and the N-tiles of one M-tile share an L2 (bijective for any grid size).
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ntile = t % a.nN;
    int mt = t / a.nN;
    const int txi = mt % a.ntx; mt /= a.ntx;
    const int tyi = mt % a.nty; mt /= a.nty;
    const int tzi = mt % a.ntz;
    const int n = mt / a.ntz;
    const int x0 = txi * BX, y0 = tyi * T::TY, z0 = tzi * T::TZ;
    const int n0 = ntile * NT;

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mb][nb][v] = 0.f;

    // per-lane LDS base (floats) of the A fragment for each M-block
    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = wave * MB + mb;
        const int line = m * T::LPB + i / BX, xx = i % BX;
        abase[mb] = (((line / T::TY) * T::HY + (line % T::TY)) * T::HX + xx) * PITCH + 4 * h;
    }

    const int nchunks = a.Cin / CK;
    const float* wlane = a.wq + (long long)ntile * nchunks * CHUNK_FLOATS + (h * NT + i) * 4;

    // ---- halo staging: global -> registers (issue early) -> LDS (write late)
    f32x4 stage[T::NITER];
    auto load_stage = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < T::NITER; ++it) {
            const int p = it * 256 + tid;
            const int vox = p >> 2, part = p & 3;
            const int hz = vox / (T::HY * T::HX), rem = vox % (T::HY * T::HX);
            const int hy = rem / T::HX, hx = rem % T::HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool ok = (p < T::NPIECE) && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const long long off = ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldx + chunk * CK + part * 4;
                v = *reinterpret_cast<const f32x4*>(a.x + off);
            }
            stage[it] = v;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < T::NITER; ++it) {
            const int p = it * 256 + tid;
            if (p < T::NPIECE) *reinterpret_cast<f32x4*>(lds + (p >> 2) * PITCH + (p & 3) * 4) = stage[it];
        }
    };

    load_stage(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();                 // every wave is done reading the previous chunk
        write_stage();
        __syncthreads();
        if (chunk + 1 < nchunks) load_stage(chunk + 1);

        const float* wp = wlane + (long long)chunk * CHUNK_FLOATS;
        f32x4 bcur[NBW], bnxt[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) bcur[nb] = *reinterpret_cast<const f32x4*>(wp + nb * 128);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int tapoff = ((dz * T::HY + dy) * T::HX + dx) * PITCH;
#pragma unroll
            for (int kk = 0; kk < CK / 8; ++kk) {
                const int step = tap * (CK / 8) + kk;
                if (step + 1 < 27 * (CK / 8)) {
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb)
                        bnxt[nb] = *reinterpret_cast<const f32x4*>(wp + (step + 1) * STEP_FLOATS + nb * 128);
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
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb][s], bcur[nb][s], acc[mb][nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) bcur[nb] = bnxt[nb];
            }
        }
    }

    // ---- epilogue: bias, store, optional BatchNorm partial statistics
    float ssum[NBW], ssq[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int col = n0 + nb * 32 + i;
        const float bv = a.bias ? a.bias[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = wave * MB + mb;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = (v & 3) + 8 * (v >> 2) + 4 * h;      // row of the 32x32 tile held in register v
                const int line = m * T::LPB + r / BX, xx = r % BX;
                const int gz = z0 + line / T::TY, gy = y0 + line % T::TY, gx = x0 + xx;
                const float val = acc[mb][nb][v] + bv;
                a.y[((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldy + col] = val;
                s1 += val; s2 += val * val;
            }
        }
        ssum[nb] = s1; ssq[nb] = s2;
    }
    if (a.spart) {
        __syncthreads();                 // LDS halo no longer needed
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            float s1 = ssum[nb] + __shfl_xor(ssum[nb], 32, 64);
            float s2 = ssq[nb] + __shfl_xor(ssq[nb], 32, 64);
            if (h == 0) { lds[(wave * NT + nb * 32 + i) * 2] = s1; lds[(wave * NT + nb * 32 + i) * 2 + 1] = s2; }
        }
        __syncthreads();
        if (tid < NT) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s1 += lds[(w * NT + tid) * 2]; s2 += lds[(w * NT + tid) * 2 + 1]; }
            const int mtile = t / a.nN;
            float* dst = a.spart + ((long long)mtile * a.Cout + n0 + tid) * 2;
            dst[0] = s1; dst[1] = s2;
        }
    }
}

// per-channel fp64 finalise of the epilogue partials: block per channel, fixed order
__global__ __launch_bounds__(256) void igemm_stats_finalize_kernel(const float* __restrict__ spart, int nM, int Cout,
                                                                    double* __restrict__ sum, double* __restrict__ sq) {
    __shared__ double sh[8];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int m = threadIdx.x; m < nM; m += 256) {
        const float* p = spart + ((long long)m * Cout + c) * 2;
        s1 += (double)p[0]; s2 += (double)p[1];
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[wv * 2] = s1; sh[wv * 2 + 1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sum[c] = sh[0] + sh[2] + sh[4] + sh[6];
        sq[c] = sh[1] + sh[3] + sh[5] + sh[7];
    }
}

// ---------------------------------------------------------------- host side
struct IgemmPlan { int BX, MB, NBW, TZ, nM, nN, ntx, nty, ntz; };

static bool igemm_plan(int N, int D, int H, int W, int Cin, int Cout, IgemmPlan* p) {
    if (Cin % CK || Cout % 32 || H % 4) return false;
    int BX = (W % 32 == 0) ? 32 : (W % 16 == 0) ? 16 : (W % 8 == 0) ? 8 : 0;
    if (!BX) return false;
    const int NBW = (Cout % 64 == 0) ? 2 : 1;
    const int nN = Cout / (32 * NBW);
    auto tiles = [&](int MB, int* tz) {
        int lines = 4 * MB * (32 / BX);
        *tz = lines / 4;
        if (D % *tz) return (long long)-1;
        return (long long)N * (D / *tz) * (H / 4) * (W / BX);
    };
    int tz2, tz1;
    long long m2 = tiles(2, &tz2), m1 = tiles(1, &tz1);
    int MB;
    if (m2 > 0 && m2 * nN >= 512) MB = 2;
    else if (m1 > 0) MB = 1;
    else if (m2 > 0) MB = 2;
    else return false;
    p->BX = BX; p->MB = MB; p->NBW = NBW; p->TZ = MB == 2 ? tz2 : tz1;
    p->ntx = W / BX; p->nty = H / 4; p->ntz = D / p->TZ;
    p->nM = N * p->ntz * p->nty * p->ntx; p->nN = nN;
    return true;
}

bool conv_mfma_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    if (k != 3 || stride != 1 || pad != 1) return false;
    if (ldx % 4) return false;
    IgemmPlan p;
    return igemm_plan(N, D, H, W, Cin, Cout, &p);
}

size_t conv_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    if (k != 3 || stride != 1 || pad != 1) return 0;
    size_t best = 0;
    // the same workspace must serve fwd (Cin->Cout) and dgrad (Cout->Cin)
    for (int pass = 0; pass < 2; ++pass) {
        int ci = pass ? Cout : Cin, co = pass ? Cin : Cout;
        IgemmPlan p;
        if (!igemm_plan(N, D, H, W, ci, co, &p)) continue;
        size_t need = align_up((size_t)27 * Cin * Cout * sizeof(float), 256) + align_up((size_t)p.nM * co * 2 * sizeof(float), 256) + 1024;
        if (need > best) best = need;
    }
    size_t wg = wgrad_mfma_ws_bytes(N, D, H, W, Cin, Cout);
    return best > wg ? best : wg;
}

template <int BX, int MB, int NBW>
static void launch_igemm(const IgemmArgs& a, int nwg, hipStream_t st) {
    using T = Tile<BX, MB>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_igemm_kernel<BX, MB, NBW>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<BX, MB, NBW>), dim3(nwg), dim3(256), T::LDS_BYTES, st, a);
}

int conv_fwd_mfma(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int N, int D, int H, int W,
                  int Cin, int Cout, int dgrad, double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st) {
    IgemmPlan p;
    SEG_CHECK_ARG(igemm_plan(N, D, H, W, Cin, Cout, &p), "conv_fwd_mfma: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0, "conv_fwd_mfma: input pointer must be 16-byte aligned");
    Carver cv(ws);
    float* wq = cv.take<float>((size_t)27 * Cin * Cout);
    float* spart = ssum ? cv.take<float>((size_t)p.nM * Cout * 2) : nullptr;
    SEG_CHECK_WS(cv.used(), ws_bytes);
    {
        long long total = (long long)27 * Cin * Cout;
        int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        hipLaunchKernelGGL(pack_wq_kernel, dim3(grid), dim3(256), 0, st, w, wq, Cin, Cout, 32 * p.NBW, dgrad);
        SEG_CHECK_LAUNCH();
    }
    IgemmArgs a{x, wq, bias, y, spart, ldx, ldy, N, D, H, W, Cin, Cout, p.ntx, p.nty, p.ntz, p.nN};
    const int nwg = p.nM * p.nN;
    const double vox = (double)N * D * H * W;
    {
        ProfScope ps(PF_IGEMM, 2.0 * vox * 27.0 * Cin * Cout, 4.0 * (vox * (Cin + Cout) + 27.0 * Cin * Cout), st);
#define IGEMM_CASE(bx, mb, nbw) \
        if (p.BX == bx && p.MB == mb && p.NBW == nbw) launch_igemm<bx, mb, nbw>(a, nwg, st)
        IGEMM_CASE(32, 2, 2); else IGEMM_CASE(32, 2, 1); else IGEMM_CASE(32, 1, 2); else IGEMM_CASE(32, 1, 1);
        else IGEMM_CASE(16, 2, 2); else IGEMM_CASE(16, 2, 1); else IGEMM_CASE(16, 1, 2); else IGEMM_CASE(16, 1, 1);
        else IGEMM_CASE(8, 2, 2); else IGEMM_CASE(8, 2, 1); else IGEMM_CASE(8, 1, 2); else IGEMM_CASE(8, 1, 1);
#undef IGEMM_CASE
        SEG_CHECK_LAUNCH();
    }
    if (ssum) {
        hipLaunchKernelGGL(igemm_stats_finalize_kernel, dim3(Cout), dim3(256), 0, st, spart, p.nM, Cout, ssum, ssq);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

}  // namespace seg
