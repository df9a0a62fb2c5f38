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
#include "igemm_kernel.h"
#include "pack.h"
#include <stdlib.h>
#include <initializer_list>

namespace seg {

// ---------------------------------------------------------------- weight packing
// fp32 policy:  wq[nt][chunk][tap][kk][h][j][s]        =  B[k = chunk*CK + kk*8 + h*4 + s][n = nt*NT + j]
// bf16 policies: wq[nt][chunk][tap][kstep][plane][h][j][e] = plane of B[k = chunk*CK + kstep*16 + 8h + e][n = nt*NT + j]
//               (NP = 3: planes h / m / l of the bf16x6 split; NP = 1: the bf16 rounding)                    with
//   mode 0 (conv fwd):     B = W[co = n][ci = k][tap]                                    W: (Cout, Cin, T)
//   mode 1 (conv dgrad):   B = W[ci_f = n .. swapped roles, taps reversed]              W: (Cin_k, Cout_k, T)
//   mode 2 (convT fwd):    n = (tapn, co): B = Wt[ci][co][tapn]                          Wt: (Cin, Cout, 8), T = 1
//   mode 3 (convT dgrad):  chunk = (tapk, cc): B = Wt[ci = n][co = cc*CK + ..][tapk]     Wt: (Cin_f, Cout_f, 8), T = 1
//   mode 5 (gather fwd):   chunk = (tap, cc):  B = W[co = n][ci = cc*CK + ..][tap]      W: (Cout, Cin, TW), aux = Cin, T = 1
//   mode 6 (gather dgrad): chunk = (slot, cc): B = W[co = cc*CK + ..][ci = n][taps.t[slot]]   aux = Cout, T = 1
// oscale (optional): per-output-channel factor folded into the packed weights -- eval-mode BatchNorm (gamma / sqrt(var + eps))
// for the fused inference forward; n is the GEMM column = output channel in the modes that use it (0: conv forward)
__global__ void pack_wq_kernel(const float* __restrict__ w, float* __restrict__ wq, int K, int Nn, int T, int NT, int mode, int aux, int CK,
                               int TW, TapList taps, const float* __restrict__ oscale) {
    const long long total = (long long)K * Nn * T;        // K = channels in the GEMM K dim (taps of mode 3 included)
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        int s = (int)(r % 4); r /= 4;
        int j = (int)(r % NT); r /= NT;
        int h = (int)(r % 2); r /= 2;
        int kk = (int)(r % (CK / 8)); r /= (CK / 8);
        int tap = (int)(r % T); r /= T;
        int chunk = (int)(r % (K / CK)); r /= (K / CK);
        int nt = (int)r;
        wq[idx] = pack_src(w, mode, nt * NT + j, chunk * CK + kk * 8 + h * 4 + s, tap, K, Nn, T, aux, TW, taps) * (oscale ? oscale[nt * NT + j] : 1.f);
    }
}

// (the low-precision form of this layout: pack_wq_lowp_body, prepack.hip)
// conv_x3s.hip layout: wq[nt][chunk][unit = (K-step s, 32-channel half nh)][plane][16-channel tile t2][lane][8]; lane = (c, g):
// element e = plane of W[co = nt*NT + nh*32 + t2*16 + c][ci = chunk*16 + 8*(g&1) + e][tap = x3s_pair_tap(s, g>>1)]  (tap 27: zero)
__global__ void pack_wq_x3s_kernel(const float* __restrict__ w, bf16* __restrict__ wq, int K, int Nn, int NBW, int mode, const float* __restrict__ oscale) {
    const int NT = 32 * NBW, NU = X3S_NPAIR * NBW, nch = K / 16;
    const long long total = (long long)(Nn / NT) * nch * NU * 1024;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long q = idx;
        const int e = (int)(q % 8); q /= 8;
        const int lane = (int)(q % 64); q /= 64;
        const int t2 = (int)(q % 2); q /= 2;
        const int u = (int)(q % NU); q /= NU;
        const int chunk = (int)(q % nch); q /= nch;
        const int nt = (int)q;
        const int s = u / NBW, nh = u % NBW, g = lane >> 4, c = lane & 15;
        const int tap = x3s_pair_tap(s, g >> 1);
        const int co = nt * NT + nh * 32 + t2 * 16 + c;
        const float v = tap < 27 ? pack_src(w, mode, co, chunk * 16 + 8 * (g & 1) + e, tap, K, Nn, 27, 0, 27, TapList{}) * (oscale ? oscale[co] : 1.f) : 0.f;
        bf16 bh, bm, bl;
        split3(v, bh, bm, bl);
        const long long base = (((long long)nt * nch + chunk) * NU + u) * 3072 + t2 * 512 + lane * 8 + e;
        wq[base] = bh; wq[base + 1024] = bm; wq[base + 2048] = bl;
    }
}

// the same for the f16x3 form: two fp16 planes of w * 2^sw
__global__ void pack_wq_x3s_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ wq, int K, int Nn, int NBW, int mode, const float* __restrict__ oscale,
                                       const float* __restrict__ amax_w) {
    const int NT = 32 * NBW, NU = X3S_NPAIR * NBW, nch = K / 16;
    const long long total = (long long)(Nn / NT) * nch * NU * 1024;
    const float sc = pow2f(f16x_scale_exp(*amax_w));
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long q = idx;
        const int e = (int)(q % 8); q /= 8;
        const int lane = (int)(q % 64); q /= 64;
        const int t2 = (int)(q % 2); q /= 2;
        const int u = (int)(q % NU); q /= NU;
        const int chunk = (int)(q % nch); q /= nch;
        const int nt = (int)q;
        const int s = u / NBW, nh = u % NBW, g = lane >> 4, c = lane & 15;
        const int tap = x3s_pair_tap(s, g >> 1);
        const int co = nt * NT + nh * 32 + t2 * 16 + c;
        const float v = tap < 27 ? pack_src(w, mode, co, chunk * 16 + 8 * (g & 1) + e, tap, K, Nn, 27, 0, 27, TapList{}) * (oscale ? oscale[co] : 1.f) * sc : 0.f;
        _Float16 bh, bl;
        split2h(v, bh, bl);
        const long long base = (((long long)nt * nch + chunk) * NU + u) * 2048 + t2 * 512 + lane * 8 + e;
        wq[base] = bh; wq[base + 1024] = bl;
    }
}

// max |x| over rows x C elements at pitch ld (C, ld multiples of 4, x 16-byte aligned; else the scalar loop), times rowscale[row / rows_per_scale]
// when given (the folded inference weights w * oscale[co]), max-combined into *slot (an ordered compare of the bit patterns of
// non-negative floats: order-independent, so the result is reproducible).  The caller zeroes the slot.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, int ld, long long rows, int C, const float* __restrict__ rowscale, unsigned* __restrict__ slot) {
    float m = 0.f;
    if ((C % 4) == 0 && (ld % 4) == 0 && ((uintptr_t)x % 16) == 0) {
        const int cw = C / 4;
        const long long total = rows * cw;
        for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
            const long long r = idx / cw;
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + r * ld + (idx - r * cw) * 4);
            const float sc = rowscale ? fabsf(rowscale[r]) : 1.f;
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) * sc);
        }
    } else {
        const long long total = rows * C;
        for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
            const long long r = idx / C;
            m = fmaxf(m, fabsf(x[r * ld + (idx - r * C)]) * (rowscale ? fabsf(rowscale[r]) : 1.f));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        // fmaxf drops NaNs: a tensor that holds one must still produce a non-finite result downstream, which the kernels that
        // consume the slot get from the NaN operand itself (h = NaN), whatever the scale
        atomicMax(slot, __builtin_bit_cast(unsigned, m));
    }
}
void tensor_amax(const float* x, int ld, long long rows, int C, const float* rowscale, float* slot, hipStream_t st) {
    const long long work = rows * (long long)C / 4;
    const int grid = (int)(work / 1024 < 1 ? 1 : (work / 1024 > 2048 ? 2048 : work / 1024));
    hipLaunchKernelGGL(amax_kernel, dim3(grid), dim3(256), 0, st, x, ld, rows, C, rowscale, reinterpret_cast<unsigned*>(slot));
}

// conv_b16s.hip layout: wq[nt][chunk][K-step s][16-channel tile tt][lane][8]; lane = (c, g), c = 4 g' + e':
// element e = bf16 of W[co = nt*NT + 32 (tt / 2) + 8 g' + 4 (tt % 2) + e'][ci = chunk*16 + 8 (g & 1) + e][tap = 2 s + (g >> 1)]  (tap >= T: zero)
__global__ void pack_wq_b16s_kernel(const float* __restrict__ w, bf16* __restrict__ wq, int K, int Nn, int T, int NT, int mode, const float* __restrict__ oscale) {
    const int nstep = (T + 1) / 2, nch = K / 16, ntt = NT / 16;
    const long long total = (long long)(Nn / NT) * nch * nstep * ntt * 512;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long q = idx;
        const int e = (int)(q % 8); q /= 8;
        const int lane = (int)(q % 64); q /= 64;
        const int tt = (int)(q % ntt); q /= ntt;
        const int s = (int)(q % nstep); q /= nstep;
        const int chunk = (int)(q % nch); q /= nch;
        const int nt = (int)q;
        const int g = lane >> 4, c = lane & 15;
        const int tap = 2 * s + (g >> 1);
        const int co = nt * NT + 32 * (tt >> 1) + 8 * (c >> 2) + 4 * (tt & 1) + (c & 3);
        wq[idx] = tap < T ? (bf16)(pack_src(w, mode, co, chunk * 16 + 8 * (g & 1) + e, tap, K, Nn, T, 0, T, TapList{}) * (oscale ? oscale[co] : 1.f)) : (bf16)0.f;
    }
}

// The two hot packings above, tiled (pack_tiled_body, prepack.hip): the descriptor of one
static PackDesc tiled_desc(int layout, const float* w, void* wq, int K, int Nn, int T, int P, int mode, const float* oscale, const float* amax_w) {
    PackDesc d{};
    d.kind = PD_TILED; d.layout = layout;
    // enough workgroups for the narrow layers, short serial work per workgroup for the 125-tap ones (runs of 4 T floats keep the 16-byte alignment)
    d.nb = (T > 27 || (long long)(Nn / 8) * (K / 16) < 256) ? 4 : 8;
    d.w = w; d.dst = wq; d.amax = amax_w; d.oscale = oscale;
    d.K = K; d.Nn = Nn; d.T = T; d.P = P; d.mode = mode;
    d.amax_elems = layout == 2 ? (long long)K * Nn * T : 0;
    return d;
}
static PackDesc lowp_desc(int np, const float* w, void* wq, int K, int Nn, int T, int NT, int mode, int aux, int CK, int TW, const TapList& taps, const float* oscale) {
    PackDesc d{};
    d.kind = PD_LOWP; d.np = np; d.w = w; d.dst = wq; d.oscale = oscale;
    d.K = K; d.Nn = Nn; d.T = T; d.P = NT; d.mode = mode; d.aux = aux; d.CK = CK; d.TW = TW; d.taps = taps;
    return d;
}

static int pack_grid(long long total) { return (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256); }
// MATH_X3: sized for the 28-tap layout of conv_x3s.hip (27 taps + one zero tap), which is the larger of its two packings
// MATH_B16 likewise for conv_b16s.hip (taps paired: one zero tap when the tap count is odd)
static size_t wq_bytes(int math, size_t nelem) { return math == MATH_F32 ? nelem * 4 : (math == MATH_X3 ? (nelem + nelem / 27 + 64) * 6 : (nelem + nelem / 27 + 64) * 2); }
// the generic layouts; the low-precision ones with fewer than 2^31 elements through a descriptor (*desc, when asked for, tells the caller
// what was launched so that a recorded step can replay it: kind 0 = not replayable)
static void launch_pack(int math, const float* w, void* wq, int K, int Nn, int T, int NT, int mode, int aux, int CK, int TW, const TapList& taps,
                        hipStream_t st, const float* oscale = nullptr, PackDesc* desc = nullptr) {
    const int grid = pack_grid((long long)K * Nn * T);
    if (desc) desc->kind = 0;
    if (math == MATH_F32) hipLaunchKernelGGL(pack_wq_kernel, dim3(grid), dim3(256), 0, st, w, (float*)wq, K, Nn, T, NT, mode, aux, CK, TW, taps, oscale);
    else {
        PackDesc d = lowp_desc(math == MATH_X3 ? 3 : 1, w, wq, K, Nn, T, NT, mode, aux, CK, TW, taps, oscale);
        d.layout = (long long)K * Nn * T < 0x7F000000LL ? 0 : 1;          // (1: 64-bit element indices)
        pack_launch(d, st);
        if (desc) *desc = d;
    }
}

// y[r][c] = bias[c] + sum_k part[k][r][c]   (split-K second stage; fixed order)
template <typename OT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ksplit, long long stride,
        const float* __restrict__ bias, OT* __restrict__ y, int ldy, long long rows, int C, int act, float slope) {
    const int cw = C / 4;
    const long long total = rows * cw;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / cw;
        const int c = (int)(idx % cw) * 4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (bias) s = *reinterpret_cast<const f32x4*>(bias + c);
        for (int k = 0; k < ksplit; ++k) s += *reinterpret_cast<const f32x4*>(part + k * stride + r * C + c);
        if (act) { s[0] = act_apply(s[0], act, slope); s[1] = act_apply(s[1], act, slope); s[2] = act_apply(s[2], act, slope); s[3] = act_apply(s[3], act, slope); }
        st4(y + r * ldy + c, s);
    }
}

// per-channel fp64 combine of the per-tile (sum, M2, n) triples: block per channel, fixed order.
// Emits sum(y) and sum(y^2) as doubles (sum^2 = M2_total + N * mean^2 is exact enough in fp64 for the
// var = E[y^2] - mean^2 that mi355seg_norm_stats_from_sums_f32 forms afterwards).
__global__ __launch_bounds__(256) void igemm_stats_finalize_kernel(const float* __restrict__ spart, int nM, int Cout,
                                                                    double* __restrict__ sum, double* __restrict__ sq) {
    __shared__ double sh[8];
    __shared__ double gmean;
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double s1 = 0.0, n = 0.0;
    for (int m = threadIdx.x; m < nM; m += 256) {
        const float* p = spart + ((long long)m * Cout + c) * 3;
        s1 += (double)p[0]; n += (double)p[2];
    }
    s1 = wave_sum(s1); n = wave_sum(n);
    if (lane == 0) { sh[wv * 2] = s1; sh[wv * 2 + 1] = n; }
    __syncthreads();
    const double S = sh[0] + sh[2] + sh[4] + sh[6], Nn = sh[1] + sh[3] + sh[5] + sh[7];
    if (threadIdx.x == 0) gmean = S / Nn;
    __syncthreads();
    const double mean = gmean;
    double m2 = 0.0;
    for (int m = threadIdx.x; m < nM; m += 256) {
        const float* p = spart + ((long long)m * Cout + c) * 3;
        const double nt = (double)p[2], d = (double)p[0] / nt - mean;
        m2 += (double)p[1] + nt * d * d;
    }
    m2 = wave_sum(m2);
    __syncthreads();
    if (lane == 0) sh[wv] = m2;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double M2 = sh[0] + sh[1] + sh[2] + sh[3];
        sum[c] = S;
        sq[c] = M2 + Nn * mean * mean;
    }
}

// ---------------------------------------------------------------- host side
static int pick_ck(int math, int KS, int Kc) {
    if (KS == 5) return math == MATH_F32 ? 8 : 16;
    if (KS == 1 && Kc % 64 == 0) return 64;
    return 16;
}

// Kc = GEMM K channels per input tap (multiple of CK), Nc = GEMM N per output tap (multiple of 32)
static bool x3s_enabled();
static bool igemm_plan(int math, int KS, int N, int D, int H, int W, int Kc, int Nc, int ntaps_out, IgemmPlan* p) {
    if (KS != 1 && KS != 3 && KS != 5) return false;
    // the split-precision igemm serves the k3 layers.  (k5 was measured: three planes of the 5^3 halo are 86-110 KB, one workgroup
    // per CU; 150 TFLOP/s on 32->32 @128^3 against 120 for the fp32 MFMA, but slower on the deep layers: V-Net fp32 step 79.6 -> 96.9 ms)
    if (math == MATH_X3 && KS != 3) return false;
    const int CK = pick_ck(math, KS, Kc);
    // ConvT with a narrow Cout: tile the flat (child, cout) axis instead of each child's channels
    const bool flat = ntaps_out > 1 && (Nc % 32) != 0 && ((long long)Nc * ntaps_out) % 32 == 0;
    if (Kc % CK || (Nc % 32 && !flat) || W < 4) return false;          // degenerate volumes stay on the generic path
    // x-extent of an M-block: the candidate with the least padding (ties -> the wider one).  The three bf16 planes of the
    // split-precision tile are 112 bytes per voxel: only the 16- and 8-wide tiles fit twice per CU.
    int BX = 0; long long best = -1;
    for (int bx : {32, 16, 8}) {
        if (math == MATH_X3 && bx == 32) continue;
        // f16x3 (r5): the 16-wide tiles carry the conv_x3s kernels (three fp16 MFMAs per product); an 8-wide volume on the generic tiles
        // runs bf16x6 (six) on a slower loop -- half-empty 16-wide tiles are the cheaper choice (profiles/r05_x3s_small_volumes_ab.log)
        if (math == MATH_X3 && bx == 8 && x3s_enabled() && x3_f16() && KS == 3 && Kc % 16 == 0 && W >= 8) continue;
        // bf16 k5: the 5^3 halo of a two-block tile fits twice per CU only for the 16- and 8-wide tiles (61 / 55 KB vs 83 KB),
        // and two M-blocks per wave halve both the halo overfetch (11x -> 5x) and the weight-fragment loads per MFMA
        if (math == MATH_B16 && KS == 5 && bx == 32 && W % 16 == 0) continue;
        long long padded = (long long)((W + bx - 1) / bx) * bx;
        if (best < 0 || padded < best) { best = padded; BX = bx; }
    }
    const int NBW = ((flat ? Nc * ntaps_out : Nc) % 64 == 0) ? 2 : 1;
    const int nN = flat ? Nc * ntaps_out / (32 * NBW) : Nc / (32 * NBW) * ntaps_out;
    p->flat = flat ? 1 : 0;
    auto tiles = [&](int MB, int* tz) {
        int lines = 4 * MB * (32 / BX);
        *tz = lines / 4;
        return (long long)N * ((D + *tz - 1) / *tz) * ((H + 3) / 4) * ((W + BX - 1) / BX);
    };
    auto waste = [&](int tz) { return (double)(((D + tz - 1) / tz) * tz) / D; };
    int tz2, tz1;
    long long m2 = tiles(2, &tz2), m1 = tiles(1, &tz1);
    (void)m1;
    int MB;
#ifdef MI355SEG_TUNE
    static const char* force = getenv("MI355SEG_IGEMM_MB");          // tuning knob: 1 / 2 force that M-block count
#else
    constexpr const char* force = nullptr;
#endif
    int tz3, tz4;
    const long long m3 = tiles(3, &tz3), m4 = tiles(4, &tz4);
    if (KS == 5 && !(math == MATH_B16 && BX != 32 && m2 * nN >= 512 && waste(tz2) <= waste(tz1) * 1.2)) MB = 1;   // the 5^3 halo of a 2-block tile does not fit twice per CU
    else if (KS == 5) MB = 2;
    else if (force && force[0] == '1') MB = 1;
    // one 32-channel N-block per tile (Cout = 32 at full resolution): a 384-voxel tile still fits twice per CU (2 x 81.6 KB),
    // cuts the halo overfetch from 3.2x to 2.7x and spreads the per-tile prologue / epilogue over 1.5x the MFMAs (+3 %)
    else if (math == MATH_F32 && !(force && force[0] == '2') && KS == 3 && BX == 32 && NBW == 1 && m3 * nN >= 2048 && waste(tz3) <= 1.05) MB = 3;
    // bf16 k3: four M-blocks per wave (512-voxel tiles, 59 KB of LDS, still two workgroups per CU): every weight fragment a
    // wave loads feeds twice the MFMAs -- at one MFMA per k-step the four waves' re-loads of the same fragments otherwise
    // saturate the L1 path -- and the halo overfetch drops from 3.2x to 2.4x
    else if (math == MATH_B16 && !(force && force[0] == '2') && KS == 3 && BX == 32 && NBW == 1 && m4 * nN >= 1024 && waste(tz4) <= 1.05) MB = 4;
    else if (math == MATH_B16 && !(force && force[0] == '2') && KS == 3 && BX == 32 && NBW == 2 && m3 * nN >= 1024 && waste(tz3) <= 1.05) MB = 3;
    else if (m2 * nN >= 512 && waste(tz2) <= waste(tz1) * 1.2) MB = 2;
    else MB = 1;
    p->KS = KS; p->CK = CK; p->BX = BX; p->MB = MB; p->NBW = NBW; p->WN = 1; p->NT = 32 * NBW; p->TY = 4;
    p->TZ = MB == 4 ? tz4 : (MB == 3 ? tz3 : (MB == 2 ? tz2 : tz1));
    int nNp = nN;
    // bf16 k3 at full width, three or more M-blocks per wave: sliding-window tiles, TY = MB lines of one z-slab per wave
    // (Tile::SLIDE), 4 / WN slabs per tile
    auto slide_waste = [&](int ty, int tz) { return (double)(((H + ty - 1) / ty) * ty) * (((D + tz - 1) / tz) * tz) / ((double)H * D); };
    // 64-channel tiles: 2 (M) x 2 (N) wave grid, six M-blocks per wave (32 x 6 x 2 voxels; eight blocks = 128 accumulator
    // registers spill)
    if (math == MATH_B16 && KS == 3 && BX == 32 && NBW == 2 && !flat && ntaps_out == 1 && m3 * nN >= 1024 && slide_waste(6, 2) <= 1.05) {
        p->WN = 2; p->NBW = 1; p->MB = 6; p->NT = 64;
    }
    if (math == MATH_B16 && KS == 3 && BX == 32 && p->MB >= 3) {
        p->TY = p->MB; p->TZ = 4 / p->WN;
        if (slide_waste(p->TY, p->TZ) > 1.10) {                      // ragged extents: back to two blocks per wave
            p->MB = 2; p->WN = 1; p->NBW = NBW; p->NT = 32 * NBW; p->TY = 4; p->TZ = tz2;
        }
    }
    p->ntx = (W + BX - 1) / BX; p->nty = (H + p->TY - 1) / p->TY; p->ntz = (D + p->TZ - 1) / p->TZ;
    p->nM = N * p->ntz * p->nty * p->ntx; p->nN = nNp;
    if (p->WN == 2) return true;
    if (math != MATH_F32 && !igemm_lowp_has(math, KS, CK, BX, MB)) return false;
    return true;
}

bool conv_pro_act_ok(int act) { return act == MI355SEG_ACT_RELU || act == MI355SEG_ACT_LRELU; }    // max(z, slope z), slope in [0, 1): positively homogeneous

static bool igemm_shape_ok(int k, int stride, int pad) {
    return stride == 1 && ((k == 1 && pad == 0) || (k == 3 && pad == 1) || (k == 5 && pad == 2));
}

bool conv_mfma_supported(int math, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    const int al = math == MATH_B16 ? 8 : 4;                           // 16-byte staged pieces
    if (!igemm_shape_ok(k, stride, pad) || (ldx % al)) return false;
    IgemmPlan p;
    return igemm_plan(math, k, N, D, H, W, Cin, Cout, 1, &p);
}

// would conv_fwd_mfma(MATH_X3, ...) run the f16x3 form of conv_x3s.hip on this geometry (16-byte aligned pointers, pitches = channels)?
bool conv_fwd_takes_amax(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    if (!x3_f16() || !igemm_shape_ok(k, stride, pad) || k != 3 || (Cin % 4) || (Cout % 4)) return false;
    IgemmPlan p;
    if (!igemm_plan(MATH_X3, k, N, D, H, W, Cin, Cout, 1, &p)) return false;
    return x3s_plan_ok(p, nullptr, Cin, nullptr, Cout, (long long)D * H * W);
}

size_t conv_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    if (!igemm_shape_ok(k, stride, pad)) return 0;
    size_t best = 0;
    const size_t T = (size_t)k * k * k;
    // the same workspace must serve fwd (Cin->Cout) and dgrad (Cout->Cin) under every arithmetic policy
    for (int math : {MATH_F32, MATH_X3, MATH_B16})
        for (int pass = 0; pass < 2; ++pass) {
            int ci = pass ? Cout : Cin, co = pass ? Cin : Cout;
            IgemmPlan p;
            if (!igemm_plan(math, k, N, D, H, W, ci, co, 1, &p)) continue;
            const int ks = pick_ksplit(p.nM * p.nN, ci / p.CK);
            size_t need = align_up(wq_bytes(math, T * Cin * Cout), 256) + align_up((size_t)p.nM * co * 3 * sizeof(float), 256) + part_reduce_ws_bytes(co) +
                          (ks > 1 ? align_up((size_t)ks * N * D * H * W * co * sizeof(float), 256) + colsum_ws_bytes(co) : 0) + 1024;
            if (need > best) best = need;
        }
    for (int pass = 0; pass < 2; ++pass) {            // the bf16 16x16x32 tiles (conv_b16s.hip) cut more, smaller M-tiles
        const size_t need = b16s_ws_bytes(N, D, H, W, pass ? Cout : Cin, pass ? Cin : Cout, k);
        if (need > best) best = need;
    }
    size_t wg = (k == 3 || k == 5) ? wgrad_mfma_ws_bytes(N, D, H, W, Cin, Cout, k) : (k == 1 ? pw_wgrad_ws_bytes((long long)N * D * H * W, Cin, Cout, 1) : 0);
    if ((k == 3 || k == 5) && wgrad_lowp_ws_bytes(N, D, H, W, Cin, Cout, k) > wg) wg = wgrad_lowp_ws_bytes(N, D, H, W, Cin, Cout, k);
    return best > wg ? best : wg;
}

template <int KS, int CK, bool ALLOW_MB2>
static void dispatch_igemm_ck(const IgemmPlan& p, const IgemmArgs& a, int nwg, hipStream_t st) {
#define IGEMM_CASE(bx, mb, nbw) \
    if (p.BX == bx && p.MB == mb && p.NBW == nbw) launch_igemm<MATH_F32, KS, bx, mb, nbw, CK>(a, nwg, st)
    if constexpr (ALLOW_MB2 && KS == 3) { IGEMM_CASE(32, 3, 1); }
    if (ALLOW_MB2) {
        IGEMM_CASE(32, 2, 2); else IGEMM_CASE(32, 2, 1); else IGEMM_CASE(16, 2, 2); else IGEMM_CASE(16, 2, 1);
        else IGEMM_CASE(8, 2, 2); else IGEMM_CASE(8, 2, 1);
    }
    IGEMM_CASE(32, 1, 2); else IGEMM_CASE(32, 1, 1); else IGEMM_CASE(16, 1, 2); else IGEMM_CASE(16, 1, 1);
    else IGEMM_CASE(8, 1, 2); else IGEMM_CASE(8, 1, 1);
#undef IGEMM_CASE
}

static int tile_block(int nt) { return nt % 4 == 0 ? 4 : (nt % 2 == 0 ? 2 : 1); }

// mi355seg_set_x3_shape(32) keeps the bf16x6 convolutions on the 32x32x16-MFMA tiles of the generic kernel
static bool x3s_enabled() { return x3_shape() == 16; }

static void dispatch_igemm(int math, const IgemmPlan& p, const IgemmArgs& a_in, int nwg, hipStream_t st, bool x3s = false) {
    IgemmArgs a = a_in;
    a.total = nwg;
#ifdef MI355SEG_TUNE
    static const char* flat_walk = getenv("MI355SEG_IGEMM_LINEAR_WALK");      // A/B knob: 1 = plain x, y, z tile order
#else
    constexpr const char* flat_walk = nullptr;
#endif
    a.by = flat_walk ? 1 : tile_block(a.nty);
    a.bz = flat_walk ? 1 : (a.ntz >= 4 ? 4 : tile_block(a.ntz));      // z may be ragged (last brick row shorter), y must divide
    if (x3s) { dispatch_x3s(p, a, nwg, st); return; }
    if (math != MATH_F32) { dispatch_igemm_lowp(math, p, a, nwg, st); return; }
    if (p.KS == 3) dispatch_igemm_ck<3, 16, true>(p, a, nwg, st);
    else if (p.KS == 5) dispatch_igemm_ck<5, 8, false>(p, a, nwg, st);
    else if (p.CK == 64) dispatch_igemm_ck<1, 64, true>(p, a, nwg, st);
    else dispatch_igemm_ck<1, 16, true>(p, a, nwg, st);
}

#ifdef MI355SEG_TUNE
static int dbg_flags() { static const char* e = getenv("MI355SEG_DBG"); return e ? atoi(e) : 0; }
#else
static constexpr int dbg_flags() { return 0; }
#endif


static size_t esize(int math) { return math == MATH_B16 ? 2 : 4; }
static double matrix_bytes(int math, double act_elems, double w_elems) { return esize(math) * act_elems + 4.0 * w_elems; }

// k in {1, 3, 5}, pad = k/2, stride 1.  x / y are fp32 (MATH_F32, MATH_X3) or bf16 (MATH_B16) NDHWC tensors.
// oscale / act / slope (inference): y = act(conv(x, w * oscale[co]) + bias[co]) -- eval-mode BatchNorm folded into the packed weights
// and the bias vector, the activation applied in the epilogue (or in the split-K reduce), no normalise pass
int conv_fwd_mfma(int math, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int N, int D, int H, int W,
                  int Cin, int Cout, int k, int dgrad, double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st,
                  const float* oscale, int act, float slope, BnBwdEpi* bne, const float* x_amax, const float* w_amax, const void* res, int ldres,
                  int* res_fused, const ConvPro* pro, float* y_amax) {
    IgemmPlan p;
    if (res_fused) *res_fused = 0;
    SEG_CHECK_ARG(igemm_plan(math, k, N, D, H, W, Cin, Cout, 1, &p), "conv_fwd_mfma: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0, "conv_fwd_mfma: input pointer must be 16-byte aligned");
    const int T = k * k * k;
    // bf16 tensors, k3 / k5, enough tiles: the 16x16x32-MFMA kernel (conv_b16s.hip) with its own tiling and weight packing
    B16sPlan bp;
    const bool b16s = math == MATH_B16 && b16s_plan(k, N, D, H, W, Cin, Cout, x, ldx, y, ldy, &bp);
    if (b16s) { p.CK = 16; p.nM = bp.nM; p.nN = bp.nN; p.ntx = bp.ntx; p.nty = bp.nty; p.ntz = bp.ntz; p.NT = bp.NT; }
    const int nchunks = Cin / p.CK;
    const long long nvox = (long long)N * D * H * W;
    const int ksplit = b16s ? bp.ksplit : ((ldy % 4 == 0) ? pick_ksplit(p.nM * p.nN, nchunks) : 1);
    // fp32 tensors under f16x3: the large layers run conv_x3w's 8 x 4 x 16 tiles (fewer, larger M-tiles: decided before anything is sized by nM)
    const bool x3w = math == MATH_X3 && x3s_enabled() && x3_f16() && x3s_plan_ok(p, x, ldx, y, ldy, (long long)D * H * W) && x3w_plan(p, N, D, H, W, Cin, Cout, ksplit);
    Carver cv(ws);
    void* wq = cv.take<char>(wq_bytes(math, (size_t)T * Cin * Cout));
    float* spart = (ssum && ksplit == 1) ? cv.take<float>((size_t)p.nM * Cout * 3) : nullptr;
    float* slabs = ksplit > 1 ? cv.take<float>((size_t)ksplit * nvox * Cout) : nullptr;
    // fp32 tensors, bf16x6, 16-wide tiles: the 16x16x32-MFMA kernel (conv_x3s.hip) with its own weight packing
    const bool x3s = x3w || (math == MATH_X3 && x3s_enabled() && x3s_plan_ok(p, x, ldx, ksplit > 1 ? (void*)slabs : y, ksplit > 1 ? Cout : ldy, (long long)D * H * W));
    // the BatchNorm-backward sums of the layer in front ride in that kernel's epilogue (whole-K launches only)
    const bool bn_epi = bne && x3s && ksplit == 1 && !ssum && !bias && !act && (bne->ldx % 4) == 0 && ((uintptr_t)bne->x % 16) == 0 && Cout % 4 == 0;
    float* bnpart = bn_epi ? cv.take<float>((size_t)p.nM * Cout * 2) : nullptr;
    double* rtmp = (bn_epi || spart) ? reinterpret_cast<double*>(cv.take<char>(part_reduce_ws_bytes(Cout))) : nullptr;
    // f16x3: the two-piece fp16 split needs max |x| and max |w| (device scalars; measured here unless the caller hands them over)
    const bool f16 = x3s && x3_f16();
    float* amax = f16 ? cv.take<float>(2) : nullptr;
    size_t tail = cv.used();
    SEG_CHECK_WS(tail + ((ssum && ksplit > 1) ? colsum_ws_bytes(Cout) : 0), ws_bytes);
    const bool w16 = ((uintptr_t)w % 16) == 0;                     // the tiled packings read W in 16-byte pieces
    // the step's prepacked copy (prepack.hip) where there is one; the key holds everything the packing below depends on
    const int pk_form = f16 ? (w16 ? 1 : 2) : (b16s ? (w16 ? 3 : 4) : (x3s ? (w16 ? 5 : 6) : 7));
    const PackKey pkey = make_pack_key(w, PK_CONV, pk_form, Cin, Cout, T, dgrad ? 1 : 0, b16s ? bp.NT : (pk_form == 7 ? p.NT * 64 + p.CK + 4096 * math : p.NBW));
    const bool pk_hit = !oscale && prepack_find(pkey, &wq, f16 ? &w_amax : nullptr);
    if (f16) {
        if (!x_amax || !w_amax) {
            if (hipMemsetAsync(amax, 0, 2 * sizeof(float), st) != hipSuccess) { set_error("conv_fwd_mfma: hipMemsetAsync failed"); return MI355SEG_EHIP; }
            if (!x_amax) { tensor_amax((const float*)x, ldx, nvox, Cin, nullptr, amax, st); x_amax = amax; }
            // oscale (forward only: W is (Cout, Cin, T)) goes by the rows of W
            if (!w_amax) {
                if (oscale && !dgrad) tensor_amax(w, Cin * T, Cout, Cin * T, oscale, amax + 1, st);
                else tensor_amax(w, T * Cin * Cout, 1, T * Cin * Cout, nullptr, amax + 1, st);
                w_amax = amax + 1;
            }
        }
    }
    if (!pk_hit) {
        PackDesc pd{};
        const int dg = dgrad ? 1 : 0;
        if (pk_form == 1) pd = tiled_desc(2, w, wq, Cin, Cout, 27, p.NBW, dg, oscale, w_amax);
        else if (pk_form == 3) pd = tiled_desc(0, w, wq, Cin, Cout, T, bp.NT, dg, oscale, nullptr);
        else if (pk_form == 5) pd = tiled_desc(1, w, wq, Cin, Cout, 27, p.NBW, dg, oscale, nullptr);
        if (pd.kind) pack_launch(pd, st);
        else if (pk_form == 2) hipLaunchKernelGGL(pack_wq_x3s_f16_kernel, dim3(pack_grid((long long)28 * Cin * Cout)), dim3(256), 0, st, w, (_Float16*)wq, Cin, Cout, p.NBW, dg, oscale, w_amax);
        else if (pk_form == 4) hipLaunchKernelGGL(pack_wq_b16s_kernel, dim3(pack_grid((long long)(T + 1) * Cin * Cout)), dim3(256), 0, st, w, (bf16*)wq, Cin, Cout, T, bp.NT, dg, oscale);
        else if (pk_form == 6) hipLaunchKernelGGL(pack_wq_x3s_kernel, dim3(pack_grid((long long)28 * Cin * Cout)), dim3(256), 0, st, w, (bf16*)wq, Cin, Cout, p.NBW, dg, oscale);
        else launch_pack(math, w, wq, Cin, Cout, T, p.NT, dg, 0, p.CK, T, TapList{}, st, oscale, &pd);
        SEG_CHECK_LAUNCH();
        if (pd.kind) prepack_note(pkey, wq_bytes(math, (size_t)T * Cin * Cout), pd);
    }
    IgemmArgs a{x, wq, ksplit > 1 ? nullptr : bias, ksplit > 1 ? (void*)slabs : y, spart, ldx, ksplit > 1 ? Cout : ldy, N, D, H, W, Cout,
                p.ntx, p.nty, p.ntz, p.nN, nchunks, nchunks, p.nN, 1, 1, p.nM, ksplit, nchunks / ksplit, nvox * Cout, dbg_flags()};
    a.Di = a.Do = D; a.Hi = a.Ho = H; a.Wi = a.Wo = W;
    a.act = ksplit > 1 ? 0 : act; a.slope = slope;
    if (f16) { a.amax_x = x_amax; a.amax_w = w_amax; }
    if (pro) {
        SEG_CHECK_ARG(f16 && !dgrad && pro->al && pro->be && conv_pro_act_ok(pro->act) && Cin <= 512, "conv_fwd_mfma: the norm + activation prologue needs the f16x3 conv_x3s kernels (Cin <= 512, act relu / leaky relu with a slope in [0, 1))");
        a.pro_al = pro->al; a.pro_be = pro->be; a.pro_act = pro->act; a.pro_slope = pro->act == MI355SEG_ACT_RELU ? 0.f : pro->slope;
        SEG_CHECK_ARG(a.pro_slope >= 0.f && a.pro_slope < 1.f, "conv_fwd_mfma: prologue slope must lie in [0, 1)");
    }
    const bool ymax_epi = y_amax && x3s && ksplit == 1 && math != MATH_B16;
    if (ymax_epi) a.amax_y = reinterpret_cast<unsigned*>(y_amax);
    if (res && res_fused && b16s && ksplit == 1 && !ssum && (ldres % 8) == 0 && ((uintptr_t)res % 16) == 0) { a.res = res; a.ldres = ldres; *res_fused = 1; }
    if (bn_epi) {
        a.bnx = bne->x; a.ldbnx = bne->ldx; a.bn_mean = bne->mean; a.bn_rstd = bne->rstd; a.bn_gamma = bne->gamma; a.bn_beta = bne->beta;
        a.bn_act = bne->act; a.bn_slope = bne->slope; a.bnpart = bnpart;
    }
    const int nwg = p.nM * p.nN * ksplit;
    const double vox = (double)nvox;
    {
        ProfScope ps(PF_IGEMM, 2.0 * vox * T * Cin * Cout, matrix_bytes(math, vox * (Cin + Cout), (double)T * Cin * Cout), st);
        if (b16s) { a.total = nwg; a.by = tile_block(a.nty); a.bz = a.ntz >= 4 ? 4 : tile_block(a.ntz); dispatch_b16s(bp, a, nwg, st); }
        else dispatch_igemm(math, p, a, nwg, st, x3s);
        SEG_CHECK_LAUNCH();
        if (ksplit > 1) {
            long long tot = nvox * (Cout / 4);
            int grid = (int)((tot + 255) / 256 > 2048 ? 2048 : (tot + 255) / 256);
            if (math == MATH_B16) hipLaunchKernelGGL(splitk_reduce_kernel<bf16>, dim3(grid), dim3(256), 0, st, slabs, ksplit, nvox * Cout, bias, (bf16*)y, ldy, nvox, Cout, act, slope);
            else hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(grid), dim3(256), 0, st, slabs, ksplit, nvox * Cout, bias, (float*)y, ldy, nvox, Cout, act, slope);
            SEG_CHECK_LAUNCH();
        }
    }
    if (bn_epi) {
        norm_bwd_finalize(bnpart, p.nM, Cout, bne->s1, bne->s2, bne->dgamma, bne->dbeta, rtmp, st);
        SEG_CHECK_LAUNCH();
        bne->done = 1;
    }
    if (y_amax && !ymax_epi) {
        SEG_CHECK_ARG(math != MATH_B16, "conv_fwd_mfma: y_amax is for fp32 tensors");
        tensor_amax((const float*)y, ldy, nvox, Cout, nullptr, y_amax, st);
        SEG_CHECK_LAUNCH();
    }
    if (ssum) {
        if (ksplit > 1) {
            if (math == MATH_B16) return channel_sums((const bf16*)y, ldy, nvox, Cout, ssum, ssq, nullptr, 0, (char*)ws + tail, ws_bytes - tail, st);
            return channel_sums((const float*)y, ldy, nvox, Cout, ssum, ssq, nullptr, 0, (char*)ws + tail, ws_bytes - tail, st);
        }
        if (!tile_stats_finalize2(spart, p.nM, Cout, ssum, ssq, rtmp, st))
            hipLaunchKernelGGL(igemm_stats_finalize_kernel, dim3(Cout), dim3(256), 0, st, spart, p.nM, Cout, ssum, ssq);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

// ---- ConvTranspose3d k2 s2 on the same kernel (KS = 1)
bool convt_mfma_supported(int math, int N, int D, int H, int W, int Cin, int Cout, int ldx, int ldy) {
    IgemmPlan p, q;
    const int al = math == MATH_B16 ? 8 : 4;
    return (ldx % al) == 0 && (ldy % al) == 0 && igemm_plan(math, 1, N, D, H, W, Cin, Cout, 8, &p) && igemm_plan(math, 1, N, D, H, W, Cout, Cin, 1, &q) &&
           Cin % 32 == 0;
}

int convt_fwd_mfma(int math, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int N, int D, int H, int W,
                   int Cin, int Cout, void* ws, size_t ws_bytes, hipStream_t st) {
    IgemmPlan p;
    SEG_CHECK_ARG(igemm_plan(math, 1, N, D, H, W, Cin, Cout, 8, &p), "convt_fwd_mfma: unsupported shape");
    Carver cv(ws);
    void* wq = cv.take<char>(wq_bytes(math, (size_t)8 * Cin * Cout));
    SEG_CHECK_WS(cv.used(), ws_bytes);
    {
        const PackKey pkey = make_pack_key(w, PK_CONVT_FWD, math, Cin, Cout, p.NT, p.CK);
        if (!prepack_find(pkey, &wq, nullptr)) {
            PackDesc pd{};
            launch_pack(math, w, wq, Cin, 8 * Cout, 1, p.NT, 2, Cout, p.CK, 8, TapList{}, st, nullptr, &pd);
            SEG_CHECK_LAUNCH();
            if (pd.kind) prepack_note(pkey, wq_bytes(math, (size_t)8 * Cin * Cout), pd);
        }
    }
    IgemmArgs a{x, wq, bias, y, nullptr, ldx, ldy, N, D, H, W, Cout, p.ntx, p.nty, p.ntz, p.nN, Cin / p.CK, Cin / p.CK,
                p.flat ? p.nN : p.nN / 8, 1, 2, p.nM, 1, Cin / p.CK, 0, 0};
    a.Di = D; a.Hi = H; a.Wi = W; a.Do = 2 * D; a.Ho = 2 * H; a.Wo = 2 * W; a.flatn = p.flat;
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_CONVT, 2.0 * vox * 8 * Cin * Cout, matrix_bytes(math, vox * (Cin + 8.0 * Cout), 8.0 * Cin * Cout), st);
    dispatch_igemm(math, p, a, p.nM * p.nN, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// dx[N,D,H,W,Cin] from dy[N,2D,2H,2W,Cout]
int convt_dgrad_mfma(int math, const void* dy, int lddy, const float* w, void* dx, int lddx, int N, int D, int H, int W,
                     int Cin, int Cout, void* ws, size_t ws_bytes, hipStream_t st) {
    IgemmPlan p;
    SEG_CHECK_ARG(igemm_plan(math, 1, N, D, H, W, Cout, Cin, 1, &p), "convt_dgrad_mfma: unsupported shape");
    Carver cv(ws);
    void* wq = cv.take<char>(wq_bytes(math, (size_t)8 * Cin * Cout));
    SEG_CHECK_WS(cv.used(), ws_bytes);
    {
        const PackKey pkey = make_pack_key(w, PK_CONVT_DGRAD, math, Cin, Cout, p.NT, p.CK);
        if (!prepack_find(pkey, &wq, nullptr)) {
            PackDesc pd{};
            launch_pack(math, w, wq, 8 * Cout, Cin, 1, p.NT, 3, Cout, p.CK, 8, TapList{}, st, nullptr, &pd);
            SEG_CHECK_LAUNCH();
            if (pd.kind) prepack_note(pkey, wq_bytes(math, (size_t)8 * Cin * Cout), pd);
        }
    }
    IgemmArgs a{dy, wq, nullptr, dx, nullptr, lddy, lddx, N, D, H, W, Cin, p.ntx, p.nty, p.ntz, p.nN, 8 * Cout / p.CK, Cout / p.CK, p.nN, 2, 1,
                p.nM, 1, 8 * Cout / p.CK, 0, 0};
    a.Di = 2 * D; a.Hi = 2 * H; a.Wi = 2 * W; a.Do = D; a.Ho = H; a.Wo = W;
    for (int t = 0; t < 8; ++t) { a.toff[t][0] = (t >> 2) & 1; a.toff[t][1] = (t >> 1) & 1; a.toff[t][2] = t & 1; }
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_CONVT, 2.0 * vox * 8 * Cin * Cout, matrix_bytes(math, vox * (Cin + 8.0 * Cout), 8.0 * Cin * Cout), st);
    dispatch_igemm(math, p, a, p.nM * p.nN, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// ---- strided / even-kernel Conv3d on the same kernel (KS = 1, no halo): every K-chunk is one (tap, channel chunk) and
// gathers its own input voxel  base * stride + tap - pad  (zero outside the volume).  Used for the k3 s2 p1 convs of the
// Residual U-Net (residual_unet3d.py:24-60) and V-Net's k2 s2 down-convolutions (vnet3d.py:66).
static bool gather_geom_ok(int k, int stride, int pad) {
    return k >= 1 && k <= 4 && stride >= 1 && stride <= k && pad >= 0 && pad < k && !(stride == 1 && (k == 1 || k == 3) && pad == k / 2);
}
static int out_extent(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

bool conv_gather_fwd_supported(int math, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    const int al = math == MATH_B16 ? 8 : 4;
    if (!gather_geom_ok(k, stride, pad) || (ldx % al) || D + 2 * pad < k || H + 2 * pad < k || W + 2 * pad < k) return false;
    IgemmPlan p;
    return igemm_plan(math, 1, N, out_extent(D, k, stride, pad), out_extent(H, k, stride, pad), out_extent(W, k, stride, pad), Cin, Cout, 1, &p);
}

// dgrad runs one launch per output phase (u mod stride): the taps that reach a phase are a fixed subset with fixed offsets
bool conv_gather_dgrad_supported(int math, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int lddy, int lddx) {
    const int al = math == MATH_B16 ? 8 : 4;
    if (!gather_geom_ok(k, stride, pad) || (lddy % al) || D + 2 * pad < k || H + 2 * pad < k || W + 2 * pad < k) return false;
    if (D < stride || H < stride || W / stride < 4) return false;
    IgemmPlan p;
    return igemm_plan(math, 1, N, D / stride, H / stride, W / stride, Cout, Cin, 1, &p);
}

size_t conv_gather_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    if (!gather_geom_ok(k, stride, pad) || D + 2 * pad < k || H + 2 * pad < k || W + 2 * pad < k) return 0;
    const size_t T = (size_t)k * k * k;
    size_t need = align_up(T * Cin * Cout * sizeof(float), 256) + 2048;
    IgemmPlan p;
    for (int math : {MATH_F32, MATH_B16}) {
        if (!igemm_plan(math, 1, N, out_extent(D, k, stride, pad), out_extent(H, k, stride, pad), out_extent(W, k, stride, pad), Cin, Cout, 1, &p)) continue;
        const int ks = pick_ksplit(p.nM * p.nN, (int)T * (Cin / p.CK));
        const size_t extra = align_up((size_t)p.nM * Cout * 3 * sizeof(float), 256) +
                             (ks > 1 ? align_up((size_t)ks * N * out_extent(D, k, stride, pad) * out_extent(H, k, stride, pad) * out_extent(W, k, stride, pad) * Cout * sizeof(float), 256) + colsum_ws_bytes(Cout) : 0);
        if (align_up(T * Cin * Cout * sizeof(float), 256) + 2048 + extra > need) need = align_up(T * Cin * Cout * sizeof(float), 256) + 2048 + extra;
    }
    return need;
}

int conv_gather_fwd_mfma(int math, const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int N, int D, int H, int W,
                         int Cin, int Cout, int k, int stride, int pad, double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st) {
    const int Do = out_extent(D, k, stride, pad), Ho = out_extent(H, k, stride, pad), Wo = out_extent(W, k, stride, pad);
    IgemmPlan p;
    SEG_CHECK_ARG(igemm_plan(math, 1, N, Do, Ho, Wo, Cin, Cout, 1, &p), "conv_gather_fwd_mfma: unsupported shape");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0, "conv_gather_fwd_mfma: input pointer must be 16-byte aligned");
    const int T = k * k * k, cpt = Cin / p.CK, nchunks = T * cpt;
    // few output tiles and a long K (the deep down-convolutions: 256 -> 512 at 10 x 12 x 10 is 9 tiles x 432 chunks): split K
    const int ksplit = (ldy % 4 == 0) ? pick_ksplit(p.nM * p.nN, nchunks) : 1;
    const long long nvo = (long long)N * Do * Ho * Wo;
    Carver cv(ws);
    void* wq = cv.take<char>(wq_bytes(math, (size_t)T * Cin * Cout));
    float* spart = (ssum && ksplit == 1) ? cv.take<float>((size_t)p.nM * Cout * 3) : nullptr;
    float* slabs = ksplit > 1 ? cv.take<float>((size_t)ksplit * nvo * Cout) : nullptr;
    const size_t tail = cv.used();
    SEG_CHECK_WS(tail + ((ssum && ksplit > 1) ? colsum_ws_bytes(Cout) : 0), ws_bytes);
    {
        const PackKey pkey = make_pack_key(w, PK_GATHER_FWD, math, Cin, Cout, T, p.NT, p.CK);
        if (!prepack_find(pkey, &wq, nullptr)) {
            PackDesc pd{};
            launch_pack(math, w, wq, T * Cin, Cout, 1, p.NT, 5, Cin, p.CK, T, TapList{}, st, nullptr, &pd);
            SEG_CHECK_LAUNCH();
            if (pd.kind) prepack_note(pkey, wq_bytes(math, (size_t)T * Cin * Cout), pd);
        }
    }
    IgemmArgs a{x, wq, ksplit > 1 ? nullptr : bias, ksplit > 1 ? (void*)slabs : y, spart, ldx, ksplit > 1 ? Cout : ldy, N, Do, Ho, Wo, Cout, p.ntx, p.nty, p.ntz, p.nN, nchunks, cpt, p.nN, stride, 1,
                p.nM, ksplit, nchunks / ksplit, nvo * Cout, 0};
    a.Di = D; a.Hi = H; a.Wi = W; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    for (int t = 0; t < T; ++t) { a.toff[t][0] = (signed char)(t / (k * k) - pad); a.toff[t][1] = (signed char)((t / k) % k - pad); a.toff[t][2] = (signed char)(t % k - pad); }
    const double vox = (double)nvo;
    {
        ProfScope ps(PF_IGEMM, 2.0 * vox * T * Cin * Cout, matrix_bytes(math, (double)N * D * H * W * Cin + vox * Cout, (double)T * Cin * Cout), st);
        dispatch_igemm(math, p, a, p.nM * p.nN * ksplit, st);
        SEG_CHECK_LAUNCH();
        if (ksplit > 1) {
            const long long tot = nvo * (Cout / 4);
            const int grid = (int)((tot + 255) / 256 > 2048 ? 2048 : (tot + 255) / 256);
            if (math == MATH_B16) hipLaunchKernelGGL(splitk_reduce_kernel<bf16>, dim3(grid), dim3(256), 0, st, slabs, ksplit, nvo * Cout, bias, (bf16*)y, ldy, nvo, Cout, 0, 0.f);
            else hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(grid), dim3(256), 0, st, slabs, ksplit, nvo * Cout, bias, (float*)y, ldy, nvo, Cout, 0, 0.f);
            SEG_CHECK_LAUNCH();
        }
    }
    if (ssum) {
        if (ksplit > 1) {
            if (math == MATH_B16) return channel_sums((const bf16*)y, ldy, nvo, Cout, ssum, ssq, nullptr, 0, (char*)ws + tail, ws_bytes - tail, st);
            return channel_sums((const float*)y, ldy, nvo, Cout, ssum, ssq, nullptr, 0, (char*)ws + tail, ws_bytes - tail, st);
        }
        hipLaunchKernelGGL(igemm_stats_finalize_kernel, dim3(Cout), dim3(256), 0, st, spart, p.nM, Cout, ssum, ssq);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

// dx[N,D,H,W,Cin] from dy[N,Do,Ho,Wo,Cout]:  dx[u] = sum over taps t with (u + pad - t) % stride == 0 of dy[(u + pad - t) / stride] W[.,.,t]
// One GEMM per output phase (u mod stride).  bf16 tensors (r6): all phases in ONE launch over ONE packing of the weights -- the taps
// partition over the phases, so the packing is the phases' tap lists back to back -- instead of eight pack + eight convolution launches
// (the deep Residual U-Net levels: eight launches of ~60 workgroups, 14-94 us each, now one of ~480); other maths, or phases whose
// plans differ, keep the per-phase launches.
int conv_gather_dgrad_mfma(int math, const void* dy, int lddy, const float* w, void* dx, int lddx, int N, int D, int H, int W,
                           int Cin, int Cout, int k, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t st) {
    const int Do = out_extent(D, k, stride, pad), Ho = out_extent(H, k, stride, pad), Wo = out_extent(W, k, stride, pad);
    SEG_CHECK_ARG(((uintptr_t)dy % 16) == 0, "conv_gather_dgrad_mfma: gradient pointer must be 16-byte aligned");
    const int T = k * k * k;
    Carver cv(ws);
    char* wq_all = cv.take<char>(wq_bytes(math, (size_t)T * Cin * Cout));
    SEG_CHECK_WS(cv.used(), ws_bytes);
    ProfScope ps(PF_IGEMM, 2.0 * (double)N * Do * Ho * Wo * T * Cin * Cout,
                 matrix_bytes(math, (double)N * D * H * W * Cin + (double)N * Do * Ho * Wo * Cout, (double)T * Cin * Cout), st);
    struct Phase { IgemmPlan p; IgemmArgs a; TapList tl; int nt; };
    Phase phs[64];
    int nph = 0;
    for (int ph = 0; ph < stride * stride * stride; ++ph) {
        const int pz = ph / (stride * stride), py = (ph / stride) % stride, px = ph % stride;
        const int Bz = (D - pz + stride - 1) / stride, By = (H - py + stride - 1) / stride, Bx = (W - px + stride - 1) / stride;
        if (Bz <= 0 || By <= 0 || Bx <= 0) continue;
        int lz[4], ly[4], lx[4], dz[4], dyo[4], dxo[4], nz = 0, ny = 0, nx = 0;
        for (int t = 0; t < k; ++t) {
            if ((pz + pad - t) % stride == 0) { lz[nz] = t; dz[nz++] = (pz + pad - t) / stride; }
            if ((py + pad - t) % stride == 0) { ly[ny] = t; dyo[ny++] = (py + pad - t) / stride; }
            if ((px + pad - t) % stride == 0) { lx[nx] = t; dxo[nx++] = (px + pad - t) / stride; }
        }
        const int nt = nz * ny * nx;
        SEG_CHECK_ARG(nt > 0, "conv_gather_dgrad_mfma: a phase without taps (k < stride)");
        Phase& q = phs[nph++];
        q.nt = nt;
        IgemmPlan& p = q.p;
        SEG_CHECK_ARG(igemm_plan(math, 1, N, Bz, By, Bx, Cout, Cin, 1, &p), "conv_gather_dgrad_mfma: unsupported shape");
        const int cpt = Cout / p.CK, nchunks = nt * cpt;
        q.a = IgemmArgs{dy, nullptr, nullptr, dx, nullptr, lddy, lddx, N, Bz, By, Bx, Cin, p.ntx, p.nty, p.ntz, p.nN, nchunks, cpt, p.nN, 1, stride,
                        p.nM, 1, nchunks, 0, 0};
        IgemmArgs& a = q.a;
        a.Di = Do; a.Hi = Ho; a.Wi = Wo; a.Do = D; a.Ho = H; a.Wo = W; a.cz = pz; a.cy = py; a.cx = px;
        q.tl = TapList{};
        int slot = 0;
        for (int iz = 0; iz < nz; ++iz)
            for (int iy = 0; iy < ny; ++iy)
                for (int ix = 0; ix < nx; ++ix, ++slot) {
                    q.tl.t[slot] = (unsigned char)((lz[iz] * k + ly[iy]) * k + lx[ix]);
                    a.toff[slot][0] = (signed char)dz[iz]; a.toff[slot][1] = (signed char)dyo[iy]; a.toff[slot][2] = (signed char)dxo[ix];
                }
    }
    // ---- one launch: the same tile shape in every phase, at most eight phases of at most eight taps, every tap in exactly one phase
    bool multi = math == MATH_B16 && nph >= 2 && nph <= 8;
    int taps_total = 0;
    for (int i = 0; i < nph && multi; ++i) {
        const IgemmPlan &p = phs[i].p, &p0 = phs[0].p;
        multi = phs[i].nt <= 8 && p.KS == p0.KS && p.CK == p0.CK && p.BX == p0.BX && p.MB == p0.MB && p.NBW == p0.NBW && p.WN == 1 && p.nN == p0.nN && !p.flat;
        taps_total += phs[i].nt;
    }
    multi = multi && taps_total == T;
    if (multi) {
        const IgemmPlan& p0 = phs[0].p;
        const int cpt = Cout / p0.CK;
        const size_t chunk_bytes = (size_t)p0.CK * p0.NT * esize(math);          // one (N-tile, chunk) of the packing (bf16: one plane)
        IgemmArgs a = phs[0].a;
        IgemmPhases pt{};
        TapList tl{};
        int slot0 = 0, first = 0;
        for (int i = 0; i < nph; ++i) {
            const Phase& q = phs[i];
            for (int sl = 0; sl < q.nt; ++sl) {
                tl.t[slot0 + sl] = q.tl.t[sl];
                for (int c = 0; c < 4; ++c) a.toff[8 * i + sl][c] = q.a.toff[sl][c];
            }
            IgemmPhase& f = pt.ph[i];
            f.wq = wq_all + (size_t)slot0 * cpt * chunk_bytes;
            f.D = q.a.D; f.H = q.a.H; f.W = q.a.W; f.ntx = q.p.ntx; f.nty = q.p.nty; f.ntz = q.p.ntz;
            f.by = tile_block(f.nty); f.bz = f.ntz >= 4 ? 4 : tile_block(f.ntz);
            f.nchunks = q.nt * cpt; f.wstride = T * cpt; f.cz = q.a.cz; f.cy = q.a.cy; f.cx = q.a.cx; f.first = first;
            first += (q.p.nM * q.p.nN + 7) / 8 * 8;
            slot0 += q.nt;
        }
        pt.n = nph;
        {
            const PackKey pkey = make_pack_key(w, PK_GATHER_DGRAD, math, Cin, Cout, k * 256 + stride * 16 + pad, p0.NT, p0.CK);
            void* hit = nullptr;
            if (prepack_find(pkey, &hit, nullptr)) {
                for (int i = 0; i < nph; ++i) pt.ph[i].wq = (const char*)hit + ((const char*)pt.ph[i].wq - wq_all);
                wq_all = (char*)hit;
            } else {
                PackDesc pd{};
                launch_pack(math, w, wq_all, T * Cout, Cin, 1, p0.NT, 6, Cout, p0.CK, T, tl, st, nullptr, &pd);
                SEG_CHECK_LAUNCH();
                if (pd.kind) prepack_note(pkey, wq_bytes(math, (size_t)T * Cin * Cout), pd);
            }
        }
        a.wq = wq_all; a.total = first;
        if (dispatch_igemm_phases_lowp(math, p0, a, pt, first, st)) {
            SEG_CHECK_LAUNCH();
            return MI355SEG_OK;
        }
    }
    size_t wq_used = 0;
    for (int i = 0; i < nph; ++i) {
        Phase& q = phs[i];
        void* wq = wq_all + wq_used;
        wq_used += align_up(wq_bytes(math, (size_t)q.nt * Cin * Cout), 16);
        q.a.wq = wq;
        launch_pack(math, w, wq, q.nt * Cout, Cin, 1, q.p.NT, 6, Cout, q.p.CK, T, q.tl, st);
        SEG_CHECK_LAUNCH();
        dispatch_igemm(math, q.p, q.a, q.p.nM * q.p.nN, st);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}

}  // namespace seg
