// conv_stem4_lowp.hip -- the four-channel stem of the Residual U-Net (conv3d_c1_1: Conv3d(4, 32, k3, p1, bias=False),
// residual_unet3d.py:22; BraTS-style 4-modality input, BASELINE cfg 4) on the bf16 matrix cores, forward and weight gradient.
//
// K = 27 taps x 4 channels = 108 is too short for the implicit-GEMM kernel (16-channel chunks) and was served by VALU kernels
// (1.6 + 3.0 ms per step at 160 x 192 x 160 against 0.1 ms of HBM time).  Here the (tap, channel) pairs ARE the GEMM axis:
//   forward   y^T[co][v]      = sum_k  Wp[co][k] * X[k][v]          k = 4 * tap + ci  (112 = seven 16-deep k-steps, 4 padded)
//   wgrad     dWp[co][n]      = sum_v  dy^T[co][v] * X'[v][n]       n = 16 * (dz, dy) + 4 * dx' + ci,  dx' in 0..3 (dx' = 3 is padding)
// The halo tile of x sits in LDS as [voxel][4 bf16] (8 bytes per voxel): the forward operand of a lane is two 8-byte reads (two
// taps x four channels); the wgrad operand uses the transposing read ds_read_b64_tr_b16, whose 16-column block is exactly
// four x-neighbours x four channels of one (dz, dy) line -- consecutive bytes of the tile.  Both kernels are HBM-bound on the
// 32-channel side (y / dy: 64 bytes per voxel).
#include "common.h"
#include "internal.h"

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int S4_BX = 32, S4_TY = 4, S4_TZ = 4;                       // 512-voxel tile: wave w owns the 4 x-lines of z-slab w
constexpr int S4_HX = S4_BX + 2, S4_HY = S4_TY + 2, S4_HZ = S4_TZ + 2;
constexpr int S4_NVOX = S4_HX * S4_HY * S4_HZ;                        // 1224 halo voxels, 8 bytes each
constexpr int S4_XBYTES = S4_NVOX * 8;

struct Stem4Args {
    const bf16* x; const bf16* wq; const float* bias; bf16* y; const bf16* dy; float* part;
    int ldy, N, D, H, W, Cout, ntx, nty, ntz, ntiles;
};

// wq[cob][kstep][h][co32][8]: the weight operand of the forward, k = 16 * kstep + 8 * h + j -> (tap, ci) = (k / 4, k % 4)
__global__ void stem4_pack_kernel(const float* __restrict__ w, bf16* __restrict__ wq, int Cout) {
    const int total = (Cout / 32) * 7 * 2 * 32 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int r = idx;
        const int j = r % 8; r /= 8;
        const int c = r % 32; r /= 32;
        const int h = r % 2; r /= 2;
        const int s = r % 7; const int cob = r / 7;
        const int k = 16 * s + 8 * h + j, tap = k / 4, ci = k % 4, co = cob * 32 + c;
        wq[idx] = (bf16)(tap < 27 ? w[((long long)co * 4 + ci) * 27 + tap] : 0.f);
    }
}

__device__ __forceinline__ void stem4_stage(unsigned char* xs, const bf16* __restrict__ x, int n, int z0, int y0, int x0, int D, int H, int W) {
    for (int p = threadIdx.x; p < S4_NVOX; p += blockDim.x) {
        const int hx = p % S4_HX, r = p / S4_HX, hy = r % S4_HY, hz = r / S4_HY;
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        bf16x4_t v = {};
        if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
            v = *reinterpret_cast<const bf16x4_t*>(x + ((((long long)n * D + gz) * H + gy) * W + gx) * 4);
        *reinterpret_cast<bf16x4_t*>(xs + p * 8) = v;
    }
}

__device__ __forceinline__ constexpr int s4_tapoff(int tap) {        // byte offset of a tap inside the halo tile (tap >= 27: padding, any valid address)
    return tap < 27 ? (((tap / 9) * S4_HY + (tap / 3) % 3) * S4_HX + tap % 3) * 8 : 0;
}

// ---------------------------------------------------------------- forward
template <int NCO>
__global__ __launch_bounds__(256) void stem4_fwd_kernel(Stem4Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char xs[S4_XBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, i = lane & 31;
    // the weights of this lane's output-channel row: seven k-steps per 32-channel block, held in registers for the whole grid walk
    bf16x8_t wr[NCO][7];
#pragma unroll
    for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
        for (int s = 0; s < 7; ++s) wr[cb][s] = *reinterpret_cast<const bf16x8_t*>(a.wq + (((cb * 7 + s) * 2 + h) * 32 + i) * 8);
    int offA[7], offB[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) { offA[s] = h ? s4_tapoff(4 * s + 2) : s4_tapoff(4 * s); offB[s] = h ? s4_tapoff(4 * s + 3) : s4_tapoff(4 * s + 1); }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int mt = tile;
        const int txi = mt % a.ntx; mt /= a.ntx;
        const int tyi = mt % a.nty; mt /= a.nty;
        const int tzi = mt % a.ntz; const int n = mt / a.ntz;
        const int x0 = txi * S4_BX, y0 = tyi * S4_TY, z0 = tzi * S4_TZ;
        __syncthreads();
        stem4_stage(xs, a.x, n, z0, y0, x0, a.D, a.H, a.W);
        __syncthreads();
#pragma unroll
        for (int ly = 0; ly < S4_TY; ++ly) {
            const int vbase = ((wave * S4_HY + ly) * S4_HX + i) * 8;      // halo voxel (z = wave, y = ly, x = i) == tap (0, 0, 0) of output voxel i
            f32x16 acc[NCO];
#pragma unroll
            for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[cb][v] = 0.f;
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(xs + vbase + offA[s]);
                const bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(xs + vbase + offB[s]);
                const bf16x8_t xb = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[cb][s], xb, acc[cb], 0, 0, 0);
            }
            const int gz = z0 + wave, gy = y0 + ly, gx = x0 + i;
            if (gz < a.D && gy < a.H && gx < a.W) {
                bf16* yp = a.y + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldy;
#pragma unroll
                for (int cb = 0; cb < NCO; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {                     // registers 4g .. 4g+3 = output channels 8g + 4h .. +3 of the block
                        const int co = cb * 32 + 8 * g + 4 * h;
                        f32x4_t o = {acc[cb][4 * g], acc[cb][4 * g + 1], acc[cb][4 * g + 2], acc[cb][4 * g + 3]};
                        if (a.bias) o += ld4(a.bias + co);
                        st4(yp + co, o);
                    }
            }
        }
    }
}

// ---------------------------------------------------------------- weight gradient
// grid = (blocks, Cout / 32).  part[blk][tap][ci][co] (the layout wgrad_reduce sums over blocks).
__global__ __launch_bounds__(256) void stem4_wgrad_kernel(Stem4Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xs = lds;                                         // [1224][8 B]
    unsigned char* ds = lds + S4_XBYTES;                             // [512 voxels][32 co] bf16 = 64-byte rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, i = lane & 31;
    const int co0 = blockIdx.y * 32;
    const int li = lane & 15, q = li >> 2, p = li & 3, cg = (lane >> 4) & 1;
    // transposing-read lane geometry: dy tile rows = voxels, 16 columns = output channels 16cg ..;  x tile rows = voxels shifted by
    // p x-neighbours (the 16 "columns" of a row are 4 neighbours x 4 channels = 32 consecutive bytes starting at the row's voxel)
    const int lane_d = (8 * h + q) * 64 + (16 * cg + 4 * p) * 2;
    int lane_x[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb) {
        int pair = 2 * nb + cg;                                      // (dz, dy) line pair of this 16-lane group; 9 = padding
        if (pair > 8) pair = 8;
        lane_x[nb] = (((pair / 3) * S4_HY + pair % 3) * S4_HX + 8 * h + q + p) * 8;
    }
    f32x16 acc[5];
#pragma unroll
    for (int nb = 0; nb < 5; ++nb)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[nb][v] = 0.f;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int mt = tile;
        const int txi = mt % a.ntx; mt /= a.ntx;
        const int tyi = mt % a.nty; mt /= a.nty;
        const int tzi = mt % a.ntz; const int n = mt / a.ntz;
        const int x0 = txi * S4_BX, y0 = tyi * S4_TY, z0 = tzi * S4_TZ;
        __syncthreads();
        stem4_stage(xs, a.x, n, z0, y0, x0, a.D, a.H, a.W);
        for (int pc = tid; pc < 512 * 4; pc += 256) {                // dy tile: 512 voxels x 4 pieces of 8 channels
            const int vox = pc >> 2, part = pc & 3;
            const int xx = vox % S4_BX, line = vox / S4_BX, gz = z0 + line / S4_TY, gy = y0 + line % S4_TY, gx = x0 + xx;
            bf16x8_t dv = {};
            if (gz < a.D && gy < a.H && gx < a.W)
                dv = *reinterpret_cast<const bf16x8_t*>(a.dy + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldy + co0 + part * 8);
            *reinterpret_cast<bf16x8_t*>(ds + vox * 64 + part * 16) = dv;
        }
        __syncthreads();
#pragma unroll
        for (int ly = 0; ly < S4_TY; ++ly)
#pragma unroll
            for (int half = 0; half < 2; ++half) {                   // two 16-voxel k-steps per x-line
                const int dbase = ((wave * S4_TY + ly) * S4_BX + half * 16) * 64 + lane_d;
                const s16x4 d0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ds + dbase));
                const s16x4 d1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ds + dbase + 4 * 64));
                const bf16x8_t df = __builtin_bit_cast(bf16x8_t, (s16x8)__builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7));
                const int xbase = ((wave * S4_HY + ly) * S4_HX + half * 16) * 8;
#pragma unroll
                for (int nb = 0; nb < 5; ++nb) {
                    const s16x4 x0v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + xbase + lane_x[nb]));
                    const s16x4 x1v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + xbase + lane_x[nb] + 4 * 8));
                    const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, (s16x8)__builtin_shufflevector(x0v, x1v, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, xf, acc[nb], 0, 0, 0);
                }
            }
    }
    // four waves hold partial sums over different voxels: add them through LDS in wave order, then scatter the valid columns
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);                      // [4 waves][32 rows][32 cols] per N-block (16 KB)
#pragma unroll
    for (int nb = 0; nb < 5; ++nb) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = (v & 3) + 8 * (v >> 2) + 4 * h;             // row = output channel of the block
            red[(wave * 32 + r) * 32 + i] = acc[nb][v];
        }
        __syncthreads();
        for (int e = tid; e < 1024; e += 256) {
            const int r = e >> 5, c = e & 31;
            const float s = red[(0 * 32 + r) * 32 + c] + red[(1 * 32 + r) * 32 + c] + red[(2 * 32 + r) * 32 + c] + red[(3 * 32 + r) * 32 + c];
            const int nn = 32 * nb + c, pair = nn >> 4, dxl = (nn & 15) >> 2, ci = nn & 3;
            if (pair < 9 && dxl < 3)
                a.part[(((long long)blockIdx.x * 27 + pair * 3 + dxl) * 4 + ci) * a.Cout + co0 + r] = s;
        }
        __syncthreads();
    }
}

bool stem4_lowp_supported(int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    return Cin == 4 && k == 3 && stride == 1 && pad == 1 && ldx == 4 && (Cout == 32 || Cout == 64) && ldy % 8 == 0;
}
size_t stem4_lowp_ws_bytes(int Cout) { return align_up((size_t)512 * 27 * 4 * Cout * sizeof(float), 256) + align_up((size_t)7 * 2 * Cout * 8 * 2, 256) + 256; }

static void stem4_geom(Stem4Args& a, int N, int D, int H, int W) {
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.ntx = (W + S4_BX - 1) / S4_BX; a.nty = (H + S4_TY - 1) / S4_TY; a.ntz = (D + S4_TZ - 1) / S4_TZ;
    a.ntiles = N * a.ntx * a.nty * a.ntz;
}

int stem4_fwd_lowp(const bf16* x, const float* w, const float* bias, bf16* y, int ldy, int N, int D, int H, int W, int Cout,
                   void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_WS(align_up((size_t)7 * 2 * Cout * 8 * 2, 256), ws_bytes);
    SEG_CHECK_ARG(((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 8) == 0, "stem4_fwd: pointers must be 8-byte aligned");
    bf16* wq = (bf16*)ws;
    hipLaunchKernelGGL(stem4_pack_kernel, dim3(8), dim3(256), 0, st, w, wq, Cout);
    SEG_CHECK_LAUNCH();
    Stem4Args a{x, wq, bias, y, nullptr, nullptr, ldy, 0, 0, 0, 0, Cout, 0, 0, 0, 0};
    stem4_geom(a, N, D, H, W);
    const int grid = a.ntiles < 2048 ? a.ntiles : 2048;
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_DIRECT, 2.0 * vox * 108.0 * Cout, 2.0 * vox * (4 + Cout), st);
    if (Cout == 32) hipLaunchKernelGGL((stem4_fwd_kernel<1>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((stem4_fwd_kernel<2>), dim3(grid), dim3(256), 0, st, a);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int stem4_wgrad_lowp(const bf16* dy, int lddy, const bf16* x, float* dw, int N, int D, int H, int W, int Cout, int accumulate,
                     void* ws, size_t ws_bytes, hipStream_t st) {
    Stem4Args a{x, nullptr, nullptr, nullptr, dy, nullptr, lddy, 0, 0, 0, 0, Cout, 0, 0, 0, 0};
    stem4_geom(a, N, D, H, W);
    int nblk = a.ntiles < 512 ? a.ntiles : 512;
    SEG_CHECK_WS((size_t)nblk * 27 * 4 * Cout * sizeof(float), ws_bytes);
    SEG_CHECK_ARG(((uintptr_t)x % 8) == 0 && ((uintptr_t)dy % 16) == 0, "stem4_wgrad: pointers must be 8 / 16-byte aligned");
    a.part = (float*)ws;
    const size_t ldsb = S4_XBYTES + 512 * 64;
    SEG_SET_LDS((stem4_wgrad_kernel), (int)ldsb);
    const double vox = (double)N * D * H * W;
    {
        ProfScope ps(PF_DIRECT, 2.0 * vox * 108.0 * Cout, 2.0 * vox * (4 + Cout), st);
        hipLaunchKernelGGL(stem4_wgrad_kernel, dim3(nblk, Cout / 32), dim3(256), ldsb, st, a);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(a.part, dw, nblk, 27, 4, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // namespace seg
