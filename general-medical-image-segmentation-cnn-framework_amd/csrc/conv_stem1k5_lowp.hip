// conv_stem1k5_lowp.hip -- V-Net's one-channel k5 stem (InputTransition.conv1: Conv3d(1, 16, k5, p2), vnet3d.py:49-58) on the
// bf16 matrix cores for bf16 tensors, forward and weight gradient (the input needs no gradient).
//
// Cin = 1: the x-taps are the GEMM's narrow axis (conv_head2_lowp.hip's scheme with one channel):
//   forward   y^T[co][v] = sum over (dz, dy) rows, dx' of  w[co][dz, dy, dx'] * x[v - 2 + (dz, dy, dx')]
//             one 16-deep k-step per PAIR of rows: lane half h takes row 2r + h, k = its eight x-neighbours (dx' > 4 zero-weighted)
//   wgrad     dW^T[co][(dz, dy, dx')] = sum_u dy[u][co] * x[u - 2 + (dz, dy, dx')]         K = voxels, transposing reads
// The one-channel tensor is 2 bytes per voxel, so the 16 bytes a forward lane reads (eight x-neighbours) start at any voxel:
// EIGHT copies of the x halo tile, copy c shifted by c voxels, lane i reads copy i % 8 at an aligned slot; the transposing
// reads of the weight gradient need 8-byte alignment: four copies, picked by voxel % 4.
#include "common.h"
#include "internal.h"

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int S1_BX = 32, S1_TY = 4, S1_TZ = 4, S1_HY = S1_TY + 4, S1_HZ = S1_TZ + 4;
constexpr int S1_ROWW = 48, S1_ROWB = S1_ROWW * 2;                    // voxels / bytes per halo row of a copy (t = gx - (x0 - 2))
constexpr int S1_COPYB = S1_HZ * S1_HY * S1_ROWB + 16;                // copies staggered by one 16-byte slot

struct Stem1Args {
    const bf16* x; const bf16* dy; const bf16* wq; const float* bias; bf16* y; float* part;
    int ldy, N, D, H, W, Cout, ntx, nty, ntz, ntiles;
};

// NCOPY copies of the x halo tile: copy c, row (hz, hy), slot j holds x at t = j + c (t = gx - (x0 - 2)).  Every source voxel is
// loaded once and written to the (up to NCOPY) slots that show it.
template <int NCOPY>
__device__ __forceinline__ void stem1_stage_x(unsigned char* xs, const bf16* __restrict__ x, int n, int z0, int y0, int x0, int D, int H, int W) {
    constexpr int SRCW = S1_ROWW + NCOPY - 1;
    for (int e = threadIdx.x; e < S1_HZ * S1_HY * SRCW; e += 256) {
        const int t = e % SRCW, r = e / SRCW;
        const int gz = z0 - 2 + r / S1_HY, gy = y0 - 2 + r % S1_HY, gx = x0 - 2 + t;
        bf16 v = (bf16)0.f;
        if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
            v = x[(((long long)n * D + gz) * H + gy) * W + gx];
#pragma unroll
        for (int c = 0; c < NCOPY; ++c) {
            const int j = t - c;
            if (j >= 0 && j < S1_ROWW) *reinterpret_cast<bf16*>(xs + c * S1_COPYB + r * S1_ROWB + j * 2) = v;
        }
    }
}

__device__ __forceinline__ void stem1_tile(const Stem1Args& a, int tile, int& n, int& z0, int& y0, int& x0) {
    int mt = tile;
    x0 = (mt % a.ntx) * S1_BX; mt /= a.ntx;
    y0 = (mt % a.nty) * S1_TY; mt /= a.nty;
    z0 = (mt % a.ntz) * S1_TZ; n = mt / a.ntz;
}

// wq[r 13][h][co 32][8]: row = 2r + h = 5 dz + dy (25 = padding), element j = dx' -> w[co][dz, dy, dx'] for dx' <= 4, co < Cout
__global__ void stem1_pack_kernel(const float* __restrict__ w, bf16* __restrict__ wq, int Cout) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 13 * 2 * 32 * 8; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int j = t % 8; t /= 8;
        const int co = t % 32; t /= 32;
        const int h = t % 2; const int r = t / 2;
        const int row = 2 * r + h;
        float v = 0.f;
        if (co < Cout && row < 25 && j <= 4) v = w[(long long)co * 125 + (row / 5) * 25 + (row % 5) * 5 + j];
        wq[idx] = (bf16)v;
    }
}

// ---------------------------------------------------------------- forward
__global__ __launch_bounds__(256) void stem1_fwd_kernel(Stem1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, i = lane & 31;
    bf16x8_t wr[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) wr[r] = *reinterpret_cast<const bf16x8_t*>(a.wq + ((r * 2 + h) * 32 + i) * 8);
    int rowoff[13];
#pragma unroll
    for (int r = 0; r < 13; ++r) {
        int row = 2 * r + h;
        if (row > 24) row = 24;                                      // padding row: zero weights, any valid address
        rowoff[r] = ((row / 5) * S1_HY + row % 5) * S1_ROWB;
    }
    const int lane_off = (i & 7) * S1_COPYB + (i & ~7) * 2;          // copy i % 8, aligned 16-byte slot: voxels x_i .. x_i + 7
    float bv[8];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int co = 8 * g + 4 * h + c; bv[4 * g + c] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f; }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int n, z0, y0, x0;
        stem1_tile(a, tile, n, z0, y0, x0);
        __syncthreads();
        stem1_stage_x<8>(lds, a.x, n, z0, y0, x0, a.D, a.H, a.W);
        __syncthreads();
#pragma unroll
        for (int ly = 0; ly < S1_TY; ++ly) {
            f32x16 acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
            const int base = (wave * S1_HY + ly) * S1_ROWB + lane_off;
#pragma unroll
            for (int r = 0; r < 13; ++r) {
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(lds + base + rowoff[r]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[r], b, acc, 0, 0, 0);
            }
            const int gz = z0 + wave, gy = y0 + ly, gx = x0 + i;
            if (gz < a.D && gy < a.H && gx < a.W) {
                bf16* yp = a.y + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldy;
#pragma unroll
                for (int g = 0; g < 2; ++g) {                        // rows 8g + 4h .. +3 = registers 4g .. 4g+3 (output channels 0..15)
                    const int co = 8 * g + 4 * h;
                    if (co < a.Cout)
                        st4(yp + co, f32x4_t{acc[4 * g] + bv[4 * g], acc[4 * g + 1] + bv[4 * g + 1], acc[4 * g + 2] + bv[4 * g + 2], acc[4 * g + 3] + bv[4 * g + 3]});
                }
            }
        }
    }
}

// ---------------------------------------------------------------- weight gradient: part[blk][tap][co]
__global__ __launch_bounds__(256) void stem1_wgrad_kernel(Stem1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* ds = lds;                                         // [512 voxels][16 co] bf16 = 32-byte rows
    unsigned char* xs = lds + 512 * 32;                              // four copies of the x halo tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int li = lane & 15, q = li >> 2, p = li & 3, cg = (lane >> 4) & 1;
    const int lane_d = (8 * h + q) * 32 + 4 * p * 2;                 // both 16-lane groups read the 16 channels (rows 16..31 of the tile: duplicates)
    const int ux0 = 8 * h + q;
    const int lane_x = (ux0 & 3) * S1_COPYB + (ux0 & ~3) * 2 + 8 * p;     // copy u % 4 at the aligned slot below it, columns 4p .. 4p+3 = voxels +4p ..
    int rowoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int row = 2 * (wave + 4 * k) + cg;
        if (row > 24) row = 24;
        rowoff[k] = ((row / 5) * S1_HY + row % 5) * S1_ROWB;
    }
    const int nblocks = wave == 0 ? 4 : 3;                           // 13 N-blocks of 2 rows x 16 dx'
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[k][v] = 0.f;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int n, z0, y0, x0;
        stem1_tile(a, tile, n, z0, y0, x0);
        __syncthreads();
        stem1_stage_x<4>(xs, a.x, n, z0, y0, x0, a.D, a.H, a.W);
        for (int pc = tid; pc < 512 * 2; pc += 256) {                // dy tile: 512 voxels x 2 pieces of 8 channels
            const int vox = pc >> 1, part = pc & 1;
            const int xx = vox % S1_BX, line = vox / S1_BX, gz = z0 + line / S1_TY, gy = y0 + line % S1_TY, gx = x0 + xx;
            bf16x8_t dv = {};
            if (gz < a.D && gy < a.H && gx < a.W)
                dv = *reinterpret_cast<const bf16x8_t*>(a.dy + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldy + part * 8);
            *reinterpret_cast<bf16x8_t*>(ds + vox * 32 + part * 16) = dv;
        }
        __syncthreads();
        for (int line = 0; line < S1_TY * S1_TZ; ++line) {
            const int lz = line / S1_TY, ly = line % S1_TY;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int db = (line * S1_BX + ks * 16) * 32 + lane_d;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ds + db));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ds + db + 4 * 32));
                const bf16x8_t df = __builtin_bit_cast(bf16x8_t, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                const int xb = (lz * S1_HY + ly) * S1_ROWB + ks * 32 + lane_x;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < nblocks) {
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + xb + rowoff[k]));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + xb + rowoff[k] + 8));
                        const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, xf, acc[k], 0, 0, 0);
                    }
                }
            }
        }
    }
    // columns of block nb: lanes 16cg + li -> row 2nb + cg, dx' = li; rows of the tile = output channels (0..15 kept)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = 2 * (wave + 4 * k) + cg;
        if (k < nblocks && row < 25 && li <= 4) {
            const int tap = (row / 5) * 25 + (row % 5) * 5 + li;
            float* dst = a.part + ((long long)blockIdx.x * 125 + tap) * a.Cout;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int co = (v & 3) + 8 * (v >> 2) + 4 * h;
                if (co < a.Cout) dst[co] = acc[k][v];
            }
        }
    }
}

// ---------------------------------------------------------------- host side
bool stem1k5_lowp_supported(int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    return Cin == 1 && k == 5 && stride == 1 && pad == 2 && ldx == 1 && Cout == 16 && ldy % 8 == 0;
}
size_t stem1k5_lowp_ws_bytes(int Cout) { return align_up((size_t)13 * 2 * 32 * 8 * 2, 256) + align_up((size_t)512 * 125 * Cout * sizeof(float), 256) + 256; }

static void stem1_geom(Stem1Args& a, int N, int D, int H, int W) {
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.ntx = (W + S1_BX - 1) / S1_BX; a.nty = (H + S1_TY - 1) / S1_TY; a.ntz = (D + S1_TZ - 1) / S1_TZ;
    a.ntiles = N * a.ntx * a.nty * a.ntz;
}

int stem1k5_fwd_lowp(const bf16* x, const float* w, const float* bias, bf16* y, int ldy, int N, int D, int H, int W, int Cout,
                     void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_ARG(((uintptr_t)x % 2) == 0 && ((uintptr_t)y % 8) == 0, "stem1k5_fwd: y must be 8-byte aligned");
    Carver cv(ws);
    bf16* wq = cv.take<bf16>(13 * 2 * 32 * 8);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    hipLaunchKernelGGL(stem1_pack_kernel, dim3(8), dim3(256), 0, st, w, wq, Cout);
    SEG_CHECK_LAUNCH();
    Stem1Args a{x, nullptr, wq, bias, y, nullptr, ldy, 0, 0, 0, 0, Cout, 0, 0, 0, 0};
    stem1_geom(a, N, D, H, W);
    const int ldsb = 8 * S1_COPYB;
    SEG_SET_LDS((stem1_fwd_kernel), ldsb);
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_DIRECT, 2.0 * vox * 125.0 * Cout, 2.0 * vox * (1 + Cout), st);
    hipLaunchKernelGGL(stem1_fwd_kernel, dim3(a.ntiles < 2048 ? a.ntiles : 2048), dim3(256), ldsb, st, a);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int stem1k5_wgrad_lowp(const bf16* dy, int lddy, const bf16* x, float* dw, int N, int D, int H, int W, int Cout, int accumulate,
                       void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_ARG(((uintptr_t)dy % 16) == 0, "stem1k5_wgrad: dy must be 16-byte aligned");
    Stem1Args a{x, dy, nullptr, nullptr, nullptr, nullptr, lddy, 0, 0, 0, 0, Cout, 0, 0, 0, 0};
    stem1_geom(a, N, D, H, W);
    const int nblk = a.ntiles < 512 ? a.ntiles : 512;
    Carver cv(ws);
    a.part = cv.take<float>((size_t)nblk * 125 * Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    const int ldsb = 512 * 32 + 4 * S1_COPYB;
    SEG_SET_LDS((stem1_wgrad_kernel), ldsb);
    const double vox = (double)N * D * H * W;
    {
        ProfScope ps(PF_DIRECT, 2.0 * vox * 125.0 * Cout, 2.0 * vox * (1 + Cout), st);
        hipLaunchKernelGGL(stem1_wgrad_kernel, dim3(nblk), dim3(256), ldsb, st, a);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(a.part, dw, nblk, 125, 1, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // namespace seg
