// conv_head2_lowp.hip -- V-Net's two-channel k5 head (OutputTransition.conv1: Conv3d(32, 2, k5, p2), vnet3d.py:107-121) on
// the bf16 matrix cores: forward, data gradient and weight gradient for bf16 tensors.
//
// With N = 2 an implicit-GEMM tile is 94 % padding, so round 1 served this layer with z-marching VALU kernels (fp32; the bf16
// path went through them on workspace copies: 1.2 + 0.9 + 1.1 ms + 0.5 ms of casts per step at [2, ., 128^3]).  Here the five
// x-taps join the two channels on the GEMM's narrow axis -- (dx, co) = 10 of 16 / 32 columns instead of 2 of 32:
//   forward   Z[u][(dx, co)] = sum_{dz, dy, ci} x[u + (dz, dy, 0)][ci] * w[co][ci][dz, dy, dx]      K = 25 * Cin
//             y[v][co]       = sum_dx Z[v + dx - 2][(dx, co)]                                       (shifted sum through LDS)
//   dgrad     dx[v][ci]      = sum_{dz, dy} sum_{(dx', co)} dy[v - 2 + (dz, dy, dx')][co] * w[co][ci][4 - dz, 4 - dy, 4 - dx']
//             one 16-deep k-step per (dz, dy) row: k = (dx' in 0..7, co), dx' > 4 zero-weighted
//   wgrad     dW[ci][(dz, dy, dx', co)] = sum_u x[u][ci] * dy[u - 2 + (dz, dy, dx')][co]            K = voxels, transposing reads
// The two-channel tensor is 4 bytes per voxel: the 16 contiguous bytes a lane needs (four x-neighbours) start at any voxel,
// so the dgrad keeps FOUR copies of the dy halo tile in LDS, copy c shifted by c voxels, and lane i reads copy i % 4 at an
// aligned slot; the wgrad's transposing reads need 8-byte alignment: two copies, picked by the parity of the row's voxel.
#include "common.h"
#include "internal.h"

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int H2_BX = 32, H2_TY = 4, H2_TZ = 4;               // dgrad / wgrad tile: 512 voxels, wave w owns z-slab w (dgrad)
constexpr int H2_HY = H2_TY + 4, H2_HZ = H2_TZ + 4;
constexpr int H2_ROWW = 44;                                    // voxels per halo row of a copy: t = gx - (x0 - 2)
constexpr int H2_ROWB = H2_ROWW * 4;                           // 176 bytes
constexpr int H2_COPYB = H2_HZ * H2_HY * H2_ROWB + 16;         // copies staggered by one 16-byte slot

struct Head2Args {
    const bf16* x; const bf16* dy; const bf16* wq; const float* bias; bf16* out; float* part;
    int ldx, lddy, ldo, N, D, H, W, Cin, ntx, nty, ntz, ntiles;
};

// NCOPY copies of the (two-channel) dy halo tile: copy c, row (hz, hy), slot j holds dy[z0 - 2 + hz][y0 - 2 + hy][x0 - 2 + j + c].
// Every source voxel is loaded once and written to the (up to NCOPY) slots that show it.
template <int NCOPY>
__device__ __forceinline__ void head2_stage_dy(unsigned char* ds, const bf16* __restrict__ dy, int lddy, int n, int z0, int y0, int x0,
                                               int D, int H, int W, int nthreads) {
    constexpr int SRCW = H2_ROWW + NCOPY - 1;
    for (int e = threadIdx.x; e < H2_HZ * H2_HY * SRCW; e += nthreads) {
        const int t = e % SRCW, r = e / SRCW;
        const int gz = z0 - 2 + r / H2_HY, gy = y0 - 2 + r % H2_HY, gx = x0 - 2 + t;
        unsigned v = 0u;
        if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
            v = *reinterpret_cast<const unsigned*>(dy + ((((long long)n * D + gz) * H + gy) * W + gx) * lddy);
#pragma unroll
        for (int c = 0; c < NCOPY; ++c) {
            const int j = t - c;
            if (j >= 0 && j < H2_ROWW) *reinterpret_cast<unsigned*>(ds + c * H2_COPYB + r * H2_ROWB + j * 4) = v;
        }
    }
}

__device__ __forceinline__ void head2_tile(const Head2Args& a, int tile, int& n, int& z0, int& y0, int& x0) {
    int mt = tile;
    x0 = (mt % a.ntx) * H2_BX; mt /= a.ntx;
    y0 = (mt % a.nty) * H2_TY; mt /= a.nty;
    z0 = (mt % a.ntz) * H2_TZ; n = mt / a.ntz;
}

// ---------------------------------------------------------------- weight packing
// mode 0 (dgrad): wq[cib][r = (sz, sy)][h][ci 32][8]   element j: dx' = 4h + j/2, co = j%2 -> w[co][ci][4 - sz, 4 - sy, 4 - dx'] (dx' <= 4)
// mode 1 (fwd):   wq[chunk][r = (sz, sy)][h][n 32][8]  element j: ci = 16 chunk + 8h + j, n = (dx, co) -> w[co][ci][sz, sy, dx]  (n < 10)
__global__ void head2_pack_kernel(const float* __restrict__ w, bf16* __restrict__ wq, int Cin, int mode) {
    const int total = (mode == 0 ? Cin / 32 : Cin / 16) * 25 * 2 * 32 * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int t = idx;
        const int j = t % 8; t /= 8;
        const int c = t % 32; t /= 32;
        const int h = t % 2; t /= 2;
        const int r = t % 25; const int blk = t / 25;
        const int sz = r / 5, sy = r % 5;
        float v = 0.f;
        if (mode == 0) {
            const int dxp = 4 * h + j / 2, co = j % 2, ci = blk * 32 + c;
            if (dxp <= 4) v = w[((long long)co * Cin + ci) * 125 + (4 - sz) * 25 + (4 - sy) * 5 + (4 - dxp)];
        } else {
            const int ci = blk * 16 + 8 * h + j, dxp = c / 2, co = c % 2;
            if (c < 10) v = w[((long long)co * Cin + ci) * 125 + sz * 25 + sy * 5 + dxp];
        }
        wq[idx] = (bf16)v;
    }
}

// ---------------------------------------------------------------- data gradient: dx [., Cin] from dy [., 2]
// grid = (tiles walked persistently, Cin / 32); the 25 weight fragments of the block's 32 input channels stay in registers
__global__ __launch_bounds__(256) void head2_dgrad_kernel(Head2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, i = lane & 31;
    const int cib = blockIdx.y;
    bf16x8_t wd[25];
#pragma unroll
    for (int r = 0; r < 25; ++r) wd[r] = *reinterpret_cast<const bf16x8_t*>(a.wq + ((((long long)cib * 25 + r) * 2 + h) * 32 + i) * 8);
    const int lane_off = (i & 3) * H2_COPYB + ((i & ~3) + 4 * h) * 4;       // copy i % 4, aligned 16-byte slot
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int n, z0, y0, x0;
        head2_tile(a, tile, n, z0, y0, x0);
        __syncthreads();
        head2_stage_dy<4>(lds, a.dy, a.lddy, n, z0, y0, x0, a.D, a.H, a.W, 256);
        __syncthreads();
#pragma unroll
        for (int ly = 0; ly < H2_TY; ++ly) {
            f32x16 acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
            for (int r = 0; r < 25; ++r) {
                const int sz = r / 5, sy = r % 5;
                const bf16x8_t b = *reinterpret_cast<const bf16x8_t*>(lds + lane_off + ((wave + sz) * H2_HY + ly + sy) * H2_ROWB);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wd[r], b, acc, 0, 0, 0);
            }
            const int gz = z0 + wave, gy = y0 + ly, gx = x0 + i;
            if (gz < a.D && gy < a.H && gx < a.W) {
                bf16* op = a.out + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldo + cib * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g)                              // registers 4g .. 4g+3 = input channels 8g + 4h .. +3 of the block
                    st4(op + 8 * g + 4 * h, f32x4_t{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]});
            }
        }
    }
}

// ---------------------------------------------------------------- weight gradient
// grid = (blocks walking tiles, Cin / 32).  part[blk][tap][ci][co] (wgrad_reduce sums the blocks in fixed order).
// 13 N-blocks of 32 columns = 2 (dz, dy) rows x (8 dx' x 2 co); wave w owns N-blocks w, w + 4, w + 8 (, 12): no cross-wave sum.
__global__ __launch_bounds__(256) void head2_wgrad_kernel(Head2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xs = lds;                                         // [512 voxels][32 ci] bf16 = 64-byte rows
    unsigned char* ds = lds + 512 * 64;                              // two copies of the dy halo tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int cib = blockIdx.y;
    const int li = lane & 15, q = li >> 2, p = li & 3, cg = (lane >> 4) & 1;
    const int lane_xs = (8 * h + q) * 64 + (16 * cg + 4 * p) * 2;
    // dy rows: voxel u_x = 16 ks + 8h + q (+ 4): copy u_x % 2 at the even slot below it, columns 4p .. 4p+3 = voxels +2p, +2p+1
    const int ux0 = 8 * h + q;
    const int lane_ds = (ux0 & 1) * H2_COPYB + ((ux0 & ~1) + 2 * p) * 4;
    int rowoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int r = 2 * (wave + 4 * k) + cg;                             // (dz, dy) row of this 16-lane group; 25 = padding
        if (r > 24) r = 24;
        rowoff[k] = ((r / 5) * H2_HY + r % 5) * H2_ROWB;
    }
    const int nblocks = wave == 0 ? 4 : 3;                           // wave-uniform
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[k][v] = 0.f;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        int n, z0, y0, x0;
        head2_tile(a, tile, n, z0, y0, x0);
        __syncthreads();
        head2_stage_dy<2>(ds, a.dy, a.lddy, n, z0, y0, x0, a.D, a.H, a.W, 256);
        for (int pc = tid; pc < 512 * 4; pc += 256) {                // x tile: 512 voxels x 4 pieces of 8 channels
            const int vox = pc >> 2, part = pc & 3;
            const int xx = vox % H2_BX, line = vox / H2_BX, gz = z0 + line / H2_TY, gy = y0 + line % H2_TY, gx = x0 + xx;
            bf16x8_t xv = {};
            if (gz < a.D && gy < a.H && gx < a.W)
                xv = *reinterpret_cast<const bf16x8_t*>(a.x + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldx + cib * 32 + part * 8);
            *reinterpret_cast<bf16x8_t*>(xs + vox * 64 + part * 16) = xv;
        }
        __syncthreads();
        for (int line = 0; line < H2_TY * H2_TZ; ++line) {
            const int lz = line / H2_TY, ly = line % H2_TY;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int xb = (line * H2_BX + ks * 16) * 64 + lane_xs;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + xb));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + xb + 4 * 64));
                const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                const int db = (lz * H2_HY + ly) * H2_ROWB + ks * 64 + lane_ds;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < nblocks) {
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ds + db + rowoff[k]));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ds + db + rowoff[k] + 16));
                        const bf16x8_t df = __builtin_bit_cast(bf16x8_t, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, df, acc[k], 0, 0, 0);
                    }
                }
            }
        }
    }
    // columns of block nb: lanes 16cg + li -> row 2nb + cg, dx' = li / 2, co = li % 2; rows of the tile = input channels
    const int dxp = li >> 1, co = li & 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = 2 * (wave + 4 * k) + cg;
        if (k < nblocks && r < 25 && dxp <= 4) {
            const int tap = (4 - r / 5) * 25 + (4 - r % 5) * 5 + (4 - dxp);
            float* dst = a.part + (((long long)blockIdx.x * 125 + tap) * a.Cin + cib * 32) * 2 + co;
#pragma unroll
            for (int v = 0; v < 16; ++v) dst[((v & 3) + 8 * (v >> 2) + 4 * h) * 2] = acc[k][v];
        }
    }
}

// ---------------------------------------------------------------- forward: y [., 2] from x [., Cin]
// tile = 28 (x outputs; 32 u-positions with the +-2 halo) x 4 (y) x 2 (z); wave (lz, ypair) owns lines y = 2 ypair, 2 ypair + 1
constexpr int HF_OX = 28, HF_TY = 4, HF_TZ = 2, HF_HY = HF_TY + 4, HF_HZ = HF_TZ + 4, HF_PITCH = 48;
constexpr int HF_XBYTES = HF_HZ * HF_HY * 32 * HF_PITCH;             // 73,728: one 16-channel chunk of the halo tile
constexpr int HF_ZP = 11;                                            // floats per (line, u) in the epilogue buffer (10 used)

__global__ __launch_bounds__(256, 2) void head2_fwd_kernel(Head2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, i = lane & 31;
    const int lz = wave >> 1, yp = wave & 1;
    int mt = blockIdx.x;
    const int x0 = (mt % a.ntx) * HF_OX; mt /= a.ntx;
    const int y0 = (mt % a.nty) * HF_TY; mt /= a.nty;
    const int z0 = (mt % a.ntz) * HF_TZ; const int n = mt / a.ntz;
    constexpr int NPIECE = HF_HZ * HF_HY * 32 * 2, NIT = NPIECE / 256;      // 16-byte pieces of one chunk: 12 per thread
    bf16x8_t stage[NIT];
    auto load_stage = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pc = it * 256 + tid, part = pc & 1, vox = pc >> 1;
            const int u = vox % 32, r = vox / 32, gz = z0 - 2 + r / HF_HY, gy = y0 - 2 + r % HF_HY, gx = x0 - 2 + u;
            bf16x8_t v = {};
            if ((unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                v = *reinterpret_cast<const bf16x8_t*>(a.x + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldx + chunk * 16 + part * 8);
            stage[it] = v;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pc = it * 256 + tid;
            *reinterpret_cast<bf16x8_t*>(lds + (pc >> 1) * HF_PITCH + (pc & 1) * 16) = stage[it];
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[mb][v] = 0.f;
    const int nchunks = a.Cin / 16;
    const int abase = ((lz * HF_HY + 2 * yp) * 32 + i) * HF_PITCH + 16 * h;
    load_stage(0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const bf16* wp = a.wq + ((long long)chunk * 25 * 2 + h) * 32 * 8 + i * 8;     // + r * 512 per (sz, sy) row
        constexpr int PFD = 3;
        bf16x8_t bq[PFD + 1];
#pragma unroll
        for (int d = 0; d < PFD; ++d) bq[d] = *reinterpret_cast<const bf16x8_t*>(wp + d * 512);
        __syncthreads();                                             // every wave is done reading the previous chunk
        write_stage();
        __syncthreads();
        if (chunk + 1 < nchunks) load_stage(chunk + 1);
#pragma unroll
        for (int r = 0; r < 25; ++r) {
            if (r + PFD < 25) bq[(r + PFD) % (PFD + 1)] = *reinterpret_cast<const bf16x8_t*>(wp + (r + PFD) * 512);
            const int sz = r / 5, sy = r % 5;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const bf16x8_t av = *reinterpret_cast<const bf16x8_t*>(lds + abase + ((sz * HF_HY + mb + sy) * 32) * HF_PITCH);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bq[r % (PFD + 1)], acc[mb], 0, 0, 0);
            }
        }
    }
    // Z -> LDS [line][u][(dx, co)], then every output sums its five shifted entries
    __syncthreads();
    float* zb = reinterpret_cast<float*>(lds);
    if (i < 10) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) zb[((wave * 2 + mb) * 32 + (v & 3) + 8 * (v >> 2) + 4 * h) * HF_ZP + i] = acc[mb][v];
    }
    __syncthreads();
    for (int t = tid; t < 8 * HF_OX * 2; t += 256) {
        const int line = t / (HF_OX * 2), rem = t % (HF_OX * 2), xl = rem >> 1, co = rem & 1;
        const int gz = z0 + line / HF_TY, gy = y0 + line % HF_TY, gx = x0 + xl;
        if (gz < a.D && gy < a.H && gx < a.W) {
            float s = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int dxp = 0; dxp < 5; ++dxp) s += zb[(line * 32 + xl + dxp) * HF_ZP + dxp * 2 + co];
            a.out[((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldo + co] = (bf16)s;
        }
    }
}

// ---------------------------------------------------------------- host side
bool head2_lowp_supported(int Cin, int Cout, int k, int stride, int pad, int ld_wide, int ld_narrow) {
    return Cout == 2 && k == 5 && stride == 1 && pad == 2 && Cin % 32 == 0 && Cin <= 256 && ld_wide % 8 == 0 && ld_narrow % 2 == 0;
}
size_t head2_lowp_ws_bytes(int Cin) {
    const size_t wq = align_up((size_t)(Cin / 16) * 25 * 2 * 32 * 8 * 2, 256);       // the larger of the two packings
    return wq + align_up((size_t)512 * 125 * Cin * 2 * sizeof(float), 256) + 256;
}

static void head2_geom(Head2Args& a, int N, int D, int H, int W, int bx, int ty, int tz) {
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.ntx = (W + bx - 1) / bx; a.nty = (H + ty - 1) / ty; a.ntz = (D + tz - 1) / tz;
    a.ntiles = N * a.ntx * a.nty * a.ntz;
}

int head2_fwd_lowp(const bf16* x, int ldx, const float* w, const float* bias, bf16* y, int ldy, int N, int D, int H, int W, int Cin,
                   void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 2) == 0, "head2_fwd: x must be 16-byte aligned");
    Carver cv(ws);
    bf16* wq = cv.take<bf16>((size_t)(Cin / 16) * 25 * 2 * 32 * 8);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    hipLaunchKernelGGL(head2_pack_kernel, dim3(32), dim3(256), 0, st, w, wq, Cin, 1);
    SEG_CHECK_LAUNCH();
    Head2Args a{x, nullptr, wq, bias, y, nullptr, ldx, 0, ldy, 0, 0, 0, 0, Cin, 0, 0, 0, 0};
    head2_geom(a, N, D, H, W, HF_OX, HF_TY, HF_TZ);
    SEG_SET_LDS((head2_fwd_kernel), HF_XBYTES);
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_DIRECT, 2.0 * vox * 125.0 * Cin * 2, 2.0 * vox * (Cin + 2), st);
    hipLaunchKernelGGL(head2_fwd_kernel, dim3(a.ntiles), dim3(256), HF_XBYTES, st, a);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int head2_dgrad_lowp(const bf16* dy, int lddy, const float* w, bf16* dx, int lddx, int N, int D, int H, int W, int Cin,
                     void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_ARG(((uintptr_t)dy % 4) == 0 && ((uintptr_t)dx % 8) == 0, "head2_dgrad: dy must be 4-byte and dx 8-byte aligned");
    Carver cv(ws);
    bf16* wq = cv.take<bf16>((size_t)(Cin / 32) * 25 * 2 * 32 * 8);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    hipLaunchKernelGGL(head2_pack_kernel, dim3(32), dim3(256), 0, st, w, wq, Cin, 0);
    SEG_CHECK_LAUNCH();
    Head2Args a{nullptr, dy, wq, nullptr, dx, nullptr, 0, lddy, lddx, 0, 0, 0, 0, Cin, 0, 0, 0, 0};
    head2_geom(a, N, D, H, W, H2_BX, H2_TY, H2_TZ);
    const int ldsb = 4 * H2_COPYB;
    SEG_SET_LDS((head2_dgrad_kernel), ldsb);
    const int grid = a.ntiles < 2048 ? a.ntiles : 2048;
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_DIRECT, 2.0 * vox * 125.0 * Cin * 2, 2.0 * vox * (Cin + 2), st);
    hipLaunchKernelGGL(head2_dgrad_kernel, dim3(grid, Cin / 32), dim3(256), ldsb, st, a);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int head2_wgrad_lowp(const bf16* dy, int lddy, const bf16* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int accumulate,
                     void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_ARG(((uintptr_t)dy % 4) == 0 && ((uintptr_t)x % 16) == 0, "head2_wgrad: dy must be 4-byte and x 16-byte aligned");
    Head2Args a{x, dy, nullptr, nullptr, nullptr, nullptr, ldx, lddy, 0, 0, 0, 0, 0, Cin, 0, 0, 0, 0};
    head2_geom(a, N, D, H, W, H2_BX, H2_TY, H2_TZ);
    const int nblk = a.ntiles < 512 ? a.ntiles : 512;
    Carver cv(ws);
    a.part = cv.take<float>((size_t)nblk * 125 * Cin * 2);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    const int ldsb = 512 * 64 + 2 * H2_COPYB;
    SEG_SET_LDS((head2_wgrad_kernel), ldsb);
    const double vox = (double)N * D * H * W;
    {
        ProfScope ps(PF_DIRECT, 2.0 * vox * 125.0 * Cin * 2, 2.0 * vox * (Cin + 2), st);
        hipLaunchKernelGGL(head2_wgrad_kernel, dim3(nblk, Cin / 32), dim3(256), ldsb, st, a);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(a.part, dw, nblk, 125, Cin, 2, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // namespace seg
