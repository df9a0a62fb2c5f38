// conv_headpw_lowp.hip -- the pointwise segmentation heads (Conv3d(C, 2 | 4, k1): unet3d.py:51, residual_unet3d.py's three
// deep-supervision heads) for bf16 tensors, forward and weight gradient.
//
// These layers are pure streams (64 bytes in, 4-8 bytes out per voxel at C = 32), but the VALU kernels they ran on split a
// voxel's channels over 8 lanes and paid three shuffle rounds per output channel: 0.40 ms for the Res-U-Net's full-resolution
// head against 0.06 ms of HBM time.  Here the matrix core does the channel contraction so the lanes only move data:
//   forward   y^T[co][v] = sum_ci W[co][ci] * x[v][ci]: B fragment = 16 bytes of the voxel straight from global memory, the
//             weight fragments (rows >= Cout zero) live in registers; lane (v, h = 0) ends up holding its voxel's outputs.
//   wgrad     dW^T[ci][co] = sum_v x[v][ci] * dy[v][co]: a K = voxels GEMM through transposing LDS reads; dy is staged four
//             channels wide (8-byte rows), so the read's 16-column block is (4 voxels) x (4 channels) and columns 0..3 are
//             the ones kept.
#include "common.h"
#include "internal.h"

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

struct HeadPwArgs { const bf16* x; const bf16* dy; const float* w; const float* bias; bf16* y; float* part; int ldx, ldy, Cin, Cout; long long nvox; int ntiles; };

// ---------------------------------------------------------------- forward
template <int KS16>          // Cin / 16
__global__ __launch_bounds__(256) void headpw_fwd_kernel(HeadPwArgs a) {
    const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
    bf16x8_t wf[KS16];
#pragma unroll
    for (int kk = 0; kk < KS16; ++kk) {
        bf16x8_t q = {};
        if (i < a.Cout) {
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = (bf16)a.w[(long long)i * a.Cin + 16 * kk + 8 * h + j];
        }
        wf[kk] = q;
    }
    float bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[c] = (a.bias && c < a.Cout) ? a.bias[c] : 0.f;
    const long long nblk = (a.nvox + 31) / 32;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    for (long long b = wid; b < nblk; b += 2 * nw) {                 // two 32-voxel blocks per trip: four loads in flight
        const long long b1 = b + nw;
        const long long v0 = b * 32 + i, v1 = b1 * 32 + i;
        const long long r0 = v0 < a.nvox ? v0 : a.nvox - 1, r1 = v1 < a.nvox ? v1 : a.nvox - 1;
        bf16x8_t x0[KS16], x1[KS16];
#pragma unroll
        for (int kk = 0; kk < KS16; ++kk) x0[kk] = *reinterpret_cast<const bf16x8_t*>(a.x + r0 * a.ldx + 16 * kk + 8 * h);
#pragma unroll
        for (int kk = 0; kk < KS16; ++kk) x1[kk] = *reinterpret_cast<const bf16x8_t*>(a.x + r1 * a.ldx + 16 * kk + 8 * h);
        f32x16 c0, c1;
#pragma unroll
        for (int v = 0; v < 16; ++v) { c0[v] = 0.f; c1[v] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < KS16; ++kk) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk], x0[kk], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk], x1[kk], c1, 0, 0, 0);
        }
        if (h == 0) {                                                // rows 0..3 of the tile = registers 0..3 of the h = 0 half
            if (v0 < a.nvox) {
                if (a.Cout == 4) st4(a.y + v0 * a.ldy, f32x4_t{c0[0] + bv[0], c0[1] + bv[1], c0[2] + bv[2], c0[3] + bv[3]});
                else { a.y[v0 * a.ldy] = (bf16)(c0[0] + bv[0]); a.y[v0 * a.ldy + 1] = (bf16)(c0[1] + bv[1]); }
            }
            if (b1 < nblk && v1 < a.nvox) {
                if (a.Cout == 4) st4(a.y + v1 * a.ldy, f32x4_t{c1[0] + bv[0], c1[1] + bv[1], c1[2] + bv[2], c1[3] + bv[3]});
                else { a.y[v1 * a.ldy] = (bf16)(c1[0] + bv[0]); a.y[v1 * a.ldy + 1] = (bf16)(c1[1] + bv[1]); }
            }
        }
    }
}

// ---------------------------------------------------------------- weight gradient
constexpr int HPW_V = 256;                                            // voxels per tile: 16 k-steps, four per wave
constexpr int HPW_DROWS = HPW_V + 8;                                  // dy rows (8 bytes each) incl. the rows a block's last read overlaps

// grid = (blocks walking tiles, Cin / 32).  part[blk][ci][co]
__global__ __launch_bounds__(256) void headpw_wgrad_kernel(HeadPwArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char xs[HPW_V * 64];
    __shared__ __attribute__((aligned(16))) unsigned char ds[HPW_DROWS * 8];
    __shared__ float red[4 * 32 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int cib = blockIdx.y;
    const int li = lane & 15, q = li >> 2, p = li & 3, cg = (lane >> 4) & 1;
    const int lane_xs = (8 * h + q) * 64 + (16 * cg + 4 * p) * 2;
    const int lane_ds = (8 * h + q + p) * 8;                         // row q, columns 4p .. 4p+3 = the four channels of voxel q + p
    for (int e = tid; e < 8; e += 256) *reinterpret_cast<unsigned long long*>(ds + (HPW_V + e) * 8) = 0ull;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const long long v0 = (long long)tile * HPW_V;
        __syncthreads();
        for (int pc = tid; pc < HPW_V * 4; pc += 256) {              // x tile: 256 voxels x 4 pieces of 8 channels
            const int vl = pc >> 2, part = pc & 3;
            bf16x8_t xv = {};
            if (v0 + vl < a.nvox) xv = *reinterpret_cast<const bf16x8_t*>(a.x + (v0 + vl) * a.ldx + cib * 32 + part * 8);
            *reinterpret_cast<bf16x8_t*>(xs + vl * 64 + part * 16) = xv;
        }
        {                                                            // dy tile, four channels wide (Cout = 2: channels 2, 3 zero)
            bf16x4_t dv = {};
            if (v0 + tid < a.nvox) {
                const bf16* dp = a.dy + (v0 + tid) * a.ldy;
                dv[0] = dp[0]; dv[1] = dp[1];
                if (a.Cout == 4) { dv[2] = dp[2]; dv[3] = dp[3]; }
            }
            *reinterpret_cast<bf16x4_t*>(ds + tid * 8) = dv;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ks = wave * 4 + s;
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + ks * 16 * 64 + lane_xs));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xs + ks * 16 * 64 + lane_xs + 4 * 64));
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ds + ks * 16 * 8 + lane_ds));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ds + ks * 16 * 8 + lane_ds + 4 * 8));
            const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            const bf16x8_t df = __builtin_bit_cast(bf16x8_t, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, df, acc, 0, 0, 0);
        }
    }
    // columns 0..3 (lanes 0..3 of the h halves: group cg = 0, li < 4) = output channels; rows = input channels
    __syncthreads();
    if (cg == 0 && li < 4) {
#pragma unroll
        for (int v = 0; v < 16; ++v) red[(wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * h) * 4 + li] = acc[v];
    }
    __syncthreads();
    for (int e = tid; e < 32 * a.Cout; e += 256) {
        const int ci = e / a.Cout, co = e % a.Cout;
        const float s = ((red[ci * 4 + co] + red[(32 + ci) * 4 + co]) + red[(64 + ci) * 4 + co]) + red[(96 + ci) * 4 + co];
        a.part[((long long)blockIdx.x * a.Cin + cib * 32 + ci) * a.Cout + co] = s;
    }
}

// ---------------------------------------------------------------- host side
bool headpw_lowp_supported(int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    return k == 1 && stride == 1 && pad == 0 && (Cout == 2 || Cout == 4) && Cin % 32 == 0 && Cin <= 128 && ldx % 8 == 0 && ldy % Cout == 0;
}
size_t headpw_lowp_ws_bytes(int Cin, int Cout) { return align_up((size_t)512 * Cin * Cout * sizeof(float), 256) + 256; }

int headpw_fwd_lowp(const bf16* x, int ldx, const float* w, const float* bias, bf16* y, int ldy, long long nvox, int Cin, int Cout, hipStream_t st) {
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % (2 * Cout)) == 0, "headpw_fwd: x must be 16-byte aligned, y aligned to a voxel");
    HeadPwArgs a{x, nullptr, w, bias, y, nullptr, ldx, ldy, Cin, Cout, nvox, 0};
    const long long nblk = (nvox + 31) / 32;
    long long g = (nblk + 15) / 16;                                   // >= 4 block pairs per wave
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    ProfScope ps(PF_DIRECT, 2.0 * nvox * Cin * Cout, 2.0 * nvox * (Cin + Cout), st);
    switch (Cin / 16) {
        case 2: hipLaunchKernelGGL((headpw_fwd_kernel<2>), dim3((unsigned)g), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((headpw_fwd_kernel<4>), dim3((unsigned)g), dim3(256), 0, st, a); break;
        case 6: hipLaunchKernelGGL((headpw_fwd_kernel<6>), dim3((unsigned)g), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((headpw_fwd_kernel<8>), dim3((unsigned)g), dim3(256), 0, st, a); break;
    }
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int headpw_wgrad_lowp(const bf16* dy, int lddy, const bf16* x, int ldx, float* dw, long long nvox, int Cin, int Cout, int accumulate,
                      void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 2) == 0, "headpw_wgrad: x must be 16-byte aligned");
    const int ntiles = (int)((nvox + HPW_V - 1) / HPW_V);
    const int nblk = ntiles < 512 ? ntiles : 512;
    SEG_CHECK_WS((size_t)nblk * Cin * Cout * sizeof(float), ws_bytes);
    HeadPwArgs a{x, dy, nullptr, nullptr, nullptr, (float*)ws, ldx, lddy, Cin, Cout, nvox, ntiles};
    {
        ProfScope ps(PF_DIRECT, 2.0 * nvox * Cin * Cout, 2.0 * nvox * (Cin + Cout), st);
        hipLaunchKernelGGL(headpw_wgrad_kernel, dim3(nblk, Cin / 32), dim3(256), 0, st, a);
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(a.part, dw, nblk, 1, Cin, Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // namespace seg
