// conv_mfma_bf16.hip -- EXPERIMENTAL, opt-in: Conv3d k3 s1 p1 forward / dgrad with bf16 MFMA operands and fp32 accumulation.
//
// Same implicit GEMM as conv_mfma.hip (halo tile in LDS, 27 taps as LDS shifts, 4 waves x MB x NBW 32x32 accumulators, weight
// fragments in a register ring, XCD-aware brick walk), but activations are rounded to bf16 (RNE) while they are staged
// and the weights are packed as bf16, so one `v_mfma_f32_32x32x16_bf16` (32 cycles) replaces the eight
// `v_mfma_f32_32x32x2_f32` (8 x 64 cycles) of a 16-channel tap: 16x the matrix rate.  Inputs and outputs stay fp32 in HBM,
// which makes the kernel staging-bound (the fp32 halo tile moves as before); it exists to put a measured number under the
// bf16 discussion in DESIGN.md and is NOT used by the parity-graded paths (results differ from fp32 by bf16 rounding of
// the operands: the test compares against an fp64 convolution of bf16-rounded tensors).
#include "common.h"
#include "internal.h"

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Bf16Args {
    const float* x; const __bf16* wq; const float* bias; float* y;
    int ldx, ldy, N, D, H, W, Cout, ntx, nty, ntz, nN, nchunks, by, bz;
};

template <int BX, int MB>
struct BTile {
    static constexpr int CK = 16, PITCH = 24;                 // bf16 elements per voxel: 16 data + 8 pad (48 B, odd 16-byte slots)
    static constexpr int LPB = 32 / BX, LINES = 4 * MB * LPB, TY = 4, TZ = LINES / TY;
    static constexpr int HX = BX + 2, HY = TY + 2, HZ = TZ + 2, NVOX = HX * HY * HZ;
    static constexpr int NPIECE = NVOX * 4, NITER = (NPIECE + 255) / 256;
    static constexpr int LDS_BYTES = NVOX * PITCH * 2;
};

// wq[nt][chunk][tap][h][j][e] = bf16(B[k = chunk*16 + 8h + e][n = nt*NT + j]) with B as in conv_mfma.hip's modes 0 / 1
__global__ void pack_wq_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wq, int K, int Nn, int NT, int dgrad) {
    const long long total = (long long)K * Nn * 27;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        const int e = (int)(r % 8); r /= 8;
        const int j = (int)(r % NT); r /= NT;
        const int h = (int)(r % 2); r /= 2;
        const int tap = (int)(r % 27); r /= 27;
        const int chunk = (int)(r % (K / 16)); r /= (K / 16);
        const int n = (int)r * NT + j, k = chunk * 16 + 8 * h + e;
        const float v = dgrad ? w[((long long)k * Nn + n) * 27 + (26 - tap)] : w[((long long)n * K + k) * 27 + tap];
        wq[idx] = (__bf16)v;
    }
}

template <int BX, int MB, int NBW>
__global__ __launch_bounds__(256, 2) void conv_igemm_bf16_kernel(Bf16Args a) {
    using T = BTile<BX, MB>;
    constexpr int NT = 32 * NBW, PITCH = T::PITCH;
    constexpr int STEP = 2 * NT * 8;                          // bf16 elements of packed weights per tap
    constexpr int CHUNK = 27 * STEP;
    extern __shared__ __attribute__((aligned(16))) __bf16 lds16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ntile = t % a.nN;
    int mt = t / a.nN;
    const int per_n = a.ntx * a.nty * a.ntz;
    const int n = mt / per_n; mt -= n * per_n;
    const int nby = a.nty / a.by, zfull = a.ntz / a.bz, rowtiles = a.ntx * a.nty * a.bz;
    int zrow = mt / rowtiles, bzz = a.bz;
    if (zrow >= zfull) { zrow = zfull; bzz = a.ntz - zfull * a.bz; }
    mt -= zrow * rowtiles;
    const int blk = a.ntx * a.by * bzz;
    const int b = mt / blk; mt -= b * blk;
    (void)nby;
    const int txi = mt % a.ntx; mt /= a.ntx;
    const int tyi = b * a.by + mt % a.by, tzi = zrow * a.bz + mt / a.by;
    const int x0 = txi * BX, y0 = tyi * T::TY, z0 = tzi * T::TZ, n0 = ntile * NT;

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mb][nb][v] = 0.f;
    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = wave * MB + mb;
        const int line = m * T::LPB + i / BX, xx = i % BX;
        abase[mb] = (((line / T::TY) * T::HY + (line % T::TY)) * T::HX + xx) * PITCH + 8 * h;
    }
    const __bf16* wlane = a.wq + (long long)ntile * a.nchunks * CHUNK + (h * NT + i) * 8;

    f32x4 stage[T::NITER];
    auto load_stage = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < T::NITER; ++it) {
            const int p = it * 256 + tid;
            const int vox = p >> 2, part = p & 3;
            const int hz = vox / (T::HY * T::HX), rem = vox % (T::HY * T::HX);
            const int hy = rem / T::HX, hx = rem % T::HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p < T::NPIECE && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                v = *reinterpret_cast<const f32x4*>(a.x + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldx + chunk * 16 + part * 4);
            stage[it] = v;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < T::NITER; ++it) {
            const int p = it * 256 + tid;
            if (p < T::NPIECE) {
                bf16x4 q;
                q[0] = (__bf16)stage[it][0]; q[1] = (__bf16)stage[it][1]; q[2] = (__bf16)stage[it][2]; q[3] = (__bf16)stage[it][3];
                *reinterpret_cast<bf16x4*>(lds16 + (p >> 2) * PITCH + (p & 3) * 4) = q;
            }
        }
    };

    load_stage(0);
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const __bf16* wp = wlane + (long long)chunk * CHUNK;
        constexpr int PFD = 4;
        bf16x8 bq[PFD + 1][NBW];
#pragma unroll
        for (int d = 0; d < PFD; ++d)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) bq[d][nb] = *reinterpret_cast<const bf16x8*>(wp + d * STEP + nb * 256);
        __syncthreads();
        write_stage();
        __syncthreads();
        if (chunk + 1 < a.nchunks) load_stage(chunk + 1);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int tapoff = ((dz * T::HY + dy) * T::HX + dx) * PITCH;
            const int cur = tap % (PFD + 1), fill = (tap + PFD) % (PFD + 1);
            if (tap + PFD < 27) {
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) bq[fill][nb] = *reinterpret_cast<const bf16x8*>(wp + (tap + PFD) * STEP + nb * 256);
            }
            bf16x8 av[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = *reinterpret_cast<const bf16x8*>(lds16 + abase[mb] + tapoff);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[mb], bq[cur][nb], acc[mb][nb], 0, 0, 0);
        }
    }

#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int col = n0 + nb * 32 + i;
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = wave * MB + mb;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
                const int line = m * T::LPB + r / BX, xx = r % BX;
                const int gz = z0 + line / T::TY, gy = y0 + line % T::TY, gx = x0 + xx;
                if (gz < a.D && gy < a.H && gx < a.W)
                    a.y[((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldy + col] = acc[mb][nb][v] + bv;
            }
        }
    }
}

template <int BX, int MB, int NBW>
static void launch_bf16(const Bf16Args& a, int nwg, hipStream_t st) {
    using T = BTile<BX, MB>;
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BX, MB, NBW>), dim3(nwg), dim3(256), T::LDS_BYTES, st, a);
}

// ---------------------------------------------------------------- bf16x6: fp32-accurate products on the bf16 matrix cores
// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) represents an fp32 number to ~2^-24; a product then needs
// the six bf16 MFMAs hh, hm, mh, mm, hl, lh (the three dropped cross terms are below 2^-23 of the product) = 6 x 32 cycles
// against 8 x 64 for the fp32 MFMA: 2.7x the matrix rate at fp32-level accuracy.  The halo tile stays fp32 in LDS (same 65 KB
// as conv_mfma.hip, two workgroups per CU); the A fragment is split in registers after the LDS read, in the shadow of the
// MFMAs; the weights are split once, in the pack kernel.  EXPERIMENTAL / opt-in like the plain bf16 kernel.
struct Bf16x6Args {
    const float* x; const __bf16* wq; const float* bias; float* y;
    int ldx, ldy, N, D, H, W, Cout, ntx, nty, ntz, nN, nchunks, by, bz;
};

// wq[nt][chunk][tap][plane][h][j][e], plane 0/1/2 = h/m/l of B[k = chunk*16 + 8h + e][n = nt*NT + j]
__global__ void pack_wq_bf16x3_kernel(const float* __restrict__ w, __bf16* __restrict__ wq, int K, int Nn, int NT, int dgrad) {
    const long long total = (long long)K * Nn * 27;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        const int e = (int)(r % 8); r /= 8;
        const int j = (int)(r % NT); r /= NT;
        const int h = (int)(r % 2); r /= 2;
        const int tap = (int)(r % 27); r /= 27;
        const int chunk = (int)(r % (K / 16)); r /= (K / 16);
        const int nt = (int)r;
        const int n = nt * NT + j, k = chunk * 16 + 8 * h + e;
        const float v = dgrad ? w[((long long)k * Nn + n) * 27 + (26 - tap)] : w[((long long)n * K + k) * 27 + tap];
        const __bf16 bh = (__bf16)v;
        const float r1 = v - (float)bh;
        const __bf16 bm = (__bf16)r1;
        const __bf16 bl = (__bf16)(r1 - (float)bm);
        const long long base = ((((long long)nt * (K / 16) + chunk) * 27 + tap) * 3) * (2 * NT * 8) + ((long long)h * NT + j) * 8 + e;
        wq[base] = bh; wq[base + 2 * NT * 8] = bm; wq[base + 4 * NT * 8] = bl;
    }
}

template <int BX, int MB, int NBW>
__global__ __launch_bounds__(256, 2) void conv_igemm_bf16x6_kernel(Bf16x6Args a) {
    constexpr int CK = 16, PITCH = CK + 4;                       // fp32 halo tile, as conv_mfma.hip
    constexpr int LPB = 32 / BX, LINES = 4 * MB * LPB, TY = 4, TZ = LINES / TY;
    constexpr int HX = BX + 2, HY = TY + 2, HZ = TZ + 2, NVOX = HX * HY * HZ;
    constexpr int NPIECE = NVOX * 4, NITER = (NPIECE + 255) / 256;
    constexpr int NT = 32 * NBW;
    constexpr int PLANE = 2 * NT * 8, STEP = 3 * PLANE, CHUNK = 27 * STEP;
    extern __shared__ __attribute__((aligned(16))) float ldsf[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ntile = t % a.nN;
    int mt = t / a.nN;
    const int per_n = a.ntx * a.nty * a.ntz;
    const int n = mt / per_n; mt -= n * per_n;
    const int zfull = a.ntz / a.bz, rowtiles = a.ntx * a.nty * a.bz;
    int zrow = mt / rowtiles, bzz = a.bz;
    if (zrow >= zfull) { zrow = zfull; bzz = a.ntz - zfull * a.bz; }
    mt -= zrow * rowtiles;
    const int blk = a.ntx * a.by * bzz;
    const int b = mt / blk; mt -= b * blk;
    const int txi = mt % a.ntx; mt /= a.ntx;
    const int tyi = b * a.by + mt % a.by, tzi = zrow * a.bz + mt / a.by;
    const int x0 = txi * BX, y0 = tyi * TY, z0 = tzi * TZ, n0 = ntile * NT;

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mb][nb][v] = 0.f;
    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = wave * MB + mb;
        const int line = m * LPB + i / BX, xx = i % BX;
        abase[mb] = (((line / TY) * HY + (line % TY)) * HX + xx) * PITCH + 8 * h;
    }
    const __bf16* wlane = a.wq + (long long)ntile * a.nchunks * CHUNK + (h * NT + i) * 8;

    f32x4 stage[NITER];
    auto load_stage = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            const int p = it * 256 + tid;
            const int vox = p >> 2, part = p & 3;
            const int hz = vox / (HY * HX), rem = vox % (HY * HX);
            const int hy = rem / HX, hx = rem % HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p < NPIECE && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                v = *reinterpret_cast<const f32x4*>(a.x + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldx + chunk * 16 + part * 4);
            stage[it] = v;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            const int p = it * 256 + tid;
            if (p < NPIECE) *reinterpret_cast<f32x4*>(ldsf + (p >> 2) * PITCH + (p & 3) * 4) = stage[it];
        }
    };

    load_stage(0);
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const __bf16* wp = wlane + (long long)chunk * CHUNK;
        constexpr int PFD = 2;
        bf16x8 bq[PFD + 1][NBW][3];
#pragma unroll
        for (int d = 0; d < PFD; ++d)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bq[d][nb][pl] = *reinterpret_cast<const bf16x8*>(wp + d * STEP + pl * PLANE + nb * 256);
        __syncthreads();
        write_stage();
        __syncthreads();
        if (chunk + 1 < a.nchunks) load_stage(chunk + 1);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int tapoff = ((dz * HY + dy) * HX + dx) * PITCH;
            const int cur = tap % (PFD + 1), fill = (tap + PFD) % (PFD + 1);
            if (tap + PFD < 27) {
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bq[fill][nb][pl] = *reinterpret_cast<const bf16x8*>(wp + (tap + PFD) * STEP + pl * PLANE + nb * 256);
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(ldsf + abase[mb] + tapoff);
                const f32x4 hi4 = *reinterpret_cast<const f32x4*>(ldsf + abase[mb] + tapoff + 4);
                bf16x8 ah, am, al;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = e < 4 ? lo4[e] : hi4[e - 4];
                    const __bf16 bh = (__bf16)v;
                    const float r1 = v - (float)bh;
                    const __bf16 bm = (__bf16)r1;
                    ah[e] = bh; am[e] = bm; al[e] = (__bf16)(r1 - (float)bm);
                }
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    f32x16 c = acc[mb][nb];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bq[cur][nb][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[cur][nb][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bq[cur][nb][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bq[cur][nb][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[cur][nb][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[cur][nb][0], c, 0, 0, 0);
                    acc[mb][nb] = c;
                }
            }
        }
    }

#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int col = n0 + nb * 32 + i;
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = wave * MB + mb;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
                const int line = m * LPB + r / BX, xx = r % BX;
                const int gz = z0 + line / TY, gy = y0 + line % TY, gx = x0 + xx;
                if (gz < a.D && gy < a.H && gx < a.W)
                    a.y[((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldy + col] = acc[mb][nb][v] + bv;
            }
        }
    }
}

// Variant with the three bf16 planes of the halo tile in LDS (split once at staging time, three 16-byte A reads per tap and no
// VALU work in the MFMA loop).  112 B per voxel: the 16-wide tile (648 halo voxels, 72.6 KB) still fits twice per CU.
template <int BX, int MB, int NBW>
__global__ __launch_bounds__(256, 2) void conv_igemm_bf16x6p_kernel(Bf16x6Args a) {
    constexpr int PITCH = 56;                                    // bf16 per voxel: three 16-channel planes (h, m, l) + 8 pad = 112 B (7 slots)
    constexpr int LPB = 32 / BX, LINES = 4 * MB * LPB, TY = 4, TZ = LINES / TY;
    constexpr int HX = BX + 2, HY = TY + 2, HZ = TZ + 2, NVOX = HX * HY * HZ;
    constexpr int NPIECE = NVOX * 4, NITER = (NPIECE + 255) / 256;
    constexpr int NT = 32 * NBW;
    constexpr int PLANE = 2 * NT * 8, STEP = 3 * PLANE, CHUNK = 27 * STEP;
    extern __shared__ __attribute__((aligned(16))) __bf16 ldsp[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ntile = t % a.nN;
    int mt = t / a.nN;
    const int per_n = a.ntx * a.nty * a.ntz;
    const int n = mt / per_n; mt -= n * per_n;
    const int zfull = a.ntz / a.bz, rowtiles = a.ntx * a.nty * a.bz;
    int zrow = mt / rowtiles, bzz = a.bz;
    if (zrow >= zfull) { zrow = zfull; bzz = a.ntz - zfull * a.bz; }
    mt -= zrow * rowtiles;
    const int blk = a.ntx * a.by * bzz;
    const int b = mt / blk; mt -= b * blk;
    const int txi = mt % a.ntx; mt /= a.ntx;
    const int tyi = b * a.by + mt % a.by, tzi = zrow * a.bz + mt / a.by;
    const int x0 = txi * BX, y0 = tyi * TY, z0 = tzi * TZ, n0 = ntile * NT;

    f32x16 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mb][nb][v] = 0.f;
    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = wave * MB + mb;
        const int line = m * LPB + i / BX, xx = i % BX;
        abase[mb] = (((line / TY) * HY + (line % TY)) * HX + xx) * PITCH + 8 * h;
    }
    const __bf16* wlane = a.wq + (long long)ntile * a.nchunks * CHUNK + (h * NT + i) * 8;

    f32x4 stage[NITER];
    auto load_stage = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            const int p = it * 256 + tid;
            const int vox = p >> 2, part = p & 3;
            const int hz = vox / (HY * HX), rem = vox % (HY * HX);
            const int hy = rem / HX, hx = rem % HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p < NPIECE && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                v = *reinterpret_cast<const f32x4*>(a.x + ((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldx + chunk * 16 + part * 4);
            stage[it] = v;
        }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
            const int p = it * 256 + tid;
            if (p < NPIECE) {                                    // split once per staged value (the register variant splits per tap)
                bf16x4 qh, qm, ql;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = stage[it][e];
                    const __bf16 bh = (__bf16)v;
                    const float r1 = v - (float)bh;
                    const __bf16 bm = (__bf16)r1;
                    qh[e] = bh; qm[e] = bm; ql[e] = (__bf16)(r1 - (float)bm);
                }
                __bf16* dst = ldsp + (p >> 2) * PITCH + (p & 3) * 4;
                *reinterpret_cast<bf16x4*>(dst) = qh;
                *reinterpret_cast<bf16x4*>(dst + 16) = qm;
                *reinterpret_cast<bf16x4*>(dst + 32) = ql;
            }
        }
    };

    load_stage(0);
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const __bf16* wp = wlane + (long long)chunk * CHUNK;
        constexpr int PFD = 2;
        bf16x8 bq[PFD + 1][NBW][3];
#pragma unroll
        for (int d = 0; d < PFD; ++d)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bq[d][nb][pl] = *reinterpret_cast<const bf16x8*>(wp + d * STEP + pl * PLANE + nb * 256);
        __syncthreads();
        write_stage();
        __syncthreads();
        if (chunk + 1 < a.nchunks) load_stage(chunk + 1);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int tapoff = ((dz * HY + dy) * HX + dx) * PITCH;
            const int cur = tap % (PFD + 1), fill = (tap + PFD) % (PFD + 1);
            if (tap + PFD < 27) {
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bq[fill][nb][pl] = *reinterpret_cast<const bf16x8*>(wp + (tap + PFD) * STEP + pl * PLANE + nb * 256);
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ldsp + abase[mb] + tapoff);
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(ldsp + abase[mb] + tapoff + 16);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(ldsp + abase[mb] + tapoff + 32);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    f32x16 c = acc[mb][nb];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bq[cur][nb][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[cur][nb][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bq[cur][nb][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bq[cur][nb][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[cur][nb][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bq[cur][nb][0], c, 0, 0, 0);
                    acc[mb][nb] = c;
                }
            }
        }
    }

#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int col = n0 + nb * 32 + i;
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const int m = wave * MB + mb;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
                const int line = m * LPB + r / BX, xx = r % BX;
                const int gz = z0 + line / TY, gy = y0 + line % TY, gx = x0 + xx;
                if (gz < a.D && gy < a.H && gx < a.W)
                    a.y[((((long long)n * a.D + gz) * a.H + gy) * a.W + gx) * a.ldy + col] = acc[mb][nb][v] + bv;
            }
        }
    }
}

template <int BX, int MB, int NBW>
static void launch_bf16x6p(const Bf16x6Args& a, int nwg, hipStream_t st) {
    constexpr int LINES = 4 * MB * (32 / BX), TZ = LINES / 4;
    constexpr int LDSB = (BX + 2) * 6 * (TZ + 2) * 56 * 2;
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)conv_igemm_bf16x6p_kernel<BX, MB, NBW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); set = true; }
    hipLaunchKernelGGL((conv_igemm_bf16x6p_kernel<BX, MB, NBW>), dim3(nwg), dim3(256), LDSB, st, a);
}

template <int BX, int MB, int NBW>
static void launch_bf16x6(const Bf16x6Args& a, int nwg, hipStream_t st) {
    constexpr int LINES = 4 * MB * (32 / BX), TZ = LINES / 4;
    constexpr int LDSB = (BX + 2) * 6 * (TZ + 2) * 20 * 4;
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)conv_igemm_bf16x6_kernel<BX, MB, NBW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB); set = true; }
    hipLaunchKernelGGL((conv_igemm_bf16x6_kernel<BX, MB, NBW>), dim3(nwg), dim3(256), LDSB, st, a);
}

}  // namespace seg

using namespace seg;

extern "C" {

size_t mi355seg_conv3d_bf16mma_ws_bytes(int Cin, int Cout) { return align_up((size_t)27 * Cin * Cout * 2, 256) + 256; }

// y = conv3d(x, w) (dgrad == 0, w = (Cout, Cin, 3,3,3), x has Cin channels) or dx = conv3d_dgrad(dy, w) (dgrad != 0: x is dy with
// Cout channels, y is dx with Cin channels), k3 s1 p1, operands rounded to bf16, fp32 accumulation.  Cin % 16 == 0 and Cout % 32 == 0
// on the GEMM's K / N side respectively, W >= 8.
int mi355seg_conv3d_bf16mma_f32(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                                int N, int D, int H, int W, int Cin, int Cout, int dgrad, void* ws, size_t ws_bytes, void* stream) {
    const int Kc = dgrad ? Cout : Cin, Nc = dgrad ? Cin : Cout;
    SEG_CHECK_ARG(x && w && y && N > 0 && D > 0 && H > 0 && W >= 8 && Kc % 16 == 0 && Nc % 32 == 0 && ldx >= Kc && ldy >= Nc && ldx % 4 == 0,
                  "conv3d_bf16mma: unsupported shape (K channels %% 16, N channels %% 32, W >= 8)");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0, "conv3d_bf16mma: input must be 16-byte aligned");
    SEG_CHECK_WS(mi355seg_conv3d_bf16mma_ws_bytes(Cin, Cout), ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    __bf16* wq = (__bf16*)ws;
    const int BX = (W % 32 == 0 || W > 48) ? 32 : 16;
    const int NBW = Nc % 64 == 0 ? 2 : 1, NT = 32 * NBW, MB = 2;
    const int TZ = (4 * MB * (32 / BX)) / 4;
    const long long total = (long long)27 * Cin * Cout;
    hipLaunchKernelGGL(pack_wq_bf16_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, st,
                       w, wq, Kc, Nc, NT, dgrad ? 1 : 0);
    SEG_CHECK_LAUNCH();
    Bf16Args a{x, wq, bias, y, ldx, ldy, N, D, H, W, Nc, (W + BX - 1) / BX, (H + 3) / 4, (D + TZ - 1) / TZ, Nc / NT, Kc / 16, 1, 1};
    a.by = a.nty % 4 == 0 ? 4 : (a.nty % 2 == 0 ? 2 : 1);
    a.bz = a.ntz >= 4 ? 4 : 1;
    const int nwg = N * a.ntx * a.nty * a.ntz * a.nN;
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_IGEMM, 2.0 * vox * 27 * Cin * Cout, 4.0 * (vox * (Cin + Cout) + 27.0 * Cin * Cout), st);
    if (BX == 32) { if (NBW == 2) launch_bf16<32, 2, 2>(a, nwg, st); else launch_bf16<32, 2, 1>(a, nwg, st); }
    else { if (NBW == 2) launch_bf16<16, 2, 2>(a, nwg, st); else launch_bf16<16, 2, 1>(a, nwg, st); }
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

size_t mi355seg_conv3d_bf16x6_ws_bytes(int Cin, int Cout) { return align_up((size_t)27 * Cin * Cout * 2 * 3, 256) + 256; }

// Same contract as mi355seg_conv3d_bf16mma_f32, but every fp32 operand is split into three bf16 parts and six MFMAs form each
// product: fp32-level accuracy (relative 2^-23 per product) at 2.7x the fp32 matrix rate.
int mi355seg_conv3d_bf16x6_f32(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                               int N, int D, int H, int W, int Cin, int Cout, int dgrad, void* ws, size_t ws_bytes, void* stream) {
    const int Kc = dgrad ? Cout : Cin, Nc = dgrad ? Cin : Cout;
    SEG_CHECK_ARG(x && w && y && N > 0 && D > 0 && H > 0 && W >= 8 && Kc % 16 == 0 && Nc % 32 == 0 && ldx >= Kc && ldy >= Nc && ldx % 4 == 0,
                  "conv3d_bf16x6: unsupported shape (K channels %% 16, N channels %% 32, W >= 8)");
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0, "conv3d_bf16x6: input must be 16-byte aligned");
    SEG_CHECK_WS(mi355seg_conv3d_bf16x6_ws_bytes(Cin, Cout), ws_bytes);
    hipStream_t st = (hipStream_t)stream;
    __bf16* wq = (__bf16*)ws;
    static const char* regsplit_env = getenv("MI355SEG_BF16X6_REGSPLIT");
    const bool planes = !(regsplit_env && regsplit_env[0] == '1');
    const int BX = (planes && W % 16 == 0) ? 16 : ((W % 32 == 0 || W > 48) ? 32 : 16);
    const int NBW = Nc % 64 == 0 ? 2 : 1, NT = 32 * NBW, MB = 2;
    const int TZ = (4 * MB * (32 / BX)) / 4;
    const long long total = (long long)27 * Cin * Cout;
    hipLaunchKernelGGL(pack_wq_bf16x3_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, st,
                       w, wq, Kc, Nc, NT, dgrad ? 1 : 0);
    SEG_CHECK_LAUNCH();
    Bf16x6Args a{x, wq, bias, y, ldx, ldy, N, D, H, W, Nc, (W + BX - 1) / BX, (H + 3) / 4, (D + TZ - 1) / TZ, Nc / NT, Kc / 16, 1, 1};
    a.by = a.nty % 4 == 0 ? 4 : (a.nty % 2 == 0 ? 2 : 1);
    a.bz = a.ntz >= 4 ? 4 : 1;
    const int nwg = N * a.ntx * a.nty * a.ntz * a.nN;
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_IGEMM, 2.0 * vox * 27 * Cin * Cout, 4.0 * (vox * (Cin + Cout) + 27.0 * Cin * Cout), st);
    static const char* regsplit = getenv("MI355SEG_BF16X6_REGSPLIT");        // A/B knob: 1 = split in registers per tap (fp32 tile in LDS)
    if (BX == 16 && !(regsplit && regsplit[0] == '1')) { if (NBW == 2) launch_bf16x6p<16, 2, 2>(a, nwg, st); else launch_bf16x6p<16, 2, 1>(a, nwg, st); }
    else if (BX == 32) { if (NBW == 2) launch_bf16x6<32, 2, 2>(a, nwg, st); else launch_bf16x6<32, 2, 1>(a, nwg, st); }
    else { if (NBW == 2) launch_bf16x6<16, 2, 2>(a, nwg, st); else launch_bf16x6<16, 2, 1>(a, nwg, st); }
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // extern "C"
