// conv_headk.hip -- Conv3d with an odd cubic kernel (k3 / k5, "same" padding, stride 1) and TWO output channels:
// V-Net's OutputTransition.conv1, k5 32 -> 2 at full resolution (vnet3d.py:107-121).  67 GFLOP per direction at
// [2, ., 128^3] but with N = 2 the MFMA tile would be 94 % padding, so forward and dgrad run on the vector ALUs.
//
// Both kernels march along z: a workgroup owns a 32 (x) x 8 (y) column segment and keeps a ring of KS input planes
// (halo included) in LDS, so every input voxel is staged once per segment (halo overhead 1.7x in x/y, KS-1 planes
// per segment in z).  A lane owns 4 adjacent x voxels of one row -- the KS x-taps of neighbouring voxels share
// their loads (KS+3 reads per 4*KS tap-voxels) -- and a WAVE owns a channel quad, which makes every weight
// wave-uniform: weights are scalar loads, never LDS traffic.
//   forward: 16 input channels per pass (4 waves x 4 channels), partial sums of the 4 waves added through LDS;
//            later passes accumulate onto y.  LDS = KS planes x 4 quad-planes, XOR-swizzled 16-byte slots.
//   dgrad:   the 2-channel dy ring is tiny; each of 8 waves produces 4 of the input channels' gradients.
// Next plane's global loads are issued before the current plane's FMAs and written to the ring slot that retires.
#include "common.h"
#include "internal.h"

namespace seg {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int HK_TX = 32, HK_TY = 8, HK_SEG = 32;      // tile in x / y, output planes per workgroup segment

struct HeadKArgs {              // pointers travel as separate __restrict__ kernel parameters: only then are the
    int ldx, ldy, N, D, H, W, Cin, ntx, nty, nseg;      // wave-uniform weight reads compiled to scalar loads
};

__device__ __forceinline__ int hk_swz(int v) { return v ^ ((v >> 3) & 7); }

// ---------------------------------------------------------------- forward: y[v][0..1] = sum_tap sum_ci x[v+tap][ci] w[co][ci][tap]
// wp[pass][quad][tap][c][co]
template <int KS>
__global__ __launch_bounds__(256, 1) void headk_fwd_kernel(const float* __restrict__ xin, const float* __restrict__ wpk,
                                                           const float* __restrict__ bias, float* __restrict__ yout, HeadKArgs a) {
    constexpr int R = KS / 2, HY = HK_TY + 2 * R, HX = HK_TX + 2 * R, SL = 40, NT = KS * KS * KS;
    constexpr int PLANE_Q = HY * SL;                               // 16-byte slots per (plane, quad)
    constexpr int ITEMS = HY * HX * 4, NIT = (ITEMS + 255) / 256;  // staged 16-byte pieces per plane
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* ring = reinterpret_cast<f32x4*>(lds);                   // [KS][4][HY][SL]
    float* red = lds + KS * 4 * PLANE_Q * 4;                       // [4 waves][64 lanes][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = channel quad of the pass
    const int xq = lane & 7, yl = lane >> 3;
    int t = blockIdx.x;
    const int txi = t % a.ntx; t /= a.ntx;
    const int tyi = t % a.nty; t /= a.nty;
    const int seg = t % a.nseg; const int n = t / a.nseg;
    const int x0 = txi * HK_TX, y0 = tyi * HK_TY, z0 = seg * HK_SEG;
    const int zend = min(a.D, z0 + HK_SEG);
    const int npass = a.Cin / 16;

    f32x4 stage[NIT];
    auto load_plane = [&](int z, int pass) {                       // global plane z (may lie outside: zeros)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid;
            const int q = idx & 3, vox = idx >> 2;
            const int hx = vox % HX, hy = vox / HX;
            const int gy = y0 - R + hy, gx = x0 - R + hx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < ITEMS && (unsigned)z < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                v = *reinterpret_cast<const f32x4*>(xin + ((((long long)n * a.D + z) * a.H + gy) * a.W + gx) * a.ldx + pass * 16 + q * 4);
            stage[it] = v;
        }
    };
    auto store_plane = [&](int slot) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 256 + tid;
            const int q = idx & 3, vox = idx >> 2;
            const int hx = vox % HX, hy = vox / HX;
            if (idx < ITEMS) ring[((slot * 4 + q) * HY + hy) * SL + hk_swz(hx)] = stage[it];
        }
    };

    for (int pass = 0; pass < npass; ++pass) {
        const float* wq = wpk + ((long long)(pass * 4 + wave) * NT) * 8;
        __syncthreads();
        for (int p = 0; p < KS; ++p) { load_plane(z0 - R + p, pass); store_plane(p); }
        __syncthreads();
        for (int zo = z0; zo < zend; ++zo) {
            const int i = zo - z0;
            load_plane(zo + R + 1, pass);                          // next plane: in flight during the FMAs below
            float acc[4][2];
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[v][0] = acc[v][1] = 0.f;
            for (int dz = 0; dz < KS; ++dz) {
                const f32x4* pl = ring + (((i + dz) % KS) * 4 + wave) * PLANE_Q;
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) {
                    const f32x4* row = pl + (yl + dy) * SL;
                    f32x4 r[KS + 3];
#pragma unroll
                    for (int j = 0; j < KS + 3; ++j) r[j] = row[hk_swz(4 * xq + j)];
                    const float* wr = wq + ((dz * KS + dy) * KS) * 8;
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float w0 = wr[dx * 8 + c * 2], w1 = wr[dx * 8 + c * 2 + 1];
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                acc[v][0] = fmaf(r[v + dx][c], w0, acc[v][0]);
                                acc[v][1] = fmaf(r[v + dx][c], w1, acc[v][1]);
                            }
                        }
                }
            }
            // add the four channel quads: wave w finishes voxel w of every lane
#pragma unroll
            for (int v = 0; v < 4; ++v) { red[(wave * 64 + lane) * 8 + v * 2] = acc[v][0]; red[(wave * 64 + lane) * 8 + v * 2 + 1] = acc[v][1]; }
            __syncthreads();
            {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { s0 += red[(w * 64 + lane) * 8 + wave * 2]; s1 += red[(w * 64 + lane) * 8 + wave * 2 + 1]; }
                const int gy = y0 + yl, gx = x0 + 4 * xq + wave;
                if (gy < a.H && gx < a.W) {
                    float* dst = yout + ((((long long)n * a.D + zo) * a.H + gy) * a.W + gx) * a.ldy;
                    if (pass == 0) { s0 += bias ? bias[0] : 0.f; s1 += bias ? bias[1] : 0.f; }
                    else { s0 += dst[0]; s1 += dst[1]; }
                    dst[0] = s0; dst[1] = s1;
                }
            }
            __syncthreads();                                       // plane i is retired, red[] is free
            store_plane(i % KS);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------- narrow -> wide: out[u][wide c] = sum_tap sum_s in[u + tap - R][s] wp[..]
// The narrow side has CS = 1 or 2 channels, the wide side a.Cin channels (a multiple of 4).  Two uses:
//   dgrad of the two-channel head:  in = dy (CS = 2), out = dx, weights flipped, no bias;
//   forward of a one/two-channel stem (V-Net in_tr, k5 1 -> 16; vnet3d.py:41-58): in = x, out = y, bias added.
// wp[quad][tap][s][c]; the narrow ring is tiny, each of 8 waves produces one quad of wide channels.
template <int KS, int CS>
__global__ __launch_bounds__(512, 1) void headk_expand_kernel(const float* __restrict__ xin, const float* __restrict__ wpk,
                                                              const float* __restrict__ bias, float* __restrict__ yout, HeadKArgs a) {
    constexpr int R = KS / 2, HY = HK_TY + 2 * R, HX = HK_TX + 2 * R, SL = 40, NT = KS * KS * KS;
    constexpr int ITEMS = HY * HX;                                 // CS-float pieces per plane
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [KS][HY][SL][CS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 8 waves = 8 wide-channel quads per group
    const int xq = lane & 7, yl = lane >> 3;
    int t = blockIdx.x;
    const int txi = t % a.ntx; t /= a.ntx;
    const int tyi = t % a.nty; t /= a.nty;
    const int seg = t % a.nseg; const int n = t / a.nseg;
    const int x0 = txi * HK_TX, y0 = tyi * HK_TY, z0 = seg * HK_SEG;
    const int zend = min(a.D, z0 + HK_SEG);
    const int nquad = a.Cin / 4;

    float stage[CS];
    auto load_plane = [&](int z) {
        const int hx = tid % HX, hy = tid / HX;
        const int gy = y0 - R + hy, gx = x0 - R + hx;
#pragma unroll
        for (int s = 0; s < CS; ++s) stage[s] = 0.f;
        if (tid < ITEMS && (unsigned)z < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W) {
            const float* src = xin + ((((long long)n * a.D + z) * a.H + gy) * a.W + gx) * a.ldx;
            if (CS == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(src); stage[0] = v.x; stage[CS - 1] = v.y; }
            else stage[0] = src[0];
        }
    };
    auto store_plane = [&](int slot) {
        if (tid < ITEMS) {
#pragma unroll
            for (int s = 0; s < CS; ++s) lds[((slot * HY + tid / HX) * SL + tid % HX) * CS + s] = stage[s];
        }
    };
    static_assert(ITEMS <= 512, "one staged piece per thread");

    for (int qg = 0; qg < nquad; qg += 8) {
        const int quad = qg + wave;
        const bool active = quad < nquad;                          // wave-uniform
        const float* wq = wpk + (long long)(active ? quad : 0) * NT * CS * 4;
        __syncthreads();
        for (int p = 0; p < KS; ++p) { load_plane(z0 - R + p); store_plane(p); }
        __syncthreads();
        for (int zo = z0; zo < zend; ++zo) {
            const int i = zo - z0;
            load_plane(zo + R + 1);
            if (active) {
                f32x4 acc[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int dz = 0; dz < KS; ++dz) {
                    const float* pl = lds + ((i + dz) % KS) * HY * SL * CS;
#pragma unroll
                    for (int dy = 0; dy < KS; ++dy) {
                        const float* row = pl + ((yl + dy) * SL + 4 * xq) * CS;
                        float r[(KS + 3) * CS];
#pragma unroll
                        for (int j = 0; j < (KS + 3) * CS / 4; ++j) {              // (KS+3)*CS is a multiple of 4 for KS = 5, 3 with CS = 2; KS = 5 with CS = 1
                            const f32x4 q = *reinterpret_cast<const f32x4*>(row + 4 * j);
                            r[4 * j] = q[0]; r[4 * j + 1] = q[1]; r[4 * j + 2] = q[2]; r[4 * j + 3] = q[3];
                        }
#pragma unroll
                        for (int j = ((KS + 3) * CS / 4) * 4; j < (KS + 3) * CS; ++j) r[j] = row[j];
                        const float* wr = wq + ((dz * KS + dy) * KS) * CS * 4;
#pragma unroll
                        for (int dx = 0; dx < KS; ++dx)
#pragma unroll
                            for (int s = 0; s < CS; ++s)
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    const float wv = wr[(dx * CS + s) * 4 + c];
#pragma unroll
                                    for (int v = 0; v < 4; ++v) acc[v][c] = fmaf(r[(v + dx) * CS + s], wv, acc[v][c]);
                                }
                    }
                }
                const int gy = y0 + yl;
                f32x4 bq = {0.f, 0.f, 0.f, 0.f};
                if (bias) bq = *reinterpret_cast<const f32x4*>(bias + quad * 4);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int gx = x0 + 4 * xq + v;
                    if (gy < a.H && gx < a.W)
                        *reinterpret_cast<f32x4*>(yout + ((((long long)n * a.D + zo) * a.H + gy) * a.W + gx) * a.ldy + quad * 4) = acc[v] + bq;
                }
            }
            __syncthreads();
            store_plane(i % KS);
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------- wgrad: dw[co][ci][tap] = sum_v dy[v][co] x[v + tap - R][ci]
// Same z-march, 16 input channels per pass, but the ring is channel-planar ([plane][hy][c][x], x fastest) so that a
// thread -- one (input channel, (dz, dy) tap row) pair, 16 x KS*KS = 400 of the 448 -- reads the KS+7 x-positions an
// 8-voxel run needs as three 16-byte LDS loads and turns them into 8*KS*2 FMAs against the run's dy values (LDS
// broadcast).  Its KS*2 sums (dx, co) live in registers for the whole segment; one slab per workgroup, summed in
// fixed order by wgrad_reduce.
template <int KS>
__global__ __launch_bounds__(448, 1) void headk_wgrad_kernel(const float* __restrict__ xin, const float* __restrict__ dyin,
                                                             float* __restrict__ part, HeadKArgs a) {
    constexpr int R = KS / 2, HY = HK_TY + 2 * R, HX = HK_TX + 2 * R, XP = 40, NT = KS * KS * KS, NTH = 448;
    constexpr int ITEMS = HY * HX * 4, NIT = (ITEMS + NTH - 1) / NTH;
    static_assert(KS + 7 <= 12, "three float4 reads cover an 8-voxel run");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ring = lds;                                             // [KS][HY][16][XP]
    float* dyl = lds + KS * HY * 16 * XP;                          // [HK_TY][HK_TX][2]
    const int tid = threadIdx.x;
    const int ci = tid & 15, trow = tid >> 4;                      // trow = dz * KS + dy (< KS*KS when active)
    const bool active = trow < KS * KS;
    const int dz = active ? trow / KS : 0, dyo = active ? trow % KS : 0;
    int t = blockIdx.x;
    const int txi = t % a.ntx; t /= a.ntx;
    const int tyi = t % a.nty; t /= a.nty;
    const int seg = t % a.nseg; const int n = t / a.nseg;
    const int x0 = txi * HK_TX, y0 = tyi * HK_TY, z0 = seg * HK_SEG;
    const int zend = min(a.D, z0 + HK_SEG);
    const int npass = a.Cin / 16;

    f32x4 stage[NIT];
    f32x2 dstage = {0.f, 0.f};
    auto load_plane = [&](int z, int pass) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * NTH + tid;
            const int q = idx & 3, vox = idx >> 2;
            const int hx = vox % HX, hy = vox / HX;
            const int gy = y0 - R + hy, gx = x0 - R + hx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < ITEMS && (unsigned)z < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                v = *reinterpret_cast<const f32x4*>(xin + ((((long long)n * a.D + z) * a.H + gy) * a.W + gx) * a.ldx + pass * 16 + q * 4);
            stage[it] = v;
        }
    };
    auto store_plane = [&](int slot) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * NTH + tid;
            const int q = idx & 3, vox = idx >> 2;
            const int hx = vox % HX, hy = vox / HX;
            if (idx < ITEMS) {
                float* dst = ring + ((slot * HY + hy) * 16 + q * 4) * XP + hx;
                dst[0] = stage[it][0]; dst[XP] = stage[it][1]; dst[2 * XP] = stage[it][2]; dst[3 * XP] = stage[it][3];
            }
        }
    };
    auto load_dy = [&](int z) {                                    // one float2 per thread: 256 voxels of the output plane
        f32x2 v = {0.f, 0.f};
        const int gy = y0 + (tid >> 5), gx = x0 + (tid & 31);
        if (tid < HK_TY * HK_TX && z < a.D && gy < a.H && gx < a.W)
            v = *reinterpret_cast<const f32x2*>(dyin + ((((long long)n * a.D + z) * a.H + gy) * a.W + gx) * a.ldy);
        dstage = v;
    };
    auto store_dy = [&]() { if (tid < HK_TY * HK_TX) *reinterpret_cast<f32x2*>(dyl + tid * 2) = dstage; };

    for (int pass = 0; pass < npass; ++pass) {
        float acc[KS][2];
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) acc[dx][0] = acc[dx][1] = 0.f;
        __syncthreads();
        for (int p = 0; p < KS; ++p) { load_plane(z0 - R + p, pass); store_plane(p); }
        load_dy(z0); store_dy();
        __syncthreads();
        for (int zo = z0; zo < zend; ++zo) {
            const int i = zo - z0;
            load_plane(zo + R + 1, pass);
            load_dy(zo + 1);
            if (active) {
                const float* pl = ring + ((((i + dz) % KS) * HY + dyo) * 16 + ci) * XP;
                for (int yy = 0; yy < HK_TY; ++yy) {
                    const float* xr = pl + yy * 16 * XP;
#pragma unroll
                    for (int run = 0; run < HK_TX / 8; ++run) {
                        float xv[12];
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const f32x4 q = *reinterpret_cast<const f32x4*>(xr + run * 8 + j * 4);
                            xv[4 * j] = q[0]; xv[4 * j + 1] = q[1]; xv[4 * j + 2] = q[2]; xv[4 * j + 3] = q[3];
                        }
                        float dv[16];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 q = *reinterpret_cast<const f32x4*>(dyl + (yy * HK_TX + run * 8) * 2 + j * 4);
                            dv[4 * j] = q[0]; dv[4 * j + 1] = q[1]; dv[4 * j + 2] = q[2]; dv[4 * j + 3] = q[3];
                        }
#pragma unroll
                        for (int v = 0; v < 8; ++v)
#pragma unroll
                            for (int dx = 0; dx < KS; ++dx) {
                                acc[dx][0] = fmaf(xv[v + dx], dv[2 * v], acc[dx][0]);
                                acc[dx][1] = fmaf(xv[v + dx], dv[2 * v + 1], acc[dx][1]);
                            }
                    }
                }
            }
            __syncthreads();
            store_plane(i % KS);
            store_dy();
            __syncthreads();
        }
        if (active) {
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
                const int tap = trow * KS + dx;
                float* dst = part + (((long long)blockIdx.x * NT + tap) * a.Cin + pass * 16 + ci) * 2;
                dst[0] = acc[dx][0]; dst[1] = acc[dx][1];
            }
        }
    }
}

// wp layouts (see the kernels); one thread per packed element
__global__ void headk_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int NT, int dgrad) {
    const long long total = (long long)Cin * NT * 2;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        if (!dgrad) {                    // [pass][quad][tap][c][co]
            const int co = (int)(r % 2); r /= 2;
            const int c = (int)(r % 4); r /= 4;
            const int tap = (int)(r % NT); r /= NT;
            const int ci = (int)r * 4 + c;                         // r = pass * 4 + quad
            wp[idx] = w[((long long)co * Cin + ci) * NT + tap];
        } else if (dgrad == 1) {         // [quad][tap'][co][c], taps flipped
            const int c = (int)(r % 4); r /= 4;
            const int co = (int)(r % 2); r /= 2;
            const int tap = (int)(r % NT); r /= NT;
            const int ci = (int)r * 4 + c;
            wp[idx] = w[((long long)co * Cin + ci) * NT + (NT - 1 - tap)];
        } else {                         // stem forward, CS = dgrad - 1 narrow input channels: [quad][tap][s][c] = W[quad*4+c][s][tap]; Cin = Cout here
            const int CS = dgrad - 1;
            if (idx >= (long long)Cin * NT * CS) return;
            const int c = (int)(r % 4); r /= 4;
            const int sidx = (int)(r % CS); r /= CS;
            const int tap = (int)(r % NT); r /= NT;
            const int co = (int)r * 4 + c;
            wp[idx] = w[((long long)co * CS + sidx) * NT + tap];
        }
    }
}

bool headk_supported(int Cin, int Cout, int k, int stride, int pad, int ldx_in, int ld_out, bool dgrad) {
    if (Cout != 2 || (k != 3 && k != 5) || stride != 1 || pad != k / 2) return false;
    if (dgrad) return Cin % 4 == 0 && Cin >= 4 && ldx_in % 2 == 0 && ld_out % 4 == 0;     // dy as float2, dx as float4
    return Cin % 16 == 0 && ldx_in % 4 == 0;
}

size_t headk_ws_bytes(int Cin, int Cout, int k) {
    return align_up((size_t)Cin * Cout * k * k * k * sizeof(float), 256) + 256;
}

template <int KS>
static void headk_launch(bool dgrad, const float* in, const float* wp, const float* bias, float* out, const HeadKArgs& a, int nwg,
                         hipStream_t st) {
    constexpr int R = KS / 2, HY = HK_TY + 2 * R;
    if (!dgrad) {
        const size_t ldsb = (size_t)KS * 4 * HY * 40 * 16 + 4 * 64 * 8 * 4;
        SEG_SET_LDS((headk_fwd_kernel<KS>), (int)ldsb);
        hipLaunchKernelGGL(headk_fwd_kernel<KS>, dim3(nwg), dim3(256), ldsb, st, in, wp, bias, out, a);
    } else {
        const size_t ldsb = (size_t)KS * HY * 40 * 8;
        hipLaunchKernelGGL((headk_expand_kernel<KS, 2>), dim3(nwg), dim3(512), ldsb, st, in, wp, (const float*)nullptr, out, a);
    }
}

// fwd:   in = x [.,Cin], out = y [.,2];   dgrad: in = dy [.,2], out = dx [.,Cin]
int headk_conv(bool dgrad, const float* in, int ld_in, const float* w, const float* bias, float* out, int ld_out,
               int N, int D, int H, int W, int Cin, int k, void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_ARG(((uintptr_t)in % 8) == 0 && ((uintptr_t)out % 8) == 0, "headk_conv: pointers must be 8-byte aligned");
    const int NT = k * k * k;
    Carver cv(ws);
    float* wp = cv.take<float>((size_t)Cin * 2 * NT);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    const long long total = (long long)Cin * NT * 2;
    hipLaunchKernelGGL(headk_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, wp, Cin, NT, dgrad ? 1 : 0);
    SEG_CHECK_LAUNCH();
    HeadKArgs a{ld_in, ld_out, N, D, H, W, Cin, (W + HK_TX - 1) / HK_TX, (H + HK_TY - 1) / HK_TY, (D + HK_SEG - 1) / HK_SEG};
    const int nwg = a.ntx * a.nty * a.nseg * N;
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_DIRECT, 2.0 * vox * NT * Cin * 2, 4.0 * vox * (Cin + 2), st);
    if (k == 5) headk_launch<5>(dgrad, in, wp, bias, out, a, nwg, st);
    else headk_launch<3>(dgrad, in, wp, bias, out, a, nwg, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// forward of a stem with Cin = 1 or 2 input channels (k5 "same"): the narrow -> wide kernel with unflipped weights
bool stemk_supported(int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    return (Cin == 1 || Cin == 2) && k == 5 && stride == 1 && pad == 2 && Cout % 4 == 0 && Cout >= 4 && ldy % 4 == 0 && (Cin == 1 || ldx % 2 == 0);
}

int stemk_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
              void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_ARG(((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 16) == 0, "stemk_fwd: x must be 8-byte and y 16-byte aligned");
    const int NT = 125;
    Carver cv(ws);
    float* wp = cv.take<float>((size_t)Cout * Cin * NT);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    const long long total = (long long)Cout * NT * 2;              // grid of the shared pack kernel; mode 2/3 trims to Cout*NT*CS
    hipLaunchKernelGGL(headk_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, wp, Cout, NT, 1 + Cin);
    SEG_CHECK_LAUNCH();
    HeadKArgs a{ldx, ldy, N, D, H, W, Cout, (W + HK_TX - 1) / HK_TX, (H + HK_TY - 1) / HK_TY, (D + HK_SEG - 1) / HK_SEG};
    const int nwg = a.ntx * a.nty * a.nseg * N;
    const double vox = (double)N * D * H * W;
    ProfScope ps(PF_DIRECT, 2.0 * vox * NT * Cin * Cout, 4.0 * vox * (Cin + Cout), st);
    const size_t ldsb = (size_t)5 * (HK_TY + 4) * 40 * 4 * Cin;
    if (Cin == 1) hipLaunchKernelGGL((headk_expand_kernel<5, 1>), dim3(nwg), dim3(512), ldsb, st, x, wp, bias, y, a);
    else hipLaunchKernelGGL((headk_expand_kernel<5, 2>), dim3(nwg), dim3(512), ldsb, st, x, wp, bias, y, a);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

bool headk_wgrad_supported(int Cin, int Cout, int k, int stride, int pad, int ldx, int lddy) {
    return Cout == 2 && (k == 3 || k == 5) && stride == 1 && pad == k / 2 && Cin % 16 == 0 && ldx % 4 == 0 && lddy % 2 == 0;
}

static int headk_nwg(int N, int D, int H, int W) {
    return ((W + HK_TX - 1) / HK_TX) * ((H + HK_TY - 1) / HK_TY) * ((D + HK_SEG - 1) / HK_SEG) * N;
}

size_t headk_wgrad_ws_bytes(int N, int D, int H, int W, int Cin, int k) {
    return align_up((size_t)headk_nwg(N, D, H, W) * k * k * k * Cin * 2 * sizeof(float), 256) + 256;
}

int headk_wgrad(const float* dy, int lddy, const float* x, int ldx, float* dw, int N, int D, int H, int W, int Cin, int k,
                int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    SEG_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 8) == 0, "headk_wgrad: x must be 16-byte and dy 8-byte aligned");
    const int NT = k * k * k, nwg = headk_nwg(N, D, H, W);
    Carver cv(ws);
    float* part = cv.take<float>((size_t)nwg * NT * Cin * 2);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    HeadKArgs a{ldx, lddy, N, D, H, W, Cin, (W + HK_TX - 1) / HK_TX, (H + HK_TY - 1) / HK_TY, (D + HK_SEG - 1) / HK_SEG};
    {
        const double vox = (double)N * D * H * W;
        ProfScope ps(PF_DIRECT, 2.0 * vox * NT * Cin * 2, 4.0 * vox * (Cin + 2), st);
        if (k == 5) {
            const size_t ldsb = ((size_t)5 * (HK_TY + 4) * 16 * 40 + HK_TY * HK_TX * 2) * 4;
            SEG_SET_LDS((headk_wgrad_kernel<5>), (int)ldsb);
            hipLaunchKernelGGL(headk_wgrad_kernel<5>, dim3(nwg), dim3(448), ldsb, st, x, dy, part, a);
        } else {
            const size_t ldsb = ((size_t)3 * (HK_TY + 2) * 16 * 40 + HK_TY * HK_TX * 2) * 4;
            SEG_SET_LDS((headk_wgrad_kernel<3>), (int)ldsb);
            hipLaunchKernelGGL(headk_wgrad_kernel<3>, dim3(nwg), dim3(448), ldsb, st, x, dy, part, a);
        }
        SEG_CHECK_LAUNCH();
    }
    wgrad_reduce(part, dw, nwg, NT, Cin, 2, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // namespace seg
