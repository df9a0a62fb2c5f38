// conv_generic.hip -- shape-generic direct Conv3d kernels (any cubic k / stride / pad)
// on NDHWC fp32.  These are the correctness path for every configuration and the
// production path for the shapes the MFMA implicit-GEMM kernels (conv_mfma.hip) do not
// cover (k5, k2s2, strided k3, k16s16, tiny channel counts).  Lanes of a wavefront map
// to consecutive output channels of one voxel, so activation reads are wave-broadcasts
// and packed-weight reads are fully coalesced.
#include "common.h"
#include "internal.h"
#include <string.h>
#include <stdlib.h>

namespace seg {

// w (Cout,Cin,T) -> wp[T][Cin][Cout]   (forward / wgrad-friendly)
__global__ void pack_w_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int T) {
    long long total = (long long)Cout * Cin * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int co = (int)(i % Cout);
        long long r = i / Cout;
        int ci = (int)(r % Cin);
        int t = (int)(r / Cin);
        wp[i] = w[((long long)co * Cin + ci) * T + t];
    }
}
// w (Cout,Cin,T) -> wd[T][Cout][Cin]   (dgrad: lanes over Cin); flip != 0 reverses the tap order
__global__ void pack_w_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wd, int Cout, int Cin, int T, int flip) {
    long long total = (long long)Cout * Cin * T;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int ci = (int)(i % Cin);
        long long r = i / Cin;
        int co = (int)(r % Cout);
        int t = (int)(r / Cout);
        int ts = flip ? (T - 1 - t) : t;
        wd[i] = w[((long long)co * Cin + ci) * T + ts];
    }
}

void pack_w_fwd(const float* w, float* wp, int Cout, int Cin, int T, hipStream_t st) {
    long long total = (long long)Cout * Cin * T;
    int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_w_fwd_kernel, dim3(grid), dim3(256), 0, st, w, wp, Cout, Cin, T);
}
void pack_w_dgrad(const float* w, float* wd, int Cout, int Cin, int T, int flip, hipStream_t st) {
    long long total = (long long)Cout * Cin * T;
    int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_w_dgrad_kernel, dim3(grid), dim3(256), 0, st, w, wd, Cout, Cin, T, flip);
}

struct ConvGeom {
    int N, D, H, W, Cin, Cout, k, stride, pad, Do, Ho, Wo;
};

// ---------------------------------------------------------------- forward
// one thread per (output voxel, cout); VOX_PER_THREAD voxels along W share weight loads.
__global__ __launch_bounds__(256) void conv_fwd_generic_kernel(const float* __restrict__ x, int ldx,
        const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, int ldy, ConvGeom g) {
    const long long nvox = (long long)g.N * g.Do * g.Ho * g.Wo;
    const long long total = nvox * g.Cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int co = (int)(i % g.Cout);
        long long v = i / g.Cout;
        int ow = (int)(v % g.Wo); long long r = v / g.Wo;
        int oh = (int)(r % g.Ho); r /= g.Ho;
        int od = (int)(r % g.Do); int n = (int)(r / g.Do);
        float acc = bias ? bias[co] : 0.f;
        const int id0 = od * g.stride - g.pad, ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
        for (int kd = 0; kd < g.k; ++kd) {
            int id = id0 + kd;
            if ((unsigned)id >= (unsigned)g.D) continue;
            for (int kh = 0; kh < g.k; ++kh) {
                int ih = ih0 + kh;
                if ((unsigned)ih >= (unsigned)g.H) continue;
                for (int kw = 0; kw < g.k; ++kw) {
                    int iw = iw0 + kw;
                    if ((unsigned)iw >= (unsigned)g.W) continue;
                    const float* xp = x + ((((long long)n * g.D + id) * g.H + ih) * g.W + iw) * ldx;
                    const float* wq = wp + ((long long)((kd * g.k + kh) * g.k + kw) * g.Cin) * g.Cout + co;
                    int ci = 0;
                    for (; ci + 4 <= g.Cin; ci += 4) {
                        acc = fmaf(xp[ci], wq[(long long)ci * g.Cout], acc);
                        acc = fmaf(xp[ci + 1], wq[(long long)(ci + 1) * g.Cout], acc);
                        acc = fmaf(xp[ci + 2], wq[(long long)(ci + 2) * g.Cout], acc);
                        acc = fmaf(xp[ci + 3], wq[(long long)(ci + 3) * g.Cout], acc);
                    }
                    for (; ci < g.Cin; ++ci) acc = fmaf(xp[ci], wq[(long long)ci * g.Cout], acc);
                }
            }
        }
        y[v * ldy + co] = acc;
    }
}

// ---------------------------------------------------------------- dgrad
// one thread per (input voxel, cin): dx = sum_{tap, co} dy[(i + pad - tap)/stride, co] * w[co, ci, tap]
__global__ __launch_bounds__(256) void conv_dgrad_generic_kernel(const float* __restrict__ dy, int lddy,
        const float* __restrict__ wd, float* __restrict__ dx, int lddx, ConvGeom g) {
    const long long nvox = (long long)g.N * g.D * g.H * g.W;
    const long long total = nvox * g.Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int ci = (int)(i % g.Cin);
        long long v = i / g.Cin;
        int iw = (int)(v % g.W); long long r = v / g.W;
        int ih = (int)(r % g.H); r /= g.H;
        int id = (int)(r % g.D); int n = (int)(r / g.D);
        float acc = 0.f;
        for (int kd = 0; kd < g.k; ++kd) {
            int td = id + g.pad - kd;
            if (td < 0 || td % g.stride) continue;
            int od = td / g.stride;
            if (od >= g.Do) continue;
            for (int kh = 0; kh < g.k; ++kh) {
                int th = ih + g.pad - kh;
                if (th < 0 || th % g.stride) continue;
                int oh = th / g.stride;
                if (oh >= g.Ho) continue;
                for (int kw = 0; kw < g.k; ++kw) {
                    int tw = iw + g.pad - kw;
                    if (tw < 0 || tw % g.stride) continue;
                    int ow = tw / g.stride;
                    if (ow >= g.Wo) continue;
                    const float* dp = dy + ((((long long)n * g.Do + od) * g.Ho + oh) * g.Wo + ow) * lddy;
                    const float* wq = wd + ((long long)((kd * g.k + kh) * g.k + kw) * g.Cout) * g.Cin + ci;
                    int co = 0;
                    for (; co + 4 <= g.Cout; co += 4) {
                        acc = fmaf(dp[co], wq[(long long)co * g.Cin], acc);
                        acc = fmaf(dp[co + 1], wq[(long long)(co + 1) * g.Cin], acc);
                        acc = fmaf(dp[co + 2], wq[(long long)(co + 2) * g.Cin], acc);
                        acc = fmaf(dp[co + 3], wq[(long long)(co + 3) * g.Cin], acc);
                    }
                    for (; co < g.Cout; ++co) acc = fmaf(dp[co], wq[(long long)co * g.Cin], acc);
                }
            }
        }
        dx[v * lddx + ci] = acc;
    }
}

// ---------------------------------------------------------------- wgrad
// grid = (pair blocks, taps, splits).  Thread = one (ci, co) pair; it walks the output
// voxels of its split.  partial[split][tap][ci][co].
__global__ __launch_bounds__(256) void conv_wgrad_generic_kernel(const float* __restrict__ dy, int lddy,
        const float* __restrict__ x, int ldx, float* __restrict__ part, ConvGeom g, long long vox_per_split) {
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    const int npair = g.Cin * g.Cout;
    const int tap = blockIdx.y;
    const int split = blockIdx.z;
    const int kw = tap % g.k, kh = (tap / g.k) % g.k, kd = tap / (g.k * g.k);
    const long long nvox = (long long)g.N * g.Do * g.Ho * g.Wo;
    long long v0 = (long long)split * vox_per_split;
    long long v1 = v0 + vox_per_split; if (v1 > nvox) v1 = nvox;
    if (pair >= npair) return;
    const int co = pair % g.Cout, ci = pair / g.Cout;
    float acc = 0.f;
    for (long long v = v0; v < v1; ++v) {
        int ow = (int)(v % g.Wo); long long r = v / g.Wo;
        int oh = (int)(r % g.Ho); r /= g.Ho;
        int od = (int)(r % g.Do); int n = (int)(r / g.Do);
        int id = od * g.stride - g.pad + kd, ih = oh * g.stride - g.pad + kh, iw = ow * g.stride - g.pad + kw;
        if ((unsigned)id >= (unsigned)g.D || (unsigned)ih >= (unsigned)g.H || (unsigned)iw >= (unsigned)g.W) continue;
        acc = fmaf(dy[v * lddy + co], x[((((long long)n * g.D + id) * g.H + ih) * g.W + iw) * ldx + ci], acc);
    }
    part[(((long long)split * gridDim.y + tap) * g.Cin + ci) * g.Cout + co] = acc;
}

// dw[co][ci][tap] (+)= sum_split part[split][tap][ci][co]
// A block owns 32 output channels x CIT input channels x TT taps (TT = T unless T > 256, then CIT = 1): the slabs are
// read along co (128-byte segments), summed in split order, transposed through LDS and written as CIT*TT-float runs of
// the PyTorch (Cout, Cin, T) layout.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits, int T, int Cin,
                                                           int Cout, int accumulate, int CIT, int TT) {
    extern __shared__ float tr[];                       // [32][CIT * TT + 1]
    const long long total = (long long)T * Cin * Cout;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * CIT, t0 = blockIdx.z * TT;
    const int tn = min(TT, T - t0);
    const int run = CIT * TT, pitch = run + 1;
    // four elements per thread and trip: their strip loads are independent, so four (x the compiler's unroll of k) are in flight --
    // with one workgroup per CU and one load at a time this kernel took 30-35 us whatever the layer
    for (int e0 = threadIdx.x; e0 < 32 * run; e0 += 1024) {
        long long idx[4]; int dst[4]; bool ok[4];
        double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            const int col = e & 31, r = e >> 5;
            const int cil = r % CIT, tl = r / CIT;
            const int co = co0 + col, ci = ci0 + cil;
            ok[u] = e < 32 * run && co < Cout && ci < Cin && tl < tn;
            idx[u] = ok[u] ? ((long long)(t0 + tl) * Cin + ci) * Cout + co : 0;
            dst[u] = e < 32 * run ? col * pitch + cil * TT + tl : -1;
        }
        for (int k = 0; k < splits; ++k) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = part[(long long)k * total + idx[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) if (ok[u]) s[u] += (double)v[u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (dst[u] >= 0) tr[dst[u]] = (float)s[u];
    }
    __syncthreads();
    const int cin_here = min(CIT, Cin - ci0);
    for (int e = threadIdx.x; e < 32 * run; e += 256) {
        const int col = e / run, j = e - col * run;
        const int cil = j / TT, tl = j - cil * TT;
        const int co = co0 + col;
        if (co < Cout && cil < cin_here && tl < tn) {
            const long long o = ((long long)co * Cin + ci0 + cil) * T + t0 + tl;      // contiguous in j when TT == T
            const float v = tr[col * pitch + j];
            dw[o] = accumulate ? dw[o] + v : v;
        }
    }
}

// few output elements (narrow layers, many strips): one thread per element, every thread walks the strips
__global__ void wgrad_reduce_flat_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits, int T, int Cin,
                                         int Cout, int accumulate) {
    long long total = (long long)T * Cin * Cout;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int co = (int)(i % Cout);
        long long r = i / Cout;
        int ci = (int)(r % Cin);
        int t = (int)(r / Cin);
        double s = 0.0;                                    // fp64: hundreds of strips, partial sums that largely cancel
        int k = 0;
        for (; k + 8 <= splits; k += 8) {                  // 8 loads in flight, added in strip order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(k + u) * total + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; k < splits; ++k) s += (double)part[(long long)k * total + i];
        long long o = ((long long)co * Cin + ci) * T + t;
        dw[o] = accumulate ? dw[o] + (float)s : (float)s;
    }
}

// many strips (the narrow full-resolution layers: 256 strips of 27 x 32 x 32): the slab stack as a [splits][T * Cin * Cout] matrix, a
// workgroup owns 32 consecutive columns (128-byte row pieces, same tap and input channel), its 8 row lanes walk the strips 8 apart with
// two accumulators each, LDS combines the lanes in a fixed order.  (wgrad_reduce_flat_kernel walked all strips in every thread:
// 27-36 us per layer whatever its size, 0.46 ms of a cfg-2 step.)
__global__ __launch_bounds__(256) void wgrad_reduce_cols_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits, int T, int Cin,
                                                                int Cout, int accumulate) {
    __shared__ double sh[8][32];
    const long long total = (long long)T * Cin * Cout;
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const long long col = (long long)blockIdx.x * 32 + c;
    double s0 = 0.0, s1 = 0.0;
    if (col < total) {
        const float* p = part + col;
        int k = q;
        for (; k + 8 < splits; k += 16) { s0 += (double)p[(long long)k * total]; s1 += (double)p[(long long)(k + 8) * total]; }
        if (k < splits) s0 += (double)p[(long long)k * total];
    }
    sh[q][c] = s0 + s1;
    __syncthreads();
    if (q == 0 && col < total) {
        double s = sh[0][c];
#pragma unroll
        for (int j = 1; j < 8; ++j) s += sh[j][c];
        const int co = (int)(col % Cout); const long long r = col / Cout;
        const int ci = (int)(r % Cin), t = (int)(r / Cin);
        const long long o = ((long long)co * Cin + ci) * T + t;
        dw[o] = accumulate ? dw[o] + (float)s : (float)s;
    }
}

// the slabs of a weight gradient that ran with the operands' ROLES SWAPPED (conv_wgrad_lowp: x as the centred operand, dy as the haloed one,
// so that a 64 -> 32 layer can use the 32 x 64 wide kernel): part[strip][T - 1 - tap][co][ci]  ->  dw[co][ci][tap]
__global__ __launch_bounds__(256) void wgrad_reduce_cols_swapped_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits, int T, int Cin,
                                                                        int Cout, int accumulate) {
    __shared__ double sh[8][32];
    const long long total = (long long)T * Cin * Cout;
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const long long col = (long long)blockIdx.x * 32 + c;
    double s0 = 0.0, s1 = 0.0;
    if (col < total) {
        const float* p = part + col;
        int k = q;
        for (; k + 8 < splits; k += 16) { s0 += (double)p[(long long)k * total]; s1 += (double)p[(long long)(k + 8) * total]; }
        if (k < splits) s0 += (double)p[(long long)k * total];
    }
    sh[q][c] = s0 + s1;
    __syncthreads();
    if (q == 0 && col < total) {
        double s = sh[0][c];
#pragma unroll
        for (int j = 1; j < 8; ++j) s += sh[j][c];
        const int ci = (int)(col % Cin); const long long r = col / Cin;          // the slab's columns: [tap'][co][ci]
        const int co = (int)(r % Cout), tp = (int)(r / Cout);
        const long long o = ((long long)co * Cin + ci) * T + (T - 1 - tp);
        dw[o] = accumulate ? dw[o] + (float)s : (float)s;
    }
}
void wgrad_reduce_swapped(const float* part, float* dw, int splits, int T, int Cin, int Cout, int accumulate, hipStream_t st) {
    const long long total = (long long)T * Cin * Cout;
    hipLaunchKernelGGL(wgrad_reduce_cols_swapped_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, part, dw, splits, T, Cin, Cout, accumulate);
}

void wgrad_reduce(const float* part, float* dw, int splits, int T, int Cin, int Cout, int accumulate, hipStream_t st) {
    const int TT = T > 256 ? 256 : T;
    int CIT = 256 / TT;
    CIT = CIT < 1 ? 1 : (CIT > 32 ? 32 : CIT);
    if (CIT > Cin) CIT = Cin;
    auto blocks = [&](int cit) { return (long long)((Cout + 31) / 32) * ((Cin + cit - 1) / cit) * ((T + TT - 1) / TT); };
    if (splits >= 8) {               // (measured: no gain below 8 strips, the transposing kernel's coalesced writes win there)
        const long long total = (long long)T * Cin * Cout;
        hipLaunchKernelGGL(wgrad_reduce_cols_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, part, dw, splits, T, Cin, Cout, accumulate);
        return;
    }
    while (CIT > 1 && blocks(CIT) < 1024) CIT = (CIT + 1) / 2;      // four workgroups per CU: each thread then walks a short run (the kernel is latency-bound)
    if (blocks(CIT) < 256) {
        long long total = (long long)T * Cin * Cout;
        int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
        hipLaunchKernelGGL(wgrad_reduce_flat_kernel, dim3(grid), dim3(256), 0, st, part, dw, splits, T, Cin, Cout, accumulate);
        return;
    }
    const size_t lds = (size_t)32 * (CIT * TT + 1) * sizeof(float);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((Cout + 31) / 32, (Cin + CIT - 1) / CIT, (T + TT - 1) / TT), dim3(256), lds, st, part, dw,
                       splits, T, Cin, Cout, accumulate, CIT, TT);
}

static int generic_splits(const ConvGeom& g) {
    const int T = g.k * g.k * g.k;
    const long long nvox = (long long)g.N * g.Do * g.Ho * g.Wo;
    int pairblocks = cdiv((long long)g.Cin * g.Cout, 256);
    long long want = 4096 / ((long long)pairblocks * T) + 1;
    long long maxs = nvox / 64 + 1;
    if (want > maxs) want = maxs;
    if (want > 128) want = 128;
    if (want < 1) want = 1;
    return (int)want;
}

size_t conv_generic_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    ConvGeom g{N, D, H, W, Cin, Cout, k, stride, pad, (D + 2 * pad - k) / stride + 1, (H + 2 * pad - k) / stride + 1,
               (W + 2 * pad - k) / stride + 1};
    const size_t T = (size_t)k * k * k;
    size_t wbytes = align_up(T * Cin * Cout * sizeof(float), 256);
    size_t part = align_up((size_t)generic_splits(g) * T * Cin * Cout * sizeof(float), 256);
    size_t red = colsum_ws_bytes(Cout);
    return wbytes + (part > red ? part : red) + 1024;
}

static int ew_blocks(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

int conv_fwd_generic(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, const ConvGeom& g,
                     double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st) {
    const int T = g.k * g.k * g.k;
    Carver cv(ws);
    float* wp = cv.take<float>((size_t)T * g.Cin * g.Cout);
    size_t off = cv.used();
    SEG_CHECK_WS(off + ((ssum || ssq) ? colsum_ws_bytes(g.Cout) : 0), ws_bytes);
    pack_w_fwd(w, wp, g.Cout, g.Cin, T, st);
    SEG_CHECK_LAUNCH();
    long long total = (long long)g.N * g.Do * g.Ho * g.Wo * g.Cout;
    {
        ProfScope ps(PF_GENERIC, 2.0 * total * T * g.Cin, 0.0, st);
        hipLaunchKernelGGL(conv_fwd_generic_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, x, ldx, wp, bias, y, ldy, g);
    }
    SEG_CHECK_LAUNCH();
    if (ssum || ssq)
        return channel_sums(y, ldy, (long long)g.N * g.Do * g.Ho * g.Wo, g.Cout, ssum, ssq, nullptr, 0, (char*)ws + off,
                            ws_bytes - off, st);
    return MI355SEG_OK;
}

int conv_dgrad_generic(const float* dy, int lddy, const float* w, float* dx, int lddx, const ConvGeom& g, void* ws,
                       size_t ws_bytes, hipStream_t st) {
    const int T = g.k * g.k * g.k;
    Carver cv(ws);
    float* wd = cv.take<float>((size_t)T * g.Cin * g.Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    pack_w_dgrad(w, wd, g.Cout, g.Cin, T, 0, st);
    SEG_CHECK_LAUNCH();
    long long total = (long long)g.N * g.D * g.H * g.W * g.Cin;
    {
        ProfScope ps(PF_GENERIC, 2.0 * total * T * g.Cout, 0.0, st);
        hipLaunchKernelGGL(conv_dgrad_generic_kernel, dim3(ew_blocks(total)), dim3(256), 0, st, dy, lddy, wd, dx, lddx, g);
    }
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int conv_wgrad_generic(const float* dy, int lddy, const float* x, int ldx, float* dw, const ConvGeom& g, int accumulate,
                       void* ws, size_t ws_bytes, hipStream_t st) {
    const int T = g.k * g.k * g.k;
    const int splits = generic_splits(g);
    Carver cv(ws);
    float* part = cv.take<float>((size_t)splits * T * g.Cin * g.Cout);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    const long long nvox = (long long)g.N * g.Do * g.Ho * g.Wo;
    long long vps = (nvox + splits - 1) / splits;
    dim3 grid(cdiv((long long)g.Cin * g.Cout, 256), T, splits);
    {
        ProfScope ps(PF_GENERIC, 2.0 * nvox * T * g.Cin * g.Cout, 0.0, st);
        hipLaunchKernelGGL(conv_wgrad_generic_kernel, grid, dim3(256), 0, st, dy, lddy, x, ldx, part, g, vps);
    }
    SEG_CHECK_LAUNCH();
    wgrad_reduce(part, dw, splits, T, g.Cin, g.Cout, accumulate, st);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

static int check_geom(ConvGeom* g, const char* who) {
    SEG_CHECK_ARG(g->N > 0 && g->D > 0 && g->H > 0 && g->W > 0 && g->Cin > 0 && g->Cout > 0, "%s: non-positive extent", who);
    SEG_CHECK_ARG(g->k >= 1 && g->k <= 16 && g->stride >= 1 && g->pad >= 0, "%s: bad k/stride/pad %d/%d/%d", who, g->k,
                  g->stride, g->pad);
    SEG_CHECK_ARG(g->D + 2 * g->pad >= g->k && g->H + 2 * g->pad >= g->k && g->W + 2 * g->pad >= g->k,
                  "%s: kernel larger than padded input", who);
    g->Do = (g->D + 2 * g->pad - g->k) / g->stride + 1;
    g->Ho = (g->H + 2 * g->pad - g->k) / g->stride + 1;
    g->Wo = (g->W + 2 * g->pad - g->k) / g->stride + 1;
    return MI355SEG_OK;
}

}  // namespace seg

using namespace seg;

// Arithmetic of the MFMA convolutions on fp32 tensors (mi355seg_set_conv_math; initial value from MI355SEG_CONV_MATH):
// MI355SEG_MATH_FP32 exact fp32 MFMA, MI355SEG_MATH_BF16X6 fp32-accurate split on the bf16 matrix cores.
static int conv_math_from_env() {
    const char* e = getenv("MI355SEG_CONV_MATH");
    if (!e || !e[0]) return MI355SEG_MATH_DEFAULT;
    if (!strcmp(e, "fp32") || !strcmp(e, "f32")) return MI355SEG_MATH_FP32;
    if (!strcmp(e, "bf16x6")) return MI355SEG_MATH_BF16X6;
    if (!strcmp(e, "f16x3")) return MI355SEG_MATH_F16X3;
    fprintf(stderr, "libmi355seg: MI355SEG_CONV_MATH=%s is not one of fp32 / bf16x6 / f16x3; using the default\n", e);
    return MI355SEG_MATH_DEFAULT;
}
static int g_conv_math = conv_math_from_env();
static int x3_shape_from_env() { const char* e = getenv("MI355SEG_X3_SHAPE"); return (e && !strcmp(e, "32")) ? 32 : 16; }
static int g_x3_shape = x3_shape_from_env();
namespace seg {
int x3_shape() { return g_x3_shape; }
// policy for a convolution on fp32 tensors: the split-precision kernels where the selected math asks for them
// (f16x3 shares the plans, tiles and launch paths of the bf16x6 policy; the kernels that have a two-piece form ask x3_f16())
int f32_conv_policy() { return g_conv_math == MI355SEG_MATH_FP32 ? MATH_F32 : MATH_X3; }
bool x3_f16() { return g_conv_math == MI355SEG_MATH_F16X3 && g_x3_shape == 16; }
}

// ---- patch embedding (kernel = stride, no padding; UNETR's k16 s16 conv, unetr.py:141-156) as a plain GEMM:
// tokens x (Cin k^3) patch matrix (one strided copy) times the weight matrix in its own (Cout, Cin k^3) layout.
namespace seg {
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, int ldx, float* __restrict__ A, int N, int D, int H, int W,
                                                       int C, int k) {
    const int pd = D / k, ph = H / k, pw = W / k;
    const long long K = (long long)C * k * k * k, total = (long long)N * pd * ph * pw * K;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        long long r = idx;
        const int dx = (int)(r % k); r /= k;
        const int dy = (int)(r % k); r /= k;
        const int dz = (int)(r % k); r /= k;
        const int c = (int)(r % C); r /= C;
        const int px = (int)(r % pw); r /= pw;
        const int py = (int)(r % ph); r /= ph;
        const int pz = (int)(r % pd); const int n = (int)(r / pd);
        A[idx] = x[((((long long)n * D + pz * k + dz) * H + py * k + dy) * W + px * k + dx) * ldx + c];
    }
}
static bool patch_embed_supported(int D, int H, int W, int Cin, int k, int stride, int pad) {
    return k == stride && pad == 0 && k >= 4 && D % k == 0 && H % k == 0 && W % k == 0 && (long long)Cin * k * k * k >= 256;
}
static size_t patch_embed_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k) {
    const long long M = (long long)N * (D / k) * (H / k) * (W / k), K = (long long)Cin * k * k * k;
    size_t g1 = mi355seg_gemm_ws_bytes((int)M, Cout, (int)K, 1, 1), g2 = mi355seg_gemm_ws_bytes(Cout, (int)K, (int)M, 1, 1);
    return align_up((size_t)M * K * sizeof(float), 256) + (g1 > g2 ? g1 : g2) + 512;
}
static int patch_embed_matrix(const float* x, int ldx, int N, int D, int H, int W, int Cin, int k, void* ws, size_t ws_bytes,
                              float** A, void** rest, size_t* rest_bytes, hipStream_t st) {
    const long long M = (long long)N * (D / k) * (H / k) * (W / k), K = (long long)Cin * k * k * k;
    Carver cv(ws);
    *A = cv.take<float>((size_t)M * K);
    SEG_CHECK_WS(cv.used(), ws_bytes);
    *rest = (char*)ws + cv.used();
    *rest_bytes = ws_bytes - cv.used();
    const long long blocks = (M * K + 255) / 256;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, st, x, ldx, *A, N, D, H, W, Cin, k);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
}  // namespace seg
using namespace seg;

extern "C" {

int mi355seg_set_conv_math(int mode) {
    SEG_CHECK_ARG(mode == MI355SEG_MATH_FP32 || mode == MI355SEG_MATH_BF16X6 || mode == MI355SEG_MATH_F16X3, "set_conv_math: unknown mode %d", mode);
    g_conv_math = mode;
    return MI355SEG_OK;
}
int mi355seg_get_conv_math(void) { return g_conv_math; }
int mi355seg_set_x3_shape(int shape) {
    SEG_CHECK_ARG(shape == 16 || shape == 32, "set_x3_shape: 16 or 32, got %d", shape);
    g_x3_shape = shape;
    return MI355SEG_OK;
}
int mi355seg_get_x3_shape(void) { return g_x3_shape; }

size_t mi355seg_conv3d_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    size_t a = conv_generic_ws_bytes(N, D, H, W, Cin, Cout, k, stride, pad);
    size_t b = conv_mfma_ws_bytes(N, D, H, W, Cin, Cout, k, stride, pad);
    size_t c = (stem_supported(Cin, Cout, k, stride, pad, 4) || head_supported(Cin, Cout, k, stride, pad, 4)) ? small_ws_bytes(Cin, Cout, k) : 0;
    size_t d = gwgrad_ws_bytes(N, D, H, W, Cin, Cout, k, stride, pad);
    if ((smallcin_wgrad_supported(Cin, Cout, k) || smallcout_wgrad_supported(Cin, Cout, k, 4)) && small_wgrad_ws_bytes(Cin, Cout, k) > d)
        d = small_wgrad_ws_bytes(Cin, Cout, k);
    size_t e = conv_gather_ws_bytes(N, D, H, W, Cin, Cout, k, stride, pad);
    if (wgrad_lowp_ws_bytes_geom(N, D, H, W, Cin, Cout, k, stride, pad) > e) e = wgrad_lowp_ws_bytes_geom(N, D, H, W, Cin, Cout, k, stride, pad);
    if (b > a) a = b;
    if (d > a) a = d;
    if (e > a) a = e;
    if (headk_supported(Cin, Cout, k, stride, pad, 4, 4, true) && headk_ws_bytes(Cin, Cout, k) > a) a = headk_ws_bytes(Cin, Cout, k);
    if (patch_embed_supported(D, H, W, Cin, k, stride, pad) && patch_embed_ws_bytes(N, D, H, W, Cin, Cout, k) > a) a = patch_embed_ws_bytes(N, D, H, W, Cin, Cout, k);
    if (stemk_supported(Cin, Cout, k, stride, pad, 2, 4) && headk_ws_bytes(Cout, Cin, k) > a) a = headk_ws_bytes(Cout, Cin, k);
    if (headk_wgrad_supported(Cin, Cout, k, stride, pad, 4, 2) && headk_wgrad_ws_bytes(N, D, H, W, Cin, k) > a) a = headk_wgrad_ws_bytes(N, D, H, W, Cin, k);
    if (k == 1 && stride == 1 && pad == 0 && pw_wgrad_lowp_ws_bytes((long long)N * D * H * W, Cin, Cout) > a) a = pw_wgrad_lowp_ws_bytes((long long)N * D * H * W, Cin, Cout);
    return a > c ? a : c;
}

// ---- inference forward with eval-mode BatchNorm and the activation folded in (predict.py:79-81,133: model.eval() forward):
// y = act(conv(x, w) * oscale[co] + oshift[co]); oscale goes into the packed weights, oshift takes the bias slot, the
// activation runs in the MFMA kernel's epilogue -- no normalise pass.  Only the matrix-core layers have this form; the caller
// asks first and keeps the two-pass form (convolution, then norm_act_fwd with the running statistics) for the rest.
int mi355seg_conv3d_fused_supported_f32(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy) {
    const int pol = f32_conv_policy();
    return (pol != MATH_F32 && conv_mfma_supported(pol, N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy)) ||
           conv_mfma_supported(MATH_F32, N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy);
}
int mi355seg_conv3d_fwd_fused_f32(const float* x, int ldx, const float* w, const float* oscale, const float* oshift, int act, float slope,
                                  float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                  void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g{N, D, H, W, Cin, Cout, k, stride, pad, 0, 0, 0};
    int rc = check_geom(&g, "conv3d_fwd_fused");
    if (rc) return rc;
    SEG_CHECK_ARG(x && w && y && oscale && oshift && ldx >= Cin && ldy >= Cout, "conv3d_fwd_fused: null pointer or pitch < channels");
    SEG_CHECK_ARG(mi355seg_conv3d_fused_supported_f32(N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy), "conv3d_fwd_fused: no fused form for this shape (ask mi355seg_conv3d_fused_supported_f32)");
    const int pol = f32_conv_policy();
    const int math = (pol != MATH_F32 && conv_mfma_supported(pol, N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy)) ? pol : MATH_F32;
    return conv_fwd_mfma(math, x, ldx, w, oshift, y, ldy, N, D, H, W, Cin, Cout, k, /*dgrad=*/0, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream, oscale, act, slope);
}

int mi355seg_conv3d_fwd_f32(const float* x, int ldx, const float* w, const float* bias,
                            float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                            int k, int stride, int pad, double* stats_sum, double* stats_sq,
                            void* ws, size_t ws_bytes, void* stream) {
    return mi355seg_conv3d_fwd_ax_f32(x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, stride, pad, stats_sum, stats_sq, nullptr, nullptr, ws, ws_bytes, stream);
}

// max |x| of a rows x C tensor at pitch ld, max-combined into the device scalar *amax (which the caller has zeroed, or which
// already holds the maximum of other parts of the same tensor): what the *_ax entry points take as x_amax / w_amax / dy_amax
int mi355seg_amax_f32(const float* x, int ld, long long rows, int C, float* amax, void* stream) {
    SEG_CHECK_ARG(x && amax && rows > 0 && C > 0 && ld >= C, "amax: bad arguments");
    tensor_amax(x, ld, rows, C, nullptr, amax, (hipStream_t)stream);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_conv_math_takes_amax(void) { return x3_f16() ? 1 : 0; }
// bit 0: the forward of this layer reads x_amax / w_amax, bit 1: its input gradient reads dy_amax / w_amax, bit 2: its weight
// gradient reads dy_amax / x_amax (0 under the other maths and for layers without an f16x3 form: nothing worth handing over)
int mi355seg_conv3d_amax_use_f32(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    if (!x3_f16()) return 0;
    int m = 0;
    if (conv_fwd_takes_amax(N, D, H, W, Cin, Cout, k, stride, pad)) m |= 1;
    if (conv_fwd_takes_amax(N, D, H, W, Cout, Cin, k, stride, pad)) m |= 2;
    if (wgrad_lowp_supported(MATH_X3, N, D, H, W, Cin, Cout, k, stride, pad, Cin, Cout)) m |= 4;
    return m;
}

}  // extern "C"

// y_amax (may be NULL): max |y| max-combined into a zeroed device scalar by the paths that can do it inside their kernel; *amax_done says
// whether the path that ran did
static int conv3d_fwd_impl(const float* x, int ldx, const float* w, const float* bias,
                            float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                            int k, int stride, int pad, double* stats_sum, double* stats_sq, const float* x_amax, const float* w_amax,
                            void* ws, size_t ws_bytes, void* stream, float* y_amax, bool* amax_done) {
    ConvGeom g{N, D, H, W, Cin, Cout, k, stride, pad, 0, 0, 0};
    int rc = check_geom(&g, "conv3d_fwd");
    if (rc) return rc;
    SEG_CHECK_ARG(x && w && y && ldx >= Cin && ldy >= Cout, "conv3d_fwd: null pointer or pitch < channels");
    SEG_CHECK_ARG((stats_sum == nullptr) == (stats_sq == nullptr), "conv3d_fwd: stats_sum/stats_sq must come together");
    hipStream_t st = (hipStream_t)stream;
    const int pol = f32_conv_policy();
    if (pol != MATH_F32 && conv_mfma_supported(pol, N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy)) {
        if (y_amax) *amax_done = true;
        return conv_fwd_mfma(pol, x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, /*dgrad=*/0, stats_sum, stats_sq, ws, ws_bytes, st,
                             nullptr, 0, 0.f, nullptr, x_amax, w_amax, nullptr, 0, nullptr, nullptr, y_amax);
    }
    if (conv_mfma_supported(MATH_F32, N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy))
        return conv_fwd_mfma(MATH_F32, x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, /*dgrad=*/0, stats_sum, stats_sq, ws, ws_bytes, st);
    if (patch_embed_supported(D, H, W, Cin, k, stride, pad)) {
        float* A; void* rest; size_t rest_bytes;
        rc = patch_embed_matrix(x, ldx, N, D, H, W, Cin, k, ws, ws_bytes, &A, &rest, &rest_bytes, st);
        if (rc) return rc;
        const int M = N * (D / k) * (H / k) * (W / k), K = Cin * k * k * k;
        rc = mi355seg_gemm_f32(A, K, 1, 0, 0, w, 1, K, 0, 0, y, ldy, 0, 0, bias, M, Cout, K, 1, 1, 1.f, 0, 0, rest, rest_bytes, stream);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, M, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    if (conv_gather_fwd_supported(MATH_F32, N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy) && ((uintptr_t)x % 16) == 0)
        return conv_gather_fwd_mfma(MATH_F32, x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, stride, pad, stats_sum, stats_sq, ws, ws_bytes, st);
    if (headk_supported(Cin, Cout, k, stride, pad, ldx, ldy, false) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0) {
        rc = headk_conv(false, x, ldx, w, bias, y, ldy, N, D, H, W, Cin, k, ws, ws_bytes, st);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, (long long)N * D * H * W, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    if (stemk_supported(Cin, Cout, k, stride, pad, ldx, ldy) && ((uintptr_t)x % 8) == 0 && ((uintptr_t)y % 16) == 0) {
        rc = stemk_fwd(x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, ws, ws_bytes, st);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, (long long)N * D * H * W, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    if (stem_supported(Cin, Cout, k, stride, pad, ldy))
        return stem_fwd(x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, stats_sum, stats_sq, ws, ws_bytes, st, y_amax, amax_done);
    if (tinypw_supported(Cin, Cout, k, stride, pad)) {
        rc = tinypw_fwd(x, ldx, w, bias, y, ldy, (long long)N * D * H * W, Cin, Cout, st);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, (long long)N * D * H * W, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    if (head_supported(Cin, Cout, k, stride, pad, ldx)) {
        rc = head_fwd(x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, st);
        if (rc || !stats_sum) return rc;
        return channel_sums(y, ldy, (long long)N * D * H * W, Cout, stats_sum, stats_sq, nullptr, 0, ws, ws_bytes, st);
    }
    return conv_fwd_generic(x, ldx, w, bias, y, ldy, g, stats_sum, stats_sq, ws, ws_bytes, st);
}

extern "C" {

int mi355seg_conv3d_fwd_ax_f32(const float* x, int ldx, const float* w, const float* bias,
                            float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                            int k, int stride, int pad, double* stats_sum, double* stats_sq, const float* x_amax, const float* w_amax,
                            void* ws, size_t ws_bytes, void* stream) {
    return conv3d_fwd_impl(x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, stride, pad, stats_sum, stats_sq, x_amax, w_amax, ws, ws_bytes, stream, nullptr, nullptr);
}

// ---- conv2 of a double-conv block reading conv1's RAW output (r5): the norm + activation between them is a prologue of conv2's staging
// (forward and weight gradient), so that activation is never written.  Supported where both passes run the f16x3 kernels.
int mi355seg_conv3d_pro_supported_f32(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int act) {
    if (!x3_f16() || f32_conv_policy() != MATH_X3 || k != 3 || stride != 1 || pad != 1 || Cin > 512 || !conv_pro_act_ok(act)) return 0;
    if (!conv_fwd_takes_amax(N, D, H, W, Cin, Cout, k, stride, pad)) return 0;
    return wgrad_lowp_supported(MATH_X3, N, D, H, W, Cin, Cout, k, stride, pad, Cin, Cout) ? 1 : 0;
}

// mi355seg_conv3d_fwd_ax_f32 that also hands back max |y| (max-combined into the zeroed device scalar *y_amax): from the kernel's
// epilogue where the f16x3 kernels run the whole K in one launch, by a pass over y elsewhere
int mi355seg_conv3d_fwd_yamax_ax_f32(const float* x, int ldx, const float* w, const float* bias,
                                     float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                                     int k, int stride, int pad, double* stats_sum, double* stats_sq, const float* x_amax, const float* w_amax,
                                     float* y_amax, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(y_amax, "conv3d_fwd_yamax: y_amax is null");
    bool done = false;
    int rc = conv3d_fwd_impl(x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, stride, pad, stats_sum, stats_sq, x_amax, w_amax, ws, ws_bytes, stream, y_amax, &done);
    if (rc || done) return rc;
    ConvGeom g{N, D, H, W, Cin, Cout, k, stride, pad, 0, 0, 0};
    rc = check_geom(&g, "conv3d_fwd_yamax");
    if (rc) return rc;
    tensor_amax(y, ldy, (long long)N * g.Do * g.Ho * g.Wo, Cout, nullptr, y_amax, (hipStream_t)stream);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

// y = conv3d(act(pro_al[c] * x + pro_be[c]), w) + bias: x is the pre-norm tensor of the layer in front (unet3d.py:80-101: conv2 behind
// norm1 + relu1), pro_al / pro_be its folded BatchNorm (mi355seg_norm_fold_f32), x_amax an upper bound of the prologue's OUTPUT
int mi355seg_conv3d_fwd_pro_ax_f32(const float* x, int ldx, const float* pro_al, const float* pro_be, int pro_act, float pro_slope,
                                   const float* w, const float* bias, float* y, int ldy, int N, int D, int H, int W, int Cin, int Cout,
                                   int k, int stride, int pad, double* stats_sum, double* stats_sq, const float* x_amax, const float* w_amax,
                                   void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g{N, D, H, W, Cin, Cout, k, stride, pad, 0, 0, 0};
    int rc = check_geom(&g, "conv3d_fwd_pro");
    if (rc) return rc;
    SEG_CHECK_ARG(x && w && y && pro_al && pro_be && x_amax && ldx >= Cin && ldy >= Cout, "conv3d_fwd_pro: null pointer or pitch < channels");
    SEG_CHECK_ARG((stats_sum == nullptr) == (stats_sq == nullptr), "conv3d_fwd_pro: stats_sum/stats_sq must come together");
    SEG_CHECK_ARG(mi355seg_conv3d_pro_supported_f32(N, D, H, W, Cin, Cout, k, stride, pad, pro_act) && conv_mfma_supported(MATH_X3, N, D, H, W, Cin, Cout, k, stride, pad, ldx, ldy),
                  "conv3d_fwd_pro: no prologue form for this layer under the selected conv math (ask mi355seg_conv3d_pro_supported_f32)");
    ConvPro pro{pro_al, pro_be, pro_act, pro_slope};
    return conv_fwd_mfma(MATH_X3, x, ldx, w, bias, y, ldy, N, D, H, W, Cin, Cout, k, /*dgrad=*/0, stats_sum, stats_sq, ws, ws_bytes, (hipStream_t)stream,
                         nullptr, 0, 0.f, nullptr, x_amax, w_amax, nullptr, 0, nullptr, &pro, nullptr);
}

// dw (+ db) of that convolution: the same prologue on the x operand of the weight-gradient kernels
int mi355seg_conv3d_wgrad_pro_ax_f32(const float* dy, int lddy, const float* x, int ldx, const float* pro_al, const float* pro_be, int pro_act, float pro_slope,
                                     float* dw, float* db, int N, int D, int H, int W, int Cin, int Cout,
                                     int k, int stride, int pad, int accumulate, const float* dy_amax, const float* x_amax,
                                     void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g{N, D, H, W, Cin, Cout, k, stride, pad, 0, 0, 0};
    int rc = check_geom(&g, "conv3d_wgrad_pro");
    if (rc) return rc;
    SEG_CHECK_ARG(dy && x && dw && pro_al && pro_be && x_amax && lddy >= Cout && ldx >= Cin, "conv3d_wgrad_pro: null pointer or pitch < channels");
    SEG_CHECK_ARG(mi355seg_conv3d_pro_supported_f32(N, D, H, W, Cin, Cout, k, stride, pad, pro_act) && wgrad_lowp_supported(MATH_X3, N, D, H, W, Cin, Cout, k, stride, pad, ldx, lddy) &&
                  ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0,
                  "conv3d_wgrad_pro: no prologue form for this layer under the selected conv math (ask mi355seg_conv3d_pro_supported_f32)");
    hipStream_t st = (hipStream_t)stream;
    if (db) {
        rc = channel_sums(dy, lddy, (long long)N * g.Do * g.Ho * g.Wo, Cout, nullptr, nullptr, db, accumulate, ws, ws_bytes, st);
        if (rc) return rc;
    }
    ConvPro pro{pro_al, pro_be, pro_act, pro_slope};
    return conv_wgrad_lowp(MATH_X3, dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, k, stride, accumulate, ws, ws_bytes, st, x_amax, dy_amax, &pro);
}

// The 1-channel stem behind a BatchNorm + activation (unet3d.py:80-89) when the stem's input needs no gradient: dw (+ db) of the stem
// straight from d(activation) da and the pre-norm tensor y -- the norm backward's apply half (dy = rstd gamma (dz - s1 / rows - xhat s2 /
// rows), dz = da act'(z)) is formed inside the weight-gradient kernel, dy is never written.  s1 / s2: the norm backward's column sums.
int mi355seg_stem_wgrad_bnbwd_supported_f32(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    return stem_wgrad_bn_supported(N, D, H, W, Cin, Cout, k, stride, pad) ? 1 : 0;
}
int mi355seg_stem_wgrad_bnbwd_f32(const float* da, int ldda, const float* y, int ldy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  int act, float slope, const float* s1, const float* s2, const float* x, int ldx, float* dw, float* db,
                                  int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(da && y && mean && rstd && s1 && s2 && x && dw && ldda >= Cout && ldy >= Cout && (ldda % 4) == 0 && (ldy % 4) == 0 && ldx >= 1 && act >= 0 && act <= 4,
                  "stem_wgrad_bnbwd: bad arguments");
    SEG_CHECK_ARG(stem_wgrad_bn_supported(N, D, H, W, Cin, Cout, k, stride, pad) && ((uintptr_t)da % 16) == 0 && ((uintptr_t)y % 16) == 0,
                  "stem_wgrad_bnbwd: unsupported geometry (ask mi355seg_stem_wgrad_bnbwd_supported_f32)");
    return stem_wgrad_bn(da, ldda, y, ldy, mean, rstd, gamma, beta, act, slope, s1, s2, x, ldx, dw, db, N, D, H, W, Cout, ws, ws_bytes, (hipStream_t)stream);
}

int mi355seg_conv3d_dgrad_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                              int N, int D, int H, int W, int Cin, int Cout,
                              int k, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
    return mi355seg_conv3d_dgrad_ax_f32(dy, lddy, w, dx, lddx, N, D, H, W, Cin, Cout, k, stride, pad, nullptr, nullptr, ws, ws_bytes, stream);
}
int mi355seg_conv3d_dgrad_ax_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                              int N, int D, int H, int W, int Cin, int Cout,
                              int k, int stride, int pad, const float* dy_amax, const float* w_amax, void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g{N, D, H, W, Cin, Cout, k, stride, pad, 0, 0, 0};
    int rc = check_geom(&g, "conv3d_dgrad");
    if (rc) return rc;
    SEG_CHECK_ARG(dy && w && dx && lddy >= Cout && lddx >= Cin, "conv3d_dgrad: null pointer or pitch < channels");
    hipStream_t st = (hipStream_t)stream;
    // k3 s1 p1: dgrad is the same convolution with flipped taps and Cin<->Cout swapped
    const int pol = f32_conv_policy();
    if (pol != MATH_F32 && conv_mfma_supported(pol, N, D, H, W, Cout, Cin, k, stride, pad, lddy, lddx))
        return conv_fwd_mfma(pol, dy, lddy, w, nullptr, dx, lddx, N, D, H, W, Cout, Cin, k, /*dgrad=*/1, nullptr, nullptr, ws, ws_bytes, st,
                             nullptr, 0, 0.f, nullptr, dy_amax, w_amax);
    if (conv_mfma_supported(MATH_F32, N, D, H, W, Cout, Cin, k, stride, pad, lddy, lddx))
        return conv_fwd_mfma(MATH_F32, dy, lddy, w, nullptr, dx, lddx, N, D, H, W, Cout, Cin, k, /*dgrad=*/1, nullptr, nullptr, ws, ws_bytes, st);
    if (conv_gather_dgrad_supported(MATH_F32, N, D, H, W, Cin, Cout, k, stride, pad, lddy, lddx) && ((uintptr_t)dy % 16) == 0)
        return conv_gather_dgrad_mfma(MATH_F32, dy, lddy, w, dx, lddx, N, D, H, W, Cin, Cout, k, stride, pad, ws, ws_bytes, st);
    if (headk_supported(Cin, Cout, k, stride, pad, lddy, lddx, true) && ((uintptr_t)dy % 8) == 0 && ((uintptr_t)dx % 16) == 0)
        return headk_conv(true, dy, lddy, w, nullptr, dx, lddx, N, D, H, W, Cin, k, ws, ws_bytes, st);
    if (head_supported(Cin, Cout, k, stride, pad, lddx))
        return head_dgrad(dy, lddy, w, dx, lddx, N, D, H, W, Cin, Cout, st);
    if (tinypw_supported(Cin, Cout, k, stride, pad)) return tinypw_dgrad(dy, lddy, w, dx, lddx, (long long)N * D * H * W, Cin, Cout, st);
    // k2 s2 p0 (V-Net's down-convolutions, vnet3d.py:66): the windows do not overlap, so the input gradient IS the forward of
    // ConvTranspose3d k2 s2 with the same weight tensor read as (Cin_T = Cout, Cout_T = Cin, 2, 2, 2) -- also for Cin = 16,
    // which the 32-column tiles of the gather dgrad cannot cut (the transposed conv tiles the flat (child, channel) axis)
    if (k == 2 && stride == 2 && pad == 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && ((uintptr_t)dy % 16) == 0 &&
        convt_mfma_supported(MATH_F32, N, D / 2, H / 2, W / 2, Cout, Cin, lddy, lddx))
        return convt_fwd_mfma(MATH_F32, dy, lddy, w, nullptr, dx, lddx, N, D / 2, H / 2, W / 2, Cout, Cin, ws, ws_bytes, st);
    return conv_dgrad_generic(dy, lddy, w, dx, lddx, g, ws, ws_bytes, st);
}

// Input gradient of a convolution whose INPUT was act(norm(bn_x)) of the layer in front, together with the two column sums that
// layer's norm backward starts with (s1 = sum dz, s2 = sum dz * xhat, dz = dx * act'(z); dgamma = s2, dbeta = s1): on the bf16x6
// k3 path they are reduced in the epilogue of the input-gradient kernel from the tile it has just computed (one read of bn_x instead
// of a separate pass over dx and bn_x); every other path runs the plain input gradient and mi355seg_norm_act_bwd_sums_f32.
// The layer in front then finishes with mi355seg_norm_act_bwd_apply_f32(dx, bn_x, ..., s1, s2).  groups == 1 (BatchNorm).
int mi355seg_conv3d_dgrad_bnsums_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                     int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                     const float* bn_x, int ld_bnx, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     int act, float slope, float* s1, float* s2, float* dgamma, float* dbeta,
                                     void* ws, size_t ws_bytes, void* stream) {
    return mi355seg_conv3d_dgrad_bnsums_ax_f32(dy, lddy, w, dx, lddx, N, D, H, W, Cin, Cout, k, stride, pad, bn_x, ld_bnx, mean, rstd, gamma, beta,
                                               act, slope, s1, s2, dgamma, dbeta, nullptr, nullptr, ws, ws_bytes, stream);
}
int mi355seg_conv3d_dgrad_bnsums_ax_f32(const float* dy, int lddy, const float* w, float* dx, int lddx,
                                     int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                     const float* bn_x, int ld_bnx, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     int act, float slope, float* s1, float* s2, float* dgamma, float* dbeta, const float* dy_amax, const float* w_amax,
                                     void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g{N, D, H, W, Cin, Cout, k, stride, pad, 0, 0, 0};
    int rc = check_geom(&g, "conv3d_dgrad_bnsums");
    if (rc) return rc;
    SEG_CHECK_ARG(dy && w && dx && lddy >= Cout && lddx >= Cin, "conv3d_dgrad_bnsums: null pointer or pitch < channels");
    SEG_CHECK_ARG(bn_x && mean && rstd && gamma && beta && s1 && s2 && ld_bnx >= Cin, "conv3d_dgrad_bnsums: the norm in front needs x, mean, rstd, gamma, beta");
    SEG_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "conv3d_dgrad_bnsums: dgamma/dbeta must come together");
    hipStream_t st = (hipStream_t)stream;
    const int pol = f32_conv_policy();
    if (pol == MATH_X3 && conv_mfma_supported(pol, N, D, H, W, Cout, Cin, k, stride, pad, lddy, lddx)) {
        BnBwdEpi e{bn_x, ld_bnx, mean, rstd, gamma, beta, act, slope, s1, s2, dgamma, dbeta, 0};
        rc = conv_fwd_mfma(pol, dy, lddy, w, nullptr, dx, lddx, N, D, H, W, Cout, Cin, k, /*dgrad=*/1, nullptr, nullptr, ws, ws_bytes, st, nullptr, 0, 0.f, &e,
                           dy_amax, w_amax);
        if (rc || e.done) return rc;
    } else {
        rc = mi355seg_conv3d_dgrad_ax_f32(dy, lddy, w, dx, lddx, N, D, H, W, Cin, Cout, k, stride, pad, dy_amax, w_amax, ws, ws_bytes, stream);
        if (rc) return rc;
    }
    return mi355seg_norm_act_bwd_sums_f32(dx, lddx, bn_x, ld_bnx, mean, rstd, gamma, beta, nullptr, 0, s1, s2, dgamma, dbeta,
                                          (long long)N * D * H * W, 1, Cin, act, slope, ws, ws_bytes, stream);
}

int mi355seg_conv3d_wgrad_f32(const float* dy, int lddy, const float* x, int ldx,
                              float* dw, float* db, int N, int D, int H, int W, int Cin, int Cout,
                              int k, int stride, int pad, int accumulate,
                              void* ws, size_t ws_bytes, void* stream) {
    return mi355seg_conv3d_wgrad_ax_f32(dy, lddy, x, ldx, dw, db, N, D, H, W, Cin, Cout, k, stride, pad, accumulate, nullptr, nullptr, ws, ws_bytes, stream);
}
int mi355seg_conv3d_wgrad_ax_f32(const float* dy, int lddy, const float* x, int ldx,
                              float* dw, float* db, int N, int D, int H, int W, int Cin, int Cout,
                              int k, int stride, int pad, int accumulate, const float* dy_amax, const float* x_amax,
                              void* ws, size_t ws_bytes, void* stream) {
    ConvGeom g{N, D, H, W, Cin, Cout, k, stride, pad, 0, 0, 0};
    int rc = check_geom(&g, "conv3d_wgrad");
    if (rc) return rc;
    SEG_CHECK_ARG(dy && x && dw && lddy >= Cout && ldx >= Cin, "conv3d_wgrad: null pointer or pitch < channels");
    hipStream_t st = (hipStream_t)stream;
    if (db) {
        rc = channel_sums(dy, lddy, (long long)N * g.Do * g.Ho * g.Wo, Cout, nullptr, nullptr, db, accumulate, ws, ws_bytes, st);
        if (rc) return rc;
    }
    if (patch_embed_supported(D, H, W, Cin, k, stride, pad)) {          // dW[co][(ci, tap)] = sum_tokens dy[token][co] * patch[token][(ci, tap)]
        float* A; void* rest; size_t rest_bytes;
        rc = patch_embed_matrix(x, ldx, N, D, H, W, Cin, k, ws, ws_bytes, &A, &rest, &rest_bytes, st);
        if (rc) return rc;
        const int M = N * (D / k) * (H / k) * (W / k), K = Cin * k * k * k;
        return mi355seg_gemm_f32(dy, 1, lddy, 0, 0, A, K, 1, 0, 0, dw, K, 0, 0, nullptr, Cout, K, M, 1, 1, 1.f, 0, accumulate, rest, rest_bytes, stream);
    }
    if (f32_conv_policy() == MATH_X3 && wgrad_lowp_supported(MATH_X3, N, D, H, W, Cin, Cout, k, stride, pad, ldx, lddy) &&
        ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0)
        return conv_wgrad_lowp(MATH_X3, dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, k, stride, accumulate, ws, ws_bytes, st, x_amax, dy_amax);
    if (wgrad_mfma_supported(N, D, H, W, Cin, Cout, k, stride, pad, ldx, lddy) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0)
        return conv_wgrad_mfma(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, k, accumulate, ws, ws_bytes, st);
    if (k == 1 && stride == 1 && pad == 0 && f32_conv_policy() == MATH_X3 && pw_wgrad_lowp_supported((long long)N * D * H * W, Cin, Cout, ldx, lddy, 4) &&
        ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0) {
        float* part; int nstrips;
        rc = pw_wgrad_lowp(dy, lddy, x, ldx, N, D, H, W, Cin, Cout, &part, &nstrips, ws, ws_bytes, st);
        if (rc) return rc;
        wgrad_reduce(part, dw, nstrips, 1, Cin, Cout, accumulate, st);
        SEG_CHECK_LAUNCH();
        return MI355SEG_OK;
    }
    if (k == 1 && stride == 1 && pad == 0 && pw_wgrad_supported((long long)N * D * H * W, Cin, Cout, 1, ldx, lddy)) {
        float* part; int nstrips;
        rc = pw_wgrad_mfma(dy, lddy, x, ldx, N, D, H, W, Cin, Cout, 1, &part, &nstrips, ws, ws_bytes, st);
        if (rc) return rc;
        wgrad_reduce(part, dw, nstrips, 1, Cin, Cout, accumulate, st);
        SEG_CHECK_LAUNCH();
        return MI355SEG_OK;
    }
    if (tinypw_supported(Cin, Cout, k, stride, pad))
        return tinypw_wgrad(dy, lddy, x, ldx, dw, (long long)N * D * H * W, Cin, Cout, accumulate, ws, ws_bytes, st);
    if (stem_supported(Cin, Cout, k, stride, pad, lddy))
        return stem_wgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, accumulate, ws, ws_bytes, st);
    if (head_supported(Cin, Cout, k, stride, pad, ldx))
        return head_wgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, accumulate, ws, ws_bytes, st);
    if (headk_wgrad_supported(Cin, Cout, k, stride, pad, ldx, lddy) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 8) == 0)
        return headk_wgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, k, accumulate, ws, ws_bytes, st);
    if (smallcin_wgrad_supported(Cin, Cout, k))
        return smallcin_wgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, k, stride, pad, accumulate, ws, ws_bytes, st);
    if (smallcout_wgrad_supported(Cin, Cout, k, ldx) && ((uintptr_t)x % 16) == 0)
        return smallcout_wgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, k, stride, pad, accumulate, ws, ws_bytes, st);
    if (gwgrad_supported(N, D, H, W, Cin, Cout, k, stride, pad, ldx, lddy) && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0)
        return conv_gwgrad(dy, lddy, x, ldx, dw, N, D, H, W, Cin, Cout, k, stride, pad, accumulate, ws, ws_bytes, st);
    return conv_wgrad_generic(dy, lddy, x, ldx, dw, g, accumulate, ws, ws_bytes, st);
}

}  // extern "C"
