// transformer.hip -- the UNETR encoder's dense ops (unetr.py:54-168): Linear / attention matmuls as one
// strided, batched fp32-MFMA GEMM, LayerNorm and row softmax with their backward.  At 96^3 the encoder is
// 216 tokens x 768 (0.09 TFLOP per step vs 4.6 TFLOP of decoder convs), so these kernels are written for
// correctness and low launch count, not tuned: 64x64 tiles, K-step 16, operands staged in LDS with a
// layout-dependent (coalesced) global read; a wavefront per row for LayerNorm / softmax (DPP shuffles).
#include "common.h"
#include "internal.h"

namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    long long a_rs, a_cs, a_b0, a_b1, b_rs, b_cs, b_b0, b_b1, c_rs, c_b0, c_b1;
    int M, N, K, nb1;
    float alpha; int relu, accumulate;
};

constexpr int GT = 64, GK = 16, LDA_S = GK + 1, LDB_S = GT + 1;

__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
    __shared__ float As[GT * LDA_S];
    __shared__ float Bs[GK * LDB_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, i = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const int b0 = blockIdx.z / g.nb1, b1 = blockIdx.z % g.nb1;
    const float* A = g.A + b0 * g.a_b0 + b1 * g.a_b1;
    const float* B = g.B + b0 * g.b_b0 + b1 * g.b_b1;
    float* C = g.C + b0 * g.c_b0 + b1 * g.c_b1;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const bool a_kfast = g.a_cs == 1, b_nfast = g.b_cs == 1;
    for (int k0 = 0; k0 < g.K; k0 += GK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = e * 256 + tid;
            int m, k;
            if (a_kfast) { m = idx / GK; k = idx % GK; } else { k = idx / GT; m = idx % GT; }
            float v = 0.f;
            if (m0 + m < g.M && k0 + k < g.K) v = A[(long long)(m0 + m) * g.a_rs + (long long)(k0 + k) * g.a_cs];
            As[m * LDA_S + k] = v;
            int kb, n;
            if (b_nfast) { kb = idx / GT; n = idx % GT; } else { n = idx / GK; kb = idx % GK; }
            float w = 0.f;
            if (k0 + kb < g.K && n0 + n < g.N) w = B[(long long)(k0 + kb) * g.b_rs + (long long)(n0 + n) * g.b_cs];
            Bs[kb * LDB_S + n] = w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
            const float av = As[(wr * 32 + i) * LDA_S + kk * 2 + h];
            const float bv = Bs[(kk * 2 + h) * LDB_S + wc * 32 + i];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wc * 32 + i;
    if (col < g.N) {
        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = m0 + wr * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
            if (row < g.M) {
                float* dst = C + (long long)row * g.c_rs + col;
                float val = g.alpha * acc[v] + bv;
                if (g.accumulate) val += *dst;
                if (g.relu) val = val > 0.f ? val : 0.f;
                *dst = val;
            }
        }
    }
}

// ---------------------------------------------------------------- LayerNorm (one wavefront per row)
constexpr int LN_MAXJ = 32;     // E <= 64 * 32

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
        const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
        long long rows, int E, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * E;
    float v[LN_MAXJ];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; v[j] = c < E ? xr[c] : 0.f; s += v[j]; }
    const float m = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; const float d = c < E ? v[j] - m : 0.f; q += d * d; }
    const float rs = 1.f / sqrtf(wave_sum(q) / (float)E + eps);
    float* yr = y + row * E;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; if (c < E) yr[c] = (v[j] - m) * rs * gamma[c] + beta[c]; }
    if (lane == 0) { mean[row] = m; rstd[row] = rs; }
}

__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
        const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
        float* __restrict__ dx, long long rows, int E) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float m = mean[row], rs = rstd[row];
    float g[LN_MAXJ], xh[LN_MAXJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int c = lane + 64 * j;
        g[j] = c < E ? dy[row * E + c] * gamma[c] : 0.f;
        xh[j] = c < E ? (x[row * E + c] - m) * rs : 0.f;
        s1 += g[j]; s2 += g[j] * xh[j];
    }
    s1 = wave_sum(s1) / (float)E; s2 = wave_sum(s2) / (float)E;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; if (c < E) dx[row * E + c] = rs * (g[j] - s1 - xh[j] * s2); }
}

// dgamma[c] = sum_rows dy * xhat, dbeta[c] = sum_rows dy   (thread per column; rows is a few hundred)
__global__ __launch_bounds__(256) void layernorm_bwd_affine_kernel(const float* __restrict__ dy, const float* __restrict__ x,
        const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
        long long rows, int E) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= E) return;
    double a = 0.0, b = 0.0;
    for (long long r = 0; r < rows; ++r) {
        const float d = dy[r * E + c];
        a += (double)(d * (x[r * E + c] - mean[r]) * rstd[r]);
        b += (double)d;
    }
    dgamma[c] = (float)a; dbeta[c] = (float)b;
}

// ---------------------------------------------------------------- row softmax (one wavefront per row)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows, int L) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[LN_MAXJ];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; v[j] = c < L ? x[row * L + c] : -INFINITY; mx = fmaxf(mx, v[j]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; v[j] = c < L ? expf(v[j] - mx) : 0.f; s += v[j]; }
    const float inv = 1.f / wave_sum(s);
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) { const int c = lane + 64 * j; if (c < L) y[row * L + c] = v[j] * inv; }
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
        float* __restrict__ dx, long long rows, int L) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float dot = 0.f;
    for (int c = lane; c < L; c += 64) dot += y[row * L + c] * dy[row * L + c];
    dot = wave_sum(dot);
    for (int c = lane; c < L; c += 64) dx[row * L + c] = y[row * L + c] * (dy[row * L + c] - dot);
}

}  // namespace seg

using namespace seg;

extern "C" {

int mi355seg_gemm_f32(const float* A, long long a_rs, long long a_cs, long long a_b0, long long a_b1,
                      const float* B, long long b_rs, long long b_cs, long long b_b0, long long b_b1,
                      float* C, long long c_rs, long long c_b0, long long c_b1, const float* bias,
                      int M, int N, int K, int nb0, int nb1, float alpha, int relu, int accumulate, void* stream) {
    SEG_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && nb0 > 0 && nb1 > 0 && (long long)nb0 * nb1 < 65536, "gemm: bad arguments");
    GemmArgs g{A, B, C, bias, a_rs, a_cs, a_b0, a_b1, b_rs, b_cs, b_b0, b_b1, c_rs, c_b0, c_b1, M, N, K, nb1, alpha, relu, accumulate};
    dim3 grid(cdiv(N, GT), cdiv(M, GT), nb0 * nb1);
    hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                               long long rows, int E, float eps, void* stream) {
    SEG_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0 && E > 0 && E <= 64 * LN_MAXJ, "layernorm_fwd: bad arguments (E <= %d)", 64 * LN_MAXJ);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, rows, E, eps);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                               float* dx, float* dgamma, float* dbeta, long long rows, int E, void* stream) {
    SEG_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && rows > 0 && E > 0 && E <= 64 * LN_MAXJ, "layernorm_bwd: bad arguments");
    hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean, rstd, dx, rows, E);
    SEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(layernorm_bwd_affine_kernel, dim3(cdiv(E, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, dgamma, dbeta, rows, E);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_softmax_rows_f32(const float* x, float* y, long long rows, int L, void* stream) {
    SEG_CHECK_ARG(x && y && rows > 0 && L > 0 && L <= 64 * LN_MAXJ, "softmax_rows: bad arguments (L <= %d)", 64 * LN_MAXJ);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, L);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, long long rows, int L, void* stream) {
    SEG_CHECK_ARG(y && dy && dx && rows > 0 && L > 0, "softmax_rows_bwd: bad arguments");
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, rows, L);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
/* out[c] = sum_rows x[r, c]  (bias gradients of Linear layers) */
int mi355seg_colsum_f32(const float* x, int ldx, long long rows, int C, float* out, void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(x && out && rows > 0 && C > 0 && ldx >= C, "colsum: bad arguments");
    return channel_sums(x, ldx, rows, C, nullptr, nullptr, out, 0, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
