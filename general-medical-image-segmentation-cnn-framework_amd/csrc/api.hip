// api.hip -- error state, version, layout helpers of the C-ABI.
#include "common.h"
#include "internal.h"
#include <string.h>

namespace seg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-family HIP-event timing ----
unsigned g_prof_mask = 0;
namespace {
struct ProfRec { hipEvent_t a, b; int family; double flops, bytes; };
constexpr int kProfMax = 1 << 16;
ProfRec* g_recs = nullptr;
int g_nrec = 0, g_nevents = 0;
}
void prof_begin(int family, double flops, double bytes, hipStream_t st) {
    if (!g_recs) g_recs = new ProfRec[kProfMax];
    if (g_nrec >= kProfMax) return;
    if (g_nrec >= g_nevents) {
        hipEventCreate(&g_recs[g_nrec].a);
        hipEventCreate(&g_recs[g_nrec].b);
        g_nevents = g_nrec + 1;
    }
    ProfRec& r = g_recs[g_nrec];
    r.family = family; r.flops = flops; r.bytes = bytes;
    hipEventRecord(r.a, st);
}
void prof_end(hipStream_t st) {
    if (!g_recs || g_nrec >= kProfMax) return;
    hipEventRecord(g_recs[g_nrec].b, st);
    ++g_nrec;
}

// [N,C,S] -> [N,S,ld] (channel-last) through a 32x33 LDS tile so both sides stay coalesced
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void ncs_to_nsc_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int ld,
                                                          int C, long long S) {
    __shared__ float tile[32][33];
    const long long s0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const long long n = blockIdx.z;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j; long long s = s0 + tx;
        tile[j][tx] = (c < C && s < S) ? ld1(src + (n * C + c) * S + s) : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        long long s = s0 + j; int c = c0 + tx;
        if (c < C && s < S) st1(dst + (n * S + s) * ld + c, tile[tx][j]);
    }
}
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void nsc_to_ncs_kernel(const TS* __restrict__ src, int ld, TD* __restrict__ dst,
                                                          int C, long long S) {
    __shared__ float tile[32][33];
    const long long s0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const long long n = blockIdx.z;
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;
    for (int j = ty; j < 32; j += 8) {
        long long s = s0 + j; int c = c0 + tx;
        tile[j][tx] = (c < C && s < S) ? ld1(src + (n * S + s) * ld + c) : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j; long long s = s0 + tx;
        if (c < C && s < S) st1(dst + (n * C + c) * S + s, tile[tx][j]);
    }
}

// Narrow channel counts (the 2 / 4-class logits, 4-modality inputs): a 32 x 32 transpose tile would be 1/16 full.  One thread
// moves four consecutive voxels of all C channels -- 4C contiguous elements on the channel-last side, one 4-voxel vector per
// channel plane on the channel-first side.  Needs S % 4 == 0, a dense channel-last side (ld == C) and 16-byte bases.
template <typename TS, typename TD, int C, bool TO_NCS>
__global__ __launch_bounds__(256) void narrow_layout_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long S, long long groups) {
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long long)gridDim.x * 256) {
        const long long v0 = g * 4, n = v0 / S, s = v0 - n * S;
        float v[4][C];
        if constexpr (TO_NCS) {
            const TS* sp = src + v0 * C;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const f32x4_t q = ld4(sp + 4 * k);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[(4 * k + j) / C][(4 * k + j) % C] = q[j];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) st4(dst + (n * C + c) * S + s, f32x4_t{v[0][c], v[1][c], v[2][c], v[3][c]});
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const f32x4_t q = ld4(src + (n * C + c) * S + s);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j][c] = q[j];
            }
            TD* dp = dst + v0 * C;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                f32x4_t q;
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = v[(4 * k + j) / C][(4 * k + j) % C];
                st4(dp + 4 * k, q);
            }
        }
    }
}
template <typename TS, typename TD, bool TO_NCS>
static bool narrow_layout(const TS* src, TD* dst, int ld, long long N, int C, long long S, hipStream_t st) {
    if (C < 2 || C > 4 || ld != C || (S % 4) || (((uintptr_t)src | (uintptr_t)dst) % 16)) return false;
    const long long groups = N * S / 4;
    const int grid = (int)((groups + 255) / 256 < 8192 ? (groups + 255) / 256 : 8192);
    if (C == 2) hipLaunchKernelGGL((narrow_layout_kernel<TS, TD, 2, TO_NCS>), dim3(grid), dim3(256), 0, st, src, dst, S, groups);
    else if (C == 3) hipLaunchKernelGGL((narrow_layout_kernel<TS, TD, 3, TO_NCS>), dim3(grid), dim3(256), 0, st, src, dst, S, groups);
    else hipLaunchKernelGGL((narrow_layout_kernel<TS, TD, 4, TO_NCS>), dim3(grid), dim3(256), 0, st, src, dst, S, groups);
    return true;
}

template <typename T, bool ADD>
__global__ __launch_bounds__(256) void rows_kernel(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd,
                                                    long long rows, int C) {
    const bool v = (C % 4 == 0) && (lds % 4 == 0) && (ldd % 4 == 0) && (((uintptr_t)src | (uintptr_t)dst) % (4 * sizeof(T)) == 0);
    const int cw = v ? C / 4 : C;
    const long long total = rows * cw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long r = i / cw;
        int c = (int)(i % cw);
        if (v) {
            float4 a = ldf4(src + r * lds + c * 4);
            T* d = dst + r * ldd + c * 4;
            if (ADD) { float4 b = ldf4(d); a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            stf4(d, a);
        } else {
            float a = ld1(src + r * lds + c);
            if (ADD) a += ld1(dst + r * ldd + c);
            st1(dst + r * ldd + c, a);
        }
    }
}
}  // namespace seg

using namespace seg;

// y[r, j*C + c] = x[r, c] (j < rep) and its adjoint dx[r, c] = sum_j dy[r, j*C + c]; C*rep is small (V-Net: 16).
template <typename T>
__global__ __launch_bounds__(256) void repeat_ch_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                        long long rows, int C, int rep) {
    const int CR = C * rep;
    const long long total = rows * CR;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / CR;
        const int c = (int)(i - r * CR);
        y[r * ldy + c] = x[r * ldx + c % C];          // a copy: no conversion either way
    }
}
template <typename T>
__global__ __launch_bounds__(256) void repeat_ch_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx,
                                                            long long rows, int C, int rep) {
    const long long total = rows * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        float s = 0.f;
        for (int j = 0; j < rep; ++j) s += ld1(dy + r * lddy + j * C + c);
        st1(dx + r * lddx + c, s);
    }
}
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                                                  long long n) {
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(o)[i] = make_float4(u.x * v.x, u.y * v.y, u.z * v.z, u.w * v.w);
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) o[i] = a[i] * b[i];
}

// ---- selective-fusion pieces (ER_net.py:36-70): per-(sample, channel) statistics and mixing of two feature maps
__global__ __launch_bounds__(256) void mix_channels_kernel(const float* __restrict__ x1, int ld1, const float* __restrict__ a,
        const float* __restrict__ x2, int ld2, const float* __restrict__ b, float* __restrict__ y, int ldy, long long rows, int groups, int C) {
    const long long total = (long long)groups * rows * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C;                       // global row (group-major)
        const int c = (int)(i - r * C);
        const int g = (int)(r / rows);
        y[r * ldy + c] = x1[r * ld1 + c] * a[g * C + c] + x2[r * ld2 + c] * b[g * C + c];
    }
}
__global__ __launch_bounds__(256) void broadcast_channels_kernel(const float* __restrict__ v, float alpha, float* __restrict__ y, int ldy,
                                                                 long long rows, int groups, int C) {
    const long long total = (long long)groups * rows * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        y[r * ldy + c] = alpha * v[(r / rows) * C + c];
    }
}
__global__ void group_sums_finish_kernel(const double* __restrict__ s, float alpha, float* __restrict__ out, int C, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) out[c] = (accumulate ? out[c] : 0.f) + (float)((double)alpha * s[c]);
}

// reverse-attention gate: thread = (row, channel quad)
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ enc, int ldenc, const float* __restrict__ t, int ldt,
                                                       float* __restrict__ y, int ldy, long long rows, int C) {
    const int cw = C / 4;
    const long long total = rows * cw;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cw;
        const int c = (int)(i - r * cw) * 4;
        const float g = 2.f - 1.f / (1.f + expf(-t[r * ldt]));
        const float4 e = *reinterpret_cast<const float4*>(enc + r * ldenc + c);
        *reinterpret_cast<float4*>(y + r * ldy + c) = make_float4(e.x * g, e.y * g, e.z * g, e.w * g);
    }
}
// one wavefront per 64 / LPV rows; LPV = lanes per row (power of two, <= 64), each lane walks its quads
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ enc, int ldenc,
                                                       const float* __restrict__ t, int ldt, float* __restrict__ denc, int lddenc,
                                                       float* __restrict__ dt, long long rows, int C, int LPV) {
    const int lane = threadIdx.x & 63, sub = lane % LPV, rpw = 64 / LPV;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long r = wave * rpw + lane / LPV;
    const bool ok = r < rows;
    float s = 0.f, dot = 0.f;
    if (ok) {
        s = 1.f / (1.f + expf(-t[r * ldt]));
        const float g = 2.f - s;
        for (int c = sub * 4; c < C; c += LPV * 4) {
            const float4 d = *reinterpret_cast<const float4*>(dy + r * lddy + c);
            const float4 e = *reinterpret_cast<const float4*>(enc + r * ldenc + c);
            dot += d.x * e.x + d.y * e.y + d.z * e.z + d.w * e.w;
            *reinterpret_cast<float4*>(denc + r * lddenc + c) = make_float4(d.x * g, d.y * g, d.z * g, d.w * g);
        }
    }
    for (int o = LPV >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (ok && sub == 0) dt[r] = -s * (1.f - s) * dot;
}

template <typename T>
__global__ __launch_bounds__(256) void add_bias_kernel(T* __restrict__ y, int ldy, const float* __restrict__ bias, long long rows, int C) {
    const long long total = rows * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        st1(y + r * ldy + c, ld1(y + r * ldy + c) + bias[c]);
    }
}

// ---- sliding-window inference (predict.py:98-147): grid patches of one volume as a device-resident batch, and the crop-mode
// aggregation of their label maps.  table[p] = {origin z, y, x, crop lo z, y, x, crop hi z, y, x} (crop window inside the patch,
// hi exclusive): torchio GridSampler locations and GridAggregator(overlap_mode='crop') windows, built once per volume on the host.
__global__ __launch_bounds__(256) void gather_patches_kernel(const float* __restrict__ vol, int C, int D, int H, int W, const int* __restrict__ table,
        int first, int count, int pd, int ph, int pw, float* __restrict__ out) {
    const long long per = (long long)C * pd * ph * pw, total = per * count;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        long long r = i - (long long)b * per;
        const int x = (int)(r % pw); r /= pw;
        const int y = (int)(r % ph); r /= ph;
        const int z = (int)(r % pd);
        const int c = (int)(r / pd);
        const int* t = table + (long long)(first + b) * 9;
        out[i] = vol[(((long long)c * D + t[0] + z) * H + t[1] + y) * W + t[2] + x];
    }
}
__global__ __launch_bounds__(256) void paste_labels_kernel(const long long* __restrict__ labels, const int* __restrict__ table, int first, int count,
        int pd, int ph, int pw, long long* __restrict__ out, int D, int H, int W) {
    const long long per = (long long)pd * ph * pw, total = per * count;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        long long r = i - (long long)b * per;
        const int x = (int)(r % pw); r /= pw;
        const int y = (int)(r % ph);
        const int z = (int)(r / ph);
        const int* t = table + (long long)(first + b) * 9;
        if (z >= t[3] && z < t[6] && y >= t[4] && y < t[7] && x >= t[5] && x < t[8])
            out[((long long)(t[0] + z) * H + t[1] + y) * W + t[2] + x] = labels[i];
    }
}

extern "C" {

const char* mi355seg_last_error(void) { return g_err; }
int mi355seg_version(void) { return 100; }

int mi355seg_prof_enable(int on) { g_prof_mask = on == 1 ? 0xFFu : (on <= 0 ? 0u : ((unsigned)on >> 1)); return MI355SEG_OK; }
int mi355seg_prof_reset(void) { g_nrec = 0; return MI355SEG_OK; }
int mi355seg_prof_records(double* out, int max_records, int* n_host) {
    SEG_CHECK_ARG(out && n_host && max_records >= 0, "prof_records: bad arguments");
    int n = 0;
    for (int i = 0; i < g_nrec && n < max_records; ++i) {
        float ms = 0.f;
        (void)hipEventSynchronize(g_recs[i].b);
        if (hipEventElapsedTime(&ms, g_recs[i].a, g_recs[i].b) != hipSuccess) continue;
        out[4 * n] = g_recs[i].family; out[4 * n + 1] = ms; out[4 * n + 2] = g_recs[i].flops; out[4 * n + 3] = g_recs[i].bytes;
        ++n;
    }
    *n_host = n;
    return MI355SEG_OK;
}
int mi355seg_prof_read(double* out, int n) {
    SEG_CHECK_ARG(out && n >= 4 * MI355SEG_PROF_FAMILIES, "prof_read: need room for %d doubles", 4 * MI355SEG_PROF_FAMILIES);
    for (int i = 0; i < 4 * MI355SEG_PROF_FAMILIES; ++i) out[i] = 0.0;
    for (int i = 0; i < g_nrec; ++i) {
        float ms = 0.f;
        hipEventSynchronize(g_recs[i].b);
        if (hipEventElapsedTime(&ms, g_recs[i].a, g_recs[i].b) != hipSuccess) continue;
        int f = g_recs[i].family;
        out[4 * f] += 1.0; out[4 * f + 1] += (double)ms; out[4 * f + 2] += g_recs[i].flops; out[4 * f + 3] += g_recs[i].bytes;
    }
    return MI355SEG_OK;
}

int mi355seg_ncdhw_to_ndhwc_f32(const float* src, float* dst, int lddst, long long N, int C, long long S, void* stream) {
    SEG_CHECK_ARG(src && dst && N > 0 && C > 0 && S > 0 && lddst >= C && N < 65536, "ncdhw_to_ndhwc: bad arguments");
    ProfScope ps(PF_POOL, 0.0, 8.0 * N * C * S, (hipStream_t)stream);
    if (narrow_layout<float, float, false>(src, dst, lddst, N, C, S, (hipStream_t)stream)) { SEG_CHECK_LAUNCH(); return MI355SEG_OK; }
    dim3 grid((unsigned)((S + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    hipLaunchKernelGGL((ncs_to_nsc_kernel<float, float>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, lddst, C, S);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_ncdhw_f32_to_ndhwc_bf16(const float* src, mi355seg_bf16* dst, int lddst, long long N, int C, long long S, void* stream) {
    SEG_CHECK_ARG(src && dst && N > 0 && C > 0 && S > 0 && lddst >= C && N < 65536, "ncdhw_f32_to_ndhwc_bf16: bad arguments");
    ProfScope ps(PF_POOL, 0.0, 6.0 * N * C * S, (hipStream_t)stream);
    if (narrow_layout<float, bf16, false>(src, dst, lddst, N, C, S, (hipStream_t)stream)) { SEG_CHECK_LAUNCH(); return MI355SEG_OK; }
    dim3 grid((unsigned)((S + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    hipLaunchKernelGGL((ncs_to_nsc_kernel<float, bf16>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, lddst, C, S);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_ndhwc_to_ncdhw_f32(const float* src, int ldsrc, float* dst, long long N, int C, long long S, void* stream) {
    SEG_CHECK_ARG(src && dst && N > 0 && C > 0 && S > 0 && ldsrc >= C && N < 65536, "ndhwc_to_ncdhw: bad arguments");
    ProfScope ps(PF_POOL, 0.0, 8.0 * N * C * S, (hipStream_t)stream);
    if (narrow_layout<float, float, true>(src, dst, ldsrc, N, C, S, (hipStream_t)stream)) { SEG_CHECK_LAUNCH(); return MI355SEG_OK; }
    dim3 grid((unsigned)((S + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    hipLaunchKernelGGL((nsc_to_ncs_kernel<float, float>), grid, dim3(256), 0, (hipStream_t)stream, src, ldsrc, dst, C, S);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_ndhwc_bf16_to_ncdhw_f32(const mi355seg_bf16* src, int ldsrc, float* dst, long long N, int C, long long S, void* stream) {
    SEG_CHECK_ARG(src && dst && N > 0 && C > 0 && S > 0 && ldsrc >= C && N < 65536, "ndhwc_bf16_to_ncdhw_f32: bad arguments");
    ProfScope ps(PF_POOL, 0.0, 6.0 * N * C * S, (hipStream_t)stream);
    if (narrow_layout<bf16, float, true>(src, dst, ldsrc, N, C, S, (hipStream_t)stream)) { SEG_CHECK_LAUNCH(); return MI355SEG_OK; }
    dim3 grid((unsigned)((S + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    hipLaunchKernelGGL((nsc_to_ncs_kernel<bf16, float>), grid, dim3(256), 0, (hipStream_t)stream, src, ldsrc, dst, C, S);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
static int rows_grid(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
int mi355seg_gather_patches_f32(const float* vol, int C, int D, int H, int W, const int* table, int first, int count,
                                int pd, int ph, int pw, float* out, void* stream) {
    SEG_CHECK_ARG(vol && table && out && C > 0 && count > 0 && first >= 0 && pd > 0 && ph > 0 && pw > 0 && pd <= D && ph <= H && pw <= W,
                  "gather_patches: bad arguments (patch %dx%dx%d in volume %dx%dx%d)", pd, ph, pw, D, H, W);
    const long long total = (long long)count * C * pd * ph * pw;
    ProfScope ps(PF_POOL, 0.0, 8.0 * total, (hipStream_t)stream);
    hipLaunchKernelGGL(gather_patches_kernel, dim3(rows_grid(total)), dim3(256), 0, (hipStream_t)stream, vol, C, D, H, W, table, first, count, pd, ph, pw, out);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_paste_labels_i64(const int64_t* labels, const int* table, int first, int count, int pd, int ph, int pw,
                              int64_t* out, int D, int H, int W, void* stream) {
    SEG_CHECK_ARG(labels && table && out && count > 0 && first >= 0 && pd > 0 && ph > 0 && pw > 0 && pd <= D && ph <= H && pw <= W,
                  "paste_labels: bad arguments");
    const long long total = (long long)count * pd * ph * pw;
    ProfScope ps(PF_POOL, 0.0, 16.0 * total, (hipStream_t)stream);
    hipLaunchKernelGGL(paste_labels_kernel, dim3(rows_grid(total)), dim3(256), 0, (hipStream_t)stream, (const long long*)labels, table, first, count,
                       pd, ph, pw, (long long*)out, D, H, W);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_copy_rows_f32(const float* src, int ldsrc, float* dst, int lddst, long long rows, int C, void* stream) {
    SEG_CHECK_ARG(src && dst && rows > 0 && C > 0 && ldsrc >= C && lddst >= C, "copy_rows: bad arguments");
    ProfScope ps(PF_POOL, 0.0, 8.0 * rows * C, (hipStream_t)stream);
    hipLaunchKernelGGL((rows_kernel<float, false>), dim3(rows_grid(rows * C / 4 + 1)), dim3(256), 0, (hipStream_t)stream, src, ldsrc,
                       dst, lddst, rows, C);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_copy_rows_bf16(const mi355seg_bf16* src, int ldsrc, mi355seg_bf16* dst, int lddst, long long rows, int C, void* stream) {
    SEG_CHECK_ARG(src && dst && rows > 0 && C > 0 && ldsrc >= C && lddst >= C, "copy_rows: bad arguments");
    ProfScope ps(PF_POOL, 0.0, 4.0 * rows * C, (hipStream_t)stream);
    hipLaunchKernelGGL((rows_kernel<bf16, false>), dim3(rows_grid(rows * C / 4 + 1)), dim3(256), 0, (hipStream_t)stream, src, ldsrc,
                       dst, lddst, rows, C);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_add_rows_f32(const float* src, int ldsrc, float* dst, int lddst, long long rows, int C, void* stream) {
    SEG_CHECK_ARG(src && dst && rows > 0 && C > 0 && ldsrc >= C && lddst >= C, "add_rows: bad arguments");
    ProfScope ps(PF_POOL, 0.0, 12.0 * rows * C, (hipStream_t)stream);
    hipLaunchKernelGGL((rows_kernel<float, true>), dim3(rows_grid(rows * C / 4 + 1)), dim3(256), 0, (hipStream_t)stream, src, ldsrc,
                       dst, lddst, rows, C);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_add_rows_bf16(const mi355seg_bf16* src, int ldsrc, mi355seg_bf16* dst, int lddst, long long rows, int C, void* stream) {
    SEG_CHECK_ARG(src && dst && rows > 0 && C > 0 && ldsrc >= C && lddst >= C, "add_rows: bad arguments");
    ProfScope ps(PF_POOL, 0.0, 6.0 * rows * C, (hipStream_t)stream);
    hipLaunchKernelGGL((rows_kernel<bf16, true>), dim3(rows_grid(rows * C / 4 + 1)), dim3(256), 0, (hipStream_t)stream, src, ldsrc,
                       dst, lddst, rows, C);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

int mi355seg_repeat_channels_f32(const float* x, int ldx, float* y, int ldy, long long rows, int C, int rep, void* stream) {
    SEG_CHECK_ARG(x && y && rows > 0 && C > 0 && rep > 0 && ldx >= C && ldy >= C * rep, "repeat_channels: bad arguments");
    hipLaunchKernelGGL(repeat_ch_kernel<float>, dim3(rows_grid(rows * C * rep)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, C, rep);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_repeat_channels_bf16(const mi355seg_bf16* x, int ldx, mi355seg_bf16* y, int ldy, long long rows, int C, int rep, void* stream) {
    SEG_CHECK_ARG(x && y && rows > 0 && C > 0 && rep > 0 && ldx >= C && ldy >= C * rep, "repeat_channels: bad arguments");
    hipLaunchKernelGGL(repeat_ch_kernel<bf16>, dim3(rows_grid(rows * C * rep)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, C, rep);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_repeat_channels_bwd_f32(const float* dy, int lddy, float* dx, int lddx, long long rows, int C, int rep, void* stream) {
    SEG_CHECK_ARG(dy && dx && rows > 0 && C > 0 && rep > 0 && lddx >= C && lddy >= C * rep, "repeat_channels_bwd: bad arguments");
    hipLaunchKernelGGL(repeat_ch_bwd_kernel<float>, dim3(rows_grid(rows * C)), dim3(256), 0, (hipStream_t)stream, dy, lddy, dx, lddx, rows, C, rep);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_repeat_channels_bwd_bf16(const mi355seg_bf16* dy, int lddy, mi355seg_bf16* dx, int lddx, long long rows, int C, int rep, void* stream) {
    SEG_CHECK_ARG(dy && dx && rows > 0 && C > 0 && rep > 0 && lddx >= C && lddy >= C * rep, "repeat_channels_bwd: bad arguments");
    hipLaunchKernelGGL(repeat_ch_bwd_kernel<bf16>, dim3(rows_grid(rows * C)), dim3(256), 0, (hipStream_t)stream, dy, lddy, dx, lddx, rows, C, rep);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_mix_channels_f32(const float* x1, int ld1, const float* a, const float* x2, int ld2, const float* b, float* y, int ldy,
                              long long rows, int groups, int C, void* stream) {
    SEG_CHECK_ARG(x1 && a && x2 && b && y && rows > 0 && groups > 0 && C > 0 && ld1 >= C && ld2 >= C && ldy >= C, "mix_channels: bad arguments");
    hipLaunchKernelGGL(mix_channels_kernel, dim3(rows_grid(groups * rows * C)), dim3(256), 0, (hipStream_t)stream, x1, ld1, a, x2, ld2, b, y, ldy,
                       rows, groups, C);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_broadcast_channels_f32(const float* v, float alpha, float* y, int ldy, long long rows, int groups, int C, void* stream) {
    SEG_CHECK_ARG(v && y && rows > 0 && groups > 0 && C > 0 && ldy >= C, "broadcast_channels: bad arguments");
    hipLaunchKernelGGL(broadcast_channels_kernel, dim3(rows_grid(groups * rows * C)), dim3(256), 0, (hipStream_t)stream, v, alpha, y, ldy, rows,
                       groups, C);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_group_sums_f32(const float* x, int ldx, long long rows, int groups, int C, float alpha, float* out, int accumulate,
                            void* ws, size_t ws_bytes, void* stream) {
    SEG_CHECK_ARG(x && out && rows > 0 && groups > 0 && C > 0 && ldx >= C && ws, "group_sums: bad arguments");
    SEG_CHECK_WS(align_up((size_t)C * sizeof(double), 256) + colsum_ws_bytes(C), ws_bytes);
    double* ds = (double*)ws;
    char* rest = (char*)ws + align_up((size_t)C * sizeof(double), 256);
    for (int g = 0; g < groups; ++g) {
        int rc = channel_sums(x + (long long)g * rows * ldx, ldx, rows, C, ds, nullptr, nullptr, 0, rest, ws_bytes - (rest - (char*)ws), (hipStream_t)stream);
        if (rc) return rc;
        hipLaunchKernelGGL(group_sums_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, ds, alpha, out + (long long)g * C, C, accumulate);
        SEG_CHECK_LAUNCH();
    }
    return MI355SEG_OK;
}
int mi355seg_gate_fwd_f32(const float* enc, int ldenc, const float* t, int ldt, float* y, int ldy, long long rows, int C, void* stream) {
    SEG_CHECK_ARG(enc && t && y && rows > 0 && C > 0 && C % 4 == 0 && ldenc >= C && ldy >= C && ldt >= 1 && ldenc % 4 == 0 && ldy % 4 == 0,
                  "gate_fwd: bad arguments (C and the pitches must be multiples of 4)");
    SEG_CHECK_ARG(((uintptr_t)enc | (uintptr_t)y) % 16 == 0, "gate_fwd: enc / y must be 16-byte aligned");
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(rows_grid(rows * C / 4)), dim3(256), 0, (hipStream_t)stream, enc, ldenc, t, ldt, y, ldy, rows, C);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_gate_bwd_f32(const float* dy, int lddy, const float* enc, int ldenc, const float* t, int ldt,
                          float* denc, int lddenc, float* dt, long long rows, int C, void* stream) {
    SEG_CHECK_ARG(dy && enc && t && denc && dt && rows > 0 && C > 0 && C % 4 == 0 && lddy >= C && ldenc >= C && lddenc >= C && ldt >= 1 &&
                  lddy % 4 == 0 && ldenc % 4 == 0 && lddenc % 4 == 0, "gate_bwd: bad arguments (C and the pitches must be multiples of 4)");
    SEG_CHECK_ARG(((uintptr_t)dy | (uintptr_t)enc | (uintptr_t)denc) % 16 == 0, "gate_bwd: dy / enc / denc must be 16-byte aligned");
    int LPV = 1;
    while (LPV * 2 <= C / 4 && LPV * 2 <= 64) LPV *= 2;
    const long long waves = (rows + 64 / LPV - 1) / (64 / LPV);
    hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, lddy, enc, ldenc, t, ldt,
                       denc, lddenc, dt, rows, C, LPV);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_add_bias_f32(float* y, int ldy, const float* bias, long long rows, int C, void* stream) {
    SEG_CHECK_ARG(y && bias && rows > 0 && C > 0 && ldy >= C, "add_bias: bad arguments");
    hipLaunchKernelGGL(add_bias_kernel<float>, dim3(rows_grid(rows * C)), dim3(256), 0, (hipStream_t)stream, y, ldy, bias, rows, C);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_add_bias_bf16(mi355seg_bf16* y, int ldy, const float* bias, long long rows, int C, void* stream) {
    SEG_CHECK_ARG(y && bias && rows > 0 && C > 0 && ldy >= C, "add_bias: bad arguments");
    hipLaunchKernelGGL(add_bias_kernel<bf16>, dim3(rows_grid(rows * C)), dim3(256), 0, (hipStream_t)stream, y, ldy, bias, rows, C);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}
int mi355seg_mul_f32(const float* a, const float* b, float* out, long long n, void* stream) {
    SEG_CHECK_ARG(a && b && out && n > 0, "mul: bad arguments");
    SEG_CHECK_ARG(((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) % 16 == 0, "mul: operands must be 16-byte aligned");
    hipLaunchKernelGGL(mul_kernel, dim3(rows_grid(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
    SEG_CHECK_LAUNCH();
    return MI355SEG_OK;
}

}  // extern "C"
