// common.h -- shared helpers for libmi355seg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <type_traits>
#include "../../include/mi355seg.h"

namespace seg {

void set_error(const char* fmt, ...);

#define SEG_CHECK_ARG(cond, ...)                         \
    do {                                                 \
        if (!(cond)) {                                   \
            seg::set_error(__VA_ARGS__);                 \
            return MI355SEG_EINVAL;                      \
        }                                                \
    } while (0)

#define SEG_CHECK_WS(need, have)                                                        \
    do {                                                                                \
        if ((size_t)(need) > (size_t)(have)) {                                          \
            seg::set_error("workspace too small: need %zu bytes, have %zu", (size_t)(need), (size_t)(have)); \
            return MI355SEG_EWORKSPACE;                                                 \
        }                                                                               \
    } while (0)

#define SEG_CHECK_LAUNCH()                                                   \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            seg::set_error("HIP launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return MI355SEG_EHIP;                                            \
        }                                                                    \
    } while (0)

// Raise a kernel's dynamic-LDS limit (kernels that need more than 48 KB), once per DEVICE: the attribute belongs to the device's
// code object, and a process may drive several GPUs.  A failure is recorded and surfaces at the launch that follows.
inline void set_max_dynamic_lds(const void* fn, int bytes, unsigned long long& done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 64;
    if (dev < 64 && ((done_mask >> dev) & 1ull)) return;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed: %s", bytes, hipGetErrorString(e));
    else if (dev < 64) done_mask |= 1ull << dev;
}
#define SEG_SET_LDS(fn, bytes)                                                  \
    do {                                                                        \
        static unsigned long long seg_lds_done__ = 0;                           \
        seg::set_max_dynamic_lds((const void*)(fn), (int)(bytes), seg_lds_done__); \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Carve 256-byte aligned pieces out of a caller-provided workspace.
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base((char*)p) {}
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* p = (T*)(base + off);
        off += n * sizeof(T);
        return p;
    }
    size_t used() const { return align_up(off, 256); }
};

constexpr int kWave = 64;

// arithmetic policies of the matrix-core convolution kernels (igemm_kernel.h, conv_wgrad_mfma.hip)
enum { MATH_F32 = 0, MATH_X3 = 1, MATH_B16 = 2 };

// ---- storage-type helpers: every tensor is fp32 or bf16 in HBM, every kernel computes in fp32 registers.
// ld4 / st4 move four consecutive elements (16 bytes of fp32, 8 bytes of bf16; the pointer must be aligned to that).
typedef __bf16 bf16;
using f32x4_t = __attribute__((ext_vector_type(4))) float;
using f32x2_t = __attribute__((ext_vector_type(2))) float;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4_t ld4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
__device__ __forceinline__ f32x4_t ld4(const bf16* p) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
    return f32x4_t{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void st4(float* p, f32x4_t v) { *reinterpret_cast<f32x4_t*>(p) = v; }
__device__ __forceinline__ void st4(bf16* p, f32x4_t v) {
    bf16x4_t q;
    q[0] = (bf16)v[0]; q[1] = (bf16)v[1]; q[2] = (bf16)v[2]; q[3] = (bf16)v[3];
    *reinterpret_cast<bf16x4_t*>(p) = q;
}
// the same for HIP's float4 struct (.x .y .z .w), which the streaming kernels are written in.  fp32 tensors: non-temporal loads and
// stores (every elementwise / reduction pass touches each byte once and the tensors are far larger than L2 + Infinity Cache;
// measured on cfg 2: norm / pool / stem-head families -3 %; the bf16 tensors of cfg 3 lost 1 % under the same policy and keep the default)
__device__ __forceinline__ float4 ldf4(const float* p) { const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p)); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float4 ldf4(const bf16* p) {
    const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
// ld8 / st8: eight consecutive elements as floats (32 bytes of fp32, 16 of bf16; the pointer must be aligned to that)
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const f32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p)), b = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p + 4));
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void ld8(const bf16* p, float (&v)[8]) {
    const bf16x8_t q = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)q[j];
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    __builtin_nontemporal_store(f32x4_t{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4_t*>(p));
    __builtin_nontemporal_store(f32x4_t{v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4_t*>(p + 4));
}
__device__ __forceinline__ void st8(bf16* p, const float (&v)[8]) {
    bf16x8_t q;
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = (bf16)v[j];
    *reinterpret_cast<bf16x8_t*>(p) = q;
}
__device__ __forceinline__ void stf4(float* p, float4 v) { __builtin_nontemporal_store(f32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_t*>(p)); }
__device__ __forceinline__ void stf4(bf16* p, float4 v) {
    bf16x4_t q;
    q[0] = (bf16)v.x; q[1] = (bf16)v.y; q[2] = (bf16)v.z; q[3] = (bf16)v.w;
    *reinterpret_cast<bf16x4_t*>(p) = q;
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16* p, float v) { *p = (bf16)v; }
// x = h + m + l with three bf16 parts (24 mantissa bits): the operand split of the bf16x6 convolutions
__device__ __forceinline__ void split3(float v, bf16& h, bf16& m, bf16& l) {
    h = (bf16)v;
    const float r1 = v - (float)h;
    m = (bf16)r1;
    l = (bf16)(r1 - (float)m);
}

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
// v = h + l with two fp16 parts (v already scaled into the fp16 range): the operand split of the f16x3 convolutions
__device__ __forceinline__ void split2h(float v, _Float16& h, _Float16& l) {
    h = (_Float16)v;
    l = (_Float16)(v - (float)h);
}
// f16x3 split (conv_x3s.hip, conv_wgrad_lowp.hip): the exponent s of the power of two that places a tensor's largest magnitude
// in [2^14, 2^15) -- below the fp16 maximum with room for the rounding, as high as possible above the fp16 underflow.  2^s stays a
// normal float (s <= 126: tensors whose maximum is below 2^-112 simply sit lower in the fp16 range); a zero / subnormal maximum
// scales by one; an infinite or NaN maximum gives s = -114 and the non-finite values propagate as they would in fp32.
__host__ __device__ __forceinline__ int f16x_scale_exp(float amax) {
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);
    const int s = e ? 141 - e : 0;
    return s > 126 ? 126 : s;
}
__host__ __device__ __forceinline__ float pow2f(int s) { return __builtin_bit_cast(float, (unsigned)(s + 127) << 23); }   // -126 <= s <= 127

// ---- optional per-family kernel timing (api.hip) ----
enum ProfFamily { PF_IGEMM = 0, PF_WGRAD = 1, PF_GENERIC = 2, PF_CONVT = 3, PF_NORM = 4, PF_POOL = 5, PF_LOSS = 6, PF_DIRECT = 7 };
extern unsigned g_prof_mask;     // bit f set: launches of ProfFamily f are bracketed by HIP events
void prof_begin(int family, double flops, double bytes, hipStream_t st);
void prof_end(hipStream_t st);
struct ProfScope {
    hipStream_t st; bool on;
    ProfScope(int family, double flops, double bytes, hipStream_t s) : st(s), on((g_prof_mask >> family) & 1u) { if (on) prof_begin(family, flops, bytes, s); }
    ~ProfScope() { if (on) prof_end(st); }
};

// Sum across the 64 lanes of a wavefront (result valid in every lane).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// max-combine a workgroup's non-negative value into a device scalar (compare of the bit patterns: order-independent, so the result
// is reproducible).  Every thread of the workgroup must call it (block size a multiple of 64, at most 1024).
__device__ __forceinline__ void block_amax_commit(float m, unsigned* slot) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float sh_amax__[16];
    if ((threadIdx.x & 63) == 0) sh_amax__[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, sh_amax__[i]);
        atomicMax(slot, __builtin_bit_cast(unsigned, m));
    }
}

// activation value / derivative in terms of the pre-activation z
__device__ __forceinline__ float act_apply(float z, int act, float slope) {
    switch (act) {
        case MI355SEG_ACT_RELU: return z > 0.f ? z : 0.f;
        case MI355SEG_ACT_ELU: return z > 0.f ? z : expm1f(z);
        case MI355SEG_ACT_LRELU: return z > 0.f ? z : z * slope;
        case MI355SEG_ACT_SIGMOID: return 1.f / (1.f + expf(-z));
        default: return z;
    }
}
// host side: f(std::integral_constant<int, A>) with A = act for the activations that have kernel instantiations of their own (the streaming
// kernels are built per activation -- the per-element run-time switch was ~100 branches of a 1,000-instruction kernel), A = -1 (the run-time
// switch) for the rest
template <class F>
inline void act_host_dispatch(int act, F&& f) {
    switch (act) {
        case MI355SEG_ACT_NONE: f(std::integral_constant<int, MI355SEG_ACT_NONE>{}); break;
        case MI355SEG_ACT_RELU: f(std::integral_constant<int, MI355SEG_ACT_RELU>{}); break;
        case MI355SEG_ACT_ELU: f(std::integral_constant<int, MI355SEG_ACT_ELU>{}); break;
        case MI355SEG_ACT_LRELU: f(std::integral_constant<int, MI355SEG_ACT_LRELU>{}); break;
        default: f(std::integral_constant<int, -1>{}); break;
    }
}
__device__ __forceinline__ float act_grad(float z, int act, float slope) {
    switch (act) {
        case MI355SEG_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case MI355SEG_ACT_ELU: return z > 0.f ? 1.f : expf(z);
        case MI355SEG_ACT_LRELU: return z > 0.f ? 1.f : slope;
        case MI355SEG_ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-z)); return s * (1.f - s); }
        default: return 1.f;
    }
}

}  // namespace seg
