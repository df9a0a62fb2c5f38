// common.h -- shared helpers for libmi355seg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
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
