// wg_turnaround.hip -- what a launch of many SHORT workgroups costs on MI355X as a function of their resources (r6 probe behind the
// conv_b16s / conv_x3s tile prologue question): grid of G workgroups of 256 threads that do (almost) nothing, with V live VGPRs and L bytes
// of dynamic LDS; prints microseconds per launch and per "round" of two workgroups per CU.   hipcc --offload-arch=gfx950 -O3 wg_turnaround.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int V>
__global__ __launch_bounds__(256, 2) void busy_regs(float* out, int n) {
    extern __shared__ float lds[];
    float r[V];
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = (float)(threadIdx.x + i);
    // keep all V registers live across a point the compiler cannot remove
#pragma unroll
    for (int i = 0; i < V; ++i) asm volatile("" : "+v"(r[i]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) s += r[i];
    if (n == -1) { lds[threadIdx.x] = s; out[blockIdx.x * 256 + threadIdx.x] = lds[255 - threadIdx.x]; }     // never true: no memory traffic
}

template <int V>
void run(int G, size_t lds, float* out) {
    hipFuncSetAttribute((const void*)busy_regs<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(busy_regs<V>, dim3(G), dim3(256), lds, 0, out, 0);
    hipEventRecord(e0, 0);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(busy_regs<V>, dim3(G), dim3(256), lds, 0, out, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("G %6d  VGPR ~%3d  LDS %6zu B : %8.2f us per launch   %6.3f us per round of 512 workgroups\n", G, V, lds, us, us / (G / 512.0));
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 20);
    for (int G : {512, 4800, 9600, 19200}) {
        run<16>(G, 0, out);
        run<16>(G, 34816, out);
        run<16>(G, 69632, out);
        run<120>(G, 34816, out);
        run<200>(G, 34816, out);
        run<200>(G, 69632, out);
    }
    return 0;
}
