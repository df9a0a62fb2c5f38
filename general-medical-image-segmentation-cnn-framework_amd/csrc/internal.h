// internal.h -- cross-translation-unit declarations (not part of the C-ABI).
#pragma once
#include "common.h"

namespace seg {

// norm.hip
size_t colsum_ws_bytes(int C);
int channel_sums(const float* x, int ldx, long long rows, int C, double* sum, double* sq, float* fsum, int accumulate,
                 void* ws, size_t ws_bytes, hipStream_t st);

// conv_generic.hip
void pack_w_fwd(const float* w, float* wp, int Cout, int Cin, int T, hipStream_t st);
void pack_w_dgrad(const float* w, float* wd, int Cout, int Cin, int T, int flip, hipStream_t st);
void wgrad_reduce(const float* part, float* dw, int splits, int T, int Cin, int Cout, int accumulate, hipStream_t st);

// conv_mfma.hip -- fp32-MFMA implicit-GEMM Conv3d k3 s1 p1
size_t conv_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad);
bool conv_mfma_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int ldy);
// dgrad != 0: `w` is still the (Cfwd_out=Cin_here ... ) torch weight of the FORWARD conv, i.e. shape (Cin, Cout, 27)
// seen from this call's Cin/Cout; taps are flipped while packing.
int conv_fwd_mfma(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int N, int D, int H, int W,
                  int Cin, int Cout, int dgrad, double* ssum, double* ssq, void* ws, size_t ws_bytes, hipStream_t st);
size_t wgrad_mfma_ws_bytes(int N, int D, int H, int W, int Cin, int Cout);
bool wgrad_mfma_supported(int N, int D, int H, int W, int Cin, int Cout, int k, int stride, int pad, int ldx, int lddy);
int conv_wgrad_mfma(const float* dy, int lddy, const float* x, int ldx, float* dw, int N, int D, int H, int W, int Cin,
                    int Cout, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);

}  // namespace seg
