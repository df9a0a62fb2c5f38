// conv_mfma.hip -- placeholder until the MFMA kernels land (next commit).
#include "common.h"
#include "internal.h"
namespace seg {
size_t conv_mfma_ws_bytes(int, int, int, int, int, int, int, int, int) { return 0; }
bool conv_mfma_supported(int, int, int, int, int, int, int, int, int, int, int) { return false; }
int conv_fwd_mfma(const float*, int, const float*, const float*, float*, int, int, int, int, int, int, int, int, double*, double*,
                  void*, size_t, hipStream_t) { set_error("conv_fwd_mfma: not built"); return MI355SEG_EINVAL; }
bool wgrad_mfma_supported(int, int, int, int, int, int, int, int, int, int, int) { return false; }
int conv_wgrad_mfma(const float*, int, const float*, int, float*, int, int, int, int, int, int, int, void*, size_t, hipStream_t) {
    set_error("conv_wgrad_mfma: not built"); return MI355SEG_EINVAL; }
}  // namespace seg
