// conv_igemm_lowp.hip -- the bf16-matrix-core instantiations of the implicit-GEMM convolution kernel (igemm_kernel.h):
//   MATH_X3  ("bf16x6", fp32 tensors, three bf16 planes per operand, six MFMAs per product) for the k3 layers of the fp32 models;
//   MATH_B16 (bf16 tensors) for k1 / k3 / k5, the gather mode and ConvTranspose3d k2 s2 of the bf16 configurations
//            (V-Net vnet3d.py:21-121, Residual U-Net residual_unet3d.py:82-107, UNETR decoder unetr.py:8-51).
// Host-side planning, weight packing and the ConvT / gather argument set-up are shared with the fp32 policy (conv_mfma.hip).
#include "common.h"
#include "internal.h"
#include "igemm_kernel.h"

namespace seg {

// which (KS, CK, BX, MB) tiles exist per policy -- igemm_plan() asks before it commits to a low-precision plan
bool igemm_lowp_has(int math, int KS, int CK, int BX, int MB) {
    if (math == MATH_X3) return KS == 3 && CK == 16 && (BX == 16 || BX == 8) && (MB == 1 || MB == 2);
    if (math == MATH_B16) {
        if (KS == 3) return CK == 16 && (MB == 1 || MB == 2 || ((MB == 3 || MB == 4) && BX == 32));
        if (KS == 5) return CK == 16 && (MB == 1 || (MB == 2 && BX != 32));
        if (KS == 1) return (CK == 16 || CK == 64) && (MB == 1 || MB == 2);
    }
    return false;
}

template <int MATH, int KS, int CK, bool ALLOW_MB2, bool ALLOW_BX32>
static void dispatch_ck(const IgemmPlan& p, const IgemmArgs& a, int nwg, hipStream_t st) {
#define IGEMM_CASE(bx, mb, nbw) \
    if (p.BX == bx && p.MB == mb && p.NBW == nbw) launch_igemm<MATH, KS, bx, mb, nbw, CK>(a, nwg, st)
    if constexpr (ALLOW_BX32) {
        if constexpr (ALLOW_MB2) { IGEMM_CASE(32, 2, 2); IGEMM_CASE(32, 2, 1); }
        IGEMM_CASE(32, 1, 2); IGEMM_CASE(32, 1, 1);
    }
    if constexpr (ALLOW_MB2) { IGEMM_CASE(16, 2, 2); IGEMM_CASE(16, 2, 1); IGEMM_CASE(8, 2, 2); IGEMM_CASE(8, 2, 1); }
    IGEMM_CASE(16, 1, 2); IGEMM_CASE(16, 1, 1); IGEMM_CASE(8, 1, 2); IGEMM_CASE(8, 1, 1);
#undef IGEMM_CASE
}

void dispatch_igemm_lowp(int math, const IgemmPlan& p, const IgemmArgs& a, int nwg, hipStream_t st) {
    if (math == MATH_X3) {
        dispatch_ck<MATH_X3, 3, 16, true, false>(p, a, nwg, st);
    } else {
        if (p.KS == 3 && p.WN == 2) launch_igemm<MATH_B16, 3, 32, 6, 1, 16, 2>(a, nwg, st);
        else if (p.KS == 3 && p.MB == 4 && p.NBW == 1) launch_igemm<MATH_B16, 3, 32, 4, 1, 16>(a, nwg, st);
        else if (p.KS == 3 && p.MB == 3 && p.NBW == 2) launch_igemm<MATH_B16, 3, 32, 3, 2, 16>(a, nwg, st);
        else if (p.KS == 3) dispatch_ck<MATH_B16, 3, 16, true, true>(p, a, nwg, st);
        else if (p.KS == 5 && p.MB == 2) {
            if (p.BX == 16) { if (p.NBW == 2) launch_igemm<MATH_B16, 5, 16, 2, 2, 16>(a, nwg, st); else launch_igemm<MATH_B16, 5, 16, 2, 1, 16>(a, nwg, st); }
            else { if (p.NBW == 2) launch_igemm<MATH_B16, 5, 8, 2, 2, 16>(a, nwg, st); else launch_igemm<MATH_B16, 5, 8, 2, 1, 16>(a, nwg, st); }
        }
        else if (p.KS == 5) dispatch_ck<MATH_B16, 5, 16, false, true>(p, a, nwg, st);
        else if (p.CK == 64) dispatch_ck<MATH_B16, 1, 64, true, true>(p, a, nwg, st);
        else dispatch_ck<MATH_B16, 1, 16, true, true>(p, a, nwg, st);
    }
}

// the output phases of a strided input gradient in ONE launch (bf16 tensors, k1 gather plans; conv_gather_dgrad_mfma)
template <int CK>
static bool dispatch_phases_ck(const IgemmPlan& p, const IgemmArgs& a, const IgemmPhases& pt, int nwg, hipStream_t st) {
#define PHASE_CASE(bx, mb, nbw) \
    if (p.BX == bx && p.MB == mb && p.NBW == nbw) { launch_igemm_phases<MATH_B16, 1, bx, mb, nbw, CK>(a, pt, nwg, st); return true; }
    PHASE_CASE(32, 2, 2) PHASE_CASE(32, 2, 1) PHASE_CASE(32, 1, 2) PHASE_CASE(32, 1, 1)
    PHASE_CASE(16, 2, 2) PHASE_CASE(16, 2, 1) PHASE_CASE(8, 2, 2) PHASE_CASE(8, 2, 1)
    PHASE_CASE(16, 1, 2) PHASE_CASE(16, 1, 1) PHASE_CASE(8, 1, 2) PHASE_CASE(8, 1, 1)
#undef PHASE_CASE
    return false;
}

bool dispatch_igemm_phases_lowp(int math, const IgemmPlan& p, const IgemmArgs& a, const IgemmPhases& pt, int nwg, hipStream_t st) {
    if (math != MATH_B16 || p.KS != 1 || p.WN != 1) return false;
    return p.CK == 64 ? dispatch_phases_ck<64>(p, a, pt, nwg, st) : (p.CK == 16 ? dispatch_phases_ck<16>(p, a, pt, nwg, st) : false);
}

}  // namespace seg
