// pack.h -- descriptors of the matrix-core convolutions' weight packings (prepack.hip): what an entry point hands to pack_launch for a
// launch of its own, what a recorded training step keeps per packing to form all of them in one launch (mi355seg_prepack_*).
#pragma once
#include "common.h"
#include "igemm_kernel.h"

namespace seg {

// B[k][n] of the generic layouts (conv_mfma.hip's packing comment lists the modes)
__device__ __forceinline__ float pack_src(const float* __restrict__ w, int mode, int n, int k, int tap, int K, int Nn, int T, int aux, int TW,
                                          const TapList& taps) {
    if (mode == 0) return w[((long long)n * K + k) * T + tap];
    if (mode == 1) return w[((long long)k * Nn + n) * T + (T - 1 - tap)];
    if (mode == 2) { const int cout = aux; const int tapn = n / cout, co = n % cout; return w[((long long)k * cout + co) * 8 + tapn]; }
    if (mode == 3 || mode == 5) { const int cout = aux; const int tapk = k / cout, co = k % cout; return w[((long long)n * cout + co) * TW + tapk]; }
    const int cout = aux; const int slot = k / cout, co = k % cout;
    return w[((long long)co * Nn + n) * TW + taps.t[slot]];
}

enum { PD_TILED = 1, PD_LOWP = 2, PD_CONVT = 3 };
// PD_TILED (pack_tiled: conv_b16s / conv_x3s layouts): layout 0 | 1 | 2, nb 4 | 8, (K, Nn, T, P, mode), amax for layout 2
// PD_LOWP  (the generic low-precision layout): np 1 | 3, (K, Nn, T, P = NT, mode, aux, CK, TW, taps)
// PD_CONVT (convt_direct.hip's planes): np 1 | 3, K = Cin, Nn = Cout, mode = gather, P = BN, aux = GEMM K, TW = GEMM columns
struct PackDesc {
    int kind, layout, nb, np;
    const float* w; void* dst; const float* amax; const float* oscale;
    int K, Nn, T, P, mode, aux, CK, TW;
    TapList taps;
    // filled in by the plan (prepack.hip): blocks [first, first + nblk) of the all-at-once launch; the measuring launch's share of a
    // packing that scales by max |w| over amax_elems elements (slot amax_idx of the arena's table; amax_slot_of >= 0: another job's slot)
    int first, nblk, amax_first, amax_nblk, amax_idx, amax_slot_of;
    long long amax_elems;
};
size_t pack_lds_bytes(const PackDesc& d);
int pack_blocks(const PackDesc& d);
void pack_launch(const PackDesc& d, hipStream_t st);

// a recorded step's packings (mi355seg_prepack_*): a pack site builds its key (weight pointer, site, the integers that fix the layout) and
// asks prepack_find -- on a hit `*wq` is the arena copy and `*w_amax`, for a packing that scales by max |w|, the scalar it was scaled
// with -- and otherwise packs in place (pack_launch) and hands the descriptor to prepack_note (kept only while a plan is being recorded)
struct PackKey { const void* w; int kind; int p[7]; };
PackKey make_pack_key(const void* w, int kind, int a = 0, int b = 0, int c = 0, int d = 0, int e = 0, int f = 0, int g = 0);
bool prepack_find(const PackKey& key, void** wq, const float** w_amax);
void prepack_note(const PackKey& key, size_t bytes, const PackDesc& d);
enum { PK_CONV = 1, PK_CONVT_FWD, PK_CONVT_DGRAD, PK_GATHER_FWD, PK_GATHER_DGRAD, PK_CONVT_DIRECT };

}  // namespace seg
