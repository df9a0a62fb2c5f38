// prepack.hip -- the weight packings of the matrix-core convolutions (the tiled conv_x3s / conv_b16s layouts, the generic low-precision
// layout, the ConvTranspose planes): one launch per layer as before, OR every packing of a training step in ONE launch.
//
// Every matrix-core convolution reads its fp32 master weights in a packed, layout- and math-specific form that its entry point used
// to build into the workspace right in front of the launch: 44 launches of 4-9 us per cfg-2 step, ~50 per UNETR step, each on the
// stream's critical path (a kernel starts when its predecessor has drained) although it depends on nothing but the weights.  Here
// every such packing is described by a PackDesc (what pack_launch runs), and a train step can have ALL of them formed at once:
//   * the first step of a model at a shape RECORDS the descriptors its convolutions pack with (key = weight pointer + kind + layout);
//   * every later step starts with mi355seg_prepack_run: one memset + one launch measuring max |w| of the f16x3-scaled weights + ONE
//     launch (pack_all_kernel: block -> job by a table in the arena) that writes every packing into a caller-owned arena, on the
//     step's own stream; a convolution that finds its key reads the arena copy and launches nothing.
// (r6 first tried the same replay on a SIDE stream beside the step's first kernels: 0.1-0.5 ms slower on every workload,
//  profiles/r06_prepack_side_stream_REVERTED.log -- small kernels on a second queue delay the main queue's launches.)
// A key that is not found (another shape, another model, inference, a misaligned weight tensor) packs in place as before.
#include "common.h"
#include "internal.h"
#include "igemm_kernel.h"
#include "pack.h"
#include <vector>
#include <cstring>

namespace seg {

namespace {

// ---------------------------------------------------------------- the packings as device functions of a virtual block index
// The two hot conv packings, tiled: a workgroup owns NB (8; 4 for k5 and for narrow layers) GEMM columns x one 16-channel K chunk x all taps, reads that block of W
// with full-width coalesced loads (the per-element kernels read W at a stride of T floats -- one 64-byte sector per 4 bytes used;
// 10-16 us per layer, 0.4-0.8 ms of a V-Net / Res-U-Net step) into LDS and emits whole 16-byte fragment slots.
// LAYOUT 0: conv_b16s.hip (P = NT), 1: conv_x3s.hip (P = NBW), 2: conv_x3s.hip f16x3 (two fp16 planes of w * 2^sw, sw from *amax_w).
// mode 0 / 1 as pack_src (conv_mfma.hip).
template <int LAYOUT, int NB>
__device__ __forceinline__ void pack_tiled_body(const int vbid, float* tile, const float* __restrict__ w, bf16* __restrict__ wq, int K, int Nn, int T, int P, int mode,
                                                const float* __restrict__ oscale, const float* __restrict__ amax_w) {
    const int nnb = Nn / NB;
    const int n0 = (vbid % nnb) * NB, chunk = vbid / nnb, k0 = chunk * 16;
    const int tid = threadIdx.x;
    // phase 1: mode 0: NB runs (n) of 16 T floats at W[n][k0 ..][.]; mode 1: 16 runs (k) of NB T floats at W[k][n0 ..][.]
    const int nrun = mode == 0 ? NB : 16, rlen = (mode == 0 ? 16 : NB) * T;
    for (int i = tid * 4; i < nrun * rlen; i += 1024) {
        const int run = i / rlen, off = i - run * rlen;
        const float* src = mode == 0 ? w + ((long long)(n0 + run) * K + k0) * T : w + ((long long)(k0 + run) * Nn + n0) * T;
        *reinterpret_cast<f32x4_t*>(tile + i) = *reinterpret_cast<const f32x4_t*>(src + off);
    }
    __syncthreads();
    const int nstep = LAYOUT == 0 ? (T + 1) / 2 : X3S_NPAIR, nch = K / 16;
    constexpr int LB = NB == 8 ? 3 : (NB == 4 ? 2 : 1);
    for (int q = tid; q < nstep * 4 * NB; q += 256) {
        const int half = q & 1, nl = (q >> 1) & (NB - 1), gh = (q >> (1 + LB)) & 1, s = q >> (2 + LB);
        const int tap = LAYOUT == 0 ? 2 * s + gh : x3s_pair_tap(s, gh);
        const int n = n0 + nl, g = 2 * gh + half;
        float v[8];
        if (tap < T) {
            float sc = (oscale && mode == 0) ? oscale[n] : 1.f;
            if (LAYOUT == 2) sc *= pow2f(f16x_scale_exp(*amax_w));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kl = 8 * half + e;
                v[e] = (mode == 0 ? tile[(nl * 16 + kl) * T + tap] : tile[(kl * NB + nl) * T + (T - 1 - tap)]) * sc;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        if (LAYOUT == 0) {
            const int NT = P, ntt = NT / 16, nt = n / NT, nin = n - nt * NT;
            const int tt = 2 * (nin / 32) + ((nin >> 2) & 1), c = 4 * ((nin & 31) >> 3) + (nin & 3);
            bf16x8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
            reinterpret_cast<bf16x8_t*>(wq)[((((long long)nt * nch + chunk) * nstep + s) * ntt + tt) * 64 + c + 16 * g] = o;
        } else if (LAYOUT == 2) {
            const int NBW = P, NT = 32 * NBW, nt = n / NT, nin = n - nt * NT;
            const int nh = nin / 32, t2 = (nin & 31) >> 4, c = nin & 15;
            f16x8_t oh, ol;
#pragma unroll
            for (int e = 0; e < 8; ++e) { _Float16 a, b; split2h(v[e], a, b); oh[e] = a; ol[e] = b; }
            _Float16* dst = reinterpret_cast<_Float16*>(wq) + (((long long)nt * nch + chunk) * (X3S_NPAIR * NBW) + s * NBW + nh) * 2048 + t2 * 512 + (c + 16 * g) * 8;
            *reinterpret_cast<f16x8_t*>(dst) = oh;
            *reinterpret_cast<f16x8_t*>(dst + 1024) = ol;
        } else {
            const int NBW = P, NT = 32 * NBW, nt = n / NT, nin = n - nt * NT;
            const int nh = nin / 32, t2 = (nin & 31) >> 4, c = nin & 15;
            bf16x8_t oh, om, ol;
#pragma unroll
            for (int e = 0; e < 8; ++e) { bf16 a, b, d; split3(v[e], a, b, d); oh[e] = a; om[e] = b; ol[e] = d; }
            bf16* dst = wq + (((long long)nt * nch + chunk) * (X3S_NPAIR * NBW) + s * NBW + nh) * 3072 + t2 * 512 + (c + 16 * g) * 8;
            *reinterpret_cast<bf16x8_t*>(dst) = oh;
            *reinterpret_cast<bf16x8_t*>(dst + 1024) = om;
            *reinterpret_cast<bf16x8_t*>(dst + 2048) = ol;
        }
    }
}

// the generic low-precision layout (igemm_kernel.h): wq[nt][chunk][tap][kstep][plane][h][j][e] = plane of B[k = chunk*CK + kstep*16 + 8h + e][n = nt*NT + j]
// (IDX: the element index type -- every packing of the networks here has fewer than 2^31 elements, and seven 64-bit divisions per
//  element were most of this kernel's time: 44 us for the 3.5 M weights of a 256 -> 512 k3 layer, 10 us for a 1 x 1 x 1 one)
template <int NP, typename IDX>
__device__ __forceinline__ void pack_wq_lowp_body(const int vbid, const int nvb, const float* __restrict__ w, bf16* __restrict__ wq, int K, int Nn, int T, int NT, int mode,
                                                  int aux, int CK, int TW, const TapList& taps, const float* __restrict__ oscale) {
    const IDX total = (IDX)K * Nn * T;
    const int plane = 2 * NT * 8;
    for (IDX idx = (IDX)vbid * blockDim.x + threadIdx.x; idx < total; idx += (IDX)nvb * blockDim.x) {
        IDX r = idx;
        const int e = (int)(r % 8); r /= 8;
        const int j = (int)(r % NT); r /= NT;
        const int h = (int)(r % 2); r /= 2;
        const int kst = (int)(r % (CK / 16)); r /= (CK / 16);
        const int tap = (int)(r % T); r /= T;
        const int chunk = (int)(r % (K / CK)); r /= (K / CK);
        const int nt = (int)r;
        const float v = pack_src(w, mode, nt * NT + j, chunk * CK + kst * 16 + 8 * h + e, tap, K, Nn, T, aux, TW, taps) * (oscale ? oscale[nt * NT + j] : 1.f);
        const long long base = ((((long long)nt * (K / CK) + chunk) * T + tap) * (CK / 16) + kst) * (NP * plane) + ((long long)h * NT + j) * 8 + e;
        if (NP == 3) {
            bf16 bh, bm, bl;
            split3(v, bh, bm, bl);
            wq[base] = bh; wq[base + plane] = bm; wq[base + 2 * plane] = bl;
        } else {
            wq[base] = (bf16)v;
        }
    }
}

// ConvTranspose3d k2 s2 (convt_direct.hip): packed planes of w (Cin, Cout, 8): slot (ntile, chunk, plane, kgroup, row) holds
// k = chunk * 64 + kgroup * 8 .. + 7 of GEMM column n = ntile * BN + row.   SCATTER: n = (t, co), k = ci.   GATHER: n = ci, k = (t, co).
template <int NP>
__device__ __forceinline__ void convt_pack_planes_body(const int vbid, const int nvb, const float* __restrict__ w, bf16x8_t* __restrict__ wq, int Cin, int Cout,
                                                       int gather, int BN, int K, int Ncols) {
    const int nchunk = K / 64;
    const long long total = (long long)(Ncols / BN) * nchunk * 8 * BN;
    for (long long i = (long long)vbid * blockDim.x + threadIdx.x; i < total; i += (long long)nvb * blockDim.x) {
        const int row = (int)(i % BN); long long q = i / BN;
        const int kgp = (int)(q % 8); q /= 8;
        const int chunk = (int)(q % nchunk); const int ntile = (int)(q / nchunk);
        const int n = ntile * BN + row, k0 = chunk * 64 + kgp * 8;
        bf16x8_t ph, pm, pl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // three planes (fp32 tensors): k-group g of a 32-wide k-step holds channels {4g .. 4g+3, 16 + 4g .. 16 + 4g+3}
            const int k = NP == 3 ? (k0 & ~31) + 4 * ((k0 >> 3) & 3) + (e < 4 ? e : 12 + e) : k0 + e;
            int ci, co, t;
            if (gather) { ci = n; t = k / Cout; co = k - t * Cout; }
            else { ci = k; t = n / Cout; co = n - t * Cout; }
            const float v = w[((long long)ci * Cout + co) * 8 + t];
            bf16 h, m, l;
            split3(v, h, m, l);
            ph[e] = h; pm[e] = m; pl[e] = l;
        }
        bf16x8_t* dst = wq + (((long long)ntile * nchunk + chunk) * NP * 8 + kgp) * BN + row;
        dst[0] = ph;
        if (NP == 3) { dst[8 * BN] = pm; dst[16 * BN] = pl; }
    }
}

// one packing, by its descriptor (`tile`: the block's dynamic LDS, pack_lds_bytes(d) of it)
__device__ __forceinline__ void pack_one(const PackDesc& d, const int vbid, const int nvb, float* tile) {
    if (d.kind == PD_TILED) {
        bf16* dst = reinterpret_cast<bf16*>(d.dst);
#define TILED_CASE(LY, NBV) if (d.layout == LY && d.nb == NBV) { pack_tiled_body<LY, NBV>(vbid, tile, d.w, dst, d.K, d.Nn, d.T, d.P, d.mode, d.oscale, d.amax); return; }
        TILED_CASE(2, 8) TILED_CASE(2, 4) TILED_CASE(0, 8) TILED_CASE(0, 4) TILED_CASE(1, 8) TILED_CASE(1, 4)
#undef TILED_CASE
    } else if (d.kind == PD_LOWP) {
        bf16* dst = reinterpret_cast<bf16*>(d.dst);
        if (d.layout) {          // (64-bit element indices: 2^31 elements and more)
            if (d.np == 3) pack_wq_lowp_body<3, long long>(vbid, nvb, d.w, dst, d.K, d.Nn, d.T, d.P, d.mode, d.aux, d.CK, d.TW, d.taps, d.oscale);
            else pack_wq_lowp_body<1, long long>(vbid, nvb, d.w, dst, d.K, d.Nn, d.T, d.P, d.mode, d.aux, d.CK, d.TW, d.taps, d.oscale);
        }
        else if (d.np == 3) pack_wq_lowp_body<3, unsigned>(vbid, nvb, d.w, dst, d.K, d.Nn, d.T, d.P, d.mode, d.aux, d.CK, d.TW, d.taps, d.oscale);
        else pack_wq_lowp_body<1, unsigned>(vbid, nvb, d.w, dst, d.K, d.Nn, d.T, d.P, d.mode, d.aux, d.CK, d.TW, d.taps, d.oscale);
    } else {
        if (d.np == 3) convt_pack_planes_body<3>(vbid, nvb, d.w, reinterpret_cast<bf16x8_t*>(d.dst), d.K, d.Nn, d.mode, d.P, d.aux, d.TW);
        else convt_pack_planes_body<1>(vbid, nvb, d.w, reinterpret_cast<bf16x8_t*>(d.dst), d.K, d.Nn, d.mode, d.P, d.aux, d.TW);
    }
}

__global__ __launch_bounds__(256) void pack_one_kernel(PackDesc d) {
    extern __shared__ float tile[];
    pack_one(d, (int)blockIdx.x, (int)gridDim.x, tile);
}

// every packing of a plan: block -> job by bisection of the jobs' first blocks (all wave-uniform: scalar loads)
__global__ __launch_bounds__(256) void pack_all_kernel(const PackDesc* __restrict__ tab, int n) {
    extern __shared__ float tile[];
    const int bid = (int)blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first <= bid) lo = mid; else hi = mid - 1;
    }
    pack_one(tab[lo], bid - tab[lo].first, tab[lo].nblk, tile);
}

// max |w| of the jobs whose packing scales by it (f16x3), max-combined into the zeroed slots the jobs point at (positive floats order
// as unsigned integers); a job takes amax_nblk blocks from amax_first
__global__ __launch_bounds__(256) void amax_all_kernel(const PackDesc* __restrict__ tab, int n) {
    const int bid = (int)blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].amax_first <= bid) lo = mid; else hi = mid - 1;
    }
    const PackDesc& d = tab[lo];
    const long long ne = d.amax_elems;
    if (ne <= 0 || d.amax_slot_of >= 0) return;            // (a second form of the same weights shares the first one's slot)
    const float* w = d.w;
    float m = 0.f;
    for (long long i = ((long long)(bid - d.amax_first) * 256 + threadIdx.x) * 4; i < ne; i += (long long)d.amax_nblk * 1024) {
        if (i + 3 < ne) { const f32x4_t v = *reinterpret_cast<const f32x4_t*>(w + i); m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3]))); }
        else for (long long j = i; j < ne; ++j) m = fmaxf(m, fabsf(w[j]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(const_cast<float*>(d.amax)), __builtin_bit_cast(unsigned, m));
}

struct PackJob { PackKey key; size_t bytes, off; PackDesc d; };
struct PackPlan {
    std::vector<PackJob> jobs;
    size_t arena_bytes = 0, amax_off = 0, tab_off = 0, lds = 0;
    int namax = 0, nblocks = 0, amax_blocks = 0;
    const void* uploaded_to = nullptr;
    bool sealed = false;
};

std::vector<PackPlan*> g_plans;      // index = plan id - 1 (nullptr once freed)
PackPlan* g_rec = nullptr;           // being recorded
PackPlan* g_act = nullptr;           // replayed for the running step
char* g_arena = nullptr;
int g_cursor = 0;

bool key_eq(const PackKey& a, const PackKey& b) { return std::memcmp(&a, &b, sizeof(PackKey)) == 0; }

int find_job(const PackPlan& p, const PackKey& k, int hint) {
    const int n = (int)p.jobs.size();
    if (hint < n && key_eq(p.jobs[hint].key, k)) return hint;
    for (int i = 0; i < n; ++i) if (key_eq(p.jobs[i].key, k)) return i;
    return -1;
}

PackPlan* plan_of(int id) { return (id >= 1 && id <= (int)g_plans.size()) ? g_plans[id - 1] : nullptr; }

}  // namespace

size_t pack_lds_bytes(const PackDesc& d) { return d.kind == PD_TILED ? (size_t)16 * d.nb * d.T * sizeof(float) : 0; }

int pack_blocks(const PackDesc& d) {
    if (d.kind == PD_TILED) return (d.Nn / d.nb) * (d.K / 16);
    long long total = d.kind == PD_LOWP ? (long long)d.K * d.Nn * d.T : (long long)d.aux * d.TW / 8;      // (ConvT: K * Ncols / 8 slots)
    long long g = (total + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

// one packing as a launch of its own (what every entry point did, and still does for a key no active plan holds)
void pack_launch(const PackDesc& d, hipStream_t st) {
    const size_t lds = pack_lds_bytes(d);
    SEG_SET_LDS(pack_one_kernel, 64 * 1024);
    hipLaunchKernelGGL(pack_one_kernel, dim3(pack_blocks(d)), dim3(256), lds, st, d);
}

PackKey make_pack_key(const void* w, int kind, int a, int b, int c, int d, int e, int f, int g) {
    PackKey k;
    std::memset(&k, 0, sizeof(k));       // (padding bytes take part in the comparison)
    k.w = w; k.kind = kind; k.p[0] = a; k.p[1] = b; k.p[2] = c; k.p[3] = d; k.p[4] = e; k.p[5] = f; k.p[6] = g;
    return k;
}

bool prepack_find(const PackKey& key, void** wq, const float** w_amax) {
    if (!g_act) return false;
    const int i = find_job(*g_act, key, g_cursor);
    if (i < 0) return false;
    const PackJob& j = g_act->jobs[i];
    g_cursor = i + 1;
    *wq = g_arena + j.off;
    if (w_amax && j.d.amax_elems > 0) *w_amax = reinterpret_cast<const float*>(g_arena + g_act->amax_off) + j.d.amax_idx;
    return true;
}

void prepack_note(const PackKey& key, size_t bytes, const PackDesc& d) {
    if (!g_rec || g_rec->sealed || d.oscale) return;
    if (find_job(*g_rec, key, (int)g_rec->jobs.size()) >= 0) return;        // a module applied twice in a step packs once
    g_rec->jobs.push_back(PackJob{key, bytes, 0, d});
}

}  // namespace seg

using namespace seg;

extern "C" {

int mi355seg_prepack_record_begin(void) {
    SEG_CHECK_ARG(!g_rec, "mi355seg_prepack_record_begin: a recording is already open");
    SEG_CHECK_ARG(!g_act, "mi355seg_prepack_record_begin: a plan is active");
    g_rec = new PackPlan();
    return MI355SEG_OK;
}

// closes the recording: *plan = its id (0 when nothing was recorded), *arena_bytes = the bytes mi355seg_prepack_run needs
int mi355seg_prepack_record_end(int* plan, size_t* arena_bytes) {
    SEG_CHECK_ARG(g_rec && plan && arena_bytes, "mi355seg_prepack_record_end: no open recording / null outputs");
    PackPlan* p = g_rec;
    g_rec = nullptr;
    *plan = 0; *arena_bytes = 0;
    if (p->jobs.empty()) { delete p; return MI355SEG_OK; }
    size_t off = 0;
    int first = 0, afirst = 0;
    for (size_t i = 0; i < p->jobs.size(); ++i) {
        PackJob& j = p->jobs[i];
        j.off = off; off += align_up(j.bytes, 256);
        PackDesc& d = j.d;
        d.first = first; d.nblk = pack_blocks(d); first += d.nblk;
        if (pack_lds_bytes(d) > p->lds) p->lds = pack_lds_bytes(d);
        d.amax_idx = -1; d.amax_slot_of = -1; d.amax_first = afirst; d.amax_nblk = 0;
        if (d.amax_elems > 0) {
            for (size_t k = 0; k < i; ++k)               // the forward and the input-gradient form of one tensor share the scalar
                if (p->jobs[k].d.amax_elems == d.amax_elems && p->jobs[k].d.w == d.w && p->jobs[k].d.amax_slot_of < 0) { d.amax_slot_of = (int)k; d.amax_idx = p->jobs[k].d.amax_idx; break; }
            if (d.amax_slot_of < 0) {
                d.amax_idx = p->namax++;
                long long nb = (d.amax_elems + 16383) / 16384;
                d.amax_nblk = (int)(nb > 256 ? 256 : nb);
            }
        }
        if (d.amax_nblk == 0) d.amax_nblk = 1;           // (every job owns at least one block of the measuring launch: the bisection needs increasing firsts)
        afirst += d.amax_nblk;
    }
    p->nblocks = first; p->amax_blocks = afirst;
    p->amax_off = off; off += align_up((size_t)(p->namax ? p->namax : 1) * sizeof(float), 256);
    p->tab_off = off; off += align_up(p->jobs.size() * sizeof(PackDesc), 256);
    p->arena_bytes = off;
    p->sealed = true;
    g_plans.push_back(p);
    *plan = (int)g_plans.size();
    *arena_bytes = p->arena_bytes;
    return MI355SEG_OK;
}

int mi355seg_prepack_jobs(int plan) { PackPlan* p = plan_of(plan); return p ? (int)p->jobs.size() : 0; }

// forms every packing of `plan` in the caller's arena on `stream` (three launches) and makes the plan the one convolutions look their
// weights up in, until mi355seg_prepack_done.  The job table is written into the arena at the first call with that arena (a blocking
// copy): inside a stream capture an arena that has not been used eagerly yet leaves the plan inactive (the convolutions pack in place).
int mi355seg_prepack_run(int plan, void* arena, size_t arena_bytes, void* stream) {
    PackPlan* p = plan_of(plan);
    SEG_CHECK_ARG(p && arena, "mi355seg_prepack_run: unknown plan / null arena");
    SEG_CHECK_ARG(!g_act && !g_rec, "mi355seg_prepack_run: another plan is active or being recorded");
    SEG_CHECK_ARG(((uintptr_t)arena % 256) == 0, "mi355seg_prepack_run: the arena must be 256-byte aligned");
    SEG_CHECK_WS(p->arena_bytes, arena_bytes);
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)arena;
    if (p->uploaded_to != arena) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return MI355SEG_OK;
        std::vector<PackDesc> tab(p->jobs.size());
        for (size_t i = 0; i < p->jobs.size(); ++i) {
            tab[i] = p->jobs[i].d;
            tab[i].dst = base + p->jobs[i].off;
            tab[i].amax = tab[i].amax_elems > 0 ? reinterpret_cast<const float*>(base + p->amax_off) + tab[i].amax_idx : nullptr;
        }
        if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(base + p->tab_off, tab.data(), tab.size() * sizeof(PackDesc), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("mi355seg_prepack_run: job table upload failed");
            return MI355SEG_EHIP;
        }
        p->uploaded_to = arena;
    }
    const PackDesc* tab = reinterpret_cast<const PackDesc*>(base + p->tab_off);
    const int n = (int)p->jobs.size();
    if (p->namax) {
        if (hipMemsetAsync(base + p->amax_off, 0, (size_t)p->namax * sizeof(float), st) != hipSuccess) { set_error("mi355seg_prepack_run: hipMemsetAsync failed"); return MI355SEG_EHIP; }
        hipLaunchKernelGGL(amax_all_kernel, dim3(p->amax_blocks), dim3(256), 0, st, tab, n);
    }
    SEG_SET_LDS(pack_all_kernel, 64 * 1024);
    hipLaunchKernelGGL(pack_all_kernel, dim3(p->nblocks), dim3(256), p->lds, st, tab, n);
    SEG_CHECK_LAUNCH();
    g_act = p; g_arena = base; g_cursor = 0;
    return MI355SEG_OK;
}

// end of the step: the plan is no longer consulted
int mi355seg_prepack_done(void* stream) {
    (void)stream;
    g_act = nullptr; g_arena = nullptr;
    return MI355SEG_OK;
}

int mi355seg_prepack_active(void) { return g_act ? 1 : 0; }

int mi355seg_prepack_free(int plan) {
    PackPlan* p = plan_of(plan);
    if (!p) return MI355SEG_OK;
    SEG_CHECK_ARG(p != g_act, "mi355seg_prepack_free: the plan is active");
    delete p;
    g_plans[plan - 1] = nullptr;
    return MI355SEG_OK;
}

}  // extern "C"
