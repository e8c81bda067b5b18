// mm_pairs_tu.hip -- translation unit of the pair kernels (mm_kernel_pairs.hip): their instances and launches.
#define MM_SECONDARY_TU
#include <algorithm>
#include "mm_internal.h"
#include "mm_kernel_pairs.hip"

namespace mm {

// The pair kernels (mm_kernel_pairs.hip): phase A, then phase B, each ONE launch that holds the forward agents (the first
// half of the grid) and the backward agents (the second half) of all pairs -- the two agents of a pair run at the same time
// because they are workgroups of one grid, not because two streams happen to share no hardware queue.  (Rounds 2 and 3
// launched the two directions as two kernels on a pair of library streams, forked from and joined into the caller's stream
// by events: every fork, join and cross-stream wait cost 10-25 us of idle device -- ~80 us of a 2.9 ms call, rocprofv3
// timeline --, needed a pool of stream pairs probed for real concurrency, and a special case inside stream captures.)
// (NJ: 64-lane passes over the pdfs in the service wave, 2 for P + 1 <= 128, 4 up to 250, 8 up to 506 -- the last with per-pdf
// arrays of twice the size and a partner-row ring of two vectors instead of three: PairLay)
// (SMALL: graphs of up to 127 states, whose service wave does a sixteenth of the copying and scanning)
template <int NJ, int PHASE, bool SMALL>
__global__ void __launch_bounds__(1024) mm_fbp_kernel(RunParams p) {
    const int npairs = (p.B + 1) / 2, dir = (int)blockIdx.x >= npairs;
    pair_agent<MM_PAIR_KA, MM_ROW_RS, PHASE, -1, NJ, 1, 2 * MM_ROW_RS, SMALL>(p, (int)blockIdx.x - (dir ? npairs : 0), 0, dir);
}
template <int NJ, int PHASE, bool SMALL>
static int launch_pair_phase(const PairLaunch *h, const RunParams &p, hipStream_t st) {
    const size_t lds = pair_lds_bytes(MM_ROW_RS, PHASE, h->slotrows, 0, pair_pc(NJ));
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "pair kernel: LDS");
    auto kernel = mm_fbp_kernel<NJ, PHASE, SMALL>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const unsigned npairs = unsigned((h->B + 1) / 2);
    hipLaunchKernelGGL(kernel, dim3(2 * npairs), dim3(64 * (h->nwc + 1)), lds, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int NJ, bool SMALL>
static int launch_pairs_ka(const PairLaunch *h, const RunParams &p, hipStream_t s0) {
    int rc = launch_pair_phase<NJ, 0, SMALL>(h, p, s0);
    if (!rc) rc = launch_pair_phase<NJ, 1, SMALL>(h, p, s0);  // (phase B of either direction needs phase A of both: stream order)
    if (rc) return rc;
    hipLaunchKernelGGL(mm_pair_finish_kernel, dim3(unsigned(h->B)), dim3(256), 0, s0, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
int mm_launch_pairs(const PairLaunch &pl, const RunParams &p, hipStream_t s0) {
    if (pl.pair_ka > MM_PAIR_KA) return MM_ERR_UNSUPPORTED;
    if (pl.max_P1 <= 128) return pl.small ? launch_pairs_ka<2, true>(&pl, p, s0) : launch_pairs_ka<2, false>(&pl, p, s0);
    if (pl.max_P1 <= 250) return launch_pairs_ka<4, false>(&pl, p, s0);
    if (pl.max_P1 > MM_PAIR_P1MAX) return MM_ERR_UNSUPPORTED;
    return launch_pairs_ka<8, false>(&pl, p, s0);
}
// ---- alpha-recursion / beta-recursion export on the pair kernels (src/inference.jl:62-74, 99-110): phase A of ONE direction over all
// N + 1 frames (pair_agent, XPT), then mm_pair_export_kernel -- instead of the item kernel's one workgroup per utterance with the
// vectors in the log domain (config 3: 26 ms).  The instances of the linear finishes only (up to 250 pdfs).
template <int NJ, bool SMALL>
__global__ void __launch_bounds__(1024) mm_fbx_kernel(RunParams p, int dir) {
    pair_agent<MM_PAIR_KA, MM_ROW_RS, 0, -1, NJ, 1, 2 * MM_ROW_RS, SMALL, true>(p, (int)blockIdx.x, 0, dir);
}
template <int NJ, bool SMALL>
static int launch_pair_export(const PairLaunch *h, const RunParams &p, int dir, hipStream_t st) {
    const size_t lds = pair_lds_bytes(MM_ROW_RS, 0, h->slotrows, 0, pair_pc(NJ));
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "pair kernel: LDS");
    auto kernel = mm_fbx_kernel<NJ, SMALL>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const unsigned npairs = unsigned((h->B + 1) / 2);
    hipLaunchKernelGGL(kernel, dim3(npairs), dim3(64 * (h->nwc + 1)), lds, st, p, dir);
    HIP_TRY(hipGetLastError());
    // the stored rows -> the reference's layout: (pairs, chunks of frames) workgroups, enough of them to keep the memory system busy
    const int chunks = std::max(1, std::min(p.N + 1, int(4096 / std::max(1u, npairs))));
    const int fpb = (p.N + 1 + chunks - 1) / chunks;
    hipLaunchKernelGGL(mm_pair_export_kernel, dim3(npairs, unsigned((p.N + 1 + fpb - 1) / fpb)), dim3(1024), size_t(3) * size_t(p.pair_s1p) * 4, st, p, dir, fpb, 1);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
bool mm_pair_export_fits(const PairLaunch &pl) { return pl.H == 1 && pl.pair_ka <= MM_PAIR_KA && pl.max_P1 <= 250; }
int mm_launch_pair_export(const PairLaunch &pl, const RunParams &p, int dir, hipStream_t s0) {
    if (!mm_pair_export_fits(pl)) return MM_ERR_UNSUPPORTED;
    if (pl.max_P1 <= 128) return pl.small ? launch_pair_export<2, true>(&pl, p, dir, s0) : launch_pair_export<2, false>(&pl, p, dir, s0);
    return launch_pair_export<4, false>(&pl, p, dir, s0);
}
size_t mm_pair_lds_bytes(int phase, int nslotrows, int max_P1) {
    return pair_lds_bytes(MM_ROW_RS, phase, nslotrows, 0, pair_pc(mm_pair_nj(max_P1)));
}
size_t mm_pair_hand_bytes() { return sizeof(PairHand); }

}  // namespace mm
