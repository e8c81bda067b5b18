// mm_dpair_tu.hip -- translation unit of the float64 exact pair kernels (mm_kernel_dpair.hip): their instances and launches.
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_dpair.hip"

namespace mm {

// One utterance per workgroup; one launch per phase: the forward agents are the first B workgroups (by rank in the
// longest-first order), the backward agents the second B (mm_pairs_tu.hip).
// (NJ: 64-lane passes over the pdfs in the service wave, 2 for P + 1 <= 128, else 4)
template <int NJ, int PHASE>
__global__ void __launch_bounds__(1024) mm_fbd_kernel(RunParams p) {
    const int dir = (int)blockIdx.x >= p.B;
    dpair_agent<MM_PAIR_KA, MM_ROW_RS, PHASE, NJ>(p, (int)blockIdx.x - (dir ? p.B : 0), dir);
}
template <int NJ, int PHASE>
static int launch_dpair_phase(const PairLaunch *h, const RunParams &p, hipStream_t st) {
    const size_t lds = pair_lds_bytes(MM_ROW_RS, PHASE, h->slotrows, 0, pair_pc(NJ));
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "exact pair kernel: LDS");
    auto kernel = mm_fbd_kernel<NJ, PHASE>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(kernel, dim3(2 * unsigned(h->B)), dim3(64 * (h->nwc + 1)), lds, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int NJ>
static int launch_dpairs_nj(const PairLaunch *h, const RunParams &p, hipStream_t s0) {
    int rc = launch_dpair_phase<NJ, 0>(h, p, s0);
    if (!rc) rc = launch_dpair_phase<NJ, 1>(h, p, s0);
    if (rc) return rc;
    hipLaunchKernelGGL(mm_dpair_finish_kernel, dim3(unsigned(h->B)), dim3(256), 0, s0, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
// ---- teams of H workgroups per utterance and direction (the split kernels' graphs: mm_split_tu.hip)
template <int H> struct DSplitGeo;
template <> struct DSplitGeo<2> { static constexpr int RS = MM_SPLIT_RS, RSH = MM_SPLIT_RSH, KA = 36; };
template <> struct DSplitGeo<4> { static constexpr int RS = MM_SPLIT4_RS, RSH = MM_SPLIT4_RSH, KA = 36; };
template <> struct DSplitGeo<8> { static constexpr int RS = MM_SPLIT8_RS, RSH = MM_SPLIT8_RSH, KA = 36; };
template <int NJ, int PHASE, int H>
__global__ void __launch_bounds__(1024) mm_fbds_kernel(RunParams p) {
    const int half = (int)gridDim.x / 2, dir = (int)blockIdx.x >= half;
    const int blk = (int)blockIdx.x - (dir ? half : 0);
    const int ui = (blk / (8 * H)) * 8 + (blk & 7), hset = (blk >> 3) % H;  // (the workgroups of a team are 8 apart: mm_split_tu.hip)
    if (ui >= p.B) return;
    if ((p.x_sleep & 0x400) && hset == 1) return;  // (test aid: a team mate that never shows up)
    dpair_agent<DSplitGeo<H>::KA, DSplitGeo<H>::RS, PHASE, NJ, H, DSplitGeo<H>::RSH>(p, ui, dir, hset);
}
template <int NJ, int PHASE, int H>
static int launch_dsplit_phase(const PairLaunch *h, const RunParams &p, hipStream_t st) {
    const size_t lds = pair_lds_bytes(DSplitGeo<H>::RS, PHASE, h->slotrows, DSplitGeo<H>::RSH, pair_pc(NJ));
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "exact split kernel: LDS");
    auto kernel = mm_fbds_kernel<NJ, PHASE, H>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(kernel, dim3(2 * ((unsigned(h->B) + 7) / 8 * 8 * H)), dim3(64 * (MM_SPLIT_NWC + 2)), lds, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int NJ, int H>
static int launch_dsplit_nj(const PairLaunch *h, const RunParams &p, hipStream_t s0) {
    int rc = launch_dsplit_phase<NJ, 0, H>(h, p, s0);
    if (!rc) rc = launch_dsplit_phase<NJ, 1, H>(h, p, s0);
    if (rc) return rc;
    hipLaunchKernelGGL(mm_dpair_finish_kernel, dim3(unsigned(h->B)), dim3(256), 0, s0, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
int mm_launch_dpairs(const PairLaunch &pl, const RunParams &p, hipStream_t s0) {
    const int nj = mm_pair_nj(pl.max_P1, pl.H);
    if (nj == 0) return MM_ERR_UNSUPPORTED;
    if (pl.H == 1) {
        if (pl.pair_ka > MM_PAIR_KA) return MM_ERR_UNSUPPORTED;
        return nj == 2 ? launch_dpairs_nj<2>(&pl, p, s0) : (nj == 4 ? launch_dpairs_nj<4>(&pl, p, s0) : launch_dpairs_nj<8>(&pl, p, s0));
    }
    if (pl.pair_ka > mm_split_ka(pl.H)) return MM_ERR_UNSUPPORTED;
    if (pl.H == 8) return nj == 2 ? launch_dsplit_nj<2, 8>(&pl, p, s0) : (nj == 4 ? launch_dsplit_nj<4, 8>(&pl, p, s0) : launch_dsplit_nj<5, 8>(&pl, p, s0));
    if (pl.H == 4) return nj == 2 ? launch_dsplit_nj<2, 4>(&pl, p, s0) : (nj == 4 ? launch_dsplit_nj<4, 4>(&pl, p, s0) : launch_dsplit_nj<8, 4>(&pl, p, s0));
    if (pl.H != 2) return mm_fail(MM_ERR_UNSUPPORTED, "exact split kernel: teams of 2, 4 or 8");
    return nj == 2 ? launch_dsplit_nj<2, 2>(&pl, p, s0) : (nj == 4 ? launch_dsplit_nj<4, 2>(&pl, p, s0) : launch_dsplit_nj<8, 2>(&pl, p, s0));
}

}  // namespace mm
