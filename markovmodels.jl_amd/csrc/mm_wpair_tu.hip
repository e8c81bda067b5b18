// mm_wpair_tu.hip -- translation unit of the wide-exponent pair kernels (mm_kernel_wpair.hip): their instances and launches.
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_wpair.hip"

namespace mm {

// Two utterances per workgroup like mm_fbp_kernel, one grid per phase: the forward agents are the first half of the grid, the
// backward agents the second.  (NJ: 64-lane passes over the pdfs in the service wave, 2 for P + 1 <= 128, 4 up to 250)
template <int NJ, int PHASE>
__global__ void __launch_bounds__(1024) mm_fbw_kernel(RunParams p) {
    const int npairs = (p.B + 1) / 2, dir = (int)blockIdx.x >= npairs;
    wpair_agent<MM_PAIR_KA, MM_ROW_RS, PHASE, NJ>(p, (int)blockIdx.x - (dir ? npairs : 0), dir);
}
template <int NJ, int PHASE>
static int launch_wpair_phase(const PairLaunch *h, const RunParams &p, hipStream_t st) {
    const size_t lds = wpair_lds_bytes(MM_ROW_RS, PHASE, h->slotrows, pair_pc(NJ));
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "wide pair kernel: LDS");
    auto kernel = mm_fbw_kernel<NJ, PHASE>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const unsigned npairs = unsigned((h->B + 1) / 2);
    hipLaunchKernelGGL(kernel, dim3(2 * npairs), dim3(64 * (h->nwc + 1)), lds, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int NJ>
static int launch_wpairs_nj(const PairLaunch *h, const RunParams &p, hipStream_t s0) {
    int rc = launch_wpair_phase<NJ, 0>(h, p, s0);
    if (!rc) rc = launch_wpair_phase<NJ, 1>(h, p, s0);
    if (rc) return rc;
    hipLaunchKernelGGL(mm_dpair_finish_kernel, dim3(unsigned(h->B)), dim3(256), 0, s0, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
// ---- teams of H workgroups per utterance pair and direction (the split kernels' graphs: mm_split_tu.hip)
template <int H> struct WSplitGeo;
template <> struct WSplitGeo<2> { static constexpr int RS = MM_SPLIT_RS, RSH = MM_SPLIT_RSH, KA = 36; };
template <> struct WSplitGeo<4> { static constexpr int RS = MM_SPLIT4_RS, RSH = MM_SPLIT4_RSH, KA = 36; };
template <> struct WSplitGeo<8> { static constexpr int RS = MM_SPLIT8_RS, RSH = MM_SPLIT8_RSH, KA = 36; };
template <int NJ, int PHASE, int H>
__global__ void __launch_bounds__(1024) mm_fbws_kernel(RunParams p) {
    const int half = (int)gridDim.x / 2, dir = (int)blockIdx.x >= half;
    const int blk = (int)blockIdx.x - (dir ? half : 0);
    const int pair = (blk / (8 * H)) * 8 + (blk & 7), hset = (blk >> 3) % H;  // (the workgroups of a team are 8 apart: mm_split_tu.hip)
    if (pair >= (p.B + 1) / 2) return;
    if ((p.x_sleep & 0x400) && hset == 1) return;  // (test aid: a team mate that never shows up)
    wpair_agent<WSplitGeo<H>::KA, WSplitGeo<H>::RS, PHASE, NJ, H, WSplitGeo<H>::RSH>(p, pair, dir, hset);
}
template <int NJ, int PHASE, int H>
static int launch_wsplit_phase(const PairLaunch *h, const RunParams &p, hipStream_t st) {
    const size_t lds = wpair_lds_bytes(WSplitGeo<H>::RS, PHASE, h->slotrows, wpair_pc(NJ, H), WSplitGeo<H>::RSH);
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "wide split kernel: LDS");
    auto kernel = mm_fbws_kernel<NJ, PHASE, H>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const unsigned npairs = unsigned((h->B + 1) / 2);
    hipLaunchKernelGGL(kernel, dim3(2 * ((npairs + 7) / 8 * 8 * H)), dim3(64 * (MM_SPLIT_NWC + 2)), lds, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int NJ, int H>
static int launch_wsplit_nj(const PairLaunch *h, const RunParams &p, hipStream_t s0) {
    int rc = launch_wsplit_phase<NJ, 0, H>(h, p, s0);
    if (!rc) rc = launch_wsplit_phase<NJ, 1, H>(h, p, s0);
    if (rc) return rc;
    hipLaunchKernelGGL(mm_dpair_finish_kernel, dim3(unsigned(h->B)), dim3(256), 0, s0, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
// does the batch fit the wide pair kernels?  (up to 250 pdfs; the LDS of phase B with per-pdf sums of two doubles; one workgroup or a team of 2)
bool mm_wpair_fits(const PairLaunch &pl) {
    if (pl.max_P1 > 250) return false;
    const int nj = mm_pair_nj(pl.max_P1);
    if (pl.H == 1) return pl.pair_ka <= MM_PAIR_KA && wpair_lds_bytes(MM_ROW_RS, 1, pl.slotrows, pair_pc(nj)) <= 160 * 1024;
    if (pl.H != 2) return false;  // (teams of 4: the kernels compile with 30 .. 45 spilled registers -- the float64 team kernels keep those graphs)
    return pl.pair_ka <= mm_split_ka(pl.H) && wpair_lds_bytes(mm_split_rs(pl.H), 1, pl.slotrows, wpair_pc(nj, pl.H), mm_split_rsh(pl.H)) <= 160 * 1024;
}
int mm_launch_wpairs(const PairLaunch &pl, const RunParams &p, hipStream_t s0) {
    if (!mm_wpair_fits(pl)) return MM_ERR_UNSUPPORTED;
    const bool two = pl.max_P1 <= 128;
    if (pl.H == 1) return two ? launch_wpairs_nj<2>(&pl, p, s0) : launch_wpairs_nj<4>(&pl, p, s0);
    return two ? launch_wsplit_nj<2, 2>(&pl, p, s0) : launch_wsplit_nj<4, 2>(&pl, p, s0);
}

}  // namespace mm
