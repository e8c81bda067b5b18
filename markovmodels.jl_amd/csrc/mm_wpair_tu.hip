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
// does the batch fit the wide pair kernels?  (one workgroup per pair: H = 1; up to 250 pdfs; the LDS of phase B with per-pdf sums
// of two doubles)
bool mm_wpair_fits(const PairLaunch &pl) {
    if (pl.H != 1 || pl.pair_ka > MM_PAIR_KA || pl.max_P1 > 250) return false;
    return wpair_lds_bytes(MM_ROW_RS, 1, pl.slotrows, pair_pc(mm_pair_nj(pl.max_P1))) <= 160 * 1024;
}
int mm_launch_wpairs(const PairLaunch &pl, const RunParams &p, hipStream_t s0) {
    if (!mm_wpair_fits(pl)) return MM_ERR_UNSUPPORTED;
    return pl.max_P1 <= 128 ? launch_wpairs_nj<2>(&pl, p, s0) : launch_wpairs_nj<4>(&pl, p, s0);
}

}  // namespace mm
