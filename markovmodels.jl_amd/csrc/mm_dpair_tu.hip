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
    const size_t lds = pair_lds_bytes(MM_ROW_RS, PHASE, h->slotrows);
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
int mm_launch_dpairs(const PairLaunch &pl, const RunParams &p, hipStream_t s0) {
    if (pl.pair_ka > MM_PAIR_KA || pl.H != 1) return MM_ERR_UNSUPPORTED;
    return pl.max_P1 <= 128 ? launch_dpairs_nj<2>(&pl, p, s0) : launch_dpairs_nj<4>(&pl, p, s0);
}

}  // namespace mm
