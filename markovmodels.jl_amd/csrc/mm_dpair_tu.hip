// mm_dpair_tu.hip -- translation unit of the float64 exact pair kernels (mm_kernel_dpair.hip): their instances and launches.
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_dpair.hip"

namespace mm {

// One utterance per workgroup: blockIdx.x = the utterance's rank in the longest-first order.
// (NJ: 64-lane passes over the pdfs in the service wave, 2 for P + 1 <= 128, else 4)
template <int NJ, int PHASE, int DIR>
__global__ void __launch_bounds__(1024) mm_fbd_kernel_dir(RunParams p) {
    dpair_agent<MM_PAIR_KA, MM_ROW_RS, PHASE, DIR, NJ>(p, blockIdx.x);
}
template <int NJ, int PHASE, int DIR>
static int launch_dpair_one(const PairLaunch *h, const RunParams &p, hipStream_t st) {
    const size_t lds = pair_lds_bytes(MM_ROW_RS, PHASE, h->slotrows);
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "exact pair kernel: LDS");
    auto kernel = mm_fbd_kernel_dir<NJ, PHASE, DIR>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(kernel, dim3(unsigned(h->B)), dim3(64 * (h->nwc + 1)), lds, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
// the same fork / join as mm_launch_pairs (mm_pairs_tu.hip): phase A of both agents side by side, then phase B of both
template <int NJ>
static int launch_dpairs_nj(const PairLaunch *h, const RunParams &p, hipStream_t s0) {
    hipStream_t sf = h->side[0], sb = h->side[1];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s0, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) sf = s0;
    HIP_TRY(hipEventRecord(h->ev[0], s0));  // fork
    HIP_TRY(hipStreamWaitEvent(sf, h->ev[0], 0));
    HIP_TRY(hipStreamWaitEvent(sb, h->ev[0], 0));
    auto body = [&]() -> int {
        int rc = launch_dpair_one<NJ, 0, 0>(h, p, sf);
        if (!rc) rc = launch_dpair_one<NJ, 0, 1>(h, p, sb);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(h->ev[1], sf));  // phase B of either direction needs phase A of both
        HIP_TRY(hipEventRecord(h->ev[2], sb));
        HIP_TRY(hipStreamWaitEvent(sf, h->ev[2], 0));
        HIP_TRY(hipStreamWaitEvent(sb, h->ev[1], 0));
        rc = launch_dpair_one<NJ, 1, 0>(h, p, sf);
        if (!rc) rc = launch_dpair_one<NJ, 1, 1>(h, p, sb);
        return rc;
    };
    const int rc = body();
    HIP_TRY(hipEventRecord(h->ev[3], sf));  // join (also after a failed launch: see mm_pairs_tu.hip)
    HIP_TRY(hipEventRecord(h->ev[4], sb));
    HIP_TRY(hipStreamWaitEvent(s0, h->ev[3], 0));
    HIP_TRY(hipStreamWaitEvent(s0, h->ev[4], 0));
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
