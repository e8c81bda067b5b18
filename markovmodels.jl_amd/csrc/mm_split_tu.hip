// mm_split_tu.hip -- translation unit of the SPLIT pair kernels: pair_agent (mm_kernel_pairs.hip) with teams of H = 2
// workgroups per utterance pair and direction, for FSMs beyond the registers / LDS of one compute unit (the reference's
// WSJ denominator graph, misc/benchmark/den_fsm_wsj.txt; the reference itself has no size limit, src/linalg.jl:170-181).
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_pairs.hip"

namespace mm {

// Workgroup -> (pair, set).  Workgroups b and b + 8 have been seen to share an XCD (its L2): the workgroups of a team are
// 8 apart, so their exchange stays inside one L2 where that holds (speed only; any placement is correct).
template <int NJ, int PHASE, int DIR, int H>
__global__ void __launch_bounds__(1024) mm_fbs_kernel_dir(RunParams p) {
    const int blk = blockIdx.x;
    const int pair = (blk / (8 * H)) * 8 + (blk & 7), hset = (blk >> 3) % H;
    if (pair >= (p.B + 1) / 2) return;
    pair_agent<MM_SPLIT_KA, MM_SPLIT_RS, PHASE, DIR, NJ, H, MM_SPLIT_RSH>(p, pair, hset);
}
template <int NJ, int PHASE, int DIR>
static int launch_split_one(const PairLaunch *h, const RunParams &p, hipStream_t st) {
    const size_t lds = pair_lds_bytes(MM_SPLIT_RS, PHASE, h->slotrows, MM_SPLIT_RSH);
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "split kernel: LDS");
    if (h->H != 2) return mm_fail(MM_ERR_UNSUPPORTED, "split kernel: teams of 2 only");
    auto kernel = mm_fbs_kernel_dir<NJ, PHASE, DIR, 2>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const unsigned npairs = unsigned((h->B + 1) / 2);
    hipLaunchKernelGGL(kernel, dim3((npairs + 7) / 8 * 8 * 2), dim3(64 * (MM_SPLIT_NWC + 2)), lds, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int NJ>
static int launch_split_nj(const PairLaunch *h, const RunParams &p, hipStream_t s0) {
    hipStream_t sf = h->side[0], sb = h->side[1];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s0, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) sf = s0;  // (see mm_pairs_tu.hip)
    HIP_TRY(hipEventRecord(h->ev[0], s0));  // fork
    HIP_TRY(hipStreamWaitEvent(sf, h->ev[0], 0));
    HIP_TRY(hipStreamWaitEvent(sb, h->ev[0], 0));
    auto body = [&]() -> int {
        int rc = launch_split_one<NJ, 0, 0>(h, p, sf);
        if (!rc) rc = launch_split_one<NJ, 0, 1>(h, p, sb);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(h->ev[1], sf));  // phase B of either direction needs phase A of both
        HIP_TRY(hipEventRecord(h->ev[2], sb));
        HIP_TRY(hipStreamWaitEvent(sf, h->ev[2], 0));
        HIP_TRY(hipStreamWaitEvent(sb, h->ev[1], 0));
        rc = launch_split_one<NJ, 1, 0>(h, p, sf);
        if (!rc) rc = launch_split_one<NJ, 1, 1>(h, p, sb);
        return rc;
    };
    const int rc = body();
    HIP_TRY(hipEventRecord(h->ev[3], sf));  // join (also after a failed launch)
    HIP_TRY(hipEventRecord(h->ev[4], sb));
    HIP_TRY(hipStreamWaitEvent(s0, h->ev[3], 0));
    HIP_TRY(hipStreamWaitEvent(s0, h->ev[4], 0));
    if (rc) return rc;
    hipLaunchKernelGGL(mm_pair_finish_kernel, dim3(unsigned(h->B)), dim3(256), 0, s0, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
int mm_launch_split(const PairLaunch &pl, const RunParams &p, hipStream_t s0) {
    if (pl.pair_ka > MM_SPLIT_KA) return MM_ERR_UNSUPPORTED;
    return pl.max_P1 <= 128 ? launch_split_nj<2>(&pl, p, s0) : launch_split_nj<4>(&pl, p, s0);
}
size_t mm_split_lds_bytes(int phase, int nslotrows) { return pair_lds_bytes(MM_SPLIT_RS, phase, nslotrows, MM_SPLIT_RSH); }

}  // namespace mm
