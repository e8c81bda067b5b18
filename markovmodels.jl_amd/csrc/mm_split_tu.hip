// mm_split_tu.hip -- translation unit of the SPLIT pair kernels: pair_agent (mm_kernel_pairs.hip) with teams of H = 2 or 4
// workgroups per utterance pair and direction, for FSMs beyond the registers / LDS of one compute unit (the reference's
// WSJ denominator graph, misc/benchmark/den_fsm_wsj.txt; the reference itself has no size limit, src/linalg.jl:170-181).
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_pairs.hip"

namespace mm {

// Workgroup -> (pair, set).  Workgroups b and b + 8 have been seen to share an XCD (its L2): the workgroups of a team are
// 8 apart, so their exchange stays inside one L2 where that holds (speed only; any placement is correct).
// (teams of 2: a vector of 24 KB, up to 3070 states; teams of 4: 32 KB, up to 4094 states, a quarter of the rows each)
template <int H> struct SplitGeo;
template <> struct SplitGeo<2> { static constexpr int RS = MM_SPLIT_RS, RSH = MM_SPLIT_RSH; };
template <> struct SplitGeo<4> { static constexpr int RS = MM_SPLIT4_RS, RSH = MM_SPLIT4_RSH; };
template <int NJ, int PHASE, int DIR, int H>
__global__ void __launch_bounds__(1024) mm_fbs_kernel_dir(RunParams p) {
    const int blk = blockIdx.x;
    const int pair = (blk / (8 * H)) * 8 + (blk & 7), hset = (blk >> 3) % H;
    if (pair >= (p.B + 1) / 2) return;
    pair_agent<MM_SPLIT_KA, SplitGeo<H>::RS, PHASE, DIR, NJ, H, SplitGeo<H>::RSH>(p, pair, hset);
}
template <int NJ, int PHASE, int DIR, int H>
static int launch_split_one(const PairLaunch *h, const RunParams &p, hipStream_t st) {
    const size_t lds = pair_lds_bytes(SplitGeo<H>::RS, PHASE, h->slotrows, SplitGeo<H>::RSH);
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "split kernel: LDS");
    auto kernel = mm_fbs_kernel_dir<NJ, PHASE, DIR, H>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const unsigned npairs = unsigned((h->B + 1) / 2);
    hipLaunchKernelGGL(kernel, dim3((npairs + 7) / 8 * 8 * H), dim3(64 * (MM_SPLIT_NWC + 2)), lds, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int NJ, int H>
static int launch_split_nj(const PairLaunch *h, const RunParams &p, hipStream_t s0) {
    hipStream_t sf = h->side[0], sb = h->side[1];
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s0, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) sf = s0;  // (see mm_pairs_tu.hip)
    HIP_TRY(hipEventRecord(h->ev[0], s0));  // fork
    HIP_TRY(hipStreamWaitEvent(sf, h->ev[0], 0));
    HIP_TRY(hipStreamWaitEvent(sb, h->ev[0], 0));
    auto body = [&]() -> int {
        int rc = launch_split_one<NJ, 0, 0, H>(h, p, sf);
        if (!rc) rc = launch_split_one<NJ, 0, 1, H>(h, p, sb);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(h->ev[1], sf));  // phase B of either direction needs phase A of both
        HIP_TRY(hipEventRecord(h->ev[2], sb));
        HIP_TRY(hipStreamWaitEvent(sf, h->ev[2], 0));
        HIP_TRY(hipStreamWaitEvent(sb, h->ev[1], 0));
        rc = launch_split_one<NJ, 1, 0, H>(h, p, sf);
        if (!rc) rc = launch_split_one<NJ, 1, 1, H>(h, p, sb);
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
    if (pl.H == 4) return pl.max_P1 <= 128 ? launch_split_nj<2, 4>(&pl, p, s0) : launch_split_nj<4, 4>(&pl, p, s0);
    if (pl.H != 2) return mm_fail(MM_ERR_UNSUPPORTED, "split kernel: teams of 2 or 4");
    return pl.max_P1 <= 128 ? launch_split_nj<2, 2>(&pl, p, s0) : launch_split_nj<4, 2>(&pl, p, s0);
}
size_t mm_split_lds_bytes(int H, int phase, int nslotrows) {
    return H == 4 ? pair_lds_bytes(MM_SPLIT4_RS, phase, nslotrows, MM_SPLIT4_RSH) : pair_lds_bytes(MM_SPLIT_RS, phase, nslotrows, MM_SPLIT_RSH);
}

}  // namespace mm
