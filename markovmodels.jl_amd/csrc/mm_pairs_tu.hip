// mm_pairs_tu.hip -- translation unit of the pair kernels (mm_kernel_pairs.hip): their instances and launches.
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_pairs.hip"

namespace mm {

// The pair kernels (mm_kernel_pairs.hip): phase A and phase B, each as a forward-agent and a backward-agent launch
// that run concurrently (the caller's stream and the batch's side stream, joined by events).
// (NJ: 64-lane passes over the pdfs in the service wave, 2 for P + 1 <= 128, else 4)
// (SMALL: graphs of up to 127 states, whose service wave does a sixteenth of the copying and scanning)
template <int NJ, int PHASE, int DIR, bool SMALL>
__global__ void __launch_bounds__(1024) mm_fbp_kernel_dir(RunParams p) {
    pair_agent<MM_PAIR_KA, MM_ROW_RS, PHASE, DIR, NJ, 1, 2 * MM_ROW_RS, SMALL>(p, blockIdx.x);
}
template <int NJ, int PHASE, int DIR, bool SMALL>
static int launch_pair_one(const PairLaunch *h, const RunParams &p, hipStream_t st) {
    const size_t lds = pair_lds_bytes(MM_ROW_RS, PHASE, h->slotrows);
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "pair kernel: LDS");
    auto kernel = mm_fbp_kernel_dir<NJ, PHASE, DIR, SMALL>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const unsigned npairs = unsigned((h->B + 1) / 2);
    hipLaunchKernelGGL(kernel, dim3(npairs), dim3(64 * (h->nwc + 1)), lds, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int NJ, bool SMALL>
static int launch_pairs_ka(const PairLaunch *h, const RunParams &p, hipStream_t s0) {
    hipStream_t sf = h->side[0], sb = h->side[1];
    // (inside a stream capture the forward agents stay on the caller's stream: ending a capture whose origin stream only
    // forks and joins crashed in hipStreamEndCapture -- ROCm 7.0; how the branches of the graph share the queues is the
    // graph executor's business then)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s0, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) sf = s0;
    HIP_TRY(hipEventRecord(h->ev[0], s0));  // fork
    HIP_TRY(hipStreamWaitEvent(sf, h->ev[0], 0));
    HIP_TRY(hipStreamWaitEvent(sb, h->ev[0], 0));
    // (from here on every path joins the side streams back into s0, also a failed launch: later work on s0 must stay
    // ordered behind what the side streams already hold, and a capture must not be left with an open fork)
    auto body = [&]() -> int {
        int rc = launch_pair_one<NJ, 0, 0, SMALL>(h, p, sf);
        if (!rc) rc = launch_pair_one<NJ, 0, 1, SMALL>(h, p, sb);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(h->ev[1], sf));  // phase B of either direction needs phase A of both
        HIP_TRY(hipEventRecord(h->ev[2], sb));
        HIP_TRY(hipStreamWaitEvent(sf, h->ev[2], 0));
        HIP_TRY(hipStreamWaitEvent(sb, h->ev[1], 0));
        rc = launch_pair_one<NJ, 1, 0, SMALL>(h, p, sf);
        if (!rc) rc = launch_pair_one<NJ, 1, 1, SMALL>(h, p, sb);
        return rc;
    };
    const int rc = body();
    HIP_TRY(hipEventRecord(h->ev[3], sf));  // join
    HIP_TRY(hipEventRecord(h->ev[4], sb));
    HIP_TRY(hipStreamWaitEvent(s0, h->ev[3], 0));
    HIP_TRY(hipStreamWaitEvent(s0, h->ev[4], 0));
    if (rc) return rc;
    hipLaunchKernelGGL(mm_pair_finish_kernel, dim3(unsigned(h->B)), dim3(256), 0, s0, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
int mm_launch_pairs(const PairLaunch &pl, const RunParams &p, hipStream_t s0) {
    if (pl.pair_ka > MM_PAIR_KA) return MM_ERR_UNSUPPORTED;
    if (pl.max_P1 <= 128) return pl.small ? launch_pairs_ka<2, true>(&pl, p, s0) : launch_pairs_ka<2, false>(&pl, p, s0);
    return launch_pairs_ka<4, false>(&pl, p, s0);
}
size_t mm_pair_lds_bytes(int phase, int nslotrows) { return pair_lds_bytes(MM_ROW_RS, phase, nslotrows); }
size_t mm_pair_hand_bytes() { return sizeof(PairHand); }

}  // namespace mm
