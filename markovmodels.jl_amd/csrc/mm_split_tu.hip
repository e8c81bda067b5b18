// mm_split_tu.hip -- translation unit of the SPLIT pair kernels: pair_agent (mm_kernel_pairs.hip) with teams of H = 2 or 4
// workgroups per utterance pair and direction, for FSMs beyond the registers / LDS of one compute unit (the reference's
// WSJ denominator graph, misc/benchmark/den_fsm_wsj.txt; the reference itself has no size limit, src/linalg.jl:170-181).
#define MM_SECONDARY_TU
#include <algorithm>
#include "mm_internal.h"
#include "mm_kernel_pairs.hip"

namespace mm {

// Workgroup -> (direction, pair, set).  Workgroups b and b + 8 have been seen to share an XCD (its L2): the workgroups of a team are
// 8 apart, so their exchange stays inside one L2 where that holds (speed only; any placement is correct).
// (teams of 2: a vector of 24 KB, up to 3070 states; teams of 4: 32 KB, up to 4094 states, a quarter of the rows each)
template <int H> struct SplitGeo;
template <> struct SplitGeo<2> { static constexpr int RS = MM_SPLIT_RS, RSH = MM_SPLIT_RSH, KA = 36; };
template <> struct SplitGeo<4> { static constexpr int RS = MM_SPLIT4_RS, RSH = MM_SPLIT4_RSH, KA = 36; };
template <> struct SplitGeo<8> { static constexpr int RS = MM_SPLIT8_RS, RSH = MM_SPLIT8_RSH, KA = 36; };
// One launch per phase: the teams of the forward agents are the first half of the grid, those of the backward agents the
// second (mm_pairs_tu.hip; a half is a multiple of 8 H workgroups: the block -> XCD pattern is the same in both).
template <int NJ, int PHASE, int H>
__global__ void __launch_bounds__(1024) mm_fbs_kernel(RunParams p) {
    const int half = (int)gridDim.x / 2, dir = (int)blockIdx.x >= half;
    const int blk = (int)blockIdx.x - (dir ? half : 0);
    const int pair = (blk / (8 * H)) * 8 + (blk & 7), hset = (blk >> 3) % H;
    if (pair >= (p.B + 1) / 2) return;
    if ((p.x_sleep & 0x400) && hset == 1) return;  // (test aid, MM_SPLIT_SLEEP bit 0x400: a team mate that never shows up)
    pair_agent<SplitGeo<H>::KA, SplitGeo<H>::RS, PHASE, -1, NJ, H, SplitGeo<H>::RSH>(p, pair, hset, dir);
}
template <int NJ, int PHASE, int H>
static int launch_split_phase(const PairLaunch *h, const RunParams &p, hipStream_t st) {
    const size_t lds = pair_lds_bytes(SplitGeo<H>::RS, PHASE, h->slotrows, SplitGeo<H>::RSH, pair_pc(NJ));
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "split kernel: LDS");
    auto kernel = mm_fbs_kernel<NJ, PHASE, H>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const unsigned npairs = unsigned((h->B + 1) / 2);
    hipLaunchKernelGGL(kernel, dim3(2 * ((npairs + 7) / 8 * 8 * H)), dim3(64 * (MM_SPLIT_NWC + 2)), lds, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int NJ, int H>
static int launch_split_nj(const PairLaunch *h, const RunParams &p, hipStream_t s0) {
    int rc = launch_split_phase<NJ, 0, H>(h, p, s0);
    if (!rc) rc = launch_split_phase<NJ, 1, H>(h, p, s0);
    if (rc) return rc;
    hipLaunchKernelGGL(mm_pair_finish_kernel, dim3(unsigned(h->B)), dim3(256), 0, s0, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
int mm_launch_split(const PairLaunch &pl, const RunParams &p, hipStream_t s0) {
    if (pl.pair_ka > mm_split_ka(pl.H)) return MM_ERR_UNSUPPORTED;
    const int nj = mm_pair_nj(pl.max_P1, pl.H);
    if (nj == 0) return MM_ERR_UNSUPPORTED;
    if (pl.H == 8) return nj == 2 ? launch_split_nj<2, 8>(&pl, p, s0) : (nj == 4 ? launch_split_nj<4, 8>(&pl, p, s0) : launch_split_nj<5, 8>(&pl, p, s0));
    if (pl.H == 4) return nj == 2 ? launch_split_nj<2, 4>(&pl, p, s0) : (nj == 4 ? launch_split_nj<4, 4>(&pl, p, s0) : launch_split_nj<8, 4>(&pl, p, s0));
    if (pl.H != 2) return mm_fail(MM_ERR_UNSUPPORTED, "split kernel: teams of 2, 4 or 8");
    return nj == 2 ? launch_split_nj<2, 2>(&pl, p, s0) : (nj == 4 ? launch_split_nj<4, 2>(&pl, p, s0) : launch_split_nj<8, 2>(&pl, p, s0));
}
// ---- alpha-recursion / beta-recursion export on the team kernels (mm_pairs_tu.hip, mm_fbx_kernel): phase A of ONE direction over all
// N + 1 frames by teams of H workgroups, then mm_pair_export_kernel.  Teams of 2 and 4, up to 128 pdfs (the reference's WSJ denominator).
template <int NJ, int H>
__global__ void __launch_bounds__(1024) mm_fbsx_kernel(RunParams p, int dir) {
    const int blk = (int)blockIdx.x;
    const int pair = (blk / (8 * H)) * 8 + (blk & 7), hset = (blk >> 3) % H;
    if (pair >= (p.B + 1) / 2) return;
    pair_agent<SplitGeo<H>::KA, SplitGeo<H>::RS, 0, -1, NJ, H, SplitGeo<H>::RSH, false, true>(p, pair, hset, dir);
}
template <int NJ, int H>
static int launch_split_export(const PairLaunch *h, const RunParams &p, int dir, hipStream_t st) {
    const size_t lds = pair_lds_bytes(SplitGeo<H>::RS, 0, h->slotrows, SplitGeo<H>::RSH, pair_pc(NJ));
    if (lds > 160 * 1024) return mm_fail(MM_ERR_UNSUPPORTED, "split kernel: LDS");
    auto kernel = mm_fbsx_kernel<NJ, H>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    const unsigned npairs = unsigned((h->B + 1) / 2);
    hipLaunchKernelGGL(kernel, dim3((npairs + 7) / 8 * 8 * H), dim3(64 * (MM_SPLIT_NWC + 2)), lds, st, p, dir);
    HIP_TRY(hipGetLastError());
    const int chunks = std::max(1, std::min(p.N + 1, int(4096 / std::max(1u, npairs))));
    const int fpb = (p.N + 1 + chunks - 1) / chunks;
    hipLaunchKernelGGL(mm_pair_export_kernel, dim3(npairs, unsigned((p.N + 1 + fpb - 1) / fpb)), dim3(1024), size_t(3) * size_t(p.pair_s1p) * 4, st, p, dir, fpb, H);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
bool mm_split_export_fits(const PairLaunch &pl) {
    return (pl.H == 2 || pl.H == 4) && pl.pair_ka <= mm_split_ka(pl.H) && mm_pair_nj(pl.max_P1, pl.H) == 2;
}
int mm_launch_split_export(const PairLaunch &pl, const RunParams &p, int dir, hipStream_t s0) {
    if (!mm_split_export_fits(pl)) return MM_ERR_UNSUPPORTED;
    return pl.H == 4 ? launch_split_export<2, 4>(&pl, p, dir, s0) : launch_split_export<2, 2>(&pl, p, dir, s0);
}
size_t mm_split_lds_bytes(int H, int phase, int nslotrows, int max_P1) {
    const int nj = mm_pair_nj(max_P1, H);
    return nj ? pair_lds_bytes(mm_split_rs(H), phase, nslotrows, mm_split_rsh(H), pair_pc(nj)) : 0;
}

}  // namespace mm
