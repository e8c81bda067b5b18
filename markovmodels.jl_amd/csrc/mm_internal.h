// mm_internal.h -- what the translation units of the engine share on the host side (not part of the C ABI).
// mm_engine.hip holds the C ABI, the handles and the item / quad / row kernels; kernel families with many template
// instances live in translation units of their own (compiled in parallel, see Makefile) behind plain launch functions.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/markovmodels_amd.h"

namespace mm {

struct RunParams;  // mm_kernels.hip

// records the message mm_last_error() returns on this thread; returns `code`
int mm_fail(int code, const std::string &msg);

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return ::mm::mm_fail(MM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

#define MM_ROW_RS 8192  // LDS bytes of one copy of the linear vector (row kernels) / half a pair vector (pair kernels)
#define MM_PAIR_KA 44   // arc slots per lane of the pair kernels
// pdfs (+ 1) of the pair / split pair / float64 pair kernels: passes of 64 lanes of their service waves (the NJ of the instances),
// and the floats of one slot of per-pdf partial sums a team publishes (PairLay::XPS, mm_kernel_pairs.hip)
#define MM_PAIR_P1MAX 506
#define MM_PAIR_SPLIT_Q10 512  // (mm_engine.hip: RunParams::split_q10 of the pair kernels)
// (teams of 8 have no LDS for the arrays of 512 pdfs: 5 passes, 314 pdfs)
inline int mm_pair_nj(int P1, int H = 1) { return P1 <= 128 ? 2 : (P1 <= 250 ? 4 : (H == 8 ? (P1 <= 314 ? 5 : 0) : (P1 <= MM_PAIR_P1MAX ? 8 : 0))); }
inline int mm_pair_xps(int P1, int H = 1) { const int nj = mm_pair_nj(P1, H); return nj <= 4 ? 512 : 128 * nj; }
// split pair kernels (teams of H workgroups per utterance pair and direction): bytes of half a pair vector, arc slots per
// lane, compute waves (+ a service wave and an exchange wave)
#define MM_SPLIT_RS 12288
#define MM_SPLIT_RSH 13312  // LDS bytes of the rows ONE workgroup of a team finishes (the sets are balanced by arcs: their row
                            // counts differ by a few per cent; 1663 rows leave 51 slot rows of LDS in phase B)
// teams of 4: the team's vector of pairs is 32 KB (up to 4094 states), a workgroup finishes a quarter of the rows
#define MM_SPLIT4_RS 16384
#define MM_SPLIT4_RSH 9216
// teams of 8: 47.5 KB (up to ~6050 states), an eighth of the rows each (a multiple of 1 KB: the partner rows come in 1 KB DMAs);
// what the LDS of a compute unit holds next to two such vectors
#define MM_SPLIT8_RS 24320
#define MM_SPLIT8_RSH 6144
inline int mm_split_rs(int H) { return H == 8 ? MM_SPLIT8_RS : (H == 4 ? MM_SPLIT4_RS : MM_SPLIT_RS); }
inline int mm_split_ka(int H) { return H == 8 ? 36 : 36; }  // arc slots per lane (teams of 8: fewer arcs per workgroup, and more registers to the exchange)
inline int mm_split_rsh(int H) { return H == 8 ? MM_SPLIT8_RSH : (H == 4 ? MM_SPLIT4_RSH : MM_SPLIT_RSH); }
#define MM_SPLIT_KA 36
#define MM_SPLIT_NWC 14

// ---- generic path (mm_generic.hip): any semiring, float32 or float64, any C_hat / V_hat
struct FsmGenView {   // host copies of one FSM in double, natural units; [0]: CSR of T_hat' (forward), [1]: of T_hat (backward)
    int semiring = 0;
    int64_t S1 = 0;
    int32_t P1 = 0;
    const int64_t *ptr[2] = {nullptr, nullptr};
    const int32_t *col[2] = {nullptr, nullptr};
    const double *val[2] = {nullptr, nullptr};
    const double *init = nullptr;   // dense alpha_hat [S1]
    const int32_t *s2p = nullptr;   // the FSM's one-hot state map [S1]
    void *dev[2] = {nullptr, nullptr};  // device copies made by the generic path (float32, float64); freed by mm_generic_free
};
FsmGenView *mm_fsm_gen_view(mm_fsm_t f);
// what the generic entry keeps with the batch between calls (owned and freed by the batch): the alpha / beta workspace,
// the utterance descriptors on the device and the host image they were uploaded from (a call with the same FSMs, maps
// and float type uploads nothing)
struct GenScratch {
    void *ws = nullptr;
    size_t ws_bytes = 0;
    void *d_utts = nullptr;
    size_t utts_bytes = 0;
    std::vector<char> host;  // what d_utts holds
    std::string last_kernels;  // what the last call of the generic entry launched (mm_batch_kernels, entry 2)
};
GenScratch *mm_batch_gen_scratch(mm_batch_t h);
int mm_batch_gen_view(mm_batch_t h, int64_t *B, const mm_fsm_t **fsms, int *semiring, int *device);
void mm_generic_free(void *dev);

// ---- pair kernels (mm_pairs_tu.hip)
struct PairLaunch {
    int64_t B = 0;
    int nwc = 1, slotrows = 0, max_P1 = 0, pair_ka = 0, H = 1;
    bool small = false;  // every FSM has at most 127 states: the instance whose service wave copies and scans one row of 64 float4
};
int mm_launch_pairs(const PairLaunch &pl, const RunParams &p, hipStream_t s0);
// alpha / beta export on the pair kernels (phase A of one direction over all frames + a layout pass): dir 0 alpha, 1 beta
bool mm_pair_export_fits(const PairLaunch &pl);
int mm_launch_pair_export(const PairLaunch &pl, const RunParams &p, int dir, hipStream_t s0);
size_t mm_pair_lds_bytes(int phase, int nslotrows, int max_P1);
size_t mm_pair_hand_bytes();
// ---- the float64 exact pair kernels (mm_dpair_tu.hip): one utterance per workgroup, for the utterances marked in p.redo
int mm_launch_dpairs(const PairLaunch &pl, const RunParams &p, hipStream_t s0);
// ---- the wide-exponent pair kernels (mm_wpair_tu.hip): two utterances per workgroup with a float64's range, whole batches only
// (every utterance marked in p.redo): what a call that goes to the exact kernels FIRST runs when the batch fits
bool mm_wpair_fits(const PairLaunch &pl);
int mm_launch_wpairs(const PairLaunch &pl, const RunParams &p, hipStream_t s0);
// ---- split pair kernels (mm_split_tu.hip): teams of pl.H workgroups
int mm_launch_split(const PairLaunch &pl, const RunParams &p, hipStream_t s0);
size_t mm_split_lds_bytes(int H, int phase, int nslotrows, int max_P1);
bool mm_split_export_fits(const PairLaunch &pl);  // alpha / beta export on the team kernels: teams of 2 and 4, up to 128 pdfs
int mm_launch_split_export(const PairLaunch &pl, const RunParams &p, int dir, hipStream_t s0);  // (0: no instance for that many pdfs)


// ---- wave kernel (mm_wave_tu.hip)
#define MM_WAVE_RS 4352  // LDS bytes of one state vector of the wave kernel
#define MM_WAVE_WAVES 4  // waves per agent (direction) of the wave kernel
struct WaveLaunch {
    int64_t B = 0;
    int nseg = 0, max_P1 = 0, n_cus = 256;
};
int mm_launch_wave(const WaveLaunch &wl, const RunParams &p, hipStream_t stream);

// ---- lane kernel (mm_lane_tu.hip): graphs of up to 64 states and 64 pdfs, one wave per direction
size_t mm_lane_dev_bytes();
void mm_lane_dev_fill(void *dst, const double *w0, const double *w1, const float *init, const float *fin, const int *s2p, const int *pdf_ptr,
                      const int *pdf_states, int S, int P, int ident);
int mm_launch_lane(int64_t B, int max_S, const RunParams &p, hipStream_t stream);

// ---- stream kernels (mm_stream.hip): graphs beyond every register-resident form -- the arcs streamed from L2 as 8-byte records,
// the vector in LDS as wide-exponent 32-bit values, one utterance per workgroup, forward launch then backward launch
struct StreamForm;
size_t mm_stream_lds_bytes(int S1, int P1);
// (H: workgroups of a team per utterance and direction -- 1, 2 or 4; the rows of a direction dealt to H sets)
int mm_stream_build(int64_t S1, int32_t P1, const int64_t *const rowptr[2], const int32_t *const col[2], const float *const val[2],
                    const float *init, const int32_t *s2p, bool upload, int H, StreamForm **out);  // *out NULL: does not fit
void mm_stream_free(StreamForm *f);
const void *mm_stream_dev(const StreamForm *f);
void mm_stream_eval(const StreamForm *f, int d, const float *in, float *out, double stats[4]);
int mm_stream_pick_h(int64_t B, int n_cus);
size_t mm_stream_slot(int max_S1);
size_t mm_stream_extra_bytes(int64_t B, int64_t total_s1p, int64_t N, size_t off[4], int H, int max_S1);
size_t mm_stream_exchange_bytes(int64_t B, int H, int max_S1);
int mm_launch_stream(int64_t B, int n_cus, int max_S1, int max_P1, int H, const RunParams &p, hipStream_t st);

// ---- Viterbi on the row-lane form (mm_vit_tu.hip)
struct VitLaunch {
    int64_t B = 0;
    int n4 = 0, n2 = 0, max_P1 = 0, max_S1p = 0, max_arcs = 0, bp_row = 0;  // n4 x n2: wide / narrow positions per wave
};
int mm_launch_viterbi(const VitLaunch &vl, const RunParams &p, hipStream_t stream);

// ---- quad kernels (mm_quad_tu.hip)
struct QuadLaunch {
    int64_t B = 0;
    int kq = 0, nw = 1, max_S1p = 0;
    size_t lds = 0;
};
int mm_launch_quad_pass(int pass, const QuadLaunch &ql, const RunParams &p, hipStream_t stream);

}  // namespace mm
