// mm_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the forward-backward /
// Viterbi engine.  Included by mm_engine.hip (single translation unit).
//
// What these kernels replace in the reference (paths under MarkovModels.jl):
//   per frame and direction  _cukernel_mul_smdv!   src/linalg.jl:213-233  (warp-per-row SpMV)
//                            elementwise (*)       src/inference.jl:71,106
//   once per call            _cukernel_mul_smdm!   src/linalg.jl:268-280  (C*V gather, C'*AB reduce)
//                            _cukernel_bc_svdv!    src/linalg.jl:320-328  (alpha_hat (*) lhs[:,1])
//                            A .* B, permutedims, sum, ./, minimum, exp   src/inference.jl:154-160
// i.e. 4 launches per frame + ~10 per call become ONE persistent launch: one
// workgroup per utterance walks all frames of the alpha-recursion
// (src/inference.jl:62-74) and then the beta-recursion (:99-110) fused with the
// posterior combine (:154-160).  beta, alpha.*beta and the state-level
// emissions C*V are never materialised in HBM; only the (normalised) alpha
// store is written once and read once.
//
// Numerics (log semiring): values live in the log2 domain, normalised per
// frame by a lagged maximum (alpha~_n = alpha_n - C_n, C_n kept in double), so
// float32 magnitudes stay O(1) instead of growing to O(N) as in the reference.
// (+) over a CSR row is a two-pass log-sum-exp: lane-local max -> lane-group
// max (cross-lane) -> sum of exp2 -> lane-group sum -> log2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mm_pack.h"

namespace mm {

#define MM_NINF (-__builtin_inff())
#define MM_LOG2E 1.4426950408889634f
#define MM_LN2 0.6931471805599453f
#define MM_MAX_WAVES 16
#define MM_SPLIT_HMAX 8  // workgroups of a team (split pair kernels)

enum { MODE_FB = 0, MODE_ALPHA = 1, MODE_BETA = 2 };

struct GraphDev {
    const ItemMeta *items;
    const RowInfo *rowinfo;
    const Slot *slots;
    int n_items;
    int n_short;  // items with <= 4 arcs per lane; they come first and can be register resident
};

struct QuadDev {  // one direction in quad form, internal numbering (mm_pack.h QuadGraph)
    const Quad *quads;
    const RowRec *recs;
    const int *rowptr;  // CSR with log2 weights for the exact fallback
    const int *col;
    const float *w;
    const unsigned short *pdfse;  // [2 * P1] (first, end) internal positions of each pdf (backward only)
    const unsigned short *dist;   // [S1] fewest arcs from an initial state (forward) / to the final state (backward)
    int nq;
    int fpos;   // internal position of the phony final state
    int ncopy;  // LDS copies of the linear vector the quad offsets refer to (mm_pack.h quad_pstride)
    int pad;
};

struct RowSched;
struct RowDev {  // one direction in row-lane form, internal numbering (mm_rows.h RowGraph)
    const float *w;               // [KA][64 * NWC] linear weights
    const unsigned *addr;         // [KA][64 * NWC] LDS byte addresses of the sources
    const unsigned *slots;        // [nslotrows][64] x 1 (forward) / 2 (backward) words
    const RowSched *sched;        // [NWC]
    const int *rowptr;            // CSR with log2 weights for the exact fallback
    const int *col;
    const float *cw;
    const unsigned short *rowpdf; // [rows] pdf of the row at each position
    const unsigned short *pdfse;  // [2 * P1] (first, end) of each pdf in pdf-major order (backward only)
    const float *init;            // [rows] alpha_hat by position, log2 domain (forward only)
    const int *order;             // [rows] position -> original state (Viterbi form: the path is reported in original states)
    const unsigned *ptab;         // wave form: the per-pdf sums as packed segments (mm_engine.hip wave_pdf_table), else NULL
    int KA, NWC, nslotrows, fpos, rows;
    float thr;  // |normalised log2 value| beyond which the linear path is not trusted (mm_kernel_rows.hip)
};

struct UttDesc {
    GraphDev g[2];  // 0: T_hat' packed (forward), 1: T_hat packed (backward)
    QuadDev q[2];   // same two matrices in quad form (log semiring only)
    RowDev r[2];    // ... and in row-lane form (log semiring only; KA == 0: not available)
    RowDev rp[2];   // ... and in the pair variant of the row-lane form (mm_rows.h RowPackOpts::pair)
    RowDev rv;      // the forward matrix in the Viterbi form (mm_kernel_vit.hip; tropical semiring)
    RowDev rw[2];   // ... and in the wave form (mm_kernel_wave.hip: one wave per direction, log domain)
    RowDev rps[2][MM_SPLIT_HMAX];  // ... and the split pair forms [direction][set] (mm_rows.h make_rows_split): rowpdf / init /
                                   // rows / fpos refer to the TEAM's vector (all sets; rowpdf 0xffff = alignment padding)
    const float *init_f;            // alpha_hat in forward numbering
    const unsigned short *map_bf;   // backward position -> forward position
    const float *init;  // dense alpha_hat [S1] (log2 domain for MM_LOG, natural for MM_TROPICAL)
    const int *s2p;     // state -> pdf [S1]
    const int *pdf_ptr, *pdf_rows;  // pdf -> states, CSR [P1 + 1], [S1]
    const struct LaneDev *lane;     // the lane form (mm_kernel_lane.hip: graphs of up to 64 states), else NULL
    const void *stream;             // the stream form (mm_stream.hip: graphs beyond the register-resident forms; StreamPairDev), else NULL
    int S1, S1p, P1, pad;
    long long state_off;   // offset of this FSM in the block-diagonal state space
    long long s1p_prefix;  // sum of S1p of the utterances before this one
};

struct RunParams {
    const UttDesc *utts;
    const float *V;
    long long vsb, vsn;
    const int *lens;
    int N, B;
    float *ws_alpha;  // [sum_b S1p_b][N+1] normalised alpha store (per utterance: [N+1][S1p])
    double *ws_c;     // [B][N+2] per-frame log2 offsets C_n
    float *gamma;
    long long gsb, gsn, gsp;
    float *ttl;
    float *out;  // alpha / beta export
    long long out_stride_n;
    int *bp;  // viterbi
    long long bp_stride_n;
    int stop_at_len;  // viterbi without an export of the back-pointers: frames beyond len_b + 1 are not computed
    // FSMs whose state vectors do not fit the LDS (item / tropical kernels, BIGV): [B][big_stride] floats of global memory
    float *ws_big;
    long long big_stride;
    int deterministic;  // item kernel: per-pdf sums over the pdf -> states lists in a fixed order instead of LDS float atomics
    int *path;
    long long path_stride_b;
    float *score;
    unsigned long long *dbg;  // diagnostic builds only (-DMM_STAMPS): per-wave cycle sums per phase
    // emission-free recursion of the total-sum family (src/algorithms.jl:8-29): every real pdf emits one(K)
    // in every frame; the phony pdf emits zero(K) up to frame N (1: totalsum -- the final state then holds
    // omega . v_N at frame N+1) or never (2: totalcumsum -- the final state accumulates omega . v_k)
    int free_run;
    int xcsr;  // quad kernel: floats of LDS holding the exact-fallback CSR (0: walk it in global memory)
    // workgroup -> utterance, longest first (NULL: identity).  With more utterances than CUs the workgroups
    // are handed out in this order, so the long utterances start first and the short ones fill the tail.
    const int *order;
    // Row kernels: redo[b] != 0 marks an utterance whose linear-domain sums left the trusted range; the exact kernels
    // launched after them skip every utterance that is not marked (NULL: run all).
    int *redo;
    // The float64 exact pair kernels (mm_kernel_dpair.hip) run the utterances marked in `redo` and mark in `redo2` the ones
    // whose values left the DOUBLE's range; mm_dpair_finish_kernel then clears or keeps redo[b] for the log-domain kernels.
    int *redo2;
    // How hard the inputs of this call were for the float32 kernels, for the engine's choice at the NEXT calls (no
    // synchronisation: the host reads whatever the last finished call left): stat_dev = {count, ticket} device counters,
    // stat_host = {count, stat_seq} in pinned host memory, written by the last workgroup of the finish kernel that reports --
    // stat_mode 0: mm_pair_finish_kernel (utterances still marked after its decision), 1: mm_dpair_finish_kernel on a call
    // that skipped the float32 kernels (utterances whose smallest overlap term is below the float32 kernels' floor).
    int *stat_dev;
    int clear_marks;  // mm_pair_finish_kernel: 1 = a range mark is cleared when the two criteria hold; 0 = it stays (see there)
    volatile int *stat_host;
    int stat_seq, stat_mode;
    int stat_xcd;  // != 0: the team kernels count their same-XCD workgroups in stat_dev[2..3] (on once mm_batch_team_xcd_stats has been called)
    // Pair kernels (mm_kernel_pairs.hip): ws_alpha holds [B + 1][N + 2][pair_s1p] state vectors, ws_c [B + 1][N + 2]
    // cumulative offsets; pair_hand [pairs][2 directions][2 utterances] what phase A hands to phase B; pair_zmin
    // [B][2 directions] the minimum over the frames of the per-frame log2 normaliser.
    int pair_s1p;
    void *pair_hand;
    double *pair_zmin;
    // Split pair kernels (teams of H workgroups per utterance pair and direction): set h computes sp_cnt[h] rows, at
    // positions sp_base[h].. of the team's vector.  xbuf: what the workgroups of a team send each other --
    // [phase][pair][direction][set][2 slots][x_slot floats] linear values of the set's rows of a step, 8-byte granules
    // {utterance 0, utterance 1} tagged in the sign bits; xps: [pair][direction][set][4 slots][512 floats] per-pdf partial
    // sums (phase B).  Both are zeroed before every call (a zero granule = not yet arrived).
    int sp_base[MM_SPLIT_HMAX], sp_cnt[MM_SPLIT_HMAX];
    float *xbuf, *xps;
    long long x_slot, x_phase;  // floats per slot; floats of one phase's area
    // ... and the same for the teams of the float64 kernels (mm_kernel_dpair.hip: one utterance per team, so B "pairs"; a granule
    // is one tagged double): [phase][utterance rank][direction][set][2 slots][x_slot floats], [rank][2][H][4 slots][512 floats]
    float *xbuf_d, *xps_d;
    long long x_phase_d;
    int x_H;  // workgroups of a team (0: no teams)
    // where the frames of a pair are cut between its two agents, in 1024ths of the frames (512: in the middle).  The backward
    // agent's steps of phase B are the dearest (its finishes: mm_engine.hip pair_variants), so it gets a little less than half
    int split_q10;
    int x_psn;  // floats of one slot of per-pdf partial sums (512; 1024 for the instances of more than 250 pdfs)
    int x_sleep;                // the exchange wave sleeps this many x 64 clocks before its first poll of a step
    // how long a poll of a team waits (ticks of s_memrealtime, 100 MHz) before it gives the team up and marks the utterances: a
    // workgroup's team mate may be dispatched a whole ROUND later (the compute units are taken by earlier workgroups of the
    // launch), so the bound follows the length of a launch -- ~4 of them: 10 us per frame, at least 2 ms, at most 0.1 s (it was
    // 0.1 s flat: a team that really does not run together -- a foreign kernel holds the compute units -- cost every call that)
    unsigned long long x_timeout;
    float lt_floor;             // mm_pair_finish_kernel: smallest accepted log2 overlap term of a frame (mm_batch_set_posterior_floor)
    // mm_batch_set_gamma_mode (wave kernel): gamma_out = g_scale * gamma (g_acc = 0) or gamma_out += g_scale * gamma (g_acc = 1: one
    // no-return float atomic per element -- every element is touched once per call, so the sum does not depend on any order; the
    // frames beyond len_b are left alone instead of zeroed)
    float g_scale;
    int g_acc;
    // stream kernels, teams of H workgroups per utterance and direction (mm_stream.hip): what the workgroups of a team send each other --
    // [utterance][direction][2 slots][sx_slot] wide values (high dwords) by position, the step's tag in their sign bits; zeroed per call
    unsigned *sx;
    long long sx_slot;
};

// one thread per workgroup of a finish kernel: add `hard` to the call's count; the last workgroup publishes it to the host
__device__ __forceinline__ void report_hard(const RunParams &p, int hard) {
    if (!p.stat_dev) return;
    atomicAdd(&p.stat_dev[0], hard);
    __threadfence();
    if (atomicAdd(&p.stat_dev[1], 1) == (int)gridDim.x - 1) {
        __threadfence();
        const int total = atomicAdd(&p.stat_dev[0], 0);
        p.stat_dev[0] = 0;
        p.stat_dev[1] = 0;
        p.stat_host[0] = total;
        p.stat_host[1] = p.stat_seq;
        __threadfence_system();
    }
}

// In-kernel cycle stamps (diagnostic build only; the shipped library executes none).
#ifdef MM_STAMPS
#define MM_STAMP(slot)                                                                     \
    do {                                                                                   \
        unsigned long long t_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        stamp_acc[slot] += t_ - stamp_last;                                                \
        stamp_last = t_;                                                                   \
    } while (0)
#define MM_STAMP_DECL unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = 0
#define MM_STAMP_RESET                                                                     \
    do {                                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory"); \
    } while (0)
#else
#define MM_STAMP(slot) do { } while (0)
#define MM_STAMP_DECL
#define MM_STAMP_RESET do { } while (0)
#endif

typedef __attribute__((address_space(3))) const float *lds_cfptr;
__device__ __forceinline__ unsigned lds_addr_of(const float *p) {
    return (unsigned)(__UINTPTR_TYPE__)(lds_cfptr)p;
}
// LDS-DMA (cdna_hip_programming.md 5.7): one wave instruction moves 4 or 16 bytes per ACTIVE lane from per-lane global
// addresses straight into the LDS block [lds_dst + lane * size): no destination register, so nothing the compiler
// would make the wave wait for (or spill); completion is counted by vmcnt, which the caller waits on by hand.
__device__ __forceinline__ void dma_b32(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_b128(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// A whole row by LDS-DMA in ONE asm block: NDM x 1 KB (64 lanes x 16 bytes each) from src + 16 * lane + 1024 * j to
// lds_dst + 1024 * j, j < NDM.  m0 is saved and restored once and advanced by s_add (dma_b128 moves it four times per
// load), the addresses are one scalar base + one vector offset + the instruction's immediate (dma_b128 takes a 64-bit vector
// address per load: a compare, a select and a 64-bit add each): 7 instructions per 4 KB instead of ~36.  The source must be
// readable for NDM KB (no clamping: the caller's buffer has that slack).
// (the instruction's immediate offset moves BOTH ends of an LDS-DMA, the global address and the LDS address: four loads of a
// group share m0 and the vector offset, both advance by 4 KB between groups)
#define MM_DMA_1(off) "global_load_lds_dwordx4 %1, %2 offset:" #off "\n\t"
#define MM_DMA_4 MM_DMA_1(0) MM_DMA_1(1024) MM_DMA_1(2048) MM_DMA_1(3072) "s_add_u32 m0, m0, 0x1000\n\tv_add_u32 %1, 0x1000, %1\n\ts_nop 0\n\t"
#define MM_DMA_G0 ""
#define MM_DMA_G1 MM_DMA_4
#define MM_DMA_G2 MM_DMA_4 MM_DMA_4
#define MM_DMA_G3 MM_DMA_4 MM_DMA_4 MM_DMA_4
#define MM_DMA_G4 MM_DMA_4 MM_DMA_4 MM_DMA_4 MM_DMA_4
#define MM_DMA_R0 ""
#define MM_DMA_R1 MM_DMA_1(0)
#define MM_DMA_R2 MM_DMA_1(0) MM_DMA_1(1024)
#define MM_DMA_R3 MM_DMA_1(0) MM_DMA_1(1024) MM_DMA_1(2048)
#define MM_DMA_CASE(G, R)                                                                                                   \
    if constexpr (NDM == 4 * G + R)                                                                                         \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" MM_DMA_G##G MM_DMA_R##R "s_mov_b32 m0, %0"          \
                     : "=&s"(keep), "+v"(voff)                                                                              \
                     : "s"(src), "s"(lds_dst)                                                                               \
                     : "memory", "scc");
template <int NDM>
__device__ __forceinline__ void dma_row_b128(const void *src, unsigned lane, unsigned lds_dst) {
    static_assert(NDM >= 1 && NDM <= 16, "rows of up to 16 KB");
    unsigned keep, voff = 16u * lane;
    MM_DMA_CASE(0, 1) MM_DMA_CASE(0, 2) MM_DMA_CASE(0, 3) MM_DMA_CASE(1, 0) MM_DMA_CASE(1, 1) MM_DMA_CASE(1, 2) MM_DMA_CASE(1, 3)
    MM_DMA_CASE(2, 0) MM_DMA_CASE(2, 1) MM_DMA_CASE(2, 2) MM_DMA_CASE(2, 3) MM_DMA_CASE(3, 0) MM_DMA_CASE(3, 1) MM_DMA_CASE(3, 2)
    MM_DMA_CASE(3, 3) MM_DMA_CASE(4, 0)
}

// LDS carve (in floats) shared by host (size) and device (offsets).
struct LdsPlan {
    int buf, stage, em, bins, part, total;
};
__host__ __device__ inline LdsPlan lds_plan(int S1p, int P1p, bool with_stage) {
    LdsPlan l;
    l.buf = 0;
    l.stage = l.buf + 2 * S1p;
    l.em = l.stage + (with_stage ? 2 * S1p : 0);
    l.bins = l.em + 2 * P1p;
    l.part = l.bins + 2 * P1p;
    l.total = l.part + 2 * MM_MAX_WAVES;
    return l;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// Cross-lane reductions inside aligned lane groups of 1 << log2g lanes (log2g is
// wave-uniform).  Butterfly steps 1, 2 are DPP quad permutes, 4 and 8 the DPP
// row_half_mirror / row_mirror (valid because the lower levels are already
// uniform inside their quads / octets); only groups wider than a 16-lane DPP row
// (rows with > 64 arcs) go through the LDS crossbar (ds_bpermute).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#define MM_DPP_XOR1 0xB1   // quad_perm [1,0,3,2]
#define MM_DPP_XOR2 0x4E   // quad_perm [2,3,0,1]
#define MM_DPP_HALF_MIRROR 0x141
#define MM_DPP_MIRROR 0x140
__device__ __forceinline__ float grp_max(float v, int log2g) {
    if (log2g == 0) return v;  // one lane per row (the common case): one branch instead of six
    if (log2g >= 1) v = fmaxf(v, dpp_mov<MM_DPP_XOR1>(v));
    if (log2g >= 2) v = fmaxf(v, dpp_mov<MM_DPP_XOR2>(v));
    if (log2g >= 3) v = fmaxf(v, dpp_mov<MM_DPP_HALF_MIRROR>(v));
    if (log2g >= 4) v = fmaxf(v, dpp_mov<MM_DPP_MIRROR>(v));
    if (log2g >= 5) v = fmaxf(v, __shfl_xor(v, 16));
    if (log2g >= 6) v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ float grp_sum(float v, int log2g) {
    if (log2g == 0) return v;
    if (log2g >= 1) v += dpp_mov<MM_DPP_XOR1>(v);
    if (log2g >= 2) v += dpp_mov<MM_DPP_XOR2>(v);
    if (log2g >= 3) v += dpp_mov<MM_DPP_HALF_MIRROR>(v);
    if (log2g >= 4) v += dpp_mov<MM_DPP_MIRROR>(v);
    if (log2g >= 5) v += __shfl_xor(v, 16);
    if (log2g >= 6) v += __shfl_xor(v, 32);
    return v;
}
// The same for a log2g known only at run time (the item kernels: one call per item): nested levels -- with the six tests in
// a row every level a group does NOT have is a taken branch over its code, four of them for a row on 4 lanes -- and the
// combine of a level as ONE DPP instruction (v_max_f32_dpp / v_add_f32_dpp) instead of a move and an operation.
#define MM_GRP_DPP(op, ctrl) asm("s_nop 1\n\t" op " %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ float grp_max_rt(float v, int log2g) {
    if (log2g == 0) return v;
    MM_GRP_DPP("v_max_f32_dpp", "quad_perm:[1,0,3,2]");
    if (log2g >= 2) {
        MM_GRP_DPP("v_max_f32_dpp", "quad_perm:[2,3,0,1]");
        if (log2g >= 3) {
            MM_GRP_DPP("v_max_f32_dpp", "row_half_mirror");
            if (log2g >= 4) {
                MM_GRP_DPP("v_max_f32_dpp", "row_mirror");
                if (log2g >= 5) {
                    v = fmaxf(v, __shfl_xor(v, 16));
                    if (log2g >= 6) v = fmaxf(v, __shfl_xor(v, 32));
                }
            }
        }
    }
    return v;
}
__device__ __forceinline__ float grp_sum_rt(float v, int log2g) {
    if (log2g == 0) return v;
    MM_GRP_DPP("v_add_f32_dpp", "quad_perm:[1,0,3,2]");
    if (log2g >= 2) {
        MM_GRP_DPP("v_add_f32_dpp", "quad_perm:[2,3,0,1]");
        if (log2g >= 3) {
            MM_GRP_DPP("v_add_f32_dpp", "row_half_mirror");
            if (log2g >= 4) {
                MM_GRP_DPP("v_add_f32_dpp", "row_mirror");
                if (log2g >= 5) {
                    v += __shfl_xor(v, 16);
                    if (log2g >= 6) v += __shfl_xor(v, 32);
                }
            }
        }
    }
    return v;
}
__device__ __forceinline__ float wave_max(float v) { return grp_max(v, 6); }
__device__ __forceinline__ float wave_sum(float v) { return grp_sum(v, 6); }

__device__ __forceinline__ Slot load_slot(const Slot *p) {
    uint2 r = *reinterpret_cast<const uint2 *>(p);
    Slot s;
    s.col = r.x;
    s.w = __uint_as_float(r.y);
    return s;
}

// (+)_k w_k (*) a[col_k] over one CSR row in the log semiring (log2 domain),
// the row being spread over a lane group: the replacement of
// _cukernel_mul_smdv! + warp_reduce (src/linalg.jl:204-233).
template <int R>
__device__ __forceinline__ float lse_small(const Slot *sp, int log2g, const float *a) {
    float x[R];
    Slot s[R];
#pragma unroll
    for (int k = 0; k < R; ++k) s[k] = load_slot(sp + k * 64);
#pragma unroll
    for (int k = 0; k < R; ++k) x[k] = s[k].w + a[s[k].col];
    float m = x[0];
#pragma unroll
    for (int k = 1; k < R; ++k) m = fmaxf(m, x[k]);
    m = grp_max_rt(m, log2g);
    float m0 = (m > MM_NINF) ? m : 0.f;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < R; ++k) sum += fast_exp2(x[k] - m0);
    sum = grp_sum_rt(sum, log2g);
    return m0 + fast_log2(sum);
}

// rows longer than 4 arcs per lane: online (running max, rescaled sum), chunks of 4
__device__ __forceinline__ float lse_long(const Slot *sp, int R, int log2g, const float *a) {
    float m = MM_NINF, sum = 0.f;
    for (int k0 = 0; k0 < R; k0 += 4) {
        float x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            Slot s = load_slot(sp + (k0 + k) * 64);
            x[k] = s.w + a[s.col];
        }
        float cm = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
        float mn = fmaxf(m, cm);
        float mn0 = (mn > MM_NINF) ? mn : 0.f;
        sum = sum * fast_exp2(m - mn0);
#pragma unroll
        for (int k = 0; k < 4; ++k) sum += fast_exp2(x[k] - mn0);
        m = mn;
    }
    float M = grp_max_rt(m, log2g);
    float M0 = (M > MM_NINF) ? M : 0.f;
    sum = grp_sum_rt(sum * fast_exp2(m - M0), log2g);
    return M0 + fast_log2(sum);
}

__device__ __forceinline__ float lse_item(const Slot *slots, const ItemMeta &im, int lane, const float *a) {
    const Slot *sp = slots + (size_t)im.slot_row * 64 + lane;
    const int R = im.R, lg = im.log2g;
    switch (R) {
        case 1: return lse_small<1>(sp, lg, a);
        case 2: return lse_small<2>(sp, lg, a);
        case 3: return lse_small<3>(sp, lg, a);
        case 4: return lse_small<4>(sp, lg, a);
        default: return lse_long(sp, R, lg, a);
    }
}

// tropical semiring: (+) = max -- the same item walk without the exponentials
__device__ __forceinline__ float max_item(const Slot *slots, const ItemMeta &im, int lane, const float *a) {
    const Slot *sp = slots + (size_t)im.slot_row * 64 + lane;
    float m = MM_NINF;
    for (int k = 0; k < im.R; ++k) {
        const Slot s = load_slot(sp + k * 64);
        m = fmaxf(m, s.w + a[s.col]);
    }
    return grp_max_rt(m, im.log2g);
}

__device__ __forceinline__ ItemMeta load_item(const ItemMeta *items, int it) {
    // `it` is wave-uniform: let the scalar unit fetch the 8-byte descriptor
    uint2 r = *reinterpret_cast<const uint2 *>(items + it);
    ItemMeta m;
    m.slot_row = __builtin_amdgcn_readfirstlane(r.x);
    unsigned q = __builtin_amdgcn_readfirstlane(r.y);
    m.R = (uint16_t)(q & 0xffff);
    m.log2g = (uint16_t)(q >> 16);
    return m;
}

// expand() (src/inference.jl:54-60) + the log2 scaling, one frame (1-based n) into LDS
__device__ __forceinline__ void stage_em(float *dst, const float *Vb, long long vsn, int n, int len, int P, int tid,
                                         int NT, float scale) {
    for (int q = tid; q <= P; q += NT) {
        float v;
        if (q < P)
            v = !Vb ? 0.f : ((n <= len) ? Vb[(long long)(n - 1) * vsn + q] * scale : MM_NINF);  // NULL: free run
        else
            v = (n <= len) ? MM_NINF : 0.f;
        dst[q] = v;
    }
}

__device__ __forceinline__ float part_max(const float *part, int NW) {
    float M = MM_NINF;
    for (int w = 0; w < NW; ++w) M = fmaxf(M, part[w]);
    return (M > MM_NINF) ? M : 0.f;
}

// Emissions of one frame (expand(), src/inference.jl:54-60): EVERY thread loads a raw value
// from a clamped, always valid address -- no branch and no arithmetic at the load, so nothing
// waits for it; em_value() turns it into the log2 emission when it is stored to LDS a phase later.
__device__ __forceinline__ float em_load_raw(const float *Vb, long long vsn, int n, int N, int P, int q) {
    const int nn = n < 1 ? 1 : (n > N ? N : n), qq = q < P ? q : P - 1;
    return Vb[(long long)(nn - 1) * vsn + qq];
}
__device__ __forceinline__ float em_value(float raw, int n, int len, int P, int q) {
    if (q < P) return (n <= len) ? raw * MM_LOG2E : MM_NINF;
    return (n <= len) ? MM_NINF : 0.f;
}

// max without the canonicalising v_max x, x the compiler puts in front of every fmaxf (the hardware
// instruction already returns the other operand for a NaN), and the 16-lane row maximum as four one-
// instruction DPP steps (s_nop: a DPP read needs two wait states after the VALU write of its source)
__device__ __forceinline__ float max_nc(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float row16_max(float v) {
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return v;
}
// wave-wide max: the 16-lane rows by DPP, then the 4 row results through readlane (no LDS crossbar trips)
__device__ __forceinline__ float wave_max_rl(float v) {
    v = row16_max(v);
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));  // scalar operands: folded on the scalar unit where possible
}

// max over the per-wave maxima of the previous frame (the lagged normaliser): one LDS read + a
// DPP row reduction
__device__ __forceinline__ float part_max_dpp(const float *part, int NW, int lane) {
    float v = (lane < NW) ? part[lane] : MM_NINF;
    v = row16_max(v);
    v = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
    return (v > MM_NINF) ? v : 0.f;
}
__device__ __forceinline__ void part_put(float *part, int wave, int lane, float wmax) {
    wmax = wave_max_rl(wmax);
    if (lane == 0) part[wave] = wmax;
}


// ---------------------------------------------------------------------------
// Register-resident graph.  The packed graph is the same for every frame, so a
// wave keeps its first NI items (those with R <= 4) in VGPRs for the whole
// time loop: 4 weights, 4 column indices packed as u16 pairs and the packed
// {row, pdf} per lane = 7 VGPRs per item.  With the whole graph on chip the
// per-frame work touches HBM/L2 only for emissions and the alpha store.
// Items beyond NI * NW (and all long rows) are streamed from L2 as before.
// ---------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int NI>
struct ItemRegs {
    float w[NI > 0 ? NI : 1][4];
    unsigned c[NI > 0 ? NI : 1][2];  // col0 | col1 << 16, col2 | col3 << 16
    unsigned ri[NI > 0 ? NI : 1];    // row | pdf << 16 ; row 0xffff = padding
    int meta[NI > 0 ? NI : 1];       // wave-uniform: R | log2g << 8 ; 0 = no item
};

template <int NI>
__device__ __forceinline__ void load_item_regs(ItemRegs<NI> &rg, const GraphDev &g, int wave, int NW, int lane) {
    static_for<0, NI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int it = wave + i * NW;
        int meta = 0;
        unsigned ri = 0xffffu, c01 = 0u, c23 = 0u;
        float w0 = MM_NINF, w1 = MM_NINF, w2 = MM_NINF, w3 = MM_NINF;
        if (it < g.n_items) {
            const ItemMeta im = load_item(g.items, it);
            if (im.R <= 4) {
                meta = (int)im.R | ((int)im.log2g << 8);
                const RowInfo r = g.rowinfo[(size_t)it * 64 + lane];
                ri = (r.row < 0 ? 0xffffu : (unsigned)r.row) | ((unsigned)r.pdf << 16);
                const Slot *sp = g.slots + (size_t)im.slot_row * 64 + lane;
                const Slot s0 = load_slot(sp);
                w0 = s0.w;
                c01 = s0.col;
                if (im.R > 1) {
                    const Slot s1 = load_slot(sp + 64);
                    w1 = s1.w;
                    c01 |= s1.col << 16;
                }
                if (im.R > 2) {
                    const Slot s2 = load_slot(sp + 128);
                    w2 = s2.w;
                    c23 = s2.col;
                }
                if (im.R > 3) {
                    const Slot s3 = load_slot(sp + 192);
                    w3 = s3.w;
                    c23 |= s3.col << 16;
                }
            }
        }
        rg.meta[i] = meta;
        rg.ri[i] = ri;
        rg.c[i][0] = c01;
        rg.c[i][1] = c23;
        rg.w[i][0] = w0;
        rg.w[i][1] = w1;
        rg.w[i][2] = w2;
        rg.w[i][3] = w3;
    });
}

// one register-resident item: two-pass log-sum-exp of <= 4 arcs per lane
__device__ __forceinline__ float lse_regs(float w0, float w1, float w2, float w3, unsigned c01, unsigned c23, int R,
                                          int lg, const float *a) {
    float x0 = w0 + a[c01 & 0xffffu];
    float x1 = w1 + a[c01 >> 16];
    float x2 = MM_NINF, x3 = MM_NINF;
    if (R > 2) {
        x2 = w2 + a[c23 & 0xffffu];
        x3 = w3 + a[c23 >> 16];
    }
    float m = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
    m = grp_max_rt(m, lg);
    const float m0 = (m > MM_NINF) ? m : 0.f;
    float sum = fast_exp2(x0 - m0) + fast_exp2(x1 - m0);
    if (R > 2) sum += fast_exp2(x2 - m0) + fast_exp2(x3 - m0);
    sum = grp_sum_rt(sum, lg);
    return m0 + fast_log2(sum);
}

__device__ __forceinline__ float max_regs(float w0, float w1, float w2, float w3, unsigned c01, unsigned c23, int R,
                                          int lg, const float *a) {
    float m = fmaxf(w0 + a[c01 & 0xffffu], w1 + a[c01 >> 16]);
    if (R > 2) m = fmaxf(m, fmaxf(w2 + a[c23 & 0xffffu], w3 + a[c23 >> 16]));
    return grp_max_rt(m, lg);
}

// Visit every item of this wave: epi(value, row, pdf) runs on the leader lane of
// each row group.  `a` = the LDS vector the arcs gather from.
// (epi also receives emn[pdf], fetched before the log-sum-exp so that its LDS round trip is not appended to it)
template <int NI, bool TROP = false, class Epi>
__device__ __forceinline__ void for_items(const ItemRegs<NI> &rg, const GraphDev &g, int wave, int NW, int lane,
                                          const float *a, const float *emn, Epi &&epi) {
    static_for<0, NI>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int meta = rg.meta[i];
        if (meta != 0) {
            int R = meta & 0xff, lg = meta >> 8;
            // (opaque per frame: hoisted out of the time loop, the comparisons with R and lg of all items become ~100
            // scalar-register pairs that are spilled to VGPR lanes and read back with v_readlane every frame)
            asm volatile("" : "+s"(R), "+s"(lg));
            const unsigned row = rg.ri[i] & 0xffffu;
            const float e = emn[row != 0xffffu ? (rg.ri[i] >> 16) : 0u];
            const float v = TROP ? max_regs(rg.w[i][0], rg.w[i][1], rg.w[i][2], rg.w[i][3], rg.c[i][0], rg.c[i][1], R, lg, a)
                                 : lse_regs(rg.w[i][0], rg.w[i][1], rg.w[i][2], rg.w[i][3], rg.c[i][0], rg.c[i][1], R, lg, a);
            if (row != 0xffffu && (lane & ((1 << lg) - 1)) == 0) epi(v, (int)row, (int)(rg.ri[i] >> 16), e);
        }
    });
    // items beyond the register window, and long rows inside it (meta = 0 there): streamed from L2
    const int resident = NI * NW < g.n_short ? NI * NW : g.n_short;  // (known without touching memory)
    for (int it = wave; it < g.n_items; it += NW) {
        if (it < resident) continue;
        const ItemMeta im = load_item(g.items, it);
        const RowInfo r = g.rowinfo[(size_t)it * 64 + lane];
        const float e = emn[r.row >= 0 ? r.pdf : 0];
        const float v = TROP ? max_item(g.slots, im, lane, a) : lse_item(g.slots, im, lane, a);
        if (r.row >= 0 && (lane & ((1 << im.log2g) - 1)) == 0) epi(v, r.row, r.pdf, e);
    }
}

// ---------------------------------------------------------------------------
// Log-semiring kernel.  MODE_FB: pdfposteriors (src/inference.jl:145-161).
// MODE_ALPHA / MODE_BETA: alpha-recursion / beta-recursion export (:62-74 / :99-110).
// grid = B workgroups (one utterance each), block = 64 * NW threads.
// ---------------------------------------------------------------------------
// PASS (MODE_FB only): 0 = the whole call in one kernel, 1 = forward half (leaves alpha, C_n and log2 Z -- in
// wsC[0] -- in the workspace), 2 = backward half.  Two kernels for the same reason as the quad path: each half
// gets its own register allocation and schedule.
// TROP (MODE_BETA only): the tropical semiring's beta-recursion -- max instead of log-sum-exp, natural-log
// values, no normalisation (float adds only, like the Viterbi kernel).
template <int MODE, int NI, int PASS = 0, bool TROP = false, bool BIGV = false>
__global__ void __launch_bounds__(NI > 8 ? 512 : 1024) mm_log_kernel(RunParams p) {
    extern __shared__ float lds[];
    const int b = p.order ? p.order[blockIdx.x] : blockIdx.x;
    if (p.redo && !p.redo[b]) return;  // launched behind the row kernels: only the utterances they marked
    const UttDesc &u = p.utts[b];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6;
    const int S1 = u.S1, S1p = u.S1p, P1 = u.P1, P = P1 - 1, P1p = (P1 + 3) & ~3;
    const int fstate = S1 - 1;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    if (p.free_run == 2) len = 0;
    // frames are 1-based n = 1 .. N+1 as in the reference; FB stops at len+1,
    // where only the phony final state is alive (everything after is constant)
    const int NF = (MODE == MODE_FB) ? len + 1 : p.N + 1;
    const LdsPlan L = lds_plan(BIGV ? 0 : S1p, P1p, MODE == MODE_FB);
    // BIGV (FSMs whose state vectors do not fit the LDS): the vectors live in global memory (L2), visible to the other
    // waves of the workgroup through a release / acquire fence pair around every barrier -- the same code, slower gathers
    float *em = lds + L.em, *bins = lds + L.bins, *part = lds + L.part;
    float *buf = BIGV ? p.ws_big + (long long)b * p.big_stride : lds + L.buf;
    float *stage = BIGV ? buf + 2 * S1p : lds + L.stage;
    auto vsync = [&]() {
        if constexpr (BIGV) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if constexpr (BIGV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    const float *Vb = p.free_run ? nullptr : p.V + (long long)b * p.vsb;
    float *wsA = p.ws_alpha ? p.ws_alpha + u.s1p_prefix * (long long)(p.N + 1) : nullptr;
    double *wsC = p.ws_c ? p.ws_c + (long long)b * (p.N + 2) : nullptr;
    const GraphDev gf = u.g[0], gb = u.g[1];
    double logZ2 = 0.0;
    ItemRegs<NI> rg;

    if (MODE != MODE_BETA && PASS != 2) {
        // ---------------- forward: alpha-recursion ----------------
        stage_em(em + 1 * P1p, Vb, p.vsn, 1, len, P, tid, NT, MM_LOG2E);
        for (int q = tid; q < 2 * S1p; q += NT) buf[q] = MM_NINF;
        vsync();
        {   // frame 1: alpha_hat (*) lhs[:,1]   (src/inference.jl:68)
            float wm = MM_NINF;
            float *a1 = buf + 1 * S1p;
            const float *e1 = em + 1 * P1p;
            for (int s = tid; s < S1; s += NT) {
                float v = u.init[s] + e1[u.s2p[s]];
                a1[s] = v;
                wm = fmaxf(wm, v);
            }
            wm = wave_max(wm);
            if (lane == 0) part[1 * MM_MAX_WAVES + wave] = wm;
            if (NF >= 2) stage_em(em + 0 * P1p, Vb, p.vsn, 2, len, P, tid, NT, MM_LOG2E);
            if (tid == 0 && wsC) wsC[1] = 0.0;
        }
        vsync();
        load_item_regs<NI>(rg, gf, wave, NW, lane);
        double C = 0.0, Cprev = 0.0;
        // the emissions travel one frame ahead in a register: loaded during step n-1, stored to LDS at the top
        // of step n, read after the barrier that ends it -- no step waits for its own global load
        float evp = Vb ? em_load_raw(Vb, p.vsn, 3, p.N, P, tid) : 0.f;
        MM_STAMP_DECL;
        MM_STAMP_RESET;
        for (int n = 2; n <= NF; ++n) {
            const float *ap = buf + ((n - 1) & 1) * S1p;
            float *an = buf + (n & 1) * S1p;
            const float *emn = em + (n & 1) * P1p;
            const float M = part_max_dpp(part + ((n - 1) & 1) * MM_MAX_WAVES, NW, lane);
            MM_STAMP(0);
            Cprev = C;
            C += (double)M;
            if (tid == 0 && wsC) wsC[n] = C;
            if (n + 1 <= NF) {
                if (tid <= P) em[((n + 1) & 1) * P1p + tid] = Vb ? em_value(evp, n + 1, len, P, tid) : ((tid < P || n + 1 > len) ? 0.f : MM_NINF);
                if (P >= NT) stage_em(em + ((n + 1) & 1) * P1p + NT, Vb ? Vb + NT : nullptr, p.vsn, n + 1, len, P - NT, tid, NT, MM_LOG2E);
            }
            if (Vb) evp = em_load_raw(Vb, p.vsn, n + 2, p.N, P, tid);
            // frame n-1 leaves the chip once (coalesced), while frame n is computed
            if (MODE == MODE_FB) {
                float4 *dst = reinterpret_cast<float4 *>(wsA + (long long)(n - 1) * S1p);
                const float4 *src = reinterpret_cast<const float4 *>(ap);
                for (int q = tid; q < (S1p >> 2); q += NT) dst[q] = src[q];
            } else {
                float *dst = p.out + (long long)(n - 2) * p.out_stride_n + u.state_off;
                const float c = (float)Cprev;
                for (int s = tid; s < S1; s += NT) dst[s] = (ap[s] + c) * MM_LN2;
            }
            MM_STAMP(1);
            float wm = MM_NINF;
            for_items<NI>(rg, gf, wave, NW, lane, ap, emn, [&](float v, int row, int pdf, float e) {
                v = v + e - M;  // (T' alpha_{n-1}) (*) lhs[:,n]   (src/inference.jl:70-71)
                an[row] = v;
                wm = max_nc(wm, v);
            });
            MM_STAMP(2);
            part_put(part + (n & 1) * MM_MAX_WAVES, wave, lane, wm);
            MM_STAMP(3);
            vsync();
            MM_STAMP(4);
        }
#ifdef MM_STAMPS
        if (p.dbg && lane == 0)
            for (int k = 0; k < 8; ++k) p.dbg[((long long)b * MM_MAX_WAVES + wave) * 16 + k] = stamp_acc[k];
#endif
        const float *alast = buf + (NF & 1) * S1p;
        logZ2 = (double)alast[fstate] + C;
        if (PASS == 1) {
            if (tid == 0) wsC[0] = logZ2;
            return;
        }
        if (MODE == MODE_ALPHA) {
            float *dst = p.out + (long long)(NF - 1) * p.out_stride_n + u.state_off;
            const float c = (float)C;
            for (int s = tid; s < S1; s += NT) dst[s] = (alast[s] + c) * MM_LN2;
            return;
        }
        vsync();
    }

    if (MODE == MODE_FB && PASS != 1) {
        // ---------------- backward: beta-recursion fused with the combine ----------------
        if (PASS == 2) logZ2 = wsC[0];
        const long long gbase = (long long)b * p.gsb;
        if (!(logZ2 > -1e300)) {  // no accepting path: gamma = 0, ttl = -inf
            for (long long q = tid; q < (long long)p.N * P; q += NT)
                p.gamma[gbase + (q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
            if (tid == 0) p.ttl[b] = MM_NINF;
            return;
        }
        // frame len+1: B (*) lhs = one for the final state only (src/inference.jl:104,106 + expand)
        for (int q = tid; q < 2 * S1p; q += NT) buf[q] = MM_NINF;
        for (int q = tid; q < 2 * P1p; q += NT) bins[q] = 0.f;
        vsync();
        if (tid == 0) buf[(NF & 1) * S1p + fstate] = 0.f;
        if (len >= 1) {
            stage_em(em + (len & 1) * P1p, Vb, p.vsn, len, len, P, tid, NT, MM_LOG2E);
            const float4 *src = reinterpret_cast<const float4 *>(wsA + (long long)len * S1p);
            float4 *dst = reinterpret_cast<float4 *>(stage + (len & 1) * S1p);
            for (int q = tid; q < (S1p >> 2); q += NT) dst[q] = src[q];
        }
        vsync();
        load_item_regs<NI>(rg, gb, wave, NW, lane);
        const bool det = __builtin_amdgcn_readfirstlane(p.deterministic) != 0;
        double D = 0.0;
        float tmin = (float)logZ2;
        // frame n-1 (emissions, alpha, C) travels in registers: loaded during step n+1, stored to LDS at the
        // top of step n, read in step n-1 -- no step waits for its own global loads
        constexpr int AK = 2;  // float4 of alpha per thread carried in registers; more rows are staged in place
        const int n4 = S1p >> 2;
        float evp = 0.f;
        float4 apre[AK];
        double Cn = len >= 1 ? __hip_atomic_load(&wsC[len], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0, Cpre = 0.0;
        auto prefetch = [&](int f) {  // frame f >= 1
            evp = em_load_raw(Vb, p.vsn, f, p.N, P, tid);
            if constexpr (BIGV) {
                const float4 *src = reinterpret_cast<const float4 *>(wsA + (long long)f * S1p);
#pragma unroll
                for (int k = 0; k < AK; ++k)
                    if (tid + k * NT < n4) apre[k] = src[tid + k * NT];
            }
            Cpre = __hip_atomic_load(&wsC[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // alpha~ of frame f straight into its staging buffer by LDS-DMA (the buffer's previous tenant, frame f + 2, is done;
        // the wave waits for its own DMAs before the barrier that ends the step).  Carried in registers, as the emissions
        // are, the rows were spilled to scratch in the loop: every reload waited for ALL outstanding loads.
        auto dma_alpha = [&](int f) {
            const float4 *src = reinterpret_cast<const float4 *>(wsA + (long long)f * S1p);
            const unsigned dst = lds_addr_of(stage + (f & 1) * S1p);
            for (int q0 = wave * 64; q0 < n4; q0 += NT)
                if (q0 + lane < n4) dma_b128(src + q0 + lane, dst + 16u * (unsigned)q0);
        };
        if (len >= 2) prefetch(len - 1);
        for (int n = len; n >= 1; --n) {
            const float *yp = buf + ((n + 1) & 1) * S1p;
            float *yn = buf + (n & 1) * S1p;
            float *ast = stage + (n & 1) * S1p;  // alpha~ of frame n (deterministic mode: replaced by the posteriors as they are made)
            const float *emn = em + (n & 1) * P1p;
            float *bn = bins + (n & 1) * P1p;
            const float M = (n == len) ? 0.f : part_max_dpp(part + ((n + 1) & 1) * MM_MAX_WAVES, NW, lane);
            D += (double)M;
            const float kappa = (float)(logZ2 - Cn - D);
            // finalise frame n+1 (one rotating wave): C' * AB, per-frame sum, divide, exp (src/inference.jl:155-160)
            // (the last wave: the host gives the workgroup one wave more than it has items when it can, so that
            // this work runs beside the items instead of in front of one)
            if (n < len && wave == NW - 1) {
                float *bf = bins + ((n + 1) & 1) * P1p;
                float s = 0.f;
                for (int q = lane; q < P1; q += 64) s += bf[q];
                s = wave_sum(s);
                const float inv = 1.f / s;
                float *gp = p.gamma + gbase + (long long)n * p.gsn;  // frame n+1 -> 0-based index n
                for (int q = lane; q < P1; q += 64) {
                    if (q < P) gp[q * p.gsp] = bf[q] * inv;
                    bf[q] = 0.f;
                }
                tmin = fminf(tmin, (float)(logZ2 + (double)fast_log2(s)));
            }
            if (n - 1 >= 1) {  // frame n-1 from the registers into the buffers frame n+1 has left
                if (tid <= P) em[((n - 1) & 1) * P1p + tid] = em_value(evp, n - 1, len, P, tid);
                if (P >= NT) stage_em(em + ((n - 1) & 1) * P1p + NT, Vb + NT, p.vsn, n - 1, len, P - NT, tid, NT, MM_LOG2E);
                if constexpr (BIGV) {
                    float4 *dst = reinterpret_cast<float4 *>(stage + ((n - 1) & 1) * S1p);
#pragma unroll
                    for (int k = 0; k < AK; ++k)
                        if (tid + k * NT < n4) dst[tid + k * NT] = apre[k];
                    if (n4 > AK * NT) {
                        const float4 *src = reinterpret_cast<const float4 *>(wsA + (long long)(n - 1) * S1p);
                        for (int q = tid + AK * NT; q < n4; q += NT) dst[q] = src[q];
                    }
                } else {
                    dma_alpha(n - 1);
                }
                Cn = Cpre;
                if (n - 2 >= 1) prefetch(n - 2);
            }
            float wm = MM_NINF;
            for_items<NI>(rg, gb, wave, NW, lane, yp, emn, [&](float v, int row, int pdf, float e) {
                const float beta = v - M;  // T (B[:,n+1] (*) lhs[:,n+1])   (src/inference.jl:106-107)
                const float q = fast_exp2(ast[row] + beta - kappa);  // state_A .* state_B / Z
                if (det) ast[row] = q;
                else if (q > 0.f) atomicAdd(&bn[pdf], q);
                const float y = beta + e;
                yn[row] = y;
                wm = max_nc(wm, y);
            });
            part_put(part + (n & 1) * MM_MAX_WAVES, wave, lane, wm);
            if constexpr (!BIGV) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of alpha~ of frame n - 1 is in LDS
            vsync();
            if (det) {
                // C' * (A .* B) (:155) without atomics: a pdf's states are listed in pdf_rows; 8 lanes per pdf add their
                // posteriors in a fixed order, a 3-step DPP reduction ends it: the same bits on every run.  The second
                // barrier also guards the buffer (the next step stages alpha~ of frame n - 2 over the posteriors).  Costs a
                // small deep graph ~20 % (WSJ numerator x128: 2.14 -> 2.6 ms), hence opt-in like PyTorch's deterministic mode.
                for (int p0 = wave * 8; p0 < P1; p0 += NW * 8) {
                    const int pdf = p0 + (lane >> 3);
                    float sacc = 0.f;
                    if (pdf < P1) {
                        const int e0 = u.pdf_ptr[pdf], e1 = u.pdf_ptr[pdf + 1];
                        for (int k = e0 + (lane & 7); k < e1; k += 8) sacc += ast[u.pdf_rows[k]];
                    }
                    sacc = grp_sum(sacc, 3);
                    if (pdf < P1 && (lane & 7) == 0) bn[pdf] = sacc;
                }
                vsync();
            }
        }
        // finalise frame 1, zero the frames beyond len, reduce ttl
        if (len >= 1 && wave == 0) {
            float *bf = bins + (1 & 1) * P1p;
            float s = 0.f;
            for (int q = lane; q < P1; q += 64) s += bf[q];
            s = wave_sum(s);
            const float inv = 1.f / s;
            float *gp = p.gamma + gbase;
            for (int q = lane; q < P; q += 64) gp[q * p.gsp] = bf[q] * inv;
            tmin = fminf(tmin, (float)(logZ2 + (double)fast_log2(s)));
        }
        for (long long q = tid; q < (long long)(p.N - len) * P; q += NT)
            p.gamma[gbase + (len + q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
        vsync();  // part[] is free again
        if (lane == 0) part[wave] = tmin;
        vsync();
        if (tid == 0) {
            float t = part[0];
            for (int w = 1; w < NW; ++w) t = fminf(t, part[w]);
            p.ttl[b] = t * MM_LN2;
        }
    }

    if (MODE == MODE_BETA) {
        // beta-recursion export (src/inference.jl:99-110): all N+1 frames, B[:,N+1] = one
        for (int q = tid; q < 2 * S1p; q += NT) buf[q] = MM_NINF;
        stage_em(em + (NF & 1) * P1p, Vb, p.vsn, NF, len, P, tid, NT, (TROP ? 1.0f : MM_LOG2E));
        vsync();
        {
            float *yl = buf + (NF & 1) * S1p;
            const float *el = em + (NF & 1) * P1p;
            float *dst = p.out + (long long)(NF - 1) * p.out_stride_n + u.state_off;
            float wm = MM_NINF;
            for (int s = tid; s < S1; s += NT) {
                float y = el[u.s2p[s]];
                yl[s] = y;
                wm = fmaxf(wm, y);
                dst[s] = 0.f;
            }
            wm = wave_max(wm);
            if (lane == 0) part[(NF & 1) * MM_MAX_WAVES + wave] = wm;
            if (NF - 1 >= 1) stage_em(em + ((NF - 1) & 1) * P1p, Vb, p.vsn, NF - 1, len, P, tid, NT, (TROP ? 1.0f : MM_LOG2E));
        }
        vsync();
        load_item_regs<NI>(rg, gb, wave, NW, lane);
        double D = 0.0;
        for (int n = NF - 1; n >= 1; --n) {
            const float *yp = buf + ((n + 1) & 1) * S1p;
            float *yn = buf + (n & 1) * S1p;
            const float *emn = em + (n & 1) * P1p;
            const float M = TROP ? 0.f : part_max(part + ((n + 1) & 1) * MM_MAX_WAVES, NW);
            D += (double)M;
            const float d = (float)D;
            if (n - 1 >= 1) stage_em(em + ((n - 1) & 1) * P1p, Vb, p.vsn, n - 1, len, P, tid, NT, (TROP ? 1.0f : MM_LOG2E));
            float *dst = p.out + (long long)(n - 1) * p.out_stride_n + u.state_off;
            float wm = MM_NINF;
            for_items<NI, TROP>(rg, gb, wave, NW, lane, yp, emn, [&](float v, int row, int pdf, float e) {
                const float beta = v - M;
                dst[row] = TROP ? beta : (beta + d) * MM_LN2;
                const float y = beta + e;
                yn[row] = y;
                wm = fmaxf(wm, y);
            });
            wm = wave_max(wm);
            if (lane == 0) part[(n & 1) * MM_MAX_WAVES + wave] = wm;
            vsync();
        }
    }
}

// ---------------------------------------------------------------------------
// Tropical-semiring forward recursion with back-pointers (Viterbi).
// Forward = alpha-recursion (src/inference.jl:62-74) with K = TropicalSemiring;
// the arithmetic is the reference's float32 adds un-normalised, so values and
// arg-max decisions are bit-identical to the CPU restatement.
// Optional alpha export through p.out (natural log).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void trop_better(float &best, int &arg, float v, int c) {
    if (v > best || (v == best && v > MM_NINF && c < arg)) {
        best = v;
        arg = c;
    }
}

template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// (max, lowest arg-max) over an aligned lane group, same butterfly as grp_max
__device__ __forceinline__ void trop_grp_reduce(float &best, int &arg, int log2g) {
    if (log2g == 0) return;
    if (log2g >= 1) trop_better(best, arg, dpp_mov<MM_DPP_XOR1>(best), dpp_mov_i<MM_DPP_XOR1>(arg));
    if (log2g >= 2) trop_better(best, arg, dpp_mov<MM_DPP_XOR2>(best), dpp_mov_i<MM_DPP_XOR2>(arg));
    if (log2g >= 3) trop_better(best, arg, dpp_mov<MM_DPP_HALF_MIRROR>(best), dpp_mov_i<MM_DPP_HALF_MIRROR>(arg));
    if (log2g >= 4) trop_better(best, arg, dpp_mov<MM_DPP_MIRROR>(best), dpp_mov_i<MM_DPP_MIRROR>(arg));
    if (log2g >= 5) trop_better(best, arg, __shfl_xor(best, 16), __shfl_xor(arg, 16));
    if (log2g >= 6) trop_better(best, arg, __shfl_xor(best, 32), __shfl_xor(arg, 32));
}

// NI items per wave live in registers for the whole time loop (ItemRegs), the rest is streamed.
// Back-pointers are collected in LDS and leave the chip as whole rows one frame later.
template <int NI, bool BIGV = false>
__global__ void __launch_bounds__(1024) mm_tropical_kernel(RunParams p) {
    extern __shared__ float lds[];
    const int b = blockIdx.x;
    const UttDesc &u = p.utts[b];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6;
    const int S1 = u.S1, S1p = u.S1p, P1 = u.P1, P = P1 - 1, P1p = (P1 + 3) & ~3;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    if (p.free_run == 2) len = 0;
    const int NF = p.N + 1;
    const LdsPlan L = lds_plan(BIGV ? 0 : S1p, P1p, true);
    float *em = lds + L.em;
    float *buf = BIGV ? p.ws_big + (long long)b * p.big_stride : lds + L.buf;  // (BIGV: see mm_log_kernel)
    auto vsync = [&]() {
        if constexpr (BIGV) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if constexpr (BIGV) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    int *bpbuf = reinterpret_cast<int *>(BIGV ? buf + 2 * S1p : lds + L.stage);  // [2][S1p]
    const float *Vb = p.free_run ? nullptr : p.V + (long long)b * p.vsb;
    const GraphDev gf = u.g[0];
    int *bpb = p.bp ? p.bp + u.state_off : nullptr;
    ItemRegs<NI> rg;

    stage_em(em + 1 * P1p, Vb, p.vsn, 1, len, P, tid, NT, 1.0f);
    for (int q = tid; q < 2 * S1p; q += NT) {
        buf[q] = MM_NINF;
        bpbuf[q] = -1;
    }
    vsync();
    {
        float *a1 = buf + 1 * S1p;
        const float *e1 = em + 1 * P1p;
        for (int s = tid; s < S1; s += NT) {
            a1[s] = u.init[s] + e1[u.s2p[s]];
            if (bpb) bpb[s] = -1;
            if (p.out) p.out[u.state_off + s] = a1[s];
        }
        if (NF >= 2) stage_em(em + 0 * P1p, Vb, p.vsn, 2, len, P, tid, NT, 1.0f);
    }
    load_item_regs<NI>(rg, gf, wave, NW, lane);
    vsync();
    if (len == 0 && tid == 0 && p.score) p.score[b] = buf[1 * S1p + S1 - 1];
    // (the best path ends in the phony final state at frame len + 1; later frames matter to an export only)
    const int n_end = (p.stop_at_len && !p.out && len + 1 < NF) ? (len + 1 < 1 ? 1 : len + 1) : NF;
    for (int n = 2; n <= n_end; ++n) {
        const float *ap = buf + ((n - 1) & 1) * S1p;
        float *an = buf + (n & 1) * S1p;
        int *bpn = bpbuf + (n & 1) * S1p;
        const float *emn = em + (n & 1) * P1p;
        // emissions of frame n+1: raw load now (every thread, clamped address: nothing waits on it),
        // stored to LDS at the end of the step
        float evraw;
        {
            const int nn = n + 1 > p.N ? p.N : n + 1, qq = tid < P ? tid : P - 1;
            evraw = Vb ? Vb[(long long)(nn - 1) * p.vsn + qq] : 0.f;
        }
        if (bpb && n > 2) {  // back-pointers of frame n-1 (0-based row n-2): whole row, coalesced
            const int *src = bpbuf + ((n - 1) & 1) * S1p;
            int *dst = bpb + (long long)(n - 2) * p.bp_stride_n;
            for (int s = tid; s < S1; s += NT) dst[s] = src[s];
        }
        auto finish = [&](float best, int arg, int row, int pdf) {
            const float v = best + emn[pdf];
            an[row] = v;
            bpn[row] = arg;
            if (p.out) p.out[(long long)(n - 1) * p.out_stride_n + u.state_off + row] = v;
        };
        static_for<0, NI>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const int meta = rg.meta[i];
            if (meta != 0) {
                const int R = meta & 0xff, lg = meta >> 8;  // (not opaque as in for_items: measured 1.5 % slower here)
                float best = MM_NINF;
                int arg = -1;
                const int c0 = rg.c[i][0] & 0xffffu, c1 = rg.c[i][0] >> 16;
                trop_better(best, arg, rg.w[i][0] + ap[c0], c0);
                trop_better(best, arg, rg.w[i][1] + ap[c1], c1);
                if (R > 2) {
                    const int c2 = rg.c[i][1] & 0xffffu, c3 = rg.c[i][1] >> 16;
                    trop_better(best, arg, rg.w[i][2] + ap[c2], c2);
                    trop_better(best, arg, rg.w[i][3] + ap[c3], c3);
                }
                trop_grp_reduce(best, arg, lg);
                const unsigned row = rg.ri[i] & 0xffffu;
                if (row != 0xffffu && (lane & ((1 << lg) - 1)) == 0) finish(best, arg, (int)row, (int)(rg.ri[i] >> 16));
            }
        });
        const int resident = NI * NW < gf.n_short ? NI * NW : gf.n_short;  // (known without touching memory)
        for (int it = wave; it < gf.n_items; it += NW) {
            if (it < resident) continue;
            const ItemMeta im = load_item(gf.items, it);
            const RowInfo ri = gf.rowinfo[(size_t)it * 64 + lane];
            const Slot *sp = gf.slots + (size_t)im.slot_row * 64 + lane;
            float best = MM_NINF;
            int arg = -1;
            for (int k = 0; k < im.R; ++k) {
                Slot s = load_slot(sp + k * 64);
                trop_better(best, arg, s.w + ap[s.col], (int)s.col);
            }
            trop_grp_reduce(best, arg, im.log2g);
            if (ri.row >= 0 && (lane & ((1 << im.log2g) - 1)) == 0) finish(best, arg, ri.row, ri.pdf);
        }
        if (n + 1 <= NF) {
            if (tid <= P) {
                float *dst = em + ((n + 1) & 1) * P1p;
                if (tid < P) dst[tid] = (!Vb || n + 1 <= len) ? evraw : MM_NINF;
                else dst[tid] = (n + 1 <= len) ? MM_NINF : 0.f;
            }
            if (P >= NT) stage_em(em + ((n + 1) & 1) * P1p + NT, Vb ? Vb + NT : nullptr, p.vsn, n + 1, len, P - NT, tid, NT, 1.0f);
        }
        vsync();
        if (n == len + 1 && tid == 0 && p.score) p.score[b] = an[S1 - 1];
    }
    if (bpb && n_end >= 2) {
        const int *src = bpbuf + (n_end & 1) * S1p;
        int *dst = bpb + (long long)(n_end - 1) * p.bp_stride_n;
        for (int s = tid; s < S1; s += NT) dst[s] = src[s];
    }
}

#ifndef MM_SECONDARY_TU  // (plain kernels: defined once, in the translation unit of mm_engine.hip)
// back-trace: one lane per utterance follows the back-pointers from the phony
// final state at frame len+1 (historical bestpath, examples/demo.ipynb cell 23)
__global__ void mm_backtrace_kernel(RunParams p) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= p.B) return;
    const UttDesc &u = p.utts[b];
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    int *path = p.path + (long long)b * p.path_stride_b;
    for (int n = len; n < p.N; ++n) path[n] = -1;
    const bool ok = p.score[b] > MM_NINF;
    int s = u.S1 - 1;
    const int *bpb = p.bp + u.state_off;
    for (int n = len; n >= 1; --n) {
        if (ok) s = bpb[(long long)n * p.bp_stride_n + s];
        path[n - 1] = ok ? s : -1;
    }
}

// total-sum family: the value of the phony final state in frame n+1 (0-based row n) of the exported recursion
__global__ void mm_pick_final_kernel(const UttDesc *utts, int B, const float *A, long long stride_n, int n, float *out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = A[(long long)n * stride_n + utts[b].state_off + utts[b].S1 - 1];
}

// maxstateposteriors (docs/src/inference.md:5): mu = alpha (*) beta (/) best in place of alpha, one workgroup per
// (utterance, frame); best[b] = alpha of the phony final state in the last frame (mm_pick_final_kernel).
__global__ void mm_maxmarginal_kernel(const UttDesc *utts, float *A, long long a_stride_n, const float *Bt, long long b_stride_n,
                                      const float *best) {
    const int b = blockIdx.x, n = blockIdx.y;
    const UttDesc &u = utts[b];
    const float bb = best[b];
    float *a = A + (long long)n * a_stride_n + u.state_off;
    const float *bt = Bt + (long long)n * b_stride_n + u.state_off;
    for (int s = threadIdx.x; s < u.S1; s += blockDim.x) a[s] = (bb > MM_NINF) ? a[s] + bt[s] - bb : MM_NINF;
}

// The launch ahead of a pdfposteriors call: order[rank] = utterance, by decreasing length (ties by index; lens / order NULL:
// none), and the marks of the linear-domain kernels: redo[0..B] = fill (0, or 1 for a call that goes to the exact kernels
// with the whole batch), redo2[0..B] = 0 (either may be NULL).  One kernel instead of a rank kernel and two or three
// memset nodes (a memset of B + 1 ints is TWO fill kernels of the runtime: an aligned body and a tail -- 5 us each on the
// caller's stream, rocprofv3 timeline).  The lengths go through LDS: B loads per thread from global memory took 14 us.
__global__ void __launch_bounds__(256) mm_prologue_kernel(const int *lens, int B, int N, int *order, int *redo, int *redo2, int fill) {
    __shared__ int sh[256];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= B) {
        if (redo) redo[i] = i < B ? fill : 0;
        if (redo2) redo2[i] = 0;
    }
    if (!lens || !order) return;
    auto clamp = [&](int l) { return l < 0 ? 0 : (l > N ? N : l); };
    const int li = i < B ? clamp(lens[i]) : 0;
    int rank = 0;
    for (int j0 = 0; j0 < B; j0 += 256) {
        __syncthreads();
        sh[threadIdx.x] = j0 + (int)threadIdx.x < B ? clamp(lens[j0 + threadIdx.x]) : -1;
        __syncthreads();
        const int n = B - j0 < 256 ? B - j0 : 256;
        for (int k = 0; k < n; ++k) {
            const int lj = sh[k], j = j0 + k;
            rank += (lj > li) || (lj == li && j < i);
        }
    }
    if (i < B) order[rank] = i;
}

// *out (pinned host memory) = the number of marked utterances
__global__ void __launch_bounds__(256) mm_count_marks_kernel(const int *redo, int B, int *out) {
    __shared__ int part[4];
    int n = 0;
    for (int b = threadIdx.x; b < B; b += 256) n += redo[b] != 0;
    for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        *out = part[0] + part[1] + part[2] + part[3];
        __threadfence_system();
    }
}

// zero n16 x 16 bytes at dst (16-byte aligned): the team kernels' exchange areas, before every call
__global__ void __launch_bounds__(256) mm_zero_kernel(char *dst, unsigned long long n16) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 *q = reinterpret_cast<f4 *>(dst);
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256ull) q[i] = f4{0.f, 0.f, 0.f, 0.f};
}

// Emission shift for the quad kernels (they normalise by a lagged state maximum only: log-likelihoods far from 0 -- GMM
// scores around -300 nats -- push every row off their linear path and onto the exact per-row fallback, 30x slower).
// Posteriors do not change when a frame's log-likelihoods are shifted by a constant, and log Z changes by the sum of the
// shifts: Vs[b][n][:] = V[b][n][:] - E[b][n], E = the frame's maximum over the real pdfs (0 if none is finite), for the
// utterances the quad kernels will compute (all, or the marked ones); mm_shift_ttl_kernel adds sum_n E back to ttl.
__global__ void mm_shift_em_kernel(RunParams p, float *Vs, float *E) {
    const int b = blockIdx.x;
    if (p.redo && !p.redo[b]) return;
    const int P = p.utts[b].P1 - 1;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    for (int n = blockIdx.y * NW + wave; n < len; n += gridDim.y * NW) {  // one wave per frame
        const float *row = p.V + (long long)b * p.vsb + (long long)n * p.vsn;
        float m = MM_NINF;
        for (int q = lane; q < P; q += 64) m = fmaxf(m, row[q]);
        m = wave_max(m);
        if (!(m > MM_NINF) || !(m < __builtin_inff())) m = 0.f;
        float *out = Vs + ((long long)b * p.N + n) * P;
        for (int q = lane; q < P; q += 64) out[q] = row[q] - m;
        if (lane == 0) E[(long long)b * p.N + n] = m;
    }
}
__global__ void mm_shift_ttl_kernel(RunParams p, const float *E) {
    const int b = blockIdx.x;
    if (p.redo && !p.redo[b]) return;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    __shared__ double part[4];
    double s = 0.0;
    for (int n = threadIdx.x; n < len; n += blockDim.x) s += (double)E[(long long)b * p.N + n];
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (unsigned w = 0; w < blockDim.x / 64; ++w) t += part[w];  // (fixed order)
        if (p.ttl[b] > MM_NINF) p.ttl[b] = (float)((double)p.ttl[b] + t);
    }
}

#endif  // MM_SECONDARY_TU

}  // namespace mm
