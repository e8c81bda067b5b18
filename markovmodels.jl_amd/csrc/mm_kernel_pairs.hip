// mm_kernel_pairs.hip -- the "pair" pdfposteriors kernels for gfx950: the row kernels' organisation
// (mm_kernel_rows.hip, mm_rows.h: one lane per row of the semiring product, register-resident graph, one barrier per
// frame, a service wave for all HBM traffic) with two changes that take the LDS gathers off the critical path:
//
//   * TWO utterances per workgroup share the graph registers.  The linear vectors of the two utterances sit side by
//     side in LDS (8 bytes per state), so ONE ds_read_b64 per arc fetches both values: per utterance and frame half
//     the gather instructions, half the address registers and half the LDS cycles of the row kernels (whose frame was
//     ~80 % LDS-array time).  Batches whose utterances share one FSM only (the denominator case,
//     examples/test_cuda.jl:112); the utterances of a pair run the same number of frames, so they are paired by length.
//   * BIDIRECTIONAL time split.  With two utterances per workgroup B utterances give B / 2 workgroups per direction --
//     half the chip for B = #CUs.  So the alpha-recursion (src/inference.jl:62-74) of a pair and its beta-recursion
//     (:99-110) run CONCURRENTLY, as two workgroups ("agents"): the forward agent walks frames 1, 2, ..., the backward
//     agent N+1, N, ...; in phase A (first launch) each stores its normalised log2 vectors for its half of the frames,
//     in phase B (second launch) each continues through the other half and combines its fresh vector with the vector
//     the other agent stored for that frame (:154-160: A .* B, C' * AB, per-frame sum, divide).  Every frame's state
//     vector is still written once and read once (the algorithmic bytes of SURVEY.md 8(d) are unchanged), the serial
//     depth of a call is N+1 steps instead of 2(N+1), and the hand-over between the phases is a kernel boundary -- no
//     in-kernel waiting of one workgroup on another.
//
// An agent's steps are numbered t = 1, 2, ...: step t handles frame t (forward) or NF + 1 - t (backward), NF = the
// pair's frames = max(len) + 1.  Step 1 is the initial vector (alpha_hat (*) lhs[:,1], or one(K) at the final state);
// all double buffers go by the parity of t, the DMA rings by t % 3 / t & 3, in both directions alike.
// Numerics as in the row kernels (frame normaliser S_t chosen two steps ahead by the service wave, emission maximum
// E_t, range check + redo marks); the posterior of frame f is normalised by the frame's own sum exactly like the
// reference (:157-158), and log Z of a frame = log2(sum) + the two agents' cumulative offsets (double), ttl = the
// minimum over the frames (:159).
#pragma once
#include "mm_kernel_rows.hip"

namespace mm {

typedef float mm_f32x2 __attribute__((ext_vector_type(2)));
// (two slot-table words at once.  Not ldsr2() + __builtin_bit_cast(unsigned, v.y): this clang folds the bit cast of an
// ext_vector ELEMENT to the vector's first element -- checked in isolation, ROCm 7.2)
typedef unsigned mm_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mm_u32x2 ldsr2u(unsigned addr) { return *(__attribute__((address_space(3))) const mm_u32x2 *)(__UINTPTR_TYPE__)addr; }
__device__ __forceinline__ mm_f32x2 ldsr2(unsigned addr) { return *(__attribute__((address_space(3))) const mm_f32x2 *)(__UINTPTR_TYPE__)addr; }
__device__ __forceinline__ void ldsw2(unsigned addr, float a, float b) {
    mm_f32x2 v = {a, b};
    *(__attribute__((address_space(3))) mm_f32x2 *)(__UINTPTR_TYPE__)addr = v;
}

// LDS byte layout of the pair kernels (absolute addresses).  RS2 = bytes of one pair vector; RSH = bytes of the rows ONE
// workgroup finishes (all of them: RS2; a set of them in the split kernels, H > 1) -- what the partner-row ring and the q
// vector hold.
// PC = the pdf capacity of the per-pdf arrays: 256, or 512 for the instances of more than 4 passes of 64 lanes over the pdfs
// (NJ > 4: graphs of 251 .. 506 pdfs).  Those pay for their arrays with the partner-row ring: NR = 2 vectors instead of 3
// -- a partner row is requested at the top of the step before the one that combines it, not two steps ahead.
constexpr int pair_pc(int NJ) { return NJ > 4 ? 64 * NJ : 256; }
// (teams of 8: the team's vector of pairs is 48 KB, twice: a ring of two partner vectors there as well)
constexpr int pair_nr(int RS, int PC) { return (PC > 256 || RS > 16384) ? 2 : 3; }
template <int RS, int PHASE, int RSH = 2 * RS, int PC = 256>
struct PairLay {
    static constexpr unsigned RS2 = 2 * RS;
    static constexpr unsigned PC4 = 4u * PC, PC8 = 8u * PC;
    static constexpr int NR = pair_nr(RS, PC);       // vectors of the partner-row ring
    static constexpr unsigned RAWS = 2u * PC4;       // bytes of one ring entry of the raw emissions (both utterances)
    static constexpr unsigned XPS = 2u * PC;         // (teams) floats of one slot of published per-pdf partial sums
    static constexpr unsigned PP(int par) { return unsigned(par) * RS2; }                          // p pairs [pos][2]
    static constexpr unsigned RAW(int k, int u) { return 2 * RS2 + unsigned(2 * k + u) * PC4; }    // raw emissions, k < 4
    static constexpr unsigned EM(int par) { return 2 * RS2 + 8u * PC4 + unsigned(par) * PC8; }     // [pdf][2] pairs
    static constexpr unsigned MS(int par) { return 2 * RS2 + 8u * PC4 + 2u * PC8 + unsigned(par) * 64u; }      // {S_0, S_1} of a step
    static constexpr unsigned OWN(int k) { return 2 * RS2 + 8u * PC4 + 2u * PC8 + 128u + unsigned(k) * 16u; }  // own offsets, k < 4: 2 doubles
    static constexpr unsigned XFLAG = 2 * RS2 + 8u * PC4 + 2u * PC8 + 192u;  // split kernels: != 0 when the whole team runs on one XCD
    // split kernels: the maxima of the vector a step writes, {utterance 0, 1} as unsigned (linear values are >= 0), gathered by
    // the waves that write it (LDS atomic max) and read -- then zeroed -- by the service wave a step later
    static constexpr unsigned MX(int par) { return 2 * RS2 + 8u * PC4 + 2u * PC8 + 200u + unsigned(par) * 8u; }
    // partner offsets, k < POFFN: requested with the partner row (two steps ahead, one with a ring of two rows), read when the
    // posteriors of the step are put out, two steps behind
    // (an entry is the double itself: its LDS-DMA runs with two lanes active -- with all 64 it wrote 256 bytes per entry, 2 to 4 KB
    // of LDS for a ring of numbers)
    static constexpr int POFFN = NR == 2 ? 4 : 8;
    static constexpr unsigned POFFB = 16u * POFFN;
    static constexpr unsigned POFF(int k, int u) { return 2 * RS2 + 8u * PC4 + 2u * PC8 + 256u + unsigned(2 * k + u) * 8u; }
    static constexpr unsigned PSUM(int par) { return 2 * RS2 + 8u * PC4 + 2u * PC8 + 256u + POFFB + unsigned(par) * PC8; }   // [pdf][2]
    static constexpr unsigned PDFSE = 2 * RS2 + 8u * PC4 + 4u * PC8 + 256u + POFFB;                // u16 [2 * P1]
    static constexpr unsigned FIX = PDFSE + PC4;
    static constexpr unsigned AL(int k) { return FIX + unsigned(k) * RSH; }                        // partner rows (phase B), k < NR
    static constexpr unsigned Q(int par) { return FIX + unsigned(NR) * RSH + unsigned(par) * RSH; }  // q pairs [qpos][2] (phase B)
    static constexpr unsigned SLOTS = PHASE ? FIX + unsigned(NR + 2) * RSH : FIX;
};
inline size_t pair_lds_bytes(int RS, int phase, int nslotrows, int RSH = 0, int PC = 256) {
    const size_t fix = size_t(4 * RS) + size_t(8 * 4 * PC) + size_t(4 * 8 * PC) + 256 + (pair_nr(RS, PC) == 2 ? 64 : 128) + size_t(4 * PC);
    return fix + (phase ? size_t(pair_nr(RS, PC) + 2) * size_t(RSH ? RSH : 2 * RS) : 0) + size_t(nslotrows) * 64 * 8;
}

// ---- split kernels (H > 1): a TEAM of H workgroups computes one direction of an utterance pair; workgroup h finishes the
// rows of set h (mm_rows.h make_rows_split) and the team exchanges its rows once per step through global memory:
//   * every finish also stores the pair of linear values of its row as ONE 8-byte write-through (sc1) store -- a granule
//     whose data is its own flag: the sign bits carry the step's tag (linear values are >= 0; exp2(-inf) = +0 becomes -0);
//   * the receiving workgroup polls the other sets' slots with sc1 loads until every granule carries the tag of the step,
//     strips the signs and writes the values into its own LDS vector; the step's barrier then releases the compute waves
//     with the complete vector.  No flags, no fences, no ordering between the stores (cdna_hip_programming.md guideline
//     16, form R2).  WHO polls: the compute waves, one chunk of 128 granules per wave, after their own arcs (MM_SPLIT_CWPOLL;
//     an exchange wave sweeping all chunks was the longest wave of every step).  The EXCHANGE wave (one more service wave)
//     finds out at the start of a launch whether the team shares an XCD, and in phase B puts out the posteriors of a frame
//     (the other sets' partial pdf sums are a memory round trip it has the time for);
//   * slots alternate by the parity of the step, the tag flips with every reuse of a slot: a slot still holding the step
//     before last shows the other tag.  A workgroup cannot be more than one step ahead of its team (it needs the others'
//     rows of every step), so two slots suffice.  The buffers are zeroed before every call (tag 0, first tag used: 1);
//   * a poll gives up after MM_SPLIT_TIMEOUT (the team is not co-resident, e.g. a foreign kernel holds the compute
//     units): the utterances are marked for the exact kernels and the workgroup runs on without waiting.
typedef unsigned long long mm_u64;
typedef __attribute__((address_space(1))) mm_u64 mm_gu64;
#ifndef MM_SPLIT_TIMEOUT
#ifndef MM_PAIR_LINFIN
#define MM_PAIR_LINFIN 1
#endif
#ifndef MM_PAIR_SCAN_BATCH
#define MM_PAIR_SCAN_BATCH 8  // 16-byte LDS reads the service wave's scan keeps in flight
#endif
#define MM_LINF_EMIN (-40.f)  // log2 of the smallest emission factor of a step that raises no mark (pair_stage_em)
#define MM_SPLIT_TIMEOUT 10000000ull  // ticks of s_memrealtime (100 MHz): 0.1 s -- the ceiling; a call passes its own (RunParams::x_timeout)
#endif
__device__ __forceinline__ void granule_store(float *base, unsigned byte_off, float a, float b) {
    const mm_u64 v = ((mm_u64)__builtin_bit_cast(unsigned, b) << 32) | __builtin_bit_cast(unsigned, a);
    __hip_atomic_store((mm_gu64 *)(__UINTPTR_TYPE__)(reinterpret_cast<char *>(base) + byte_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ mm_u64 granule_load(const float *base, unsigned byte_off) {
    return __hip_atomic_load((mm_gu64 *)(__UINTPTR_TYPE__)(reinterpret_cast<const char *>(base) + byte_off), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
// tag of step t (steps of a launch are numbered from t0 + 1) for a ring of 1 << LG slots: 1 for the first use of a slot
__device__ __forceinline__ unsigned split_tag(int t, int t0, int LG) { return (((unsigned)(t - t0 - 1) >> LG) & 1u) ^ 1u; }

// Exchange wave: the rows of set g of step t -> LDS.  src: that set's slot of the step (n granules), dst: LDS byte address of
// the set's region in the vector being written.  Sweeps the granules that have not arrived -- two per lane and load (16-byte
// sc1 loads: half the memory instructions of 8-byte ones, and what the hand-off costs sits in this compute unit's memory
// queue) -- until all carry `tag`.  Returns false on timeout.  NG2 = 16-byte loads per lane (128 * NG2 >= n).
template <int NG2>
__device__ __forceinline__ bool split_receive(const float *src, int n, unsigned dst, unsigned tag, int lane, unsigned first_sleep) {
    typedef unsigned mm_u32x4 __attribute__((ext_vector_type(4)));
    mm_u32x4 v[NG2];
    unsigned pending = 0;  // bit j: the granules 2 (lane + 64 j), + 1 of this lane have not both arrived
#pragma unroll
    for (int j = 0; j < NG2; ++j) {
        v[j] = mm_u32x4{0u, 0u, 0u, 0u};
        if (2 * (lane + 64 * j) < n) pending |= 1u << j;
    }
    for (unsigned i = 0; i < first_sleep; ++i) __builtin_amdgcn_s_sleep(1);
    const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
    for (;;) {
#pragma unroll
        for (int j = 0; j < NG2; ++j)
            if ((pending >> j) & 1u)
                asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v[j]) : "v"(16u * (unsigned)(lane + 64 * j)), "s"(src) : "memory");
        // (the loads are inline asm: the wait carries every destination, so that no use is scheduled in front of it)
#pragma unroll
        for (int j = 0; j < NG2; ++j) asm volatile("" : "+v"(v[j]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NG2; ++j) asm volatile("" : "+v"(v[j]));
#pragma unroll
        for (int j = 0; j < NG2; ++j)
            if ((pending >> j) & 1u) {
                const bool second = 2 * (lane + 64 * j) + 1 < n;
                // (one store wrote both floats of a granule: one sign tells)
                if ((v[j].x >> 31) == tag && (!second || (v[j].z >> 31) == tag)) {
                    mm_f32x4 w;
                    w.x = __builtin_bit_cast(float, v[j].x & 0x7fffffffu);
                    w.y = __builtin_bit_cast(float, v[j].y & 0x7fffffffu);
                    w.z = second ? __builtin_bit_cast(float, v[j].z & 0x7fffffffu) : 0.f;
                    w.w = second ? __builtin_bit_cast(float, v[j].w & 0x7fffffffu) : 0.f;
                    *(__attribute__((address_space(3))) mm_f32x4 *)(__UINTPTR_TYPE__)(dst + 16u * (unsigned)(lane + 64 * j)) = w;
                    pending &= ~(1u << j);
                }
            }
        if (__builtin_amdgcn_ballot_w64(pending != 0u) == 0ull) return true;
        if (__builtin_amdgcn_s_memrealtime() - tstart > MM_SPLIT_TIMEOUT) return false;
        __builtin_amdgcn_s_sleep(4);
    }
}

// The same for teams of more than two: the rows of ALL the other sets of the step in ONE pipelined sweep -- the items (set,
// chunk of 128 granules) are loaded in a fixed order with K loads in flight, each checked and written as it returns; a
// sweep is repeated until every granule has carried the tag.  One set after the other (split_receive per set) cost a team
// of 4 three memory round trips per step behind the arrival of the last row: 12 k cycles per step against 2.5 k of
// arithmetic (cycle stamps).  Every item is loaded in every sweep, pending or not (clamped to the set's first granule
// where a lane has none): the number of loads in flight is then a constant of the code, which the counted waits need.
template <int N>
__device__ __forceinline__ void split_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int NG2, int NP, int K>
__device__ __forceinline__ bool split_receive_multi(const float *const (&src)[NP], const int (&n)[NP], const unsigned (&dst)[NP], unsigned tag,
                                                    int lane, unsigned first_sleep) {
    typedef unsigned mm_u32x4 __attribute__((ext_vector_type(4)));
    constexpr int I = NP * NG2;
    static_assert(I <= 32 && K <= I && K <= 12, "pending mask / wait constants");
    mm_u32x4 v[K];
    unsigned pending = 0u;  // bit i = item (set i / NG2, chunk i % NG2): the granules 2 (lane + 64 chunk), + 1 of the set have not both arrived
#pragma unroll
    for (int i = 0; i < I; ++i)
        if (2 * (lane + 64 * (i % NG2)) < n[i / NG2]) pending |= 1u << i;
#pragma unroll
    for (int s = 0; s < K; ++s) v[s] = mm_u32x4{0u, 0u, 0u, 0u};
    for (unsigned i = 0; i < first_sleep; ++i) __builtin_amdgcn_s_sleep(1);
    const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
    for (;;) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int g = i / NG2, j = i % NG2;
            const unsigned off = 2 * (lane + 64 * j) < n[g] ? 16u * (unsigned)(lane + 64 * j) : 0u;
            asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v[i]) : "v"(off), "s"(src[g]) : "memory");
        }
#pragma unroll
        for (int i = 0; i < I; ++i) {
            const int g = i / NG2, j = i % NG2, slot = i % K;
            // loads issued so far: min(i + K, I); item i is the oldest of the ones not yet looked at
            constexpr int dummy = 0;
            (void)dummy;
            if (i + K <= I) {
                split_wait_vm<K - 1>();
            } else {
                switch (I - i - 1) {
                    case 11: split_wait_vm<11>(); break;
                    case 10: split_wait_vm<10>(); break;
                    case 9: split_wait_vm<9>(); break;
                    case 8: split_wait_vm<8>(); break;
                    case 7: split_wait_vm<7>(); break;
                    case 6: split_wait_vm<6>(); break;
                    case 5: split_wait_vm<5>(); break;
                    case 4: split_wait_vm<4>(); break;
                    case 3: split_wait_vm<3>(); break;
                    case 2: split_wait_vm<2>(); break;
                    case 1: split_wait_vm<1>(); break;
                    default: split_wait_vm<0>(); break;
                }
            }
            asm volatile("" : "+v"(v[slot]));
            if ((pending >> i) & 1u) {
                const bool second = 2 * (lane + 64 * j) + 1 < n[g];
                if ((v[slot].x >> 31) == tag && (!second || (v[slot].z >> 31) == tag)) {
                    mm_f32x4 w;
                    w.x = __builtin_bit_cast(float, v[slot].x & 0x7fffffffu);
                    w.y = __builtin_bit_cast(float, v[slot].y & 0x7fffffffu);
                    w.z = second ? __builtin_bit_cast(float, v[slot].z & 0x7fffffffu) : 0.f;
                    w.w = second ? __builtin_bit_cast(float, v[slot].w & 0x7fffffffu) : 0.f;
                    *(__attribute__((address_space(3))) mm_f32x4 *)(__UINTPTR_TYPE__)(dst[g] + 16u * (unsigned)(lane + 64 * j)) = w;
                    pending &= ~(1u << i);
                }
            }
            if (i + K < I) {  // the slot is free: the next item of the sweep
                const int g2 = (i + K) / NG2, j2 = (i + K) % NG2;
                const unsigned off = 2 * (lane + 64 * j2) < n[g2] ? 16u * (unsigned)(lane + 64 * j2) : 0u;
                asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v[slot]) : "v"(off), "s"(src[g2]) : "memory");
            }
        }
        if (__builtin_amdgcn_ballot_w64(pending != 0u) == 0ull) return true;
        if (__builtin_amdgcn_s_memrealtime() - tstart > MM_SPLIT_TIMEOUT) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

struct PairUtt {  // one utterance of the pair (scalar registers)
    const float *Vb;
    double *offs;     // [N + 2] cumulative offset of the stored vector of every frame
    int b, len, valid;
};

// emissions of one frame for utterance u of the pair: raw values from LDS (DMA), log2 domain relative to the frame's
// maximum E (row_stage_em for interleaved pairs).  Returns E.
// LIN: the LINEAR factor of the step's finishes instead, 2^(v - E - S) with the step's normaliser S folded in -- a finish is then
// one multiplication (no v_log_f32 / v_exp_f32 in the compute waves: quarter-rate instructions, 4 to 6 per finish until round 5).
template <int NJ, bool LIN = false>  // NJ * 64 >= P + 1
__device__ __forceinline__ float pair_stage_em(unsigned dst, unsigned rawsrc, int u, int n, int len, int P, int lane, float S = 0.f, int *mark = nullptr) {
    float v[NJ], E = MM_NINF;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        v[j] = em_value(ldsr(rawsrc + 256u * j + 4u * lane), n, len, P, q);
        if (q < P) E = max_nc(E, v[j]);
    }
    E = wave_max_rl(E);
    if (!(E > MM_NINF)) E = 0.f;
    bool tiny = false;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        if (q <= P) ldsw(dst + 8u * q + 4u * u, LIN ? fast_exp2(v[j] - E - S) : v[j] - E);
        // (a finite emission whose factor is about to leave the float range: if a whole vector dies of it the utterance looks like
        // one without a path -- the mark tells mm_pair_finish_kernel not to believe that)
        if (LIN) tiny = tiny || (q <= P && v[j] - E - S < MM_LINF_EMIN && v[j] > MM_NINF);
    }
    if (LIN && tiny) *mark = 1;
    return E;
}

// two wave-wide maxima at once: the DPP steps of the two chains alternate (each fills the other's wait states), then the 4 row
// results of each through readlane
__device__ __forceinline__ void wave_max_rl2(float &a, float &b) {
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1"
        : "+v"(a), "+v"(b));
    const int ia = __builtin_bit_cast(int, a), ib = __builtin_bit_cast(int, b);
    const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ia, 0)), b0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ib, 0));
    const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ia, 16)), b1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ib, 16));
    const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ia, 32)), b2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ib, 32));
    const float a3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ia, 48)), b3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ib, 48));
    a = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
    b = fmaxf(fmaxf(b0, b1), fmaxf(b2, b3));
}

// the wave-wide maximum of max(a, b) with ONE scalar result (lane 63 after two row broadcasts; wave_max_rl2 takes eight scalar
// registers for its readlanes, which the instances of the large teams do not have)
__device__ __forceinline__ float wave_max_of2(float a, float b) {
    float v = max_nc(a, b);
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// pair_stage_em<NJ, true> for BOTH utterances in one pass (up to 4 passes of 64 pdfs): all raw values are read before anything is
// written -- called once per utterance, the second call's LDS reads wait behind the first call's LDS writes (the compiler cannot
// tell them apart), and the service wave's own chain of round trips is what bounds a step since the finishes went linear --, the
// two maxima share one DPP ladder, and a pdf's two factors leave in ONE 8-byte write.
// (ADAPT: X = minus the smallest finite log2 factor of the step; else a fixed bound, MM_LINF_EMIN, and marks for what is below it)
template <int NJ, bool ADAPT>
__device__ __forceinline__ void pair_stage_em2(unsigned dst, unsigned raw0, unsigned raw1, int n, int len0, int len1, int P, int lane, float S0,
                                               float S1, float (&E)[2], float &X, int *mark0, int *mark1) {
    float v0[NJ], v1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        v0[j] = ldsr(raw0 + 256u * j + 4u * lane);
        v1[j] = ldsr(raw1 + 256u * j + 4u * lane);
    }
    float e0 = MM_NINF, e1 = MM_NINF;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        v0[j] = em_value(v0[j], n, len0, P, q);
        v1[j] = em_value(v1[j], n, len1, P, q);
        if (q < P) {
            e0 = max_nc(e0, v0[j]);
            e1 = max_nc(e1, v1[j]);
        }
    }
    wave_max_rl2(e0, e1);
    if (!(e0 > MM_NINF)) e0 = 0.f;
    if (!(e1 > MM_NINF)) e1 = 0.f;
    // X: minus the smallest finite log2 factor of the step (0 if there is none) -- how much of the float range the emissions take
    float n0 = 0.f, n1 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        if (q <= P) {
            const float x0 = v0[j] - e0 - S0, x1 = v1[j] - e1 - S1;
            ldsw2(dst + 8u * (unsigned)q, fast_exp2(x0), fast_exp2(x1));
            n0 = max_nc(n0, v0[j] > MM_NINF ? -x0 : 0.f);
            n1 = max_nc(n1, v1[j] > MM_NINF ? -x1 : 0.f);
        }
    }
    if constexpr (ADAPT) {
        X = wave_max_of2(n0, n1);
    } else {
        if (n0 > -MM_LINF_EMIN) *mark0 = 1;
        if (n1 > -MM_LINF_EMIN) *mark1 = 1;
    }
    E[0] = e0;
    E[1] = e1;
}

// The same for BOTH utterances at once with 16-byte reads: a lane takes 4 consecutive pdfs per pass (the instances of more
// than 4 passes of 64 pdfs, whose service wave is the longest actor of every step: 2000 states / 400 pdfs spent 3100 of a
// 4800-cycle phase-A step here with 32 four-byte LDS accesses; now 4 reads of 16 bytes and 8 writes of 8).
template <int NJ, bool LIN = false>
__device__ __forceinline__ void pair_stage_em_wide(unsigned dst, unsigned raw0, unsigned raw1, int n, int len0, int len1, int P, int lane,
                                                   float (&E)[2], float S0 = 0.f, float S1 = 0.f, int *mark0 = nullptr, int *mark1 = nullptr) {
    constexpr int NP = (NJ + 3) / 4;
    mm_f32x4 v0[NP], v1[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const unsigned o = 4u * (unsigned)(256 * j + 4 * lane);
        v0[j] = *(__attribute__((address_space(3))) const mm_f32x4 *)(__UINTPTR_TYPE__)(raw0 + o);
        v1[j] = *(__attribute__((address_space(3))) const mm_f32x4 *)(__UINTPTR_TYPE__)(raw1 + o);
    }
    float e0 = MM_NINF, e1 = MM_NINF;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = 256 * j + 4 * lane + i;
            v0[j][i] = em_value(v0[j][i], n, len0, P, q);
            v1[j][i] = em_value(v1[j][i], n, len1, P, q);
            if (q < P) {
                e0 = max_nc(e0, v0[j][i]);
                e1 = max_nc(e1, v1[j][i]);
            }
        }
    }
    wave_max_rl2(e0, e1);
    if (!(e0 > MM_NINF)) e0 = 0.f;
    if (!(e1 > MM_NINF)) e1 = 0.f;
    bool tiny0 = false, tiny1 = false;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = 256 * j + 4 * lane + i;
            if (q <= P) {
                if constexpr (LIN) {
                    ldsw2(dst + 8u * (unsigned)q, fast_exp2(v0[j][i] - e0 - S0), fast_exp2(v1[j][i] - e1 - S1));
                    tiny0 = tiny0 || (v0[j][i] - e0 - S0 < MM_LINF_EMIN && v0[j][i] > MM_NINF);
                    tiny1 = tiny1 || (v1[j][i] - e1 - S1 < MM_LINF_EMIN && v1[j][i] > MM_NINF);
                } else ldsw2(dst + 8u * (unsigned)q, v0[j][i] - e0, v1[j][i] - e1);
            }
        }
    }
    if (LIN && tiny0) *mark0 = 1;
    if (LIN && tiny1) *mark1 = 1;
    E[0] = e0;
    E[1] = e1;
}

// service wave: log2 of the maxima of both linear vectors (pairs [pos][2]; n2 float4s = 2 states each, n2 <= 64 * NB).
// All loads are issued before the first maximum (clamped indices: a duplicate changes no maximum) -- a loop with one
// load per trip costs the wave one LDS round trip per trip, and the service wave is the one wave whose own latency
// chain every step waits for.
template <int NB, int BATCH = 4>
__device__ __forceinline__ void pair_scan_max(unsigned pbase, int n2, int lane, float &m0, float &m1) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j0 = 0; j0 < NB; j0 += BATCH) {  // (batches of 4 loads: 16 registers; more spill in the phase B kernels)
        mm_f32x4 v[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int q = lane + 64 * (j0 + j);
            if (j0 + j < NB) v[j] = *(__attribute__((address_space(3))) const mm_f32x4 *)(__UINTPTR_TYPE__)(pbase + 16u * (q < n2 ? q : n2 - 1));
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            if (j0 + j < NB) {
                a = max_nc(a, max_nc(v[j].x, v[j].z));
                b = max_nc(b, max_nc(v[j].y, v[j].w));
            }
        }
    }
    wave_max_rl2(a, b);
    m0 = fast_log2(a);
    m1 = fast_log2(b);
}

// wave-wide sum without the LDS crossbar: 16-lane rows by DPP, then the 4 row results through readlane
__device__ __forceinline__ float wave_sum_rl(float v) {
    v += dpp_mov<MM_DPP_XOR1>(v);
    v += dpp_mov<MM_DPP_XOR2>(v);
    v += dpp_mov<MM_DPP_HALF_MIRROR>(v);
    v += dpp_mov<MM_DPP_MIRROR>(v);
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// one wave, both utterances of the pair: per-frame sums over the pdfs (psum pairs [pdf][2]), divide, store gamma
// (src/inference.jl:156-160); lt[u] = log2 of the sum (-inf, and gamma = 0, if nothing is alive)
// (split kernels: xp[g] = the slot of the step in which set g of the team published its partial sums, NULL for the own
// set; the sum of a pdf is the sum of the sets' partial sums in the order of the sets -- the same bits in every workgroup)
template <int NJ, int H = 1>  // NJ * 64 >= P + 1
__device__ __forceinline__ bool pair_finish_frames(unsigned psum, int P1, int P, int lane, float *gp0, float *gp1, long long gsp,
                                                   bool store0, bool store1, float (&lt)[2], const float *const *xp = nullptr,
                                                   unsigned tag = 0u, unsigned long long tmo = MM_SPLIT_TIMEOUT) {
    mm_f32x2 s[NJ];
    float t0 = 0.f, t1 = 0.f;
    bool arrived = true;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        s[j] = ldsr2(psum + 8u * (q < P1 ? q : 0));
    }
    if constexpr (H > 1) {
        // the other sets' partial sums: requested together, four sets at a time, before the first is looked at (one memory round
        // trip per frame for teams of up to 4, two for teams of 8 -- not one per set; all seven sets of a team of 8 at once do
        // not fit the registers), summed in the order of the sets -- the same bits in every workgroup
#ifndef MM_XPS_GB
#define MM_XPS_GB 4
#endif
#ifndef MM_XPS_GB2
#define MM_XPS_GB2 1
#endif
        constexpr int GB = (NJ <= 2 && MM_XPS_GB2) ? H : (H < MM_XPS_GB ? H : MM_XPS_GB);  // (up to 128 pdfs: the registers hold all seven sets)
        mm_f32x2 tot[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) tot[j] = mm_f32x2{0.f, 0.f};
#pragma unroll
        for (int g0 = 0; g0 < H; g0 += GB) {
            mm_u64 v[GB][NJ];
            const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int g = 0; g < GB; ++g) {
                    if (xp[g0 + g] == nullptr) continue;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int q = lane + 64 * j;
                        v[g][j] = granule_load(xp[g0 + g], 8u * (unsigned)(q < P1 ? q : 0));
                    }
                }
#pragma unroll
                for (int g = 0; g < GB; ++g) {
                    if (xp[g0 + g] == nullptr) continue;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) ok = ok && ((unsigned)(v[g][j] >> 31) % 2u == tag);
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (!arrived || __builtin_amdgcn_s_memrealtime() - tstart >= tmo) {
                    arrived = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int g = 0; g < GB; ++g) {
                if (xp[g0 + g] == nullptr) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) tot[j] += s[j];
                    continue;
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    tot[j].x += __builtin_bit_cast(float, (unsigned)v[g][j] & 0x7fffffffu);
                    tot[j].y += __builtin_bit_cast(float, (unsigned)(v[g][j] >> 32) & 0x7fffffffu);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) s[j] = tot[j];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        if (q < P1) {
            t0 += s[j].x;
            t1 += s[j].y;
        }
    }
    t0 = wave_sum_rl(t0);
    t1 = wave_sum_rl(t1);
    const float i0 = t0 > 0.f ? 1.f / t0 : 0.f, i1 = t1 > 0.f ? 1.f / t1 : 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        if (q < P) {
            if (store0) gp0[q * gsp] = s[j].x * i0;
            if (store1) gp1[q * gsp] = s[j].y * i1;
        }
    }
    lt[0] = fast_log2(t0);
    lt[1] = fast_log2(t1);
    return arrived;
}

// The same with 16-byte reads (a lane takes 2 consecutive pdfs of both utterances per pass) and, where the rows of gamma
// allow it (pdfs contiguous, rows 8-byte aligned: `wide`), 8-byte stores: the instances of more than 4 passes, H = 1.
template <int NJ>
__device__ __forceinline__ void pair_finish_frames_wide(unsigned psum, int P1, int P, int lane, float *gp0, float *gp1, long long gsp, bool store0,
                                                        bool store1, bool wide, float (&lt)[2]) {
    constexpr int NP = (NJ + 1) / 2;
    mm_f32x4 s[NP];
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int q = 128 * j + 2 * lane;
        s[j] = *(__attribute__((address_space(3))) const mm_f32x4 *)(__UINTPTR_TYPE__)(psum + 8u * (unsigned)(q < P1 ? q : 0));
        if (q < P1) {
            t0 += s[j].x;
            t1 += s[j].y;
        }
        if (q + 1 < P1) {
            t0 += s[j].z;
            t1 += s[j].w;
        }
    }
    t0 = wave_sum_rl(t0);
    t1 = wave_sum_rl(t1);
    const float i0 = t0 > 0.f ? 1.f / t0 : 0.f, i1 = t1 > 0.f ? 1.f / t1 : 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int q = 128 * j + 2 * lane;
        if (wide && q + 1 < P) {
            if (store0) *reinterpret_cast<mm_f32x2 *>(gp0 + q) = mm_f32x2{s[j].x * i0, s[j].z * i0};
            if (store1) *reinterpret_cast<mm_f32x2 *>(gp1 + q) = mm_f32x2{s[j].y * i1, s[j].w * i1};
        } else {
            if (q < P) {
                if (store0) gp0[q * gsp] = s[j].x * i0;
                if (store1) gp1[q * gsp] = s[j].y * i1;
            }
            if (q + 1 < P) {
                if (store0) gp0[(q + 1) * gsp] = s[j].z * i0;
                if (store1) gp1[(q + 1) * gsp] = s[j].w * i1;
            }
        }
    }
    lt[0] = fast_log2(t0);
    lt[1] = fast_log2(t1);
}

// pdf sums of both utterances (q pairs in pdf-major order): 8 pdfs per wave and pass, 8 lanes per pdf
// (split kernels: xs = the slot the partial sums are published in for the team, sg = +-1 carrying the tag)
// (LG: log2 of the lanes per pdf -- 8 lanes, 8 pdfs per wave and pass; the instances of more than 256 pdfs, whose pdfs have a
// handful of states each: 2 lanes, 32 pdfs per wave and pass -- 400 pdfs in one pass of the 15 compute waves instead of four;
// 129 .. 250 pdfs: 4 lanes)
template <int LG = 3>
__device__ __forceinline__ void pair_pdf_sums(unsigned qbase, unsigned pdfse_base, unsigned psum_base, int P1, int wave, int NWC, int lane,
                                              float *xs = nullptr, float sg = 1.f) {
    constexpr int LP = 1 << LG, PPW = 64 >> LG;
    constexpr unsigned STR = 8u * LP;
    for (int p0 = wave * PPW; p0 < P1; p0 += NWC * PPW) {
        const int pdf = p0 + (lane >> LG);
        float s0 = 0.f, s1 = 0.f;
        if (pdf < P1) {
            const unsigned se = ldsru(pdfse_base + 4u * pdf);  // first | end << 16
            const unsigned a0 = 8u * ((se & 0xffffu) + (lane & (LP - 1))), a1 = 8u * (se >> 16);
            mm_f32x2 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ldsr2(qbase + (a0 + STR * k < a1 ? a0 + STR * k : 0u));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (a0 + STR * k < a1) {
                    s0 += v[k].x;
                    s1 += v[k].y;
                }
            for (unsigned a = a0 + 4u * STR; a < a1; a += STR) {
                const mm_f32x2 w = ldsr2(qbase + a);
                s0 += w.x;
                s1 += w.y;
            }
        }
        s0 = grp_sum(s0, LG);
        s1 = grp_sum(s1, LG);
        if (pdf < P1 && (lane & (LP - 1)) == 0) {
            ldsw2(psum_base + 8u * pdf, s0, s1);
            if (xs) granule_store(xs, 8u * (unsigned)pdf, s0 * sg, s1 * sg);
        }
    }
}

// The arcs of a compute wave for two utterances (8-byte gathers, D pairs of arcs ahead of the FMAs), statically unrolled
// (the graph registers need static indices).  A segment may end after any pair of arcs; pair_two() / MM_PAIR_TWO test
// the end mask once per TWO pairs.  Every wave runs the whole window: the pairs behind its last segment have weight 0
// (a few wasted gathers); leaving early costs more -- a `goto` out of the rare path made the compiler add an
// s_cbranch_execz to every test.
// acc += w (broadcast) * x for the two utterances in ONE packed FMA (v_pk_fma_f32 issues at the rate of v_fma_f32 on
// gfx950): the weights of an arc pair share an aligned register pair, op_sel picks the half that both lanes of the
// packed operation use.
__device__ __forceinline__ unsigned min3_u32(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void pk_fma_wlo(mm_f32x2 &acc, const mm_f32x2 &w2, const mm_f32x2 &x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w2), "v"(x));
}
__device__ __forceinline__ void pk_fma_whi(mm_f32x2 &acc, const mm_f32x2 &w2, const mm_f32x2 &x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w2), "v"(x));
}

__device__ __forceinline__ void pk_mul_wlo(mm_f32x2 &acc, const mm_f32x2 &w2, const mm_f32x2 &x) {
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "v"(w2), "v"(x));
}

template <int KA>
struct PairRegs {  // what a compute wave keeps across the steps
    mm_f32x2 w2[KA / 2];
    unsigned a[KA];
};

template <int K2, int KA, int D, bool PK>
__device__ __forceinline__ void pair_one(const mm_f32x2 (&wr)[KA / 2], const unsigned (&ar)[KA], mm_f32x2 (&x)[2 * D], mm_f32x2 &accA,
                                         mm_f32x2 &accB, unsigned rdoff) {
    constexpr int s0 = (2 * K2) % (2 * D);
    if constexpr (PK) {
        pk_fma_wlo(accA, wr[K2], x[s0]);
        pk_fma_whi(accB, wr[K2], x[s0 + 1]);
    } else {  // (phase B has no room for the aligned register pairs of the packed form)
        accA.x = fmaf(wr[K2].x, x[s0].x, accA.x);
        accA.y = fmaf(wr[K2].x, x[s0].y, accA.y);
        accA.x = fmaf(wr[K2].y, x[s0 + 1].x, accA.x);
        accA.y = fmaf(wr[K2].y, x[s0 + 1].y, accA.y);
    }
    if constexpr (2 * (K2 + D) < KA) {
        x[s0] = ldsr2(ar[2 * (K2 + D)] + rdoff);
        x[s0 + 1] = ldsr2(ar[2 * (K2 + D) + 1] + rdoff);
    }
}
// Two pairs whose products are all in straight-line code: pair K2 goes to the running sums, pair K2 + 1 to sums of
// its own (accN), which the caller adds to the running sums unless a segment ends between the two.
template <int K2, int KA, int D>
__device__ __forceinline__ void pair_two(const mm_f32x2 (&wr)[KA / 2], const unsigned (&ar)[KA], mm_f32x2 (&x)[2 * D], mm_f32x2 &accA,
                                         mm_f32x2 &accN, unsigned rdoff) {
    constexpr int s0 = (2 * K2) % (2 * D), s1 = (2 * K2 + 2) % (2 * D);
    pk_fma_wlo(accA, wr[K2], x[s0]);
    pk_mul_wlo(accN, wr[K2 + 1], x[s1]);
    pk_fma_whi(accA, wr[K2], x[s0 + 1]);
    pk_fma_whi(accN, wr[K2 + 1], x[s1 + 1]);
#ifdef MM_PAIR_DUMMY
    {   // timing experiment (never shipped): the instruction mix of a two-utterance float64 form -- per arc one more 64-bit FMA and
        // a move next to the (here: packed) FMA that exists; results go nowhere
        double da;
        unsigned dt;
        asm volatile("" : "=v"(da), "=v"(dt));
        asm volatile("v_mov_b32 %1, %6\n\tv_fma_f64 %0, %2, %3, %0\n\tv_mov_b32 %1, %7\n\tv_fma_f64 %0, %2, %4, %0\n\t"
                     "v_mov_b32 %1, %6\n\tv_fma_f64 %0, %5, %3, %0\n\tv_mov_b32 %1, %7\n\tv_fma_f64 %0, %5, %4, %0"
                     : "+v"(da), "+v"(dt)
                     : "v"(wr[K2]), "v"(x[s0]), "v"(x[s0 + 1]), "v"(wr[K2 + 1]), "v"(x[s1].x), "v"(x[s1 + 1].x));
    }
#endif
    if constexpr (2 * (K2 + D) < KA) {
        x[s0] = ldsr2(ar[2 * (K2 + D)] + rdoff);
        x[s0 + 1] = ldsr2(ar[2 * (K2 + D) + 1] + rdoff);
    }
    if constexpr (2 * (K2 + 1 + D) < KA) {
        x[s1] = ldsr2(ar[2 * (K2 + 1 + D)] + rdoff);
        x[s1 + 1] = ldsr2(ar[2 * (K2 + 1 + D) + 1] + rdoff);
    }
}
#define MM_PAIR_FMA(k) pair_one<(2 * (k) < KA ? (k) : 0), KA, D, PHASE == 0>(rg.w2, rg.a, x, accA, accB, rdoff);
#define MM_PAIR_END(k) ((((k) < 32 ? em_lo : em_hi) >> ((k) & 31)) & 1u)
#define MM_PAIR_ONE(k)                   \
    if constexpr (2 * (k) < KA) {        \
        MM_PAIR_FMA(k)                   \
        if (MM_PAIR_END(k)) finish();    \
    }
// Two pairs per wave-uniform test of the end mask (a taken branch costs a wave about 30 cycles with its bit test, as
// much as the pair's gathers): the bits are looked at one by one only when one of the two pairs ends a segment.
#define MM_PAIR_TWO(k)                                                                                       \
    if constexpr (2 * (k) + 2 < KA && MM_PAIR_DOUBLE) {                                                      \
        pair_two<(2 * (k) + 2 < KA ? (k) : 0), KA, D>(rg.w2, rg.a, x, accA, accN, rdoff);                    \
        if (__builtin_expect(((((k) < 32 ? em_lo : em_hi) >> ((k) & 31)) & 3u) != 0u, 0)) {                  \
            if (MM_PAIR_END(k)) finish();                                                                    \
            accA += accN;                                                                                    \
            accN = mm_f32x2{0.f, 0.f};                                                                       \
            if (MM_PAIR_END((k) + 1)) finish();                                                              \
        }                                                                                                    \
        accA += accN;                                                                                        \
    } else {                                                                                                 \
        MM_PAIR_ONE(k)                                                                                       \
        MM_PAIR_ONE((k) + 1)                                                                                 \
    }
// A wave lowers its issue priority as it advances through the step (MM_PAIR_PRIO): the arbiter serves the highest
// priority, then the OLDEST wave, so with equal priorities the four waves of a SIMD finish one after the other and the
// youngest runs the tail of the step alone, latency bound, while the LDS idles; with priorities that fall with the
// progress a wave that is ahead yields to the ones behind and all reach the barrier together.
#ifndef MM_PAIR_PRIO_A
#define MM_PAIR_PRIO_A 8
#define MM_PAIR_PRIO_B 16
#endif
#define MM_PAIR_PRIO(k, lvl) if constexpr (2 * (k) < KA) __builtin_amdgcn_s_setprio(lvl);
// (a wave whose last segment ends early leaves the window at one of three points -- plain nested blocks: an exit from
// inside the rare path made the compiler test exec at every pair; its remaining pairs would be gathers of weight 0,
// ~10 % of the gathers of config 3)
#define MM_PAIR_CASES(M)                                                                                     \
    M(0) M(2) M(4) M(6) MM_PAIR_PRIO(MM_PAIR_PRIO_A, 1) M(8) M(10) M(12)                                     \
    if (lastp >= 14) {                                                                                       \
        M(14) MM_PAIR_PRIO(MM_PAIR_PRIO_B, 0) M(16)                                                          \
        if (lastp >= 18) {                                                                                   \
            M(18)                                                                                            \
            if (lastp >= 20) { M(20) M(22) }                                                                 \
        }                                                                                                    \
    }

struct PairHand {  // what an agent hands from phase A to phase B, per utterance
    float m_prev, s_cur, s_prev, cbar;
    int seen, pad;
    double cum;
};

// One agent: direction DIR (0: forward / alpha, 1: backward / beta) of pair `pair`, phase PHASE (0: A, 1: B).
// H > 1 (split kernels): the agent is a team of H workgroups, this one finishes the rows of set `hset`.
// DIRT: the direction as a template constant, or -1: the direction is `rdir`, a run-time value that is the same for the whole
// workgroup -- one kernel then holds the forward and the backward agents of a launch (mm_pairs_tu.hip) in ONE body of code
// and ONE register allocation (the two specialised bodies behind a branch on blockIdx made the register allocator spill 50
// scalar registers in phase B: the loads of the parameters are hoisted above the branch and stay live through both bodies).
// What depends on the direction at run time is scalar arithmetic (frame_of, indices) and one select per finished row.
// XPT (phase A only; mm_fbx_kernel): the agent of ONE direction walks ALL N + 1 frames and stores every frame's vector -- the
// alpha-recursion / beta-recursion export (src/inference.jl:62-74, 99-110): forward p = alpha_n with the frame's emission, as the
// reference's alpha has it; backward s, the sum BEFORE the frame's emission and normaliser, as the reference's beta has it (:107:
// B[:, n] = T (B[:, n+1] (*) lhs[:, n+1])), with the offset that goes with it.  mm_pair_export_kernel turns the stored rows into the
// reference's layout in natural logarithms.
template <int KA, int RS, int PHASE, int DIRT, int NJ, int H = 1, int RSH = 2 * RS, bool SMALL = false, bool XPT = false>
__device__ __forceinline__ void pair_agent(const RunParams &p, int pair, int hset = 0, int rdir = 0) {
// (teams, tried in round 4 and left OFF: the waves that write a step's vector -- finishes and received rows -- gather its maxima
// with LDS atomics instead of the service wave scanning the team's whole vector (3300 .. 4200 cycles of its step for teams of
// 8).  The ~40 instructions per compute wave and step cost more than the scan they replace, which is not on the critical
// path: 5000 states / 100 pdfs 7.7 -> 8.7 ms, the 4000-state graph 3.45 -> 3.63, 6000 / 300 9.7 -> 9.6, WSJ 1.79 -> 1.78.)
#ifndef MM_SPLIT_MAXTRACK
#define MM_SPLIT_MAXTRACK 0
#endif
    extern __shared__ float lds[];
    const int DIR = DIRT < 0 ? __builtin_amdgcn_readfirstlane(rdir) : DIRT;
    const unsigned long long x_tmo = p.x_timeout;  // (teams) ticks of s_memrealtime a poll waits before it gives the team up
    using L = PairLay<RS, PHASE, RSH, pair_pc(NJ)>;
#ifndef MM_PAIR_DA
#define MM_PAIR_DA 3
#endif
#ifndef MM_PAIR_DB
#define MM_PAIR_DB 3
#endif
#ifndef MM_PAIR_EXITS_B
#define MM_PAIR_EXITS_B 1  // phase B leaves the arc window after the wave's last segment like phase A (until the finishes went linear: 8 spilled registers)
#endif
    constexpr bool MM_PAIR_DOUBLE = true;
    // LINF: the finishes stay in the linear domain (round 5): the service wave stages the step's emissions as factors 2^(v - E - S),
    // a finish is p = s * factor (and, in phase B, q = s * partner), BOTH directions store p -- the vector with the frame's emission --
    // and combine with their own s (the vector without it): alpha_n beta_n = (T' alpha_{n-1}) (beta_n (*) lhs_n) either way.
    // Until then a finish went through log2: v_log_f32 of the sum, the normaliser and the emission added, v_exp_f32 back, and a
    // second v_exp_f32 for the combine -- 4 (phase A) / 6 (phase B) quarter-rate instructions per finish for the two utterances,
    // ~1100 of a SIMD's ~3000 busy cycles per phase-B step (SQ_INSTS_VALU, profiles/r05_pmc_lfmmi_den.json).
    // (not the instances of more than 4 passes over the pdfs, 251 .. 506 pdfs: their service wave bounds the step, and 16 more
    // exponentials per lane and step in it cost more than the compute waves save -- 2000 states / 400 pdfs: 4.65 against 3.62 ms)
    constexpr bool LINF = MM_PAIR_LINFIN != 0 && NJ <= 4;
    // The range check of the linear finishes: the smallest non-zero sum a step accepts.  ADAPT: from the step's emissions (the service
    // wave posts it: stage()); the teams' instances, which have no register for anything more: a fixed split of the range, sums down to
    // 2^-(thr + MM_LINF_EMIN), factors down to 2^MM_LINF_EMIN (smaller ones are marked by the service wave).
    constexpr bool ADAPT = LINF && H == 1;
    constexpr int D = PHASE ? MM_PAIR_DB : MM_PAIR_DA;  // gather pairs in flight ahead of the FMAs
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6, NWC = NW - (H > 1 ? 2 : 1);
    const bool service = wave == NWC;
    const bool xwave = H > 1 && wave == NWC + 1;  // the exchange wave of a team's workgroup
    if (service || xwave) __builtin_amdgcn_s_setprio(3);
    // ---- the two utterances
    PairUtt U[2];
    int NFp = 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = 2 * pair + u;
        const bool valid = i < p.B;
        const int ii = valid ? i : p.B - 1;  // (odd batch: the last pair runs its first utterance twice, the copy writes nothing outside the workspace)
        const int b = uni(p.order ? p.order[ii] : ii);
        int len = uni(p.lens ? p.lens[b] : p.N);
        len = len < 0 ? 0 : (len > p.N ? p.N : len);
        const int slot = valid ? b : p.B;  // workspace slot
        U[u].b = b;
        U[u].len = len;
        U[u].valid = valid;
        U[u].Vb = p.V + (long long)b * p.vsb;
        U[u].offs = p.ws_c + (long long)slot * (p.N + 2);
        NFp = len + 1 > NFp ? len + 1 : NFp;
    }
    static_assert(!XPT || PHASE == 0, "the export runs phase A");
    if constexpr (XPT) NFp = p.N + 1;  // (every frame of the reference's (sum S1) x (N + 1) matrix, whatever the lengths: expand() pads)
    // state vectors of the pair's frames (alpha~ up to the split, beta~ beyond), the two utterances side by side like
    // in LDS: [N + 2][S1p][2] -- one 8-byte store per finish in phase A, and in phase B ONE ds_read_b64 fetches both
    // partner values of a row (two 4-byte reads at the partner's scattered positions cost the backward agent 9 % of a step)
    float *rowsP = p.ws_alpha + (long long)pair * (long long)(p.N + 2) * 2 * p.pair_s1p;
    const UttDesc &ud = p.utts[U[0].b];
    const RowU r = uni(H > 1 ? ud.rps[DIR][hset] : ud.rp[DIR]);
    const int S1 = r.rows, S1p = p.pair_s1p, P1 = uni(ud.P1), P = P1 - 1, P1p = (P1 + 3) & ~3;
    const float thr = r.thr;
    unsigned sthr = 0u;  // (a scalar register by force)
    if constexpr (LINF && !ADAPT)
        asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sthr) : "v"(((unsigned)(127 - (int)(thr + MM_LINF_EMIN < 1.f ? 1.f : thr + MM_LINF_EMIN)) << 23) - 1u));
    // (Steering the frame maxima to 2^(thr - 20) instead of 2^0 -- to use the upper half of the float exponent range and keep
    // states up to ~190 log2 below the maximum on the linear path -- was tried and dropped: v_log_f32 returns log2 of sums
    // near 2^87 as floats 7.6e-6 apart, the per-frame normalisers of one utterance then scatter by 2e-4..7e-4 log2 instead of
    // ~1e-5, and the sharp emissions it was meant for -- log-softmax of 10 N(0,1) -- need 160 log2 in the COMBINE, which no
    // level gives: those inputs go to the exact kernels, mm_pair_finish_kernel.)
    // split: forward steps 1..m are phase A, backward steps 1..NFp-m
    int m = (int)(((long long)NFp * (p.split_q10 > 0 ? p.split_q10 : 512)) >> 10);
    m = m < 1 ? 1 : (m > NFp - 1 && NFp > 1 ? NFp - 1 : m);
    const int tA = XPT ? NFp : (DIR ? NFp - m : m), tEnd = NFp;
    auto frame_of = [&](int t) { return DIR ? NFp + 1 - t : t; };
    PairHand *hand = reinterpret_cast<PairHand *>(p.pair_hand) + ((long long)pair * 2 + DIR) * 2;
    if (lds_addr_of(lds) != 0u) __builtin_trap();
    MM_STAMP_DECL;

    // ---- LDS set-up
    for (unsigned q = tid * 4u; q < 2u * L::RS2; q += NT * 4u) ldsw(L::PP(0) + q, 0.f);
    if constexpr (PHASE == 1)
        for (unsigned q = tid * 4u; q < 2u * RSH; q += NT * 4u) ldsw(L::Q(0) + q, 0.f);
    if (tid < 32) ldsw(L::MS(0) + 4u * tid, 0.f);
    if (tid < 4) ldswu(L::MX(0) + 4u * tid, 0u);
    if (tid < 4) ldsw(L::EM(tid >> 1) + 8u * P1p + 4u * (tid & 1), LINF ? 0.f : MM_NINF);  // the emission slot of lanes without a row
    const int nslotwords = r.nslotrows * 128;
    for (int q = tid; q < nslotwords; q += NT) ldswu(L::SLOTS + 4u * q, as_global(r.slots)[q]);
    if constexpr (PHASE == 1)
        for (int q = tid; q < P1; q += NT) ldswu(L::PDFSE + 4u * q, as_global(reinterpret_cast<const unsigned *>(r.pdfse))[q]);
    int *redo0 = p.redo + U[0].b, *redo1 = p.redo + (U[1].valid ? U[1].b : p.B);
    // ---- the team (split kernels): own set's region, own / others' slots of this launch
    const int xbase = H > 1 ? p.sp_base[hset] : 0, xcnt = H > 1 ? p.sp_cnt[hset] : 0;
    float *xsend = nullptr, *xps_send = nullptr;
    // The whole team on ONE XCD (found out at the start of the launch, see the exchange wave): plain stores reach the L2
    // that the team's sc1 polls read, and they leave the compute unit as whole lines instead of one fabric write per lane
    // (measured on the WSJ graph: 2.8 against 3.3 ms per call).  Anything else: write-through (sc1) granules.
    bool xplain = false;
    const float *xrecv[H], *xps_recv[H];
    if constexpr (H > 1) {
        float *xb = p.xbuf + (long long)PHASE * p.x_phase + ((long long)pair * 2 + DIR) * H * 2 * p.x_slot;
        float *xq = p.xps + ((long long)pair * 2 + DIR) * H * 4 * (int)L::XPS;
#pragma unroll
        for (int g = 0; g < H; ++g) {
            xrecv[g] = g == hset ? nullptr : xb + (long long)g * 2 * p.x_slot;
            xps_recv[g] = g == hset ? nullptr : xq + (long long)g * 4 * (int)L::XPS;
        }
        xsend = xb + (long long)hset * 2 * p.x_slot;
        xps_send = xq + (long long)hset * 4 * (int)L::XPS;
    }
    unsigned long long endmask = 0, lgw0 = 0;
    int nslots = 0, start2 = 0;
    unsigned slot_base = 0;
    if (!service && wave < r.NWC) {
        const RowSched &sc = r.sched[wave];
        endmask = sc.endmask;
        lgw0 = sc.lg;
        nslots = (int)(sc.nslots & 0xffffu);
        start2 = (int)(sc.nslots >> 16);
        slot_base = L::SLOTS + (sc.slot0 * 64u + lane) * 8u;
    }
    unsigned em_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)endmask);
    unsigned em_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(endmask >> 32));
    // the wave's last pair of arcs (phase B runs the whole window: the exits cost it 8 spilled VGPRs)
    const int lastp = (PHASE && !(MM_PAIR_EXITS_B && NJ <= 4)) ? 63 : (em_hi ? 63 - __builtin_clz(em_hi) : (em_lo ? 31 - __builtin_clz(em_lo) : -1));
    lgw0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(lgw0 >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((unsigned)lgw0);
    nslots = __builtin_amdgcn_readfirstlane(nslots);
    start2 = __builtin_amdgcn_readfirstlane(start2);
    PairRegs<KA> rg;
    auto load_graph = [&]() {
        static_assert(KA <= MM_ROW_KA_PAD, "register window larger than the padding of the device arrays");
        const int nt = 64 * r.NWC;
        const bool mine = wave < r.NWC;
        const auto wp = as_global(r.w);
        const auto ap = as_global(r.addr);
        const int t0 = mine ? tid : 0;
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            if (k & 1) rg.w2[k / 2].y = wp[k * nt + t0];
            else rg.w2[k / 2].x = wp[k * nt + t0];
            rg.a[k] = ap[k * nt + t0];
        }
        if (!mine) {
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                if (k & 1) rg.w2[k / 2].y = 0.f;
                else rg.w2[k / 2].x = 0.f;
                rg.a[k] = 0u;
            }
        }
    };
    // steps of this launch: (t0, t1]; the vector of step t0 is the starting point
    const int t0 = PHASE ? tA : 1, t1 = PHASE ? tEnd : tA;
    __syncthreads();

    if (service) {
        // ================= service wave =================
        // (split kernels: `sl` is made opaque at the top of every step, so that nothing derived from the lane index is
        // hoisted out of the step loop -- with the 24 KB vectors of those kernels the hoisted LDS addresses did not fit the
        // registers, and every reload of a spilled one is a scratch load whose wait also waits for the LDS-DMAs in flight)
        int sl = lane;
        float thr_v = thr;  // (a vector register: the instances of the large teams have no scalar register to keep it in across the steps)
        asm volatile("" : "+v"(thr_v));
        RowNorm norm[2];
        double cum[2] = {0.0, 0.0};
        double zmin[2] = {__builtin_inf(), __builtin_inf()}, zmax[2] = {-__builtin_inf(), -__builtin_inf()};
        float ltmin[2] = {__builtin_inff(), __builtin_inff()};  // smallest log2 of a frame's sum of 2^(a~ + b~) (see mm_pair_finish_kernel)
        bool xdead = H > 1 && (p.x_sleep & 0x300) != 0;  // (split kernels) a poll of the team's partial sums timed out
        // SMALL (an instance of its own: graphs of up to 127 states, BASELINE config 2): the row of pairs is at most 64 float4 --
        // one partner DMA per step instead of the region's 16, 4 scan loads per lane instead of 16.  The service wave is the
        // longest actor of a small graph's step; as run-time branches in the one kernel they cost config 3 3 %.
        constexpr bool small_graph = SMALL;
        static_assert(!SMALL || H == 1, "teams are for large graphs");
        auto dma_raw = [&](int t) {  // raw emissions of step t (clamped) -> RAW(t & 3, u)
            const int tt = t < 1 ? 1 : (t > tEnd ? tEnd : t);
#pragma unroll
            for (int u = 0; u < 2; ++u) row_dma_em<NJ>(L::RAW(0, u) + L::RAWS * (unsigned)(t & 3), U[u].Vb, p.vsn, frame_of(tt), p.N, P, sl);
        };
        auto dma_partner = [&](int t) {  // the other agent's vector + offset of step t's frame -> AL(t % 3, u), POFF(t & 3, u)
            const int tt = t < 1 ? 1 : (t > tEnd ? tEnd : t);
            int f = frame_of(tt);
            f = f > p.N ? p.N : f;  // (frame N+1 is never combined)
            // (split kernels: the rows of the own set only -- the other direction's workgroup of the same set stored them,
            // contiguously, at the set's base)
            const int n4 = H > 1 ? (xcnt + 1) >> 1 : S1p >> 1;  // float4s of the row of pairs
            constexpr int NDM = RSH / 1024;
            const mm_f32x4 *src = reinterpret_cast<const mm_f32x4 *>(rowsP + (long long)f * 2 * S1p + 2 * xbase);
            const unsigned dst = L::AL(0) + (unsigned)(tt % L::NR) * (unsigned)RSH;
            if constexpr (small_graph) {  // (one DMA covers the row: the others would copy element 0 again)
                dma_b128(src + (sl < n4 ? sl : 0), dst);
            } else {
                // (the whole region in one asm block, no clamping: what lies behind the row -- the next frame's row, or the
                // workspace's slack behind the last one -- lands in the part of the LDS region no finish reads)
                (void)n4;
                dma_row_b128<NDM>(uni(src), (unsigned)sl, dst);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (sl < 2) dma_b32(reinterpret_cast<const unsigned *>(U[u].offs + f) + sl, L::POFF(0, u) + 16u * (unsigned)(t & (L::POFFN - 1)));
        };
        constexpr int NDMA = 2 * NJ + (PHASE ? RSH / 1024 + 2 : 0);  // DMAs issued per step (a lower bound of the VMEM operations)
        constexpr int NDMA_SMALL = 2 * NJ + (PHASE ? 1 + 2 : 0);     // ... of a small graph (see dma_partner)
        // stage the emissions of step t into EM(t & 1) and account its offsets; S = the normaliser the step subtracts
        auto stage = [&](int t, const float (&S)[2]) {
            float E[2], X = 0.f;
#ifndef MM_PAIR_WIDE_SERVICE
#define MM_PAIR_WIDE_SERVICE 1
#endif
            if constexpr (NJ > 4 && MM_PAIR_WIDE_SERVICE) {
                pair_stage_em_wide<NJ, LINF>(L::EM(t & 1), L::RAW(0, 0) + L::RAWS * (unsigned)(t & 3), L::RAW(0, 1) + L::RAWS * (unsigned)(t & 3), frame_of(t),
                                             U[0].len, U[1].len, P, sl, E, S[0], S[1], redo0, redo1);
            } else if constexpr (LINF) {
                pair_stage_em2<NJ, ADAPT>(L::EM(t & 1), L::RAW(0, 0) + L::RAWS * (unsigned)(t & 3), L::RAW(0, 1) + L::RAWS * (unsigned)(t & 3), frame_of(t),
                                          U[0].len, U[1].len, P, sl, S[0], S[1], E, X, redo0, redo1);
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    E[u] = pair_stage_em<NJ, LINF>(L::EM(t & 1), L::RAW(0, u) + L::RAWS * (unsigned)(t & 3), u, frame_of(t), U[u].len, P, sl, S[u], u ? redo1 : redo0);
            }
            double before[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                before[u] = cum[u];
                cum[u] += (double)S[u] + (double)E[u];
            }
            if constexpr (ADAPT) {
                // the smallest sum the step's finishes accept: with every non-zero s >= 2^-(thr - X) and every non-zero factor >=
                // 2^-X, every non-zero p is >= 2^-thr and no product of the next step's sums leaves the float range.  Posted as
                // the bits of that float - 1 (the finishes compare integers); emissions that leave the sums less than 2^-16 of
                // room: marked here.
                float room = thr_v - X;
                if (room < 16.f) {
                    room = 16.f;
                    if (sl == 0) {
                        *redo0 = 1;
                        *redo1 = 1;
                    }
                }
                if (sl == 0) ldswu(L::MS(t & 1), ((unsigned)(127 - (int)room) << 23) - 1u);
            }
            if (sl == 0) {
                if constexpr (!LINF) ldsw2(L::MS(t & 1), S[0], S[1]);
                // the offsets that turn the step's vectors into log2 values.  LINF: the stored vector p includes the step's
                // normaliser and emission (both directions); what is combined with the partner's is s, the sum before either
                // (OWN).  Else: forward alpha~ includes the emission, backward beta~ does not, and both are what is combined.
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const double off = LINF ? cum[u] : (DIR ? cum[u] - (double)E[u] : cum[u]);
                    *(__attribute__((address_space(3))) double *)(__UINTPTR_TYPE__)(L::OWN(t & 3) + 8u * u) = LINF ? before[u] : off;
                    // (XPT, backward: the stored vector is s, the sum before this step's normaliser and emission)
                    if (PHASE == 0) U[u].offs[frame_of(t)] = (XPT && DIR) ? before[u] : off;
                }
            }
        };
        // ---- prologue: everything step t0 + 1 needs
        for (int t = t0; t <= t0 + 3; ++t) dma_raw(t);
        if constexpr (PHASE == 1) {
            dma_partner(t0 + 1);
            if constexpr (L::NR == 3) dma_partner(t0 + 2);  // (a ring of 2: step t0 + 1 requests the row of t0 + 2 at its top)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const PairHand h = hand[u];
                norm[u].m_prev = h.m_prev;
                norm[u].s_cur = h.s_cur;
                norm[u].s_prev = h.s_prev;
                norm[u].cbar = h.cbar;
                norm[u].seen = h.seen;
                cum[u] = h.cum;
            }
        }
        MM_ROW_VMCNT(0);
        if (PHASE == 0 || (DIR == 1 && !LINF)) {  // emissions of the starting step: the initial alpha needs them, and so does
            float E[2];                           // rebuilding the backward agent's linear vector from its stored beta~ (!LINF)
#pragma unroll
            for (int u = 0; u < 2; ++u) E[u] = pair_stage_em<NJ>(L::EM(t0 & 1), L::RAW(0, u) + L::RAWS * (unsigned)(t0 & 3), u, frame_of(t0), U[u].len, P, sl);
            if (PHASE == 0) {  // step 1 subtracts nothing but E
#pragma unroll
                for (int u = 0; u < 2; ++u) cum[u] = (double)E[u];
                if (DIR == 0 && sl == 0) {
                    U[0].offs[1] = cum[0];
                    U[1].offs[1] = cum[1];
                }
            }
        }
        __syncthreads();  // (1) emissions of step t0 staged
        if (t0 + 1 <= t1) {
            const float S[2] = {norm[0].s_cur, norm[1].s_cur};  // (phase A: 0 -- step 2 subtracts nothing but E)
            stage(t0 + 1, S);
        }
        dma_raw(t0 + 4);
        __syncthreads();  // (2) starting vector in LDS, step t0 + 1 prepared
        // posteriors and per-frame log Z of step ts (its per-pdf sums are complete)
        // (8-byte stores of the posteriors where every row of gamma allows them: pdfs contiguous, rows and batch strides even, the
        // base 8-byte aligned)
        const bool gwide = p.gsp == 1 && ((p.gsn | p.gsb) & 1) == 0 && ((__UINTPTR_TYPE__)p.gamma & 7) == 0;
        auto frames_of_step = [&](int ts, unsigned psum) {
            const int f = frame_of(ts);
            const bool live0 = f >= 1 && f <= U[0].len, live1 = f >= 1 && f <= U[1].len;
            float lt[2];
            const float *xp[H];
#pragma unroll
            for (int g = 0; g < H; ++g) xp[g] = (H > 1 && g != hset) ? xps_recv[g] + (ts & 3) * (int)L::XPS : nullptr;
            const bool writer = H == 1 || hset == 0;  // (every workgroup of a team has the sums of all pdfs: the first stores gamma)
            if constexpr (H == 1 && NJ > 4 && MM_PAIR_WIDE_SERVICE) {
                pair_finish_frames_wide<NJ>(psum, P1, P, sl, p.gamma + (long long)U[0].b * p.gsb + (long long)(f - 1) * p.gsn,
                                            p.gamma + (long long)U[1].b * p.gsb + (long long)(f - 1) * p.gsn, p.gsp, live0 && U[0].valid,
                                            live1 && U[1].valid, gwide, lt);
            } else if (!pair_finish_frames<NJ, H>(psum, P1, P, sl, p.gamma + (long long)U[0].b * p.gsb + (long long)(f - 1) * p.gsn,
                                           p.gamma + (long long)U[1].b * p.gsb + (long long)(f - 1) * p.gsn, p.gsp,
                                           live0 && U[0].valid && writer, live1 && U[1].valid && writer, lt, xp, split_tag(ts, t0, 2),
                                           xdead ? 0ull : x_tmo)) {
                if (sl == 0) {
                    *redo0 = 2;  // (the team is not running together: the exact kernels compute these utterances)
                    *redo1 = 2;
                }
                xdead = true;  // ... and nothing is waited for any more
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u ? live1 : live0) {
                    const double own = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(L::OWN(ts & 3) + 8u * u);
                    const double oth = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(L::POFF(0, u) + 16u * (unsigned)(ts & (L::POFFN - 1)));
                    const double z = (double)lt[u] + own + oth;
                    zmin[u] = z < zmin[u] ? z : zmin[u];
                    zmax[u] = z > zmax[u] ? z : zmax[u];
                    if (!(z == z)) zmax[u] = __builtin_inf();  // (an overflow somewhere: inf * 0; the finish kernel sends the utterance to the exact kernels)
                    ltmin[u] = lt[u] < ltmin[u] ? lt[u] : ltmin[u];
                }
        };
        auto step = [&](auto RDc, int t) {
            constexpr int RD = decltype(RDc)::value, WR = 1 - RD;  // RD = parity of steps t - 1 and t + 1
            // (NJ = 4, P + 1 > 128: with the addresses of 8 emission DMAs and 8 gamma stores hoisted the phase B kernels spilled
            // 24 / 33 VGPRs, and this wave waited for its DMAs at every reload: 4.0 -> 3.6 ms per call on config 3's graph
            // with 200 pdfs.  NJ = 2 keeps its hoisted addresses: 2.94 against 3.12 ms with the opaque lane -- -DMM_PAIR_OPAQUE_B=1)
#ifndef MM_PAIR_OPAQUE_B
#define MM_PAIR_OPAQUE_B 0
#endif
            if constexpr (H > 1 || NJ > 2 || (MM_PAIR_OPAQUE_B && PHASE == 1)) asm volatile("" : "+v"(sl));
            // A partner ring of TWO vectors (NJ > 4): the row of step t + 1 is requested here, at the top of step t -- its buffer
            // was read until the barrier that ended step t - 1 -- and awaited at the bottom with the 2 NJ emission DMAs issued
            // behind it allowed in flight; those are then all that is in flight at the top of a step, and the emissions of
            // step t + 1 (requested at step t - 3) have landed.
            constexpr bool ring2 = PHASE == 1 && L::NR == 2;
            if constexpr (ring2) dma_partner(t + 1);
            // emissions of step t + 1 (requested at step t - 2: the DMAs of step t - 1 may still be in flight)
            // (a small graph issues ONE partner DMA per step, not RSH / 1024: the bound of what may be in flight shrinks with it)
            else if constexpr (small_graph) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA_SMALL) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            MM_STAMP(2);
            // the normaliser of step t + 1 from the maxima of step t - 1 (complete since the last barrier)
            float mx[2];
            // (a graph of up to 511 states: 4 loads per lane instead of the region's 16 -- the service wave is the longest
            // actor of a small graph's step, BASELINE config 2)
            // (teams: the waves that wrote the vector left its maxima in MX -- a scan of a team's whole vector, 48 KB for teams of 8,
            // was the longest thing this wave did: 3300 .. 4200 of the 6500 cycles that bounded a step of theirs; the starting
            // vector of a launch is scanned)
            if constexpr (H > 1 && MM_SPLIT_MAXTRACK) {
                if (t == t0 + 1) {
                    pair_scan_max<(RS / 8 + 63) / 64>(L::PP(RD), (S1 + 2) >> 1, sl, mx[0], mx[1]);
                } else {
                    const mm_u32x2 w = ldsr2u(L::MX(RD));
                    mx[0] = fast_log2(__builtin_bit_cast(float, (unsigned)__builtin_amdgcn_readfirstlane((int)w.x)));
                    mx[1] = fast_log2(__builtin_bit_cast(float, (unsigned)__builtin_amdgcn_readfirstlane((int)w.y)));
                }
                if (sl == 0) {
                    ldswu(L::MX(RD), 0u);
                    ldswu(L::MX(RD) + 4u, 0u);
                }
            } else if constexpr (small_graph) pair_scan_max<4>(L::PP(RD), (S1 + 2) >> 1, sl, mx[0], mx[1]);
            else pair_scan_max<(RS / 8 + 63) / 64, MM_PAIR_SCAN_BATCH>(L::PP(RD), (S1 + 2) >> 1, sl, mx[0], mx[1]);
            MM_STAMP(3);
            if (t + 1 <= tEnd) {
                const float S[2] = {norm[0].next(mx[0]), norm[1].next(mx[1])};
                if (t + 1 <= t1) stage(t + 1, S);
            }
            MM_STAMP(4);
            dma_raw(t + 4);
            if constexpr (PHASE == 1) {
                if constexpr (!ring2) dma_partner(t + 2);
                MM_STAMP(5);
                // gamma of step t - 2: its per-pdf sums were completed in the previous step
                // (split kernels: by the exchange wave, which has nothing else to do since the compute waves receive the rows --
                // this wave was the longest actor of phase B, and the other sets' partial sums are a memory round trip each)
                if constexpr (H == 1)
                    if (t - 2 > t0) frames_of_step(t - 2, L::PSUM(WR));
                MM_STAMP(6);
                // the partner vector of step t + 1 (requested at step t - 1) must be in LDS when the compute waves leave the barrier
                if constexpr (ring2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
                else if constexpr (small_graph) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA_SMALL) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
                MM_STAMP(7);
            }
            MM_STAMP(0);
            MM_STEP_SYNC();
            MM_STAMP(1);
        };
        MM_STAMP_RESET;
        for (int t = t0 + 1; t <= t1; t += 2) {
            if (t & 1) step(std::integral_constant<int, 0>{}, t);
            else step(std::integral_constant<int, 1>{}, t);
            if (t + 1 <= t1) {
                if ((t + 1) & 1) step(std::integral_constant<int, 0>{}, t + 1);
                else step(std::integral_constant<int, 1>{}, t + 1);
            }
        }
        if constexpr (PHASE == 0) {
            if (sl == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    PairHand h;
                    h.m_prev = norm[u].m_prev;
                    h.s_cur = norm[u].s_cur;
                    h.s_prev = norm[u].s_prev;
                    h.cbar = norm[u].cbar;
                    h.seen = norm[u].seen;
                    h.pad = 0;
                    h.cum = cum[u];
                    hand[u] = h;
                }
            }
        } else {
            // the last two steps' posteriors: (a) sums of step t1 by the compute waves, gamma of step t1 - 1 here; (b) gamma of step t1
            MM_ROW_VMCNT(0);
            for (int k = 1; k >= 0; --k) {
                const int t = t1 - k;
                if (k == 0) __syncthreads();  // (a)
                if (H == 1 && t > t0) frames_of_step(t, L::PSUM(t & 1));
            }
            if (H == 1 && sl == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (U[u].valid) {
                        p.pair_zmin[(long long)U[u].b * 6 + DIR] = zmin[u];
                        p.pair_zmin[(long long)U[u].b * 6 + 2 + DIR] = zmax[u];
                        p.pair_zmin[(long long)U[u].b * 6 + 4 + DIR] = (double)ltmin[u];
                    }
            }
        }
    } else if (xwave) {
        // ================= exchange wave (split kernels) =================
        __syncthreads();  // (1)
        bool dead = (p.x_sleep & 0x200) != 0;
        {   // which XCD is the team on?  Every workgroup publishes the id of its own (a granule in the unused tail of
            // psum slot PHASE of its set) and reads the others'; workgroup -> XCD placement is the hardware's business,
            // only what the workgroups SEE decides how they store.
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 15u;
            if (lane == 0) granule_store(xps_send + PHASE * (int)L::XPS, 8u * (L::XPS / 2u - 1u), __builtin_bit_cast(float, xcc + 1u), 0.f);
            bool same = true;
            const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
#pragma unroll
            for (int g = 0; g < H; ++g) {
                if (g == hset) continue;
                unsigned other = 0u;
                while (!dead) {
                    other = (unsigned)granule_load(xps_recv[g] + PHASE * (int)L::XPS, 8u * (L::XPS / 2u - 1u));
                    if (other != 0u) break;
                    if (__builtin_amdgcn_s_memrealtime() - tstart > x_tmo) dead = true;
                    __builtin_amdgcn_s_sleep(8);
                }
                same = same && other == xcc + 1u;
            }
            if (dead && lane == 0) {
                *redo0 = 2;
                *redo1 = 2;
            }
            if (lane == 0) ldswu(L::XFLAG, (same && !dead && !(p.x_sleep & 0x800)) ? 1u : 0u);
            // (how often the hardware's placement puts a whole team on one XCD: counted per workgroup of the phase-A launch once
            // mm_batch_team_xcd_stats has been called on the batch, read and cleared by it -- bench.py prints it for the team workloads)
            if (PHASE == 0 && lane == 0 && p.stat_dev && p.stat_xcd) {
                atomicAdd(&p.stat_dev[3], 1);
                if (same && !dead) atomicAdd(&p.stat_dev[2], 1);
            }
        }
        __syncthreads();  // (2)
        constexpr int NG2 = (RSH / 16 + 63) / 64;
        // posteriors and per-frame log Z of step ts (phase B; the service wave's frames_of_step for H = 1)
        double xzmin[2] = {__builtin_inf(), __builtin_inf()}, xzmax[2] = {-__builtin_inf(), -__builtin_inf()};
        float xltmin[2] = {__builtin_inff(), __builtin_inff()};
        auto xframes = [&](int ts, unsigned psum) {
            const int f = frame_of(ts);
            const bool live0 = f >= 1 && f <= U[0].len, live1 = f >= 1 && f <= U[1].len;
            float lt[2];
            const float *xp[H];
#pragma unroll
            for (int g = 0; g < H; ++g) xp[g] = g != hset ? xps_recv[g] + (ts & 3) * (int)L::XPS : nullptr;
            const bool writer = hset == 0;  // (every workgroup of a team has the sums of all pdfs: the first stores gamma)
            if (!pair_finish_frames<NJ, H>(psum, P1, P, lane, p.gamma + (long long)U[0].b * p.gsb + (long long)(f - 1) * p.gsn,
                                           p.gamma + (long long)U[1].b * p.gsb + (long long)(f - 1) * p.gsn, p.gsp,
                                           live0 && U[0].valid && writer, live1 && U[1].valid && writer, lt, xp, split_tag(ts, t0, 2),
                                           dead ? 0ull : x_tmo)) {
                if (lane == 0) {
                    *redo0 = 2;  // (the team is not running together: the exact kernels compute these utterances)
                    *redo1 = 2;
                }
                dead = true;  // ... and nothing is waited for any more
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u ? live1 : live0) {
                    const double own = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(L::OWN(ts & 3) + 8u * u);
                    const double oth = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(L::POFF(0, u) + 16u * (unsigned)(ts & (L::POFFN - 1)));
                    const double z = (double)lt[u] + own + oth;
                    xzmin[u] = z < xzmin[u] ? z : xzmin[u];
                    xzmax[u] = z > xzmax[u] ? z : xzmax[u];
                    if (!(z == z)) xzmax[u] = __builtin_inf();
                    xltmin[u] = lt[u] < xltmin[u] ? lt[u] : xltmin[u];
                }
        };
        MM_STAMP_RESET;
#ifndef MM_SPLIT_CWPOLL
#define MM_SPLIT_CWPOLL 1
#endif
        for (int t = t0 + 1; t <= t1; ++t) {
            if constexpr (!MM_SPLIT_CWPOLL) if (!dead) {
                // (teams of 2 as well: the chunks of the other set checked and written as they return instead of all loaded, all
                // awaited, all written -- 2.73 -> 2.30 ms on the reference's WSJ denominator.  Loads in flight, measured on that
                // graph / on a 4000-state graph with teams of 4: 2: 2.52 / 7.03 ms, 3: 2.29 / 6.00, 4: 2.29 / 5.77, 5: 2.30 / 5.48,
                // 6: 2.33 / 5.67, 9: 2.46 / 6.10, 12: 2.47 / 6.12 -- what a hand-off costs sits in this compute unit's memory queue)
#ifndef MM_SPLIT_PIPE2
#define MM_SPLIT_PIPE2 1
#endif
#ifndef MM_SPLIT_K
#define MM_SPLIT_K 5
#endif
                if constexpr (H == 2 && MM_SPLIT_PIPE2) {
                    const int g = 1 - hset;
                    const float *src[1] = {xrecv[g] + (long long)(t & 1) * p.x_slot};
                    const int cnt[1] = {p.sp_cnt[g]};
                    const unsigned dsts[1] = {L::PP(t & 1) + 8u * (unsigned)p.sp_base[g]};
                    if (!split_receive_multi<NG2, 1, (NG2 < MM_SPLIT_K ? NG2 : MM_SPLIT_K)>(src, cnt, dsts, split_tag(t, t0, 1), lane, (unsigned)p.x_sleep & 0xffu)) {
                        dead = true;
                        if (lane == 0) {
                            *redo0 = 2;
                            *redo1 = 2;
                        }
                    }
                } else if constexpr (H == 2) {
#pragma unroll
                    for (int g = 0; g < H; ++g) {
                        if (g == hset) continue;
                        if (!split_receive<NG2>(xrecv[g] + (long long)(t & 1) * p.x_slot, p.sp_cnt[g], L::PP(t & 1) + 8u * (unsigned)p.sp_base[g],
                                               split_tag(t, t0, 1), lane, (unsigned)p.x_sleep & 0xffu)) {
                            dead = true;  // the team is not running together: mark the utterances for the exact kernels, wait no more
                            if (lane == 0) {
                                *redo0 = 2;
                                *redo1 = 2;
                            }
                        }
                    }
                } else if constexpr (H > 2) {  // (the other sets' rows in one pipelined sweep: split_receive_multi)
                    constexpr int NP = H - 1;
                    const float *src[NP];
                    int cnt[NP];
                    unsigned dsts[NP];
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const int g = q < hset ? q : q + 1;
                        src[q] = xrecv[g] + (long long)(t & 1) * p.x_slot;
                        cnt[q] = p.sp_cnt[g];
                        dsts[q] = L::PP(t & 1) + 8u * (unsigned)p.sp_base[g];
                    }
                    if (!split_receive_multi<NG2, NP, MM_SPLIT_K>(src, cnt, dsts, split_tag(t, t0, 1), lane, (unsigned)p.x_sleep & 0xffu)) {
                        dead = true;
                        if (lane == 0) {
                            *redo0 = 2;
                            *redo1 = 2;
                        }
                    }
                }
            }
            // (the posteriors and the per-frame statistics are the FIRST workgroup's business: the others publish their partial
            // sums and read nobody's -- eight workgroups polling seven sets each was eight times the traffic for the same result)
            if constexpr (PHASE == 1)
                if (t - 2 > t0 && hset == 0) xframes(t - 2, L::PSUM(t & 1));
            MM_STAMP(0);
            MM_STEP_SYNC();
            MM_STAMP(1);
        }
        if constexpr (PHASE == 1) {
            // the last two steps' posteriors: (a) sums of step t1 by the compute waves, gamma of step t1 - 1 here; (b) gamma of step t1
            for (int k = 1; k >= 0; --k) {
                const int t = t1 - k;
                if (k == 0) __syncthreads();  // (a)
                if (t > t0 && hset == 0) xframes(t, L::PSUM(t & 1));
            }
            if (lane == 0 && hset == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (U[u].valid) {
                        p.pair_zmin[(long long)U[u].b * 6 + DIR] = xzmin[u];
                        p.pair_zmin[(long long)U[u].b * 6 + 2 + DIR] = xzmax[u];
                        p.pair_zmin[(long long)U[u].b * 6 + 4 + DIR] = (double)xltmin[u];
                    }
            }
        }
    } else {
        // ================= compute waves =================
        __syncthreads();  // (1)
        // the starting vector (step t0)
        if (PHASE == 0 && DIR == 0) {  // alpha_hat (*) lhs[:,1]   (src/inference.jl:68)
            for (int i = tid; i < S1; i += 64 * NWC) {
                unsigned pdfi = as_global(r.rowpdf)[i];
                if (H > 1 && pdfi == 0xffffu) pdfi = (unsigned)P1p;  // (alignment padding between the sets' regions: init = -inf)
                const unsigned e8 = 8u * pdfi;
                const mm_f32x2 e = ldsr2(L::EM(1) + e8);
                const float a = as_global(r.init)[i];
                float v0 = a + e.x, v1 = a + e.y;
                if (LINF && H > 1 && pdfi == (unsigned)P1p) v0 = v1 = MM_NINF;  // (LINF: the slot of lanes without a row holds the linear 0)
                if (row_out_of_range(v0, thr)) *redo0 = 1;
                if (row_out_of_range(v1, thr)) *redo1 = 1;
                ldsw2(L::PP(1) + 8u * i, fast_exp2(v0), fast_exp2(v1));
                if constexpr (LINF) *reinterpret_cast<mm_f32x2 *>(rowsP + ((long long)1 * S1p + i) * 2) = mm_f32x2{fast_exp2(v0), fast_exp2(v1)};
                else *reinterpret_cast<mm_f32x2 *>(rowsP + ((long long)1 * S1p + i) * 2) = mm_f32x2{v0, v1};
            }
        } else if (DIR == 1 && t0 == 1) {  // B[:, N+1] = one at the final state   (src/inference.jl:104)
            if (tid == 0) ldsw2(L::PP(1) + 8u * r.fpos, 1.f, 1.f);
        } else {  // phase B: the vector this agent stored at the end of phase A
            const int f = frame_of(t0);
            for (int i = tid; i < S1; i += 64 * NWC) {
                const mm_f32x2 vv = *reinterpret_cast<const mm_f32x2 *>(rowsP + ((long long)f * S1p + i) * 2);
                float v0 = vv.x, v1 = vv.y;
                const unsigned pdfi = as_global(r.rowpdf)[i];
                if constexpr (LINF) {  // the stored vector IS the linear vector of the step
                    if (H > 1 && pdfi == 0xffffu) v0 = v1 = 0.f;  // (padding: never stored)
                    ldsw2(L::PP(t0 & 1) + 8u * i, v0, v1);
                    continue;
                }
                if (DIR == 1) {  // beta~ is stored without the frame's emission
                    const mm_f32x2 e = ldsr2(L::EM(t0 & 1) + 8u * (H > 1 && pdfi == 0xffffu ? (unsigned)P1p : pdfi));
                    v0 += e.x;
                    v1 += e.y;
                }
                if (H > 1 && pdfi == 0xffffu) v0 = v1 = MM_NINF;  // (padding: never stored)
                ldsw2(L::PP(t0 & 1) + 8u * i, fast_exp2(v0), fast_exp2(v1));
            }
        }
        load_graph();
        __syncthreads();  // (2)
        if constexpr (H > 1) xplain = __builtin_amdgcn_readfirstlane(ldsru(L::XFLAG)) != 0u;
        bool cdead = H > 1 && (p.x_sleep & 0x200) != 0;  // (split kernels) a poll of this wave timed out: it waits no more
        auto step = [&](auto RDc, int t) {
            constexpr int RD = decltype(RDc)::value, WR = 1 - RD;
            float vmx0 = 0.f, vmx1 = 0.f;  // (teams) maxima of what this wave writes into the step's vector
            if (nslots > 0) {
                constexpr unsigned rdoff = L::PP(RD);
                mm_f32x2 x[2 * D];
#pragma unroll
                for (int j = 0; j < D; ++j) {  // the first gathers leave before anything else
                    x[2 * j] = ldsr2(rg.a[(2 * j < KA) ? 2 * j : 0] + rdoff);
                    x[2 * j + 1] = ldsr2(rg.a[(2 * j + 1 < KA) ? 2 * j + 1 : 0] + rdoff);
                }
                // the slot table runs one segment ahead (infoN / info2N), so that the emission and partner reads of a segment,
                // which need its slot word as their address, never wait for a load issued just before them
                unsigned sa = slot_base;
                unsigned info, info2 = 0u, infoN, info2N = 0u;
                if constexpr (PHASE == 1) {  // (both words of a slot in one 8-byte read: one LDS instruction less per finish)
                    const mm_u32x2 w0 = ldsr2u(sa), w1 = ldsr2u(sa + 512u);
                    info = w0.x;
                    info2 = w0.y;
                    infoN = w1.x;
                    info2N = w1.y;
                } else {
                    info = ldsru(sa);
                    infoN = ldsru(sa + 512u);
                }
                mm_f32x2 S = {0.f, 0.f};
                if constexpr (!LINF) S = ldsr2(L::MS(WR));  // the step's normalisers, posted by the service wave
                mm_f32x2 e = ldsr2((info >> 16) + L::EM(WR));
                const int f = frame_of(t);
                const unsigned alb = L::AL(0) + (unsigned)(t % L::NR) * (unsigned)RSH;
                mm_f32x2 al = {0.f, 0.f};
                // (split kernels) where the team reads this step's rows, and the step's tag as a sign
                float *xw = H > 1 ? xsend + (long long)(t & 1) * p.x_slot - 2 * xbase : nullptr;
                const float xsg = (H > 1 && split_tag(t, t0, 1)) ? -1.f : 1.f;
                if constexpr (PHASE == 1) al = ldsr2((info2 & 0xffffu) + alb);
                float *rowP = rowsP + (long long)(f <= p.N + (XPT ? 1 : 0) ? f : 0) * 2 * S1p;  // (frame N + 1 is stored by the export only)
                // even / odd arcs (phase B has no registers to spare: one chain there)
                float worst = 0.f;
                unsigned smin = 0xffffffffu;
                mm_f32x2 accA = {0.f, 0.f}, accN = {0.f, 0.f};
                mm_f32x2 &accB = accA;
                unsigned long long lgw = lgw0;
                auto finish = [&]() {
                    const int lg = (int)(lgw & 15ull);
                    lgw >>= 4;
                    float s0 = accA.x, s1 = accA.y;
                    if (lg) grp_sum_last2(s0, s1, lg);
                    const unsigned pos8 = info & 0xffffu;
                    if constexpr (LINF) {
                        // p = s * 2^(emission - E - S): the step's vector, forward (T' alpha) (*) lhs (src/inference.jl:70-71),
                        // backward B (*) lhs, the operand of the next product (:106-107).
                        // Range check, deferred to the end of the step: the smallest non-zero sum of the lane, as integers (a
                        // non-negative float orders like its bits; bits - 1 sends the semiring's zero to the top), against the
                        // step's threshold (the service wave's stage()).  An overflow ends as a NaN frame sum
                        // (mm_pair_finish_kernel).
                        smin = min3_u32(smin, __builtin_bit_cast(unsigned, s0) - 1u, __builtin_bit_cast(unsigned, s1) - 1u);
                        mm_f32x2 sv = {s0, s1}, pv, qv;
                        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(pv) : "v"(sv), "v"(e));
                        ldsw2(pos8 + L::PP(WR), pv.x, pv.y);
                        if constexpr (H > 1) {
                            if (xplain) *reinterpret_cast<mm_f32x2 *>(reinterpret_cast<char *>(xw) + pos8) = mm_f32x2{pv.x * xsg, pv.y * xsg};
                            else granule_store(xw, pos8, pv.x * xsg, pv.y * xsg);
                        }
                        if constexpr (PHASE == 0) {
                            if constexpr (XPT) *reinterpret_cast<mm_f32x2 *>(reinterpret_cast<char *>(rowP) + pos8) = DIR ? sv : pv;
                            else *reinterpret_cast<mm_f32x2 *>(reinterpret_cast<char *>(rowP) + pos8) = pv;
                        } else {
                            asm("v_pk_mul_f32 %0, %1, %2" : "=v"(qv) : "v"(sv), "v"(al));  // A .* B   (:154)
                            ldsw2((info2 >> 16) + L::Q(WR), qv.x, qv.y);
                        }
                    } else {
                    // forward: (T' alpha) (*) lhs (src/inference.jl:70-71); backward: T (B (*) lhs) (:106-107), the emission is
                    // added for the next step's product only
                    const float b0 = fast_log2(s0) - S.x, b1 = fast_log2(s1) - S.y;
                    const float y0 = b0 + e.x, y1 = b1 + e.y;
                    // range check, deferred to the end of the step: the largest finite |y| of the lane.  |y| * 0 + |y| is NaN
                    // for -inf (the semiring's zero: in range) and v_max ignores NaN operands.
                    worst = __builtin_fmaxf(worst, __builtin_fmaxf(__builtin_fmaf(__builtin_fabsf(y0), 0.f, __builtin_fabsf(y0)),
                                                                   __builtin_fmaf(__builtin_fabsf(y1), 0.f, __builtin_fabsf(y1))));
                    const float p0 = fast_exp2(y0), p1 = fast_exp2(y1);
                    ldsw2(pos8 + L::PP(WR), p0, p1);
                    if constexpr (H > 1 && MM_SPLIT_MAXTRACK) {
                        vmx0 = max_nc(vmx0, p0);
                        vmx1 = max_nc(vmx1, p1);
                    }
                    if constexpr (H > 1) {
                        if (xplain) *reinterpret_cast<mm_f32x2 *>(reinterpret_cast<char *>(xw) + pos8) = mm_f32x2{p0 * xsg, p1 * xsg};
                        else granule_store(xw, pos8, p0 * xsg, p1 * xsg);
                    }
                    const float st0 = DIR ? b0 : y0, st1 = DIR ? b1 : y1;  // the vector that is stored / combined
                    if constexpr (PHASE == 0) {
                        *reinterpret_cast<mm_f32x2 *>(reinterpret_cast<char *>(rowP) + pos8) = mm_f32x2{st0, st1};
                    } else {
                        ldsw2((info2 >> 16) + L::Q(WR), fast_exp2(st0 + al.x), fast_exp2(st1 + al.y));  // A .* B   (:154)
                    }
                    }
                    accA = mm_f32x2{0.f, 0.f};
                    sa += 512u;
                    // (a plain copy is coalesced away and paid for with register moves on the no-finish path of EVERY pair)
                    asm volatile("v_mov_b32 %0, %1" : "=v"(info) : "v"(infoN));
                    e = ldsr2((info >> 16) + L::EM(WR));
                    if constexpr (PHASE == 1) {
                        asm volatile("v_mov_b32 %0, %1" : "=v"(info2) : "v"(info2N));
                        al = ldsr2((info2 & 0xffffu) + alb);
                        const mm_u32x2 w1 = ldsr2u(sa + 512u);
                        infoN = w1.x;
                        info2N = w1.y;
                    } else {
                        infoN = ldsru(sa + 512u);
                    }
                };
                asm volatile("" : "+s"(em_lo), "+s"(em_hi));
                __builtin_amdgcn_s_setprio(2);
                MM_PAIR_CASES(MM_PAIR_TWO)
                // out of the linear range somewhere: both utterances go to the exact kernels (the check does not tell
                // them apart; it only costs time)
                // (LINF: bits - 1 of the smallest sum the step's finishes accept, posted by the service wave -- read here, at the end:
                // a register that lives across the arcs is one the instances of the large teams do not have)
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(LINF ? smin < (ADAPT ? ldsru(L::MS(WR)) : sthr) : worst > thr) != 0ull, 0)) {
                    *redo0 = 1;
                    *redo1 = 1;
                }
            }
            if constexpr (PHASE == 1)  // C' * (A .* B) of the previous step (:155)
                if (t - 1 > t0)
                    pair_pdf_sums<(NJ > 4 ? 1 : (NJ > 2 ? 2 : 3))>(L::Q(RD), L::PDFSE, L::PSUM(RD), P1, wave, NWC, lane, H > 1 ? xps_send + ((t - 1) & 3) * (int)L::XPS : nullptr,
                                  (H > 1 && split_tag(t - 1, t0, 2)) ? -1.f : 1.f);
            if constexpr (H > 1 && MM_SPLIT_CWPOLL) {
                // The rows of the other sets of this step: chunk j (128 granules) of the q-th other set is item q * NG2 + j, and
                // compute wave w receives the items w, w + NWC, ... -- all chunks are polled at the same time, one load in flight
                // per wave, by waves that have nothing else to do until the barrier (one exchange wave sweeping all chunks was
                // the longest wave of every step: arrival of the last row + a sweep of 13 loads).
                constexpr int NG2 = (RSH / 16 + 63) / 64, I = (H - 1) * NG2;
                typedef unsigned mm_u32x4 __attribute__((ext_vector_type(4)));
                const unsigned tg = split_tag(t, t0, 1);
                if constexpr (H > 2) {
                    // (teams of 4: 27 chunks on 14 waves -- a wave's two chunks are polled together, both loads in flight: one
                    // after the other, the second cost another round trip behind the first; teams of 8: 49 chunks, four per wave)
                    constexpr int NI = (I + 13) / 14;
                    static_assert(NI >= 2 && NI <= 4, "items per wave");
                    const float *srcs[NI];
                    unsigned offs2[NI], dst2[NI];
                    bool pnd[NI], sec[NI], any = false;
#pragma unroll
                    for (int e = 0; e < NI; ++e) {
                        const int i = wave + e * NWC, ic = i < I ? i : 0;
                        const int q = ic / NG2, j = ic % NG2, g = q < hset ? q : q + 1;
                        srcs[e] = uni(xrecv[g] + (long long)(t & 1) * p.x_slot);
                        const int ng = p.sp_cnt[g];
                        dst2[e] = L::PP(WR) + 8u * (unsigned)p.sp_base[g] + 16u * (unsigned)(lane + 64 * j);
                        pnd[e] = i < I && 2 * (lane + 64 * j) < ng;
                        sec[e] = 2 * (lane + 64 * j) + 1 < ng;
                        offs2[e] = pnd[e] ? 16u * (unsigned)(lane + 64 * j) : 0u;
                        any = any || pnd[e];
                    }
                    if (!cdead && __builtin_amdgcn_ballot_w64(any) != 0ull) {
                        const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
                        for (;;) {
                            // (the loads of a poll and their wait in ONE asm block: the compiler may copy a register an
                            // asynchronous load has not filled yet)
                            mm_u32x4 v[NI];
                            if constexpr (NI == 2)
                                asm volatile("global_load_dwordx4 %0, %2, %3 sc1\n\tglobal_load_dwordx4 %1, %4, %5 sc1\n\ts_waitcnt vmcnt(0)"
                                             : "=&v"(v[0]), "=&v"(v[1])
                                             : "v"(offs2[0]), "s"(srcs[0]), "v"(offs2[1]), "s"(srcs[1])
                                             : "memory");
                            else if constexpr (NI == 3)
                                asm volatile("global_load_dwordx4 %0, %3, %4 sc1\n\tglobal_load_dwordx4 %1, %5, %6 sc1\n\t"
                                             "global_load_dwordx4 %2, %7, %8 sc1\n\ts_waitcnt vmcnt(0)"
                                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
                                             : "v"(offs2[0]), "s"(srcs[0]), "v"(offs2[1]), "s"(srcs[1]), "v"(offs2[2]), "s"(srcs[2])
                                             : "memory");
                            else
                                asm volatile("global_load_dwordx4 %0, %4, %5 sc1\n\tglobal_load_dwordx4 %1, %6, %7 sc1\n\t"
                                             "global_load_dwordx4 %2, %8, %9 sc1\n\tglobal_load_dwordx4 %3, %10, %11 sc1\n\ts_waitcnt vmcnt(0)"
                                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                                             : "v"(offs2[0]), "s"(srcs[0]), "v"(offs2[1]), "s"(srcs[1]), "v"(offs2[2]), "s"(srcs[2]), "v"(offs2[3]),
                                               "s"(srcs[3])
                                             : "memory");
                            any = false;
#pragma unroll
                            for (int e = 0; e < NI; ++e) {
                                if (pnd[e] && (v[e].x >> 31) == tg && (!sec[e] || (v[e].z >> 31) == tg)) {
                                    mm_f32x4 w;
                                    w.x = __builtin_bit_cast(float, v[e].x & 0x7fffffffu);
                                    w.y = __builtin_bit_cast(float, v[e].y & 0x7fffffffu);
                                    w.z = sec[e] ? __builtin_bit_cast(float, v[e].z & 0x7fffffffu) : 0.f;
                                    w.w = sec[e] ? __builtin_bit_cast(float, v[e].w & 0x7fffffffu) : 0.f;
                                    *(__attribute__((address_space(3))) mm_f32x4 *)(__UINTPTR_TYPE__)dst2[e] = w;
                                    vmx0 = max_nc(vmx0, max_nc(w.x, w.z));
                                    vmx1 = max_nc(vmx1, max_nc(w.y, w.w));
                                    pnd[e] = false;
                                }
                                any = any || pnd[e];
                            }
                            if (__builtin_amdgcn_ballot_w64(any) == 0ull) break;
                            if (__builtin_amdgcn_s_memrealtime() - tstart > x_tmo) {
                                cdead = true;
                                if (lane == 0) {
                                    *redo0 = 2;
                                    *redo1 = 2;
                                }
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                } else
                for (int i = wave; i < I; i += NWC) {
                    const int q = i / NG2, j = i % NG2, g = q < hset ? q : q + 1;
                    const float *src = xrecv[g] + (long long)(t & 1) * p.x_slot;
                    const int ng = p.sp_cnt[g];
                    const unsigned dsta = L::PP(WR) + 8u * (unsigned)p.sp_base[g] + 16u * (unsigned)(lane + 64 * j);
                    const bool have = 2 * (lane + 64 * j) < ng, second = 2 * (lane + 64 * j) + 1 < ng;
                    const unsigned off = have ? 16u * (unsigned)(lane + 64 * j) : 0u;
                    bool pend = have;
                    if (__builtin_amdgcn_ballot_w64(pend) == 0ull || cdead) continue;
                    const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        mm_u32x4 v;
                        asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(src) : "memory");
                        if (pend && (v.x >> 31) == tg && (!second || (v.z >> 31) == tg)) {
                            mm_f32x4 w;
                            w.x = __builtin_bit_cast(float, v.x & 0x7fffffffu);
                            w.y = __builtin_bit_cast(float, v.y & 0x7fffffffu);
                            w.z = second ? __builtin_bit_cast(float, v.z & 0x7fffffffu) : 0.f;
                            w.w = second ? __builtin_bit_cast(float, v.w & 0x7fffffffu) : 0.f;
                            *(__attribute__((address_space(3))) mm_f32x4 *)(__UINTPTR_TYPE__)dsta = w;
                            vmx0 = max_nc(vmx0, max_nc(w.x, w.z));
                            vmx1 = max_nc(vmx1, max_nc(w.y, w.w));
                            pend = false;
                        }
                        if (__builtin_amdgcn_ballot_w64(pend) == 0ull) break;
                        if (__builtin_amdgcn_s_memrealtime() - tstart > x_tmo) {
                            cdead = true;  // the team is not running together: the exact kernels compute these utterances
                            if (lane == 0) {
                                *redo0 = 2;
                                *redo1 = 2;
                            }
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            if constexpr (H > 1 && MM_SPLIT_MAXTRACK) {
                // (a finished or received value that is a NaN is dropped by v_max: the range marks catch what produced it)
                const float m0 = wave_max_rl(vmx0), m1 = wave_max_rl(vmx1);
                if (lane == 0) {
                    typedef __attribute__((address_space(3))) unsigned lds_u32;
                    (void)__hip_atomic_fetch_max((lds_u32 *)(__UINTPTR_TYPE__)L::MX(WR), __builtin_bit_cast(unsigned, m0), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
                    (void)__hip_atomic_fetch_max((lds_u32 *)(__UINTPTR_TYPE__)(L::MX(WR) + 4u), __builtin_bit_cast(unsigned, m1), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            MM_STAMP(0);
            MM_STEP_SYNC();
            MM_STAMP(1);
        };
        MM_STAMP_RESET;
        for (int t = t0 + 1; t <= t1; t += 2) {
            if (t & 1) step(std::integral_constant<int, 0>{}, t);
            else step(std::integral_constant<int, 1>{}, t);
            if (t + 1 <= t1) {
                if ((t + 1) & 1) step(std::integral_constant<int, 0>{}, t + 1);
                else step(std::integral_constant<int, 1>{}, t + 1);
            }
        }
        if constexpr (PHASE == 1) {
            if (t1 > t0)
                pair_pdf_sums<(NJ > 4 ? 1 : (NJ > 2 ? 2 : 3))>(L::Q(t1 & 1), L::PDFSE, L::PSUM(t1 & 1), P1, wave, NWC, lane, H > 1 ? xps_send + (t1 & 3) * (int)L::XPS : nullptr,
                              (H > 1 && split_tag(t1, t0, 2)) ? -1.f : 1.f);
            __syncthreads();  // (a)
        }
    }
#ifdef MM_STAMPS
    if (p.dbg && lane == 0)  // [pair][wave][phase * 2 + dir][work, barrier]
        for (int k = 0; k < 2; ++k) p.dbg[(((long long)pair * MM_MAX_WAVES + wave) * 4 + PHASE * 2 + DIR) * 2 + k] = stamp_acc[k];
    if (p.dbg && lane == 0 && service)  // the service wave's sections, behind the per-wave table: [pair][phase * 2 + dir][section]
        for (int k = 0; k < 8; ++k)
            p.dbg[(long long)((p.B + 1) / 2) * MM_MAX_WAVES * 8 + ((long long)pair * 4 + PHASE * 2 + DIR) * 8 + k] = stamp_acc[k];
#endif
}

// ttl = min over the frames of the per-frame log-normaliser (src/inference.jl:159); zeros beyond the sequence lengths.
//
// Also decides what a range mark (redo[b] == 1: some finished value of the linear-domain kernels was finite but outside
// the range in which float32 products keep every term) means for the utterance.  The linear path loses such a value's
// mass at most (sums of non-negative terms: flushing can only remove mass); in exact arithmetic the per-frame normaliser
// z_n = log sum_s alpha_n(s) beta_n(s) is the SAME for every frame (= log Z), and frame n lacks exactly the mass of the
// paths that met a flushed value (forward at a frame <= n, backward at a frame >= n).  So the spread of z_n over the frames
// measures what was lost: below MM_Z_SPREAD_TOL the marked values carried no mass that matters -- states of a component
// that decays against the rest of the graph (the initial contexts of the reference's WSJ denominator graph fall 2 log2 per
// frame behind: 2^-1400 after 700 frames) -- and the result stands; otherwise (or if z is not finite) the mark stays and the
// exact kernels compute the utterance again.  Marks of value 2 (a team of the split kernels did not run together) stay.
// Second condition, for the parity bar on SMALL posteriors (log gamma within 1e-4 relative wherever gamma > 1e-30): a term
// that dropped out of the linear path was below 2^-126 (times the few log2 the predicted normalisers are off by) of its
// frame's scale, so the posterior it would have had is below 2^(-120 - L_n), L_n = log2 sum_s 2^(a~_n(s) + b~_n(s)) as the
// kernels compute it (both factors on their own frame scale: L_n is far below 0 when the forward and the backward mass
// sit on different states, as under a sharp acoustic model -- log-softmax of 10 N(0,1): -160).  With L_n >= the floor (default
// MM_LT_FLOOR = -20) in every frame nothing above 2^-100 < 1e-30 can have been lost; a caller who accepts a larger posterior
// floor lowers it (mm_batch_set_posterior_floor: 1e-12 -> -80).
#ifndef MM_Z_SPREAD_TOL
#define MM_Z_SPREAD_TOL 2e-4  // log2 units (1.4e-4 nats on log Z); unmarked utterances of 1500 frames scatter by 2e-5..5e-5
#define MM_LT_FLOOR (-20.0)
#endif
static __global__ void mm_pair_finish_kernel(RunParams p) {
    const int b = blockIdx.x;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int P = p.utts[b].P1 - 1;
    if (threadIdx.x == 0) {
        const double z0 = p.pair_zmin[6 * b], z1 = p.pair_zmin[6 * b + 1];
        const double z = z0 < z1 ? z0 : z1;
        const double y0 = p.pair_zmin[6 * b + 2], y1 = p.pair_zmin[6 * b + 3];
        const double zM = y0 > y1 ? y0 : y1;
        const double l0 = p.pair_zmin[6 * b + 4], l1 = p.pair_zmin[6 * b + 5];
        const double lm = l0 < l1 ? l0 : l1;
        p.ttl[b] = (z < __builtin_inf()) ? (float)(z * (double)MM_LN2) : MM_NINF;  // (no frame: no path of length 0)
        int mark = p.redo[b];
        const bool sound = z > -__builtin_inf() && zM < __builtin_inf() && zM - z <= MM_Z_SPREAD_TOL && lm >= (double)p.lt_floor;
        if (mark == 1 && sound && p.clear_marks) p.redo[b] = mark = 0;
        // MM_PAIR_LINFIN: an overflow raises no mark in the kernels; it ends as a frame sum that is not a number (zM = inf)
        if (MM_PAIR_LINFIN && mark == 0 && len >= 1 && !(zM < __builtin_inf())) p.redo[b] = mark = 1;
        // The second condition holds for UNMARKED utterances too (round 5): a range mark says that a value of a VECTOR may have been
        // flushed; the product of the combine, 2^(a~ + b~), is flushed below 2^-126 whatever the vectors' ranges -- no mark -- and what
        // it was worth is bounded by the same overlap term.  (Until then an unmarked utterance whose forward and backward mass
        // overlapped at 2^-50 could lose a posterior of 1e-28: found by tools/fuzz_round3.py, SEED=1.)  All frames without mass: an
        // utterance without a path, Z = 0 like the reference's.
        if (mark == 0 && len >= 1 && !(z == -__builtin_inf() && zM == -__builtin_inf()) && !(lm >= (double)p.lt_floor)) p.redo[b] = mark = 1;
        if (p.stat_mode == 0) report_hard(p, mark != 0);
    }
    const long long gbase = (long long)b * p.gsb;
    for (long long q = threadIdx.x; q < (long long)(p.N - len) * P; q += blockDim.x)
        p.gamma[gbase + (len + q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
    // (teams) an utterance that stays marked goes to the float64 team kernels: their exchange areas must read "nothing has arrived"
    // -- zeroed here, for the marked utterances only (a memset node over all of them was 12 .. 26 us of every call)
    if (p.x_H > 0 && p.xbuf_d) {
        __shared__ int marked;
        __syncthreads();
        if (threadIdx.x == 0) marked = p.redo[b];
        __syncthreads();
        if (marked) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const long long nrow = 2ll * p.x_H * 2 * p.x_slot, nps = 2ll * p.x_H * 4 * p.x_psn;  // floats per utterance and phase / of partial sums
            for (int ph = 0; ph < 2; ++ph) {
                f4 *q = reinterpret_cast<f4 *>(p.xbuf_d + ph * p.x_phase_d + (long long)b * nrow);
                for (long long i = threadIdx.x; i < nrow / 4; i += blockDim.x) q[i] = f4{0.f, 0.f, 0.f, 0.f};
            }
            f4 *q = reinterpret_cast<f4 *>(p.xps_d + (long long)b * nps);
            for (long long i = threadIdx.x; i < nps / 4; i += blockDim.x) q[i] = f4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// alpha-recursion / beta-recursion export, second half (mm_fbx_kernel first): the rows phase A stored -- [frame][position][2
// utterances] linear float32 values + a float64 log2 offset per utterance and frame -- into the reference's layout, element (b, n, s)
// at out[(n - 1) * out_stride_n + state_off_b + s] in natural logarithms (src/inference.jl:62-74, 99-110: state_A / state_B).
// One workgroup per (pair, chunk of frames): a row of pairs is read as it lies (coalesced), scattered to state order through LDS
// (position -> state: RowDev::order), and leaves as two contiguous rows.  dir 1: frame N + 1 is B[:, N+1] = one(K) (:103).
// Utterances the kernel marked (values beyond float32's range) are skipped: the item kernel computes them behind this launch.
static __global__ void __launch_bounds__(1024) mm_pair_export_kernel(RunParams p, int dir, int frames_per_block, int H) {
    extern __shared__ float xs[];  // [2][S1p] values by state, then [S1p] ints: position -> state (-1: padding / a state the forms dropped)
    constexpr int NT = 1024, NV = 5;  // (a team of 4 has a vector of up to ~4100 positions with the padding between its regions)
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int S1p = p.pair_s1p;
    int *posmap = reinterpret_cast<int *>(xs + 2 * S1p);
    int b[2], valid[2];
    for (int u = 0; u < 2; ++u) {
        const int i = 2 * pair + u;
        valid[u] = i < p.B;
        b[u] = valid[u] ? i : p.B - 1;
        if (valid[u] && p.redo[b[u]] != 0) valid[u] = 0;
    }
    if (!valid[0] && !valid[1]) return;  // (both marked: the item kernel's)
    const UttDesc &ud = p.utts[b[0]];
    const int S1 = ud.S1;
    const float *rowsP = p.ws_alpha + (long long)pair * (long long)(p.N + 2) * 2 * S1p;
    const int f0 = 1 + (int)blockIdx.y * frames_per_block, f1 = min(p.N + 1, f0 + frames_per_block - 1);
    // position -> state over the whole vector: the sets' regions one after the other, each in its own form's finishing order
    for (int q = tid; q < S1p; q += NT) posmap[q] = -1;
    __syncthreads();
    for (int h = 0; h < H; ++h) {
        const RowDev &r = H > 1 ? ud.rps[dir][h] : ud.rp[dir];
        const int base = H > 1 ? p.sp_base[h] : 0, cnt = H > 1 ? p.sp_cnt[h] : r.rows;
        for (int i = tid; i < cnt; i += NT) posmap[base + i] = r.order[i];
    }
    __syncthreads();
    int st[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) st[k] = tid + k * NT < S1p ? posmap[tid + k * NT] : -1;
    const long long slot1 = 2 * pair + 1 < p.B ? b[1] : p.B;  // (the offsets' slots are pair_agent's: the spare slot B for the copy that fills an odd batch's last pair)
    auto load_row = [&](int f, mm_f32x2 (&v)[NV]) __attribute__((always_inline)) {
        const mm_f32x2 *row = reinterpret_cast<const mm_f32x2 *>(rowsP + (long long)f * 2 * S1p);
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = st[k] >= 0 ? row[tid + k * NT] : mm_f32x2{0.f, 0.f};
    };
    mm_f32x2 nxt[NV];
    load_row(f0 <= p.N + (dir ? 0 : 1) ? f0 : 1, nxt);
    for (int f = f0; f <= f1; ++f) {
        mm_f32x2 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = nxt[k];
        const bool ones = dir == 1 && f == p.N + 1;  // fill!(B[:, end], one(K))   (src/inference.jl:103: every state, not only the final one)
        const double o0 = ones ? 0.0 : p.ws_c[(long long)b[0] * (p.N + 2) + f], o1 = ones ? 0.0 : p.ws_c[slot1 * (p.N + 2) + f];
        if (f + 1 <= f1 && !(dir == 1 && f + 1 == p.N + 1)) load_row(f + 1, nxt);  // (the next frame's row travels while this one is written)
        if (H > 1 || ones)  // (states without a position: zero(K); the last column of beta: one(K))
            for (int i = tid; i < S1; i += NT) xs[i] = xs[S1p + i] = ones ? 0.f : MM_NINF;
        if (H > 1 || ones) __syncthreads();
        if (!ones) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                if (st[k] >= 0) {
                    // (log2 of a float in double: the offsets reach thousands of log2, whose float ulp is what a posterior is compared at)
                    xs[st[k]] = v[k].x > 0.f ? (float)(((double)fast_log2(v[k].x) + o0) * (double)MM_LN2) : MM_NINF;
                    xs[S1p + st[k]] = v[k].y > 0.f ? (float)(((double)fast_log2(v[k].y) + o1) * (double)MM_LN2) : MM_NINF;
                }
        }
        __syncthreads();
        for (int u = 0; u < 2; ++u)
            if (valid[u]) {
                float *dst = p.out + (long long)(f - 1) * p.out_stride_n + p.utts[b[u]].state_off;
                for (int i = tid; i < S1; i += NT) dst[i] = xs[u * S1p + i];
            }
        __syncthreads();
    }
}

}  // namespace mm
