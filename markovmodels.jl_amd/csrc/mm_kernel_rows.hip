// mm_kernel_rows.hip -- the "row" pdfposteriors kernels for gfx950: forward (alpha-recursion,
// src/inference.jl:62-74) and backward (beta-recursion :99-110 fused with the combine :154-160).
// Included by mm_engine.hip after mm_kernel_quad.hip.
//
// Same contract and numerics as the quad kernels (linear-domain row products with a frame-level shift,
// exact log-sum-exp fallback per row), different organisation of a frame (mm_rows.h):
//
//   * a row of the semiring product belongs to ONE lane (an aligned group of 2..64 lanes when it is long),
//     which holds the row's arcs in registers (linear weight + absolute LDS byte address: two VGPRs per
//     arc, no unpacking) and finishes the row itself -- (group sum by DPP,) log2, emission, normaliser, 2^x,
//     stores.  The quad kernels write per-lane partial sums to LDS, cross a barrier and read them back in a
//     second phase; here a frame has ONE barrier, no partial sums in LDS, and the LDS-bound gathers of one
//     wave overlap the issue-bound finishing code of another instead of all waves being in the same phase.
//   * the linear vector p = 2^a~ is double buffered by frame parity (the buffer being read / written is a
//     compile-time immediate offset of the ds instructions: the frame loop is unrolled by two);
//   * one SERVICE wave per workgroup (the last) owns all traffic that would otherwise make the compute waves
//     wait on vmcnt: it stages the emissions (one frame ahead) and, backward, the alpha rows (two frames
//     ahead) from HBM into LDS, normalises and stores the posteriors, and keeps the normaliser sums.  The
//     compute waves issue only fire-and-forget stores (forward: the alpha store, straight from registers,
//     coalesced in the forward numbering) and no global loads at all in steady state;
//   * the per-frame emission maximum E_n is part of the frame normaliser, so log-likelihoods far from 0 (GMM
//     scores around -300 nats) stay on the linear path; the other part S_n is chosen by the service wave TWO steps
//     ahead, from the maximum over the states of frame n-2, which it finds by scanning the log2 vector while the
//     compute waves work on frame n-1 (struct RowNorm; a~_n = a_n - C_n, C_n = sum_{k<=n} (E_k + S_k)).  No wave
//     reduction and no dependent LDS round trip sits between the barrier and the first gathers of a frame.
#pragma once
#include "mm_kernel_quad.hip"
#include "mm_rows.h"

namespace mm {

typedef float mm_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float *lds_fptr;
typedef __attribute__((address_space(3))) unsigned *lds_uptr;
__device__ __forceinline__ float ldsr(unsigned addr) { return *(lds_cfptr)(__UINTPTR_TYPE__)addr; }
__device__ __forceinline__ unsigned ldsru(unsigned addr) { return *(lds_uptr)(__UINTPTR_TYPE__)addr; }
__device__ __forceinline__ void ldsw(unsigned addr, float v) { *(lds_fptr)(__UINTPTR_TYPE__)addr = v; }
__device__ __forceinline__ void ldswu(unsigned addr, unsigned v) { *(lds_uptr)(__UINTPTR_TYPE__)addr = v; }

// The barrier that ends a step.  The scheduling fences matter: with a bare __syncthreads() the compiler moves code of
// the neighbouring steps across it in a way that cost the pair kernels 20 % (3.6 -> 4.4 ms per call, measured).
#ifndef MM_STEP_SYNC
#define MM_STEP_SYNC()                     \
    do {                                   \
        __builtin_amdgcn_sched_barrier(0); \
        __syncthreads();                   \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)
#endif

#define MM_ROW_EMS 1024  // bytes per emission buffer: P1p + 4 <= 256 floats
#define MM_ROW_KA_PAD 48 // arc-slot rows of the device arrays (>= every instantiated register window)

// LDS byte layout (absolute addresses; the kernels have no static LDS, the dynamic segment starts at 0).
// Everything a finish touches sits at a compile-time offset from the row's 4 * position.
template <int RS, int PASS>
struct RowLay {
    static constexpr unsigned P(int par, int c) { return unsigned(par * (2 * RS + 128) + c * (RS + 64)); }
    // targets of the service wave's LDS-DMA (kept below 64 KiB): raw emissions of 4 frames in flight (frame & 3),
    // backward also the alpha rows of 3 frames (frame % 3)
    static constexpr unsigned AL(int k) { return unsigned(4 * RS + 256 + k * RS); }  // backward only
    static constexpr unsigned RAW(int k) { return unsigned((PASS ? 7 : 4) * RS + 256 + k * 1024); }
    static constexpr unsigned Q(int par) { return RAW(4) + unsigned(par * RS); }     // backward only
    static constexpr unsigned EMB = PASS ? Q(2) : RAW(4);
    static constexpr unsigned EM(int par) { return EMB + unsigned(par * MM_ROW_EMS); }
    static constexpr unsigned MS(int par) { return EMB + 2 * MM_ROW_EMS + unsigned(par * 64); }  // normaliser of a frame
    static constexpr unsigned PSUM(int par) { return EMB + 2 * MM_ROW_EMS + 128 + unsigned(par * MM_ROW_EMS); }
    static constexpr unsigned PDFSE = EMB + 4 * MM_ROW_EMS + 128;          // u16 [2 * P1] <= 1024 bytes
    static constexpr unsigned SLOTS = PDFSE + (PASS ? 1024u : 0u);
};
inline size_t row_lds_bytes(int RS, int pass, int nslotrows) {
    const size_t slots = size_t(nslotrows) * 64 * 4 * (pass ? 2 : 1);
    const size_t emb = size_t(pass ? 9 : 4) * RS + 256 + 4096;
    return emb + 4 * MM_ROW_EMS + 128 + (pass ? 1024 : 0) + slots;
}

// emissions of one frame (expand(), src/inference.jl:54-60) in the log2 domain relative to the frame's
// maximum E over the real pdfs: lane handles pdfs lane, lane + 64, ...  Returns E (0 when no real pdf emits).
__device__ __forceinline__ float row_stage_em(unsigned dst, const float (&raw)[4], int n, int len, int P, int lane) {
    float v[4], E = MM_NINF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = lane + 64 * j;
        v[j] = em_value(raw[j], n, len, P, q);
        if (q < P) E = max_nc(E, v[j]);
    }
    E = wave_max_rl(E);
    if (!(E > MM_NINF)) E = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = lane + 64 * j;
        if (q <= P) ldsw(dst + 4u * q, v[j] - E);
    }
    return E;
}
// LDS-DMA (cdna_hip_programming.md 5.7): one wave instruction moves 4 or 16 bytes per lane from per-lane global
// addresses straight into the LDS block [dst + 64 * lane-size): no destination register, so nothing the compiler
// would make the wave wait for; completion is counted by vmcnt, which the service wave waits on by hand.
#define MM_ROW_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// raw emissions of frame n (clamped to a valid frame and pdf: expand() decides later what they mean): always 4 DMAs
// of 256 bytes, so that the number of outstanding operations per step is a constant
template <int NJ = 4>  // NJ * 64 >= P + 1
__device__ __forceinline__ void row_dma_em(unsigned dst, const float *Vb, long long vsn, int n, int N, int P, int lane) {
    const int nn = n < 1 ? 1 : (n > N ? N : n);
    const float *row = Vb + (long long)(nn - 1) * vsn;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        dma_b32(row + (q < P ? q : P - 1), dst + 256u * j);
    }
}
__device__ __forceinline__ void row_read_em(float (&raw)[4], unsigned src, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = ldsr(src + 256u * j + 4u * lane);
}

// Range of the linear path.  A state whose normalised log2 value v is finite but below -thr is alive with a linear
// image 2^v that products with the arc weights (>= 2^wmin) could no longer represent as normal floats: sums over it
// would silently lose it.  The row kernels do not handle that case: they mark the UTTERANCE in `redo`, and the exact
// kernels (quad / item: per-row log-sum-exp fallback) run it again after them (mm_engine.hip) -- results are the log
// semiring's for every input, only speed depends on the data.  thr = 125 + wmin (RowDev::thr), so every product of
// an unmarked utterance is >= 2^-125; a row sum that is exactly 0 then means that every source is exactly zero(K),
// and log2 of it (-inf) is the semiring's answer.  (An exact row walk inlined in every finish -- what the quad
// kernels do -- costs the compute waves ~40 VGPRs, which the register-resident graph needs.)
__device__ __forceinline__ bool row_out_of_range(float v, float thr) { return __builtin_fabsf(v) > thr && v != MM_NINF; }

// Sum over aligned groups of 1 << lg lanes, valid in the LAST lane of every group (the lane that finishes the row):
// five one-instruction DPP steps at most, no LDS crossbar and no lane-index registers (the butterfly of the item
// kernel needs ds_bpermute and three VGPRs of lane arithmetic for groups wider than a DPP row).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, true));
}
__device__ __forceinline__ float grp_sum_last(float v, int lg) {
    v = dpp_add<0x111, 0xF>(v);                // row_shr:1
    if (lg >= 2) v = dpp_add<0x112, 0xF>(v);   // row_shr:2
    if (lg >= 3) v = dpp_add<0x114, 0xF>(v);   // row_shr:4
    if (lg >= 4) v = dpp_add<0x118, 0xF>(v);   // row_shr:8
    if (lg >= 5) v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    if (lg >= 6) v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

// The same for the two utterances of a pair workgroup, with ONE wave-uniform branch per level that is still needed
// (a branch costs a wave about three VALU instructions of issue time on gfx950; grp_sum_last() called twice tests
// every level for both values).  The steps commute (I + shift^k), so the nesting order is free.
__device__ __forceinline__ void grp_sum_last2(float &a, float &b, int lg) {
    a = dpp_add<0x111, 0xF>(a);
    b = dpp_add<0x111, 0xF>(b);
    if (lg >= 2) {
        a = dpp_add<0x112, 0xF>(a);
        b = dpp_add<0x112, 0xF>(b);
        if (lg >= 3) {
            a = dpp_add<0x114, 0xF>(a);
            b = dpp_add<0x114, 0xF>(b);
            if (lg >= 4) {
                a = dpp_add<0x118, 0xF>(a);
                b = dpp_add<0x118, 0xF>(b);
                if (lg >= 5) {
                    a = dpp_add<0x142, 0xA>(a);
                    b = dpp_add<0x142, 0xA>(b);
                    if (lg >= 6) {
                        a = dpp_add<0x143, 0xC>(a);
                        b = dpp_add<0x143, 0xC>(b);
                    }
                }
            }
        }
    }
}

// Service wave: log2 of the maximum of the linear vector at LDS byte address pbase (n4 float4s, n4 <= 64 * NB); -inf if
// nothing is alive.  (The log2 vector itself is not kept in LDS: one store per finish less.)  All loads are issued
// before the first maximum (clamped indices: a duplicate changes no maximum): a loop with one load per trip costs the
// wave one LDS round trip per trip.
template <int NB>
__device__ __forceinline__ float row_scan_max(unsigned pbase, int n4, int lane) {
    mm_f32x4 v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = lane + 64 * j;
        v[j] = *(__attribute__((address_space(3))) const mm_f32x4 *)(__UINTPTR_TYPE__)(pbase + 16u * (q < n4 ? q : n4 - 1));
    }
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) m = max_nc(max_nc(m, max_nc(v[j].x, v[j].y)), max_nc(v[j].z, v[j].w));
    return fast_log2(wave_max_rl(m));
}

// Service wave: the frame normaliser.  Frame k subtracts S_k, chosen two steps ahead (the maximum m_{k-2} of frame k-2
// is the newest one known when S_k is posted).  With m_k = m_{k-1} + c_k - S_k (c_k: the frame's own growth),
// subtracting the stale maximum itself (S_k = m_{k-2}) gives m_k = m_{k-1} - m_{k-2} + c_k: an undamped oscillator
// driven by the noise of c, whose amplitude random-walks out of the float range over a long utterance.  Instead the
// missing step is predicted with a running mean cbar of the observed growths:
//     S_{k} = m_{k-2} - S_{k-1} + 2 cbar   =>   m_k = (c_{k-1} - cbar) + (c_k - cbar):  bounded, nothing accumulates.
// Whatever float is posted is exactly what the compute waves subtract and what the bookkeeping adds up.
struct RowNorm {
    float m_prev = 0.f, s_cur = 0.f, s_prev = 0.f, cbar = 0.f;
    int seen = 0;
    // m: maximum of the newest complete frame; returns the normaliser of the frame after next
    __device__ __forceinline__ float next(float m) {
        if (!(m > MM_NINF)) {  // nothing alive: the utterance has no path; keep the state
            s_prev = s_cur;
            s_cur = 0.f;
            return 0.f;
        }
        if (seen >= 1) {
            const float c = m - m_prev + s_prev;
            cbar = seen == 1 ? c : cbar + 0.25f * (c - cbar);
        }
        const float s_next = m - s_cur + 2.f * cbar;
        m_prev = m;
        s_prev = s_cur;
        s_cur = s_next;
        ++seen;
        return s_next;
    }
};

// C' * (A .* B) (src/inference.jl:154-155) for the row kernels: states of one pdf are contiguous in the pdf-major
// order of Q; a wave sums 8 pdfs with 8 lanes each.  Split in two so that no dependent LDS round trip sits at the
// top of a step: the loads are issued right behind the first gathers of the frame (up to 4 per lane, independent;
// the pdf's range is loop invariant), the additions and the 3-step DPP reduction follow the last segment.
struct PdfLane {
    unsigned first4, end4;  // byte offsets of this lane's pdf range in Q (first4 includes the lane's own start)
    int pdf;                // -1: none
    bool lead;              // the lane of its group of 8 that stores the sum
};
__device__ __forceinline__ PdfLane pdf_lane(unsigned pdfse_base, int P1, int wave, int lane) {
    PdfLane pl;
    pl.pdf = wave * 8 + (lane >> 3);
    pl.lead = (lane & 7) == 0;
    pl.first4 = pl.end4 = 0u;
    if (pl.pdf < P1) {
        const unsigned se = ldsru(pdfse_base + 4u * pl.pdf);  // first | end << 16
        pl.first4 = 4u * ((se & 0xffffu) + (lane & 7));
        pl.end4 = 4u * (se >> 16);
    } else {
        pl.pdf = -1;
    }
    return pl;
}
__device__ __forceinline__ void pdf_load(float (&pq)[4], const PdfLane &pl, unsigned qbase) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned a = pl.first4 + 32u * k;
        pq[k] = ldsr(qbase + (a < pl.end4 ? a : 0u));  // (clamped: an always valid address, masked below)
    }
}
__device__ __forceinline__ void pdf_finish(const float (&pq)[4], const PdfLane &pl, unsigned qbase, unsigned psum_base) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (pl.first4 + 32u * k < pl.end4) ? pq[k] : 0.f;
    for (unsigned a = pl.first4 + 128u; a < pl.end4; a += 32u) s += ldsr(qbase + a);  // pdfs with more than 32 states
    s = grp_sum(s, 3);
    if (pl.pdf >= 0 && pl.lead) ldsw(psum_base + 4u * pl.pdf, s);
}
// pdfs beyond the first pass (more than 8 per compute wave): the plain dependent walk
__device__ __forceinline__ void pdf_sums_rest(unsigned qbase, unsigned pdfse_base, unsigned psum_base, int P1, int wave, int NWC,
                                              int lane) {
    for (int p0 = 8 * NWC + wave * 8; p0 < P1; p0 += NWC * 8) {
        const int pdf = p0 + (lane >> 3);
        float s = 0.f;
        if (pdf < P1) {
            const unsigned se = ldsru(pdfse_base + 4u * pdf);
            for (unsigned a = 4u * ((se & 0xffffu) + (lane & 7)); a < 4u * (se >> 16); a += 32u) s += ldsr(qbase + a);
        }
        s = grp_sum(s, 3);
        if (pdf < P1 && (lane & 7) == 0) ldsw(psum_base + 4u * pdf, s);
    }
}

// The arcs of a compute wave, statically unrolled (the graph registers need static indices); a segment may end after
// any pair of arcs.  The gathers run D pairs ahead of the FMAs.  Two pairs share ONE wave-uniform test of the end mask
// (a branch costs a wave 12-25 cycles of issue time, as much as a pair's gathers): the products of the second pair go
// to a side sum that joins the running sum unless a segment ends between the two; all products are in straight-line
// code, only the rare path (a segment ends here) looks at the bits one by one.  Every wave runs the whole window: the
// pairs behind its last segment have weight 0.
// The wave lowers its issue priority as it advances (s_setprio 2, 1, 0 by thirds of the window): the arbiter serves the
// highest priority, then the OLDEST wave, so with equal priorities the four waves of a SIMD finish one after the other
// and the youngest runs the tail of the step alone, latency bound, while the LDS idles; with priorities that fall with
// the progress a wave that is ahead yields to the ones behind and all reach the step's barrier together.
template <int K2, int KA, int D, unsigned RDOFF, class F>
__device__ __forceinline__ void row_pairs(const float (&wr)[KA], const unsigned (&ar)[KA], float (&x)[2 * D], float &acc,
                                          unsigned em_lo, unsigned em_hi, F &&finish) {
    constexpr int NP = KA / 2, T1 = ((NP + 2) / 3 + 1) & ~1, T2 = ((2 * NP + 2) / 3 + 1) & ~1;
    if constexpr (K2 == 0) __builtin_amdgcn_s_setprio(2);
    if constexpr (K2 == T1 && T1 > 0) __builtin_amdgcn_s_setprio(1);
    if constexpr (K2 == T2 && T2 > T1) __builtin_amdgcn_s_setprio(0);
    constexpr int s0 = (2 * K2) % (2 * D);
    const unsigned em = K2 < 32 ? em_lo : em_hi;
    if constexpr (K2 + 1 < NP) {
        constexpr int s1 = (2 * K2 + 2) % (2 * D);
        acc = fmaf(wr[2 * K2], x[s0], acc);
        float accN = wr[2 * K2 + 2] * x[s1];
        acc = fmaf(wr[2 * K2 + 1], x[s0 + 1], acc);
        accN = fmaf(wr[2 * K2 + 3], x[s1 + 1], accN);
        if constexpr (2 * (K2 + D) < KA) {
            x[s0] = ldsr(ar[2 * (K2 + D)] + RDOFF);
            x[s0 + 1] = ldsr(ar[2 * (K2 + D) + 1] + RDOFF);
        }
        if constexpr (2 * (K2 + 1 + D) < KA) {
            x[s1] = ldsr(ar[2 * (K2 + 1 + D)] + RDOFF);
            x[s1 + 1] = ldsr(ar[2 * (K2 + 1 + D) + 1] + RDOFF);
        }
        if (__builtin_expect(((em >> (K2 & 31)) & 3u) != 0u, 0)) {
            if ((em >> (K2 & 31)) & 1u) finish();  // (zeroes acc)
            acc += accN;
            accN = 0.f;
            if ((em >> ((K2 + 1) & 31)) & 1u) finish();
        }
        acc += accN;
        if constexpr (K2 + 2 < NP) row_pairs<K2 + 2, KA, D, RDOFF>(wr, ar, x, acc, em_lo, em_hi, finish);
    } else {
        acc = fmaf(wr[2 * K2], x[s0], acc);
        acc = fmaf(wr[2 * K2 + 1], x[s0 + 1], acc);
        if ((em >> (K2 & 31)) & 1u) finish();
    }
}

// A value that is the same in every lane, moved to scalar registers.  (The utterance descriptor is read through an
// index that may come from memory -- the longest-first order -- so the compiler keeps everything derived from it
// in vector registers otherwise: ~30 VGPRs that the graph needs.)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T *uni(T *ptr) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
struct RowU {  // RowDev in scalar registers
    const float *w;
    const unsigned *addr;
    const unsigned *slots;
    const RowSched *sched;
    const unsigned short *rowpdf, *pdfse;
    const float *init;
    int KA, NWC, nslotrows, fpos, rows;
    float thr;
};
__device__ __forceinline__ RowU uni(const RowDev &d) {
    RowU r;
    r.w = uni(d.w);
    r.addr = uni(d.addr);
    r.slots = uni(d.slots);
    r.sched = uni(d.sched);
    r.rowpdf = uni(d.rowpdf);
    r.pdfse = uni(d.pdfse);
    r.init = uni(d.init);
    r.KA = uni(d.KA);
    r.NWC = uni(d.NWC);
    r.nslotrows = uni(d.nslotrows);
    r.fpos = uni(d.fpos);
    r.rows = uni(d.rows);
    r.thr = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, d.thr)));
    return r;
}

// what a compute wave keeps across the frames
template <int KA>
struct RowRegs {
    float w[KA];
    unsigned a[KA];
};

template <int KA, int RS, int PASS>
__global__ void __launch_bounds__(1024) mm_fbr_kernel(RunParams p) {
    extern __shared__ float lds[];
    using L = RowLay<RS, PASS>;
#ifndef MM_ROW_D
#define MM_ROW_D 3
#endif
    constexpr int D = MM_ROW_D;  // gather pairs in flight ahead of the FMAs
    const int b = uni(p.order ? p.order[blockIdx.x] : (int)blockIdx.x);
    const UttDesc &u = p.utts[b];
    const RowU r = uni(u.r[PASS]);
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6, NWC = NW - 1;
    const bool service = wave == NWC;
    const int S1 = r.rows, S1p = uni(u.S1p), P1 = uni(u.P1), P = P1 - 1, P1p = (P1 + 3) & ~3;
    int len = uni(p.lens ? p.lens[b] : p.N);
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int NF = len + 1;
    const float *Vb = p.V + (long long)b * p.vsb;
    const long long s1p_prefix = ((long long)uni((int)(u.s1p_prefix >> 32)) << 32) | (unsigned)uni((int)u.s1p_prefix);
    float *wsA = p.ws_alpha + s1p_prefix * (long long)(p.N + 1);
    float *wsM = reinterpret_cast<float *>(p.ws_c + (long long)b * (p.N + 2));
    double *hand = p.ws_c + (long long)b * (p.N + 2) + p.N;
    if (lds_addr_of(lds) != 0u) __builtin_trap();  // the layout uses absolute LDS addresses
    MM_STAMP_DECL;
    // Issue arbitration between the four waves of a SIMD goes by priority, then age.  The service wave is the youngest
    // of its SIMD and would get the leftover slots only (measured with cycle stamps: its ~300 instructions took a whole
    // frame); its work is short, so it goes first.  (The compute waves lower their own priority as they advance
    // through a step, row_pairs(): the four waves of a SIMD then progress together whatever their age.)
    if (service) __builtin_amdgcn_s_setprio(3);

    // ---- set-up common to both directions
    const int fpos = r.fpos;
    if constexpr (PASS == 1) {
        if (uni(p.redo[b])) return;  // marked by the forward kernel: the exact kernels compute this utterance
        const double logZ2 = hand[1];
        if (!(logZ2 > -1e300)) {  // no accepting path: gamma = 0, ttl = -inf
            const long long gbase = (long long)b * p.gsb;
            for (long long q = tid; q < (long long)p.N * P; q += NT) p.gamma[gbase + (q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
            if (tid == 0) p.ttl[b] = MM_NINF;
            return;
        }
    }
    for (unsigned q = tid * 4u; q < 4u * RS + 256u; q += NT * 4u) ldsw(q, 0.f);                  // p, both parities
    if constexpr (PASS == 1)
        for (unsigned q = tid * 4u; q < 2u * RS; q += NT * 4u) ldsw(L::Q(0) + q, 0.f);
    if (tid < 32) ldsw(L::MS(0) + 4u * tid, 0.f);
    if (tid < 2) ldsw(L::EM(tid) + 4u * P1p, MM_NINF);  // the emission slot of lanes without a row
    const int nslotwords = r.nslotrows * 64 * (PASS ? 2 : 1);
    for (int q = tid; q < nslotwords; q += NT) ldswu(L::SLOTS + 4u * q, as_global(r.slots)[q]);
    int *redo = p.redo + b;
    const float thr = r.thr;
    if constexpr (PASS == 1)
        for (int q = tid; q < P1; q += NT) ldswu(L::PDFSE + 4u * q, as_global(reinterpret_cast<const unsigned *>(r.pdfse))[q]);
    // schedule of a compute wave
    unsigned long long endmask = 0, lgw0 = 0;
    int nslots = 0;
    unsigned slot_base = 0;
    if (!service && wave < r.NWC) {
        const RowSched &sc = r.sched[wave];
        endmask = sc.endmask;
        lgw0 = sc.lg;
        nslots = (int)(sc.nslots & 0xffffu);  // (the wave's arcs are right-aligned in the window, mm_rows.h: the leading pairs are zero)
        slot_base = L::SLOTS + (sc.slot0 * 64u + lane) * (PASS ? 8u : 4u);
    }
    // graph registers of a compute wave: loaded at the top of the compute branch of each direction, so that they are
    // never live together with the staging registers of the service wave
    RowRegs<KA> rg;
    auto load_graph = [&]() {
        // straight-line, no per-slot conditions: the device arrays are padded with zero rows up to MM_ROW_KA_PAD arc
        // slots (mm_engine.hip), so every wave loads KA rows whatever the graph's own KA is
        static_assert(KA <= MM_ROW_KA_PAD, "register window larger than the padding of the device arrays");
        const int nt = 64 * r.NWC;
        const bool mine = wave < r.NWC;
        const auto wp = as_global(r.w);
        const auto ap = as_global(r.addr);
        const int t0 = mine ? tid : 0;
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            rg.w[k] = wp[k * nt + t0];
            rg.a[k] = ap[k * nt + t0];
        }
        if (!mine) {
#pragma unroll
            for (int k = 0; k < KA; ++k) {
                rg.w[k] = 0.f;
                rg.a[k] = 0u;
            }
        }
    };
    unsigned em_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)endmask);
    unsigned em_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(endmask >> 32));
    lgw0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(lgw0 >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((unsigned)lgw0);
    nslots = __builtin_amdgcn_readfirstlane(nslots);
    __syncthreads();

    // The two roles run separate code from here on, each with the same sequence of barriers (two in the prologue, one
    // per frame): their register-resident state -- the graph of a compute wave, the staging registers of the service
    // wave -- is then never live together.
    if constexpr (PASS == 0) {
        // =================== forward: alpha-recursion (src/inference.jl:62-74) ===================
        // frame 1: alpha_hat (*) lhs[:,1]   (src/inference.jl:68), by all threads once its emissions are staged
        auto frame1 = [&]() {
            for (int i = tid; i < S1; i += NT) {
                const float v = as_global(r.init)[i] + ldsr(L::EM(1) + 4u * as_global(r.rowpdf)[i]);
                if (row_out_of_range(v, thr)) *redo = 1;
                const float pv = fast_exp2(v);
                ldsw(L::P(1, 0) + 4u * i, pv);
                ldsw(L::P(1, 1) + 4u * i, pv);
                wsA[(long long)1 * S1p + i] = v;
            }
        };
        if (service) {
            // raw emissions of frame f arrive by LDS-DMA in RAW(f & 3), requested three frames before they are staged
            float raw[4];
            double C = 0.0;
            for (int f = 1; f <= 4; ++f) row_dma_em(L::RAW(0) + 1024u * (f & 3), Vb, p.vsn, f, p.N, P, lane);
            MM_ROW_VMCNT(0);
            row_read_em(raw, L::RAW(1), lane);
            C += (double)row_stage_em(L::EM(1), raw, 1, len, P, lane);
            row_dma_em(L::RAW(1), Vb, p.vsn, 5, p.N, P, lane);  // (step n requests frame n + 4)
            __syncthreads();
            frame1();
            if (NF >= 2) {
                row_read_em(raw, L::RAW(2), lane);
                C += (double)row_stage_em(L::EM(0), raw, 2, len, P, lane);
            }
            __syncthreads();
            RowNorm norm;
            auto step = [&](auto RDc, int n) {
                constexpr int RD = decltype(RDc)::value;  // = parity of frame n+1
                // frame n+1: its emissions (requested at step n-2; the 4 DMAs of step n-1 may still be in flight), and its
                // normaliser from the maximum of frame n-1 (complete since the last barrier)
                MM_ROW_VMCNT(4);
                if (n + 1 <= NF) {
                    row_read_em(raw, L::RAW(0) + 1024u * ((n + 1) & 3), lane);
                    const float M = norm.next(row_scan_max<(RS / 16 + 63) / 64>(L::P(RD, 0), (S1 + 3) >> 2, lane));
                    C += (double)M;
                    if (lane == 0) {
                        ldsw(L::MS(RD), M);
                        wsM[n + 1] = M;  // what frame n+1 subtracts (frames 1 and 2: nothing)
                    }
                    C += (double)row_stage_em(L::EM(RD), raw, n + 1, len, P, lane);
                }
                row_dma_em(L::RAW(0) + 1024u * ((n + 4) & 3), Vb, p.vsn, n + 4, p.N, P, lane);  // (its buffer was read at step n-1)
                MM_STAMP(0);
                MM_STEP_SYNC();
                MM_STAMP(1);
            };
            MM_STAMP_RESET;
            for (int n = 2; n <= NF; n += 2) {
                step(std::integral_constant<int, 1>{}, n);
                if (n + 1 <= NF) step(std::integral_constant<int, 0>{}, n + 1);
            }
            if (lane == 0) {
                const float afin = fast_log2(ldsr(L::P(NF & 1, 0) + 4u * fpos));  // normalised log2 value of the final state, last frame
                hand[0] = (double)afin;
                hand[1] = (double)afin + C;
            }
        } else {
            __syncthreads();
            frame1();
            __syncthreads();
            load_graph();
            auto step = [&](auto RDc, int n) {
                constexpr int RD = decltype(RDc)::value, WR = 1 - RD;
                if (nslots > 0) {
                    // the first gathers leave before anything else
                    float x[2 * D];
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        x[2 * j] = (2 * j < KA) ? ldsr(rg.a[(2 * j < KA) ? 2 * j : 0] + L::P(RD, 0)) : 0.f;
                        x[2 * j + 1] = (2 * j + 1 < KA) ? ldsr(rg.a[(2 * j + 1 < KA) ? 2 * j + 1 : 0] + L::P(RD, 0)) : 0.f;
                    }
                    unsigned sa = slot_base;
                    unsigned info = ldsru(sa);
                    const float M = ldsr(L::MS(WR));  // M_{n-2}, posted by the service wave during the previous step
                    float e = ldsr((info >> 16) + L::EM(WR));
                    // (the alpha store has rows 0..N; frame len+1 is not needed by the backward pass: it goes to the unused row 0)
                    float *wsAn = wsA + (long long)(n <= len ? n : 0) * S1p;
                    float acc = 0.f;
                    unsigned long long lgw = lgw0;
                    float worst = 0.f;  // largest finite |value| of the lane in this step: range check at its end
                    auto finish = [&]() {
                        const int lg = (int)(lgw & 15ull);
                        lgw >>= 4;
                        float s = acc;
                        if (lg) s = grp_sum_last(s, lg);
                        const unsigned pos4 = info & 0xffffu;
                        const float v = fast_log2(s) + e - M;  // (T' alpha_{n-1}) (*) lhs[:,n]   (src/inference.jl:70-71)
                        worst = __builtin_fmaxf(worst, __builtin_fmaf(__builtin_fabsf(v), 0.f, __builtin_fabsf(v)));  // (NaN for -inf: ignored)
                        const float pv = fast_exp2(v);
                        ldsw(pos4 + L::P(WR, 0), pv);
                        ldsw(pos4 + L::P(WR, 1), pv);
                        *reinterpret_cast<float *>(reinterpret_cast<char *>(wsAn) + pos4) = v;
                        acc = 0.f;
                        sa += 256u;
                        info = ldsru(sa);
                        e = ldsr((info >> 16) + L::EM(WR));
                    };
                    // (opaque per step: hoisted out of the frame loop, the bit tests of all pairs would each occupy a scalar
                    // register pair for the whole loop)
                    asm volatile("" : "+s"(em_lo), "+s"(em_hi));
                    row_pairs<0, KA, D, L::P(RD, 0)>(rg.w, rg.a, x, acc, em_lo, em_hi, finish);
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(worst > thr) != 0ull, 0)) *redo = 1;
                }
                MM_STAMP(0);
                MM_STEP_SYNC();
                MM_STAMP(1);
            };
            MM_STAMP_RESET;
            for (int n = 2; n <= NF; n += 2) {
                step(std::integral_constant<int, 1>{}, n);
                if (n + 1 <= NF) step(std::integral_constant<int, 0>{}, n + 1);
            }
        }
    } else {
        // ============ backward: beta-recursion (src/inference.jl:99-110) fused with the combine (:154-160) ============
        const float afin = (float)hand[0];
        const long long gbase = (long long)b * p.gsb;
        float tmin = 0.f;
        if (tid == 0) {  // frame len+1: B (*) lhs = one for the final state only
            ldsw(L::P(NF & 1, 0) + 4u * fpos, 1.f);
            ldsw(L::P(NF & 1, 1) + 4u * fpos, 1.f);
        }
        // run steps n = len .. 1; step n reads y_{n+1} from parity (n + 1) & 1
        auto run = [&](auto &&step) {
            int n = len;
            if (n >= 1 && ((n + 1) & 1) == 0) {
                step(std::integral_constant<int, 0>{}, n);
                --n;
            }
            for (; n >= 1; n -= 2) {
                step(std::integral_constant<int, 1>{}, n);
                if (n - 1 >= 1) step(std::integral_constant<int, 0>{}, n - 1);
            }
        };
        if (service) {
            // raw emissions (RAW(f & 3)) and alpha rows (AL(f % 3)) arrive by LDS-DMA, requested three and two steps
            // before the step that uses them
            float raw[4];
            const int n4 = S1p >> 2;
            constexpr int NA = (RS / 4 + 255) / 256;
            auto dma_arow = [&](int f) {
                const int ff = f < 1 ? 1 : f;
                const mm_f32x4 *src = reinterpret_cast<const mm_f32x4 *>(wsA + (long long)ff * S1p);
                const unsigned dst = L::AL(0) + (unsigned)(ff % 3) * RS;
#pragma unroll
                for (int j = 0; j < NA; ++j) {
                    const int q = lane + 64 * j;
                    dma_b128(src + (q < n4 ? q : 0), dst + 1024u * j);  // (always NA DMAs: a constant number in flight)
                }
            };
            if (len >= 1) {
                for (int f = len; f >= len - 2; --f) row_dma_em(L::RAW(0) + 1024u * (f & 3), Vb, p.vsn, f, p.N, P, lane);
                dma_arow(len);
                dma_arow(len - 1);
                MM_ROW_VMCNT(0);
                row_read_em(raw, L::RAW(0) + 1024u * (len & 3), lane);
                (void)row_stage_em(L::EM(len & 1), raw, len, len, P, lane);
            }
            // kappa_n = log2 Z - C_n - D_n (what a~ + b~ of frame n still lacks to a log2 posterior) = afin + G_n with
            // G_n = sum_{k>n} S_k(forward) - sum_{k>=n} S_k(backward): accumulated in double (its magnitude stays small),
            // posted to the compute waves together with the frame's normaliser.  Frame m of the forward pass subtracted
            // wsM[m] (frames 1 and 2 nothing); the first two backward frames subtract nothing.
            double G = len + 1 >= 3 ? (double)wsM[len + 1] : 0.0;  // G_len
            if (lane == 0) ldsw(L::MS(len & 1) + 4u, afin + (float)G);
            __syncthreads();
            RowNorm norm;
            MM_STAMP_RESET;
            run([&](auto RDc, int n) {
                constexpr int RD = decltype(RDc)::value, WR = 1 - RD;  // RD = parity of frames n+1 and n-1
                // emissions of frame n-1 (requested at step n+2; the 4 + NA DMAs of step n+1 may still be in flight)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NA) : "memory");
                if (n - 1 >= 1) {
                    row_read_em(raw, L::RAW(0) + 1024u * ((n - 1) & 3), lane);
                    (void)row_stage_em(L::EM(RD), raw, n - 1, len, P, lane);
                }
                row_dma_em(L::RAW(0) + 1024u * ((n - 3) & 3), Vb, p.vsn, n - 3, p.N, P, lane);
                dma_arow(n - 2);
                // the normaliser of frame n-1, from the maximum of y_{n+1} (complete since the last barrier)
                if (n - 1 >= 1) {
                    const float M = norm.next(row_scan_max<(RS / 16 + 63) / 64>(L::P(RD, 0), (S1 + 3) >> 2, lane));
                    G += (double)(n >= 3 ? wsM[n] : 0.f) - (double)M;  // G_{n-1} = G_n + S_n(forward) - S_{n-1}(backward)
                    if (lane == 0) {
                        ldsw(L::MS(RD), M);
                        ldsw(L::MS(RD) + 4u, afin + (float)G);
                    }
                }
                // gamma of frame n+2: its per-pdf sums were completed in the previous step
                if (n + 2 <= len) {
                    const float s = finish_frame(reinterpret_cast<float *>(lds) + L::PSUM(WR) / 4, P1, P, lane,
                                                 p.gamma + gbase + (long long)(n + 1) * p.gsn, p.gsp);
                    tmin = fminf(tmin, fast_log2(s));
                }
                // the alpha row of frame n-1 (requested at step n+1) must be in LDS when the compute waves leave the barrier
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NA) : "memory");
                MM_STAMP(0);
                MM_STEP_SYNC();
                MM_STAMP(1);
            });
        } else {
            __syncthreads();
            load_graph();
            const PdfLane pl = pdf_lane(L::PDFSE, P1, wave, lane);  // the pdf this lane sums (first pass: 8 pdfs per wave)
            MM_STAMP_RESET;
            run([&](auto RDc, int n) {
                constexpr int RD = decltype(RDc)::value, WR = 1 - RD;
                float x[2 * D], pq[4];
                if (nslots > 0) {  // the first gathers leave before anything else
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        x[2 * j] = (2 * j < KA) ? ldsr(rg.a[(2 * j < KA) ? 2 * j : 0] + L::P(RD, 0)) : 0.f;
                        x[2 * j + 1] = (2 * j + 1 < KA) ? ldsr(rg.a[(2 * j + 1 < KA) ? 2 * j + 1 : 0] + L::P(RD, 0)) : 0.f;
                    }
                }
                const float M = ldsr(L::MS(WR));           // the normaliser of this step (chosen two steps ago; 0 for the first two)
                const float kappa = ldsr(L::MS(WR) + 4u);  // ... and what a~ + b~ lacks to a log2 posterior, both from the service wave
                const unsigned alb = L::AL(0) + (unsigned)(n % 3) * RS;  // where the service wave put the alpha row of frame n
                pdf_load(pq, pl, L::Q(RD));  // frame n+1, per pdf (finished after the segments)
                if (nslots > 0) {
                    float acc = 0.f;
                    unsigned long long lgw = lgw0;
                    unsigned sa = slot_base;
                    // (both words of a slot in one 8-byte read; an integer vector type: see mm_kernel_pairs.hip ldsr2u)
                    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                    auto slot2 = [](unsigned a) { return *(__attribute__((address_space(3))) const u32x2_t *)(__UINTPTR_TYPE__)a; };
                    const u32x2_t w0 = slot2(sa);
                    unsigned info = w0.x, info2 = w0.y;
                    float e = ldsr((info >> 16) + L::EM(WR));
                    float al = ldsr((info2 & 0xffffu) + alb);
                    float worst = 0.f;  // largest finite |value| of the lane in this step: range check at its end
                    auto finish = [&]() {
                        const int lg = (int)(lgw & 15ull);
                        lgw >>= 4;
                        float s = acc;
                        if (lg) s = grp_sum_last(s, lg);
                        const unsigned pos4 = info & 0xffffu;
                        const float beta = fast_log2(s) - M;  // T (B[:,n+1] (*) lhs[:,n+1])  (src/inference.jl:106-107)
                        ldsw((info2 >> 16) + L::Q(WR), fast_exp2(al + beta - kappa));
                        const float y = beta + e;
                        worst = __builtin_fmaxf(worst, __builtin_fmaf(__builtin_fabsf(y), 0.f, __builtin_fabsf(y)));  // (NaN for -inf: ignored)
                        const float py = fast_exp2(y);
                        ldsw(pos4 + L::P(WR, 0), py);
                        ldsw(pos4 + L::P(WR, 1), py);
                        acc = 0.f;
                        sa += 512u;
                        const u32x2_t w1 = slot2(sa);
                        info = w1.x;
                        info2 = w1.y;
                        e = ldsr((info >> 16) + L::EM(WR));
                        al = ldsr((info2 & 0xffffu) + alb);
                    };
                    asm volatile("" : "+s"(em_lo), "+s"(em_hi));
                    row_pairs<0, KA, D, L::P(RD, 0)>(rg.w, rg.a, x, acc, em_lo, em_hi, finish);
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(worst > thr) != 0ull, 0)) *redo = 1;
                }
                if (n < len) {
                    pdf_finish(pq, pl, L::Q(RD), L::PSUM(RD));
                    if (P1 > 8 * NWC) pdf_sums_rest(L::Q(RD), L::PDFSE, L::PSUM(RD), P1, wave, NWC, lane);
                }
                MM_STAMP(0);
                MM_STEP_SYNC();
                MM_STAMP(1);
            });
        }
        // gamma of frames 2 and 1, zeros beyond len, ttl
        if (service && len >= 2) {  // frame 2: summed in the last step
            const float s = finish_frame(reinterpret_cast<float *>(lds) + L::PSUM(0) / 4, P1, P, lane, p.gamma + gbase + p.gsn, p.gsp);
            tmin = fminf(tmin, fast_log2(s));
        }
        if (len >= 1) {
            if (!service)
                pdf_sums(reinterpret_cast<float *>(lds) + L::Q(1) / 4, reinterpret_cast<unsigned short *>(lds) + L::PDFSE / 2,
                         reinterpret_cast<float *>(lds) + L::PSUM(1) / 4, P1, wave, NWC, lane);
            __syncthreads();
            if (service) {
                const float s = finish_frame(reinterpret_cast<float *>(lds) + L::PSUM(1) / 4, P1, P, lane, p.gamma + gbase, p.gsp);
                tmin = fminf(tmin, fast_log2(s));
            }
        }
        for (long long q = tid; q < (long long)(p.N - len) * P; q += NT)
            p.gamma[gbase + (len + q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
        if (service && lane == 0)  // ttl = log Z + min over frames of log(per-frame sum)   (src/inference.jl:159)
            p.ttl[b] = (float)((hand[1] + (double)tmin) * (double)MM_LN2);
    }
#ifdef MM_STAMPS
    if (p.dbg && lane == 0)
        for (int k = 0; k < 8; ++k) p.dbg[((long long)b * MM_MAX_WAVES + wave) * 16 + 8 * PASS + k] = stamp_acc[k];
#endif
}

}  // namespace mm
