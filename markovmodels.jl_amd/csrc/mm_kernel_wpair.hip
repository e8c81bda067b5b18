// mm_kernel_wpair.hip -- the pair kernels (mm_kernel_pairs.hip) on "wide-exponent 32-bit" values: TWO utterances per workgroup
// with a float64's exponent range, for whole batches of sharp emissions (a trained acoustic model: the forward and the backward
// mass of a frame overlap ~160 log2 below their maxima, beyond float32's 126).
//
// The float64 exact pair kernels (mm_kernel_dpair.hip) take such batches with ONE utterance per workgroup: the 8 bytes of a
// state in LDS hold one double, so B = 256 utterances are two rounds of 256 workgroups (5.5 ms where the float32 pair kernels
// need 2.8).  What bounds a step of either kernel is the LDS -- a gather of 64 lanes costs ~1.45 cycles of the LDS whether the
// lanes read 4 or 8 bytes (tools/dev/lds_test.hip; 32 lanes per pass, bank = position mod 32 for both widths) -- not the
// vector ALU: the pair kernels with a move and a second, 64-bit, FMA added to every arc ran 19 % longer (2.8 -> 3.4 ms,
// -DMM_PAIR_DUMMY).  So the 8 bytes of a state hold TWO utterances again, each as the HIGH dword of its double (sign, 11
// exponent bits, 20 mantissa bits):
//   * an arc is ONE ds_read_b64 (both utterances' high dwords, into an aligned register pair) + a v_mov_b32 + TWO v_fma_f64:
//     the pair as it landed is utterance 1's operand -- its low dword is utterance 0's high dword, up to 2^-20 relative on a
//     term, and the frames are renormalised by their own sums --, utterance 0's operand is a second pair whose high register the
//     move fills; the weight operand is the pair {LDS address of the arc, high dword of the weight's double} (mm_kernel_dpair.hip,
//     MM_DPAIR_W32): no conversion, the float32 kernels' 2 registers per arc;
//   * accumulators, group sums, log2 / exp2 of a finish are the float64 kernels' (the exponent through v_frexp / integer ops,
//     the mantissa through v_log_f32 / v_exp_f32); what a finish writes back is the high dword of 2^y, rounded to nearest;
//   * the stored vectors stay float32 log2 values [N + 2][S1p][2] (the float32 pair kernels' workspace, one 8-byte store per
//     finished row), the combine A .* B writes high dwords into the q vector, the per-pdf sums and the frame's normaliser are
//     doubles;
//   * range marks go to redo2 and are decided by mm_dpair_finish_kernel like the float64 kernels' (1022 log2 below a frame's
//     maximum instead of 126).
// Everything else -- agents, phases, service wave, LDS-DMA rings, normaliser prediction, slot tables, the pair form of the
// graph -- is pair_agent's; the layout differs in the per-pdf sums (16 bytes per pdf: two doubles).
// Precision: operands carry 20 mantissa bits (relative 5e-7 rounded, up to 1e-6 on utterance 1's terms), accumulated in
// float64; the bar is 1e-4 relative on log gamma.  GPU tests: tests/test_gpu_exact.py (against the float64 oracle).
#pragma once
#include "mm_kernel_dpair.hip"

namespace mm {

// LDS layout: PairLay with per-pdf sums of two doubles (RSH: bytes of the rows ONE workgroup finishes -- all of them, or a team's set)
template <int RS, int PHASE, int PC = 256, int RSH_ = 2 * RS>
struct WPairLay {
    using B = PairLay<RS, PHASE, RSH_, PC>;
    static constexpr unsigned RS2 = B::RS2, RSH = RSH_, PC4 = B::PC4, PC8 = B::PC8, RAWS = B::RAWS;
    static constexpr int NR = B::NR, POFFN = B::POFFN;
    // (teams) floats of one slot of published per-pdf partial sums: two doubles per pdf, the XCD handshake in the slot's last 8 bytes
    // (128 pdfs fill an array of 128: 16 bytes more)
    static constexpr unsigned XPS = 4u * PC + (PC == 128 ? 4u : 0u);
    static constexpr unsigned XFLAG = B::XFLAG;
    static constexpr unsigned PP(int par) { return B::PP(par); }
    static constexpr unsigned RAW(int k, int u) { return B::RAW(k, u); }
    static constexpr unsigned EM(int par) { return B::EM(par); }
    static constexpr unsigned MS(int par) { return B::MS(par); }
    static constexpr unsigned OWN(int k) { return B::OWN(k); }
    static constexpr unsigned POFF(int k, int u) { return B::POFF(k, u); }
    static constexpr unsigned PSUM(int par) { return B::PSUM(0) + unsigned(par) * 2u * PC8; }  // [pdf][2] doubles
    static constexpr unsigned PDFSE = B::PDFSE + 2u * PC8;
    static constexpr unsigned FIX = PDFSE + PC4;
    static constexpr unsigned AL(int k) { return FIX + unsigned(k) * RSH; }
    static constexpr unsigned Q(int par) { return FIX + unsigned(NR) * RSH + unsigned(par) * RSH; }
    static constexpr unsigned SLOTS = PHASE ? FIX + unsigned(NR + 2) * RSH : FIX;
};
inline size_t wpair_lds_bytes(int RS, int phase, int nslotrows, int PC = 256, int RSH = 0) {
    return pair_lds_bytes(RS, phase, nslotrows, RSH, PC) + size_t(16) * PC;
}
// pdf capacity of the per-pdf arrays: the teams' instances of up to 128 pdfs take arrays of half the size (their LDS is the tightest)
constexpr int wpair_pc(int NJ, int H) { return (H > 1 && NJ <= 2) ? 128 : pair_pc(NJ); }

__device__ __forceinline__ void ldsw2u(unsigned addr, unsigned a, unsigned b) {
    mm_u32x2 v = {a, b};
    *(__attribute__((address_space(3))) mm_u32x2 *)(__UINTPTR_TYPE__)addr = v;
}
typedef double mm_f64x2 __attribute__((ext_vector_type(2)));
// the high dword of 2^y as a double: the fraction through v_exp_f32, its 23 mantissa bits CUT to the 20 of the high dword by
// v_cvt_f64_f32 (the bias of the cut is the same for every state, frame and direction: the frames are renormalised), the
// integer part added to the exponent field; 0 below 2^-1022 and for -inf / NaN
__device__ __forceinline__ unsigned w_exp2_hi(float y) {
    const float fl = __builtin_floorf(y);
    const double m = (double)fast_exp2(y - fl);  // [1, 2]: exponent field 1023 (2.0: 1024)
    const unsigned hi = __builtin_bit_cast(mm_u32x2, m).y + ((unsigned)(int)fl << 20);
    return y > -1022.f ? hi : 0u;
}
__device__ __forceinline__ double w_from_hi(unsigned hi) { return __builtin_bit_cast(double, (unsigned long long)hi << 32); }
// log2 of a non-negative wide value given by its high dword (20 mantissa bits): exponent + v_log_f32 of the mantissa
__device__ __forceinline__ float w_log2_hi(unsigned hi) {
    const int ex = (int)((hi >> 20) & 0x7ffu);
    if (ex == 0) return MM_NINF;
    const float mant = __builtin_bit_cast(float, 0x3f800000u | ((hi & 0xfffffu) << 3));  // [1, 2)
    return (float)(ex - 1023) + fast_log2(mant);
}

// log2 of a non-negative double accumulator: the exponent field + v_log_f32 of the top 23 mantissa bits -- integer operations on
// the two words instead of v_frexp_exp_i32_f64 / v_frexp_mant_f64 / v_cvt_f32_f64 (quarter-rate instructions, three per finish
// and utterance).  0 and denormals -> -inf.  (The exponent is unbiased as an integer: 1023 folded into the float normaliser the
// caller subtracts would cost that normaliser its low bits -- the per-frame log Z then scatter by 1e-4.)
__device__ __forceinline__ float w_log2_acc(const double &s) {
    const mm_u32x2 w = __builtin_bit_cast(mm_u32x2, s);
    const unsigned ex = (w.y >> 20) & 0x7ffu;
    const unsigned m23 = __builtin_amdgcn_alignbit(w.y, w.x, 29) & 0x7fffffu;  // (hi << 3 | lo >> 29): mantissa bits 51 .. 29
    const float l = fast_log2(__builtin_bit_cast(float, m23 | 0x3f800000u)) + (float)((int)ex - 1023);
    return ex != 0u ? l : MM_NINF;
}

#ifndef MM_WPAIR_LINFIN
#define MM_WPAIR_LINFIN 1
#endif
#ifndef MM_WPAIR_EXITS_B
#define MM_WPAIR_EXITS_B 0  // 1: phase B leaves the arc window after the last segment like pair_agent -- this compiler (ROCm 7.2) then fails ("illegal VGPR to SGPR copy")
#endif
#define MM_WLINF_EMIN (-500.f)  // log2 of the smallest emission factor of a step that raises no mark (wpair_stage_em)
// pair_stage_em<NJ, LIN> for the wide kernels: the step's emission factors 2^(v - E - S) as wide values (high dwords)
template <int NJ>
__device__ __forceinline__ float wpair_stage_em(unsigned dst, unsigned rawsrc, int u, int n, int len, int P, int lane, float S, int *mark) {
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
        if (q <= P) {
            ldswu(dst + 8u * q + 4u * u, w_exp2_hi(v[j] - E - S));
            tiny = tiny || (v[j] - E - S < MM_WLINF_EMIN && v[j] > MM_NINF);
        }
    }
    if (tiny) *mark = 1;
    return E;
}

// ... for BOTH utterances in one pass (pair_stage_em2): all raw values are read before anything is written, the two maxima share one
// DPP ladder, a pdf's two factors leave in ONE 8-byte write -- the service wave is what a step of phase B waits for in these kernels
// (profiles/r05_stamps_wide.txt)
template <int NJ>
__device__ __forceinline__ void wpair_stage_em2(unsigned dst, unsigned raw0, unsigned raw1, int n, int len0, int len1, int P, int lane, float S0,
                                                float S1, int *mark0, int *mark1, float (&E)[2]) {
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
    bool tiny0 = false, tiny1 = false;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        if (q <= P) {
            const float x0 = v0[j] - e0 - S0, x1 = v1[j] - e1 - S1;
            ldsw2u(dst + 8u * (unsigned)q, w_exp2_hi(x0), w_exp2_hi(x1));
            tiny0 = tiny0 || (x0 < MM_WLINF_EMIN && v0[j] > MM_NINF);
            tiny1 = tiny1 || (x1 < MM_WLINF_EMIN && v1[j] > MM_NINF);
        }
    }
    if (tiny0) *mark0 = 1;
    if (tiny1) *mark1 = 1;
    E[0] = e0;
    E[1] = e1;
}

// One arc for the two utterances: acc0 += w * x_0, acc1 += w * x_1 with xx = {high dword of x_0, high dword of x_1} as gathered.
// `tmp`: two register pairs that live across the steps; only their high registers are written here, their low registers stay 0.
// (The pair as it landed would do as utterance 1's operand without a move -- its low dword is then utterance 0's high dword: up to
// 2^-20 relative on a term, and NOT the same for all states: it follows utterance 0's value at the state.  Measured: the per-frame
// log Z of utterance 1 then scatter by 3e-3 log2 over 1500 frames, 60 times the other kernels' and beyond what
// mm_dpair_finish_kernel accepts as "no mass lost".)
__device__ __forceinline__ void w_fma2(double &acc0, double &acc1, const double &wa, const mm_u32x2 &xx, double (&tmp)[2]) {
    mm_u32x2 t0 = __builtin_bit_cast(mm_u32x2, tmp[0]), t1 = __builtin_bit_cast(mm_u32x2, tmp[1]);
    t0.y = xx.x;
    t1.y = xx.y;
    tmp[0] = __builtin_bit_cast(double, t0);
    tmp[1] = __builtin_bit_cast(double, t1);
    asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc0) : "v"(wa), "v"(tmp[0]));
    asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc1) : "v"(wa), "v"(tmp[1]));
}
__device__ __forceinline__ void w_mul2(double &acc0, double &acc1, const double &wa, const mm_u32x2 &xx, double (&tmp)[2]) {
    mm_u32x2 t0 = __builtin_bit_cast(mm_u32x2, tmp[0]), t1 = __builtin_bit_cast(mm_u32x2, tmp[1]);
    t0.y = xx.x;
    t1.y = xx.y;
    tmp[0] = __builtin_bit_cast(double, t0);
    tmp[1] = __builtin_bit_cast(double, t1);
    asm("v_mul_f64 %0, %1, %2" : "=v"(acc0) : "v"(wa), "v"(tmp[0]));
    asm("v_mul_f64 %0, %1, %2" : "=v"(acc1) : "v"(wa), "v"(tmp[1]));
}
// the two utterances' products with two different left factors: a0 = s0 * x_0, a1 = s1 * x_1 (the finishes)
__device__ __forceinline__ void w_mul2s(double &a0, double &a1, const double &s0, const double &s1, const mm_u32x2 &xx, double (&tmp)[2]) {
    mm_u32x2 t0 = __builtin_bit_cast(mm_u32x2, tmp[0]), t1 = __builtin_bit_cast(mm_u32x2, tmp[1]);
    t0.y = xx.x;
    t1.y = xx.y;
    tmp[0] = __builtin_bit_cast(double, t0);
    tmp[1] = __builtin_bit_cast(double, t1);
    asm("v_mul_f64 %0, %1, %2" : "=v"(a0) : "v"(s0), "v"(tmp[0]));
    asm("v_mul_f64 %0, %1, %2" : "=v"(a1) : "v"(s1), "v"(tmp[1]));
}
// (Tried and dropped, round 5: an arc as TWO ds_read_b32, each straight into the high register of a persistent operand pair whose low
// register stays 0 -- no moves, 2 LDS + 2 vector instructions per arc instead of 1 + 4.  Slower: 5.8 against 5.2 ms on the
// same box -- twice the gather instructions on an LDS that the float32 kernels already keep 75 % busy.)
template <int K2, int KA, int D>
__device__ __forceinline__ void wpair_one(const double (&wa)[KA], mm_u32x2 (&x)[2 * D], double (&tmp)[2], double &a0, double &a1, unsigned rdoff) {
    constexpr int s0 = (2 * K2) % (2 * D);
    w_fma2(a0, a1, wa[2 * K2], x[s0], tmp);
    w_fma2(a0, a1, wa[2 * K2 + 1], x[s0 + 1], tmp);
    if constexpr (2 * (K2 + D) < KA) {
        x[s0] = ldsr2u(d_waddr(wa[2 * (K2 + D)]) + rdoff);
        x[s0 + 1] = ldsr2u(d_waddr(wa[2 * (K2 + D) + 1]) + rdoff);
    }
}
// two pairs of arcs in straight-line code: pair K2 to the running sums, pair K2 + 1 to sums of its own (pair_two)
template <int K2, int KA, int D>
__device__ __forceinline__ void wpair_two(const double (&wa)[KA], mm_u32x2 (&x)[2 * D], double (&tmp)[2], double &a0, double &a1, double &n0, double &n1,
                                          unsigned rdoff) {
    constexpr int s0 = (2 * K2) % (2 * D), s1 = (2 * K2 + 2) % (2 * D);
    w_fma2(a0, a1, wa[2 * K2], x[s0], tmp);
    w_mul2(n0, n1, wa[2 * K2 + 2], x[s1], tmp);
    w_fma2(a0, a1, wa[2 * K2 + 1], x[s0 + 1], tmp);
    w_fma2(n0, n1, wa[2 * K2 + 3], x[s1 + 1], tmp);
    if constexpr (2 * (K2 + D) < KA) {
        x[s0] = ldsr2u(d_waddr(wa[2 * (K2 + D)]) + rdoff);
        x[s0 + 1] = ldsr2u(d_waddr(wa[2 * (K2 + D) + 1]) + rdoff);
    }
    if constexpr (2 * (K2 + 1 + D) < KA) {
        x[s1] = ldsr2u(d_waddr(wa[2 * (K2 + 1 + D)]) + rdoff);
        x[s1 + 1] = ldsr2u(d_waddr(wa[2 * (K2 + 1 + D) + 1]) + rdoff);
    }
}
#define MM_WPAIR_ONE(k)                                                                \
    if constexpr (2 * (k) < KA) {                                                      \
        wpair_one<(2 * (k) < KA ? (k) : 0), KA, D>(wa, x, tmp, accA0, accA1, rdoff);   \
        if (MM_PAIR_END(k)) finish();                                                  \
    }
#define MM_WPAIR_TWO(k)                                                                               \
    if constexpr (2 * (k) + 2 < KA && PHASE == 0) {                                                                 \
        wpair_two<(2 * (k) + 2 < KA ? (k) : 0), KA, D>(wa, x, tmp, accA0, accA1, accN0, accN1, rdoff); \
        if (__builtin_expect(((((k) < 32 ? em_lo : em_hi) >> ((k) & 31)) & 3u) != 0u, 0)) {           \
            if (MM_PAIR_END(k)) finish();                                                             \
            accA0 += accN0;                                                                           \
            accA1 += accN1;                                                                           \
            accN0 = accN1 = 0.0;                                                                      \
            if (MM_PAIR_END((k) + 1)) finish();                                                       \
        }                                                                                             \
        accA0 += accN0;                                                                               \
        accA1 += accN1;                                                                               \
    } else {                                                                                          \
        MM_WPAIR_ONE(k)                                                                               \
        MM_WPAIR_ONE((k) + 1)                                                                         \
    }

// service wave: log2 of the maxima of both linear vectors (pairs of high dwords [pos][2]; n2 16-byte reads = 2 states each).
// Non-negative doubles order like their high dwords.
template <int NB, int BATCH = 4>
__device__ __forceinline__ void wpair_scan_max(unsigned pbase, int n2, int lane, float &m0, float &m1) {
    typedef unsigned mm_u32x4 __attribute__((ext_vector_type(4)));
    static_assert(NB % BATCH == 0, "batches of the scan");
    unsigned a = 0u, b = 0u;
#pragma unroll
    for (int j0 = 0; j0 < NB; j0 += BATCH) {
        mm_u32x4 v[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int q = lane + 64 * (j0 + j);
            v[j] = *(__attribute__((address_space(3))) const mm_u32x4 *)(__UINTPTR_TYPE__)(pbase + 16u * (q < n2 ? q : n2 - 1));
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const unsigned ma = v[j].x > v[j].z ? v[j].x : v[j].z, mb = v[j].y > v[j].w ? v[j].y : v[j].w;
            a = a > ma ? a : ma;
            b = b > mb ? b : mb;
        }
    }
    m0 = w_log2_hi(wave_max_u32(a));
    m1 = w_log2_hi(wave_max_u32(b));
}

// a granule of two tagged doubles (teams: the per-pdf partial sums of a set; the sign bits carry the step's tag)
__device__ __forceinline__ void wgranule_store2(float *base, unsigned byte_off, double a, double b, bool neg) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) u64x2 gu64x2;
    const unsigned long long sg = neg ? 0x8000000000000000ull : 0ull;
    u64x2 v = {__builtin_bit_cast(unsigned long long, a) | sg, __builtin_bit_cast(unsigned long long, b) | sg};
    // (one 16-byte write-through store: both doubles of a pdf arrive together or not at all -- the tag of the first tells)
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"((gu64x2 *)(__UINTPTR_TYPE__)(reinterpret_cast<char *>(base) + byte_off)), "v"(v) : "memory");
}
// one wave, both utterances: per-frame sums over the pdfs (psum: two doubles per pdf), divide, store gamma
// (src/inference.jl:156-160); lt[u] = log2 of the sum (-inf, and gamma = 0, if nothing is alive)
// (teams: xp[g] = the slot of the step in which set g published its partial sums, NULL for the own set; the sum of a pdf is the
// sum of the sets' parts in the order of the sets -- the same bits in every workgroup; returns false if a poll timed out)
template <int NJ, int H = 1>
__device__ __forceinline__ bool wpair_finish_frames(unsigned psum, int P1, int P, int lane, float *gp0, float *gp1, long long gsp, bool store0,
                                                    bool store1, float (&lt)[2], const float *const *xp = nullptr, unsigned tag = 0u,
                                                    unsigned long long tmo = MM_SPLIT_TIMEOUT) {
    mm_f64x2 s[NJ];
    double t0 = 0.0, t1 = 0.0;
    bool arrived = true;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        s[j] = *(__attribute__((address_space(3))) const mm_f64x2 *)(__UINTPTR_TYPE__)(psum + 16u * (unsigned)(q < P1 ? q : 0));
    }
    if constexpr (H > 1) {
        typedef unsigned mm_u32x4 __attribute__((ext_vector_type(4)));
        mm_f64x2 tot[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) tot[j] = mm_f64x2{0.0, 0.0};
#pragma unroll
        for (int g = 0; g < H; ++g) {  // (one set at a time: 2 NJ registers pairs per set in flight)
            if (xp[g] == nullptr) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) tot[j] += s[j];
                continue;
            }
            mm_u32x4 v[NJ];
            const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int q = lane + 64 * j;
                    asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v[j]) : "v"(16u * (unsigned)(q < P1 ? q : 0)), "s"(xp[g]) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    asm volatile("" : "+v"(v[j]));
                    ok = ok && ((v[j].y >> 31) == tag) && ((v[j].w >> 31) == tag);
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (!arrived || __builtin_amdgcn_s_memrealtime() - tstart >= tmo) {
                    arrived = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const unsigned long long a = ((unsigned long long)(v[j].y & 0x7fffffffu) << 32) | v[j].x, b = ((unsigned long long)(v[j].w & 0x7fffffffu) << 32) | v[j].z;
                tot[j].x += __builtin_bit_cast(double, a);
                tot[j].y += __builtin_bit_cast(double, b);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) s[j] = tot[j];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (lane + 64 * j < P1) {
            t0 += s[j].x;
            t1 += s[j].y;
        }
    t0 = dwave_sum_rl(t0);
    t1 = dwave_sum_rl(t1);
    // gamma = s / t through floats on the scale of t (dpair_finish_frame)
    const int e0 = __builtin_amdgcn_frexp_exp(t0), e1 = __builtin_amdgcn_frexp_exp(t1);
    const float f0 = (float)__builtin_amdgcn_ldexp(t0, -e0), f1 = (float)__builtin_amdgcn_ldexp(t1, -e1);
    const float i0 = f0 > 0.f ? 1.f / f0 : 0.f, i1 = f1 > 0.f ? 1.f / f1 : 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        if (q < P) {
            if (store0) gp0[q * gsp] = (float)__builtin_amdgcn_ldexp(s[j].x, -e0) * i0;
            if (store1) gp1[q * gsp] = (float)__builtin_amdgcn_ldexp(s[j].y, -e1) * i1;
        }
    }
    lt[0] = dlog2(t0);
    lt[1] = dlog2(t1);
    return arrived;
}

// pdf sums of both utterances (q: pairs of high dwords in pdf-major order), 1 << LG lanes per pdf (pair_pdf_sums)
template <int LG = 3>
__device__ __forceinline__ void wpair_pdf_sums(unsigned qbase, unsigned pdfse_base, unsigned psum_base, int P1, int wave, int NWC, int lane,
                                               float *xs = nullptr, bool neg = false) {
    constexpr int LP = 1 << LG, PPW = 64 >> LG;
    constexpr unsigned STR = 8u * LP;
    for (int p0 = wave * PPW; p0 < P1; p0 += NWC * PPW) {
        const int pdf = p0 + (lane >> LG);
        double s0 = 0.0, s1 = 0.0;
        if (pdf < P1) {
            const unsigned se = ldsru(pdfse_base + 4u * pdf);  // first | end << 16
            const unsigned a0 = 8u * ((se & 0xffffu) + (lane & (LP - 1))), a1 = 8u * (se >> 16);
            mm_u32x2 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ldsr2u(qbase + (a0 + STR * k < a1 ? a0 + STR * k : 0u));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (a0 + STR * k < a1) {
                    s0 += w_from_hi(v[k].x);
                    s1 += w_from_hi(v[k].y);
                }
            for (unsigned a = a0 + 4u * STR; a < a1; a += STR) {
                const mm_u32x2 w = ldsr2u(qbase + a);
                s0 += w_from_hi(w.x);
                s1 += w_from_hi(w.y);
            }
        }
        static_assert(LG >= 1 && LG <= 3, "lanes per pdf: 2, 4 or 8");
        s0 = dpp_add_d<MM_DPP_XOR1, 0xF>(s0);
        s1 = dpp_add_d<MM_DPP_XOR1, 0xF>(s1);
        if constexpr (LG >= 2) {
            s0 = dpp_add_d<MM_DPP_XOR2, 0xF>(s0);
            s1 = dpp_add_d<MM_DPP_XOR2, 0xF>(s1);
        }
        if constexpr (LG >= 3) {
            s0 = dpp_add_d<MM_DPP_HALF_MIRROR, 0xF>(s0);
            s1 = dpp_add_d<MM_DPP_HALF_MIRROR, 0xF>(s1);
        }
        if (pdf < P1 && (lane & (LP - 1)) == 0) {
            mm_f64x2 w = {s0, s1};
            *(__attribute__((address_space(3))) mm_f64x2 *)(__UINTPTR_TYPE__)(psum_base + 16u * (unsigned)pdf) = w;
            if (xs) wgranule_store2(xs, 16u * (unsigned)pdf, s0, s1, neg);
        }
    }
}

// One agent: direction rdir (0: forward / alpha, 1: backward / beta) of pair `pair`, phase PHASE (0: A, 1: B).  pair_agent
// with the arithmetic above.  H > 1: the agent is a TEAM of H workgroups, this one finishes the rows of set `hset` (the split
// kernels, mm_kernel_pairs.hip 4.1b: graphs beyond the registers / LDS of one compute unit); the exchange is pair_agent's bit for
// bit -- a granule is the pair of high dwords of a row, the step's tag in their sign bits (the values are >= 0) -- in the float64
// team kernels' exchange areas (p.xbuf_d / p.xps_d: B slots, (B + 1) / 2 pairs use them), the per-pdf partial sums two tagged
// doubles per pdf.
template <int KA, int RS, int PHASE, int NJ, int H = 1, int RSH = 2 * RS>
__device__ __forceinline__ void wpair_agent(const RunParams &p, int pair, int rdir, int hset = 0) {
    extern __shared__ float lds[];
    const int DIR = __builtin_amdgcn_readfirstlane(rdir);
    const unsigned long long x_tmo = p.x_timeout;  // (teams) ticks of s_memrealtime a poll waits before it gives the team up
    using L = WPairLay<RS, PHASE, wpair_pc(NJ, H), RSH>;
    constexpr int D = PHASE ? 2 : 3;  // gather pairs in flight ahead of the FMAs (phase B: the registers of two)
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
        const int ii = valid ? i : p.B - 1;  // (odd batch: the last pair runs its first utterance twice)
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
    float *rowsP = p.ws_alpha + (long long)pair * (long long)(p.N + 2) * 2 * p.pair_s1p;  // [N + 2][S1p][2] float32 log2 values
    const UttDesc &ud = p.utts[U[0].b];
    const RowU r = uni(H > 1 ? ud.rps[DIR][hset] : ud.rp[DIR]);
    const int S1 = r.rows, S1p = p.pair_s1p, P1 = uni(ud.P1), P = P1 - 1, P1p = (P1 + 3) & ~3;
    const float thr = r.thr + MM_DPAIR_THR_EXTRA;
    // LINF (pair_agent): the finishes stay in the linear domain -- p = s * factor, q = s * partner as v_mul_f64 on wide values,
    // no logarithms or exponentials in the compute waves (a finish was ~55 vector instructions, 4 to 6 of them quarter-rate)
    constexpr bool LINF = MM_WPAIR_LINFIN != 0;
    // bits - 1 of the high dword of the smallest sum a finish accepts: 2^-(thr + MM_WLINF_EMIN)
    unsigned sthr;  // (a scalar register by force: as a vector register it was spilled in the phase-B instance of the wide kernels)
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sthr) : "v"(((unsigned)(1023 - (int)(thr + MM_WLINF_EMIN < 1.f ? 1.f : thr + MM_WLINF_EMIN)) << 20) - 1u));
    int m = (int)(((long long)NFp * (p.split_q10 > 0 ? p.split_q10 : 512)) >> 10);
    m = m < 1 ? 1 : (m > NFp - 1 && NFp > 1 ? NFp - 1 : m);
    const int tA = DIR ? NFp - m : m, tEnd = NFp;
    auto frame_of = [&](int t) { return DIR ? NFp + 1 - t : t; };
    PairHand *hand = reinterpret_cast<PairHand *>(p.pair_hand) + ((long long)pair * 2 + DIR) * 2;
    if (lds_addr_of(lds) != 0u) __builtin_trap();
    MM_STAMP_DECL;

    // ---- LDS set-up
    for (unsigned q = tid * 4u; q < 2u * L::RS2; q += NT * 4u) ldsw(L::PP(0) + q, 0.f);
    if constexpr (PHASE == 1)
        for (unsigned q = tid * 4u; q < 2u * RSH; q += NT * 4u) ldsw(L::Q(0) + q, 0.f);
    if (tid < 32) ldsw(L::MS(0) + 4u * tid, 0.f);
    if (tid < 4) ldsw(L::EM(tid >> 1) + 8u * P1p + 4u * (tid & 1), LINF ? 0.f : MM_NINF);  // the emission slot of lanes without a row
    const int nslotwords = r.nslotrows * 128;
    for (int q = tid; q < nslotwords; q += NT) ldswu(L::SLOTS + 4u * q, as_global(r.slots)[q]);
    if constexpr (PHASE == 1)
        for (int q = tid; q < P1; q += NT) ldswu(L::PDFSE + 4u * q, as_global(reinterpret_cast<const unsigned *>(r.pdfse))[q]);
    // (marks: values beyond the DOUBLE's range -- redo2, decided by mm_dpair_finish_kernel)
    int *redo0 = p.redo2 + U[0].b, *redo1 = p.redo2 + (U[1].valid ? U[1].b : p.B);
    // ---- the team: own set's region, own / others' slots of this launch (the float64 team kernels' exchange areas)
    const int xbase = H > 1 ? p.sp_base[hset] : 0, xcnt = H > 1 ? p.sp_cnt[hset] : 0;
    (void)xcnt;
    float *xsend = nullptr, *xps_send = nullptr;
    bool xplain = false;
    const float *xrecv[H], *xps_recv[H];
    if constexpr (H > 1) {
        float *xb = p.xbuf_d + (long long)PHASE * p.x_phase_d + ((long long)pair * 2 + DIR) * H * 2 * p.x_slot;
        float *xq = p.xps_d + ((long long)pair * 2 + DIR) * H * 4 * (int)L::XPS;
#pragma unroll
        for (int g = 0; g < H; ++g) {
            xrecv[g] = g == hset ? nullptr : xb + (long long)g * 2 * p.x_slot;
            xps_recv[g] = g == hset ? nullptr : xq + (long long)g * 4 * (int)L::XPS;
        }
        xsend = xb + (long long)hset * 2 * p.x_slot;
        xps_send = xq + (long long)hset * 4 * (int)L::XPS;
    }
    unsigned long long endmask = 0, lgw0 = 0;
    int nslots = 0;
    unsigned slot_base = 0;
    if (!service && wave < r.NWC) {
        const RowSched &sc = r.sched[wave];
        endmask = sc.endmask;
        lgw0 = sc.lg;
        nslots = (int)(sc.nslots & 0xffffu);
        slot_base = L::SLOTS + (sc.slot0 * 64u + lane) * 8u;
    }
    unsigned em_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)endmask);
    unsigned em_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(endmask >> 32));
    const int lastp = (PHASE && !MM_WPAIR_EXITS_B) ? 63 : (em_hi ? 63 - __builtin_clz(em_hi) : (em_lo ? 31 - __builtin_clz(em_lo) : -1));
    lgw0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(lgw0 >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((unsigned)lgw0);
    nslots = __builtin_amdgcn_readfirstlane(nslots);
    double wa[KA];  // {LDS address, high dword of the weight} per arc slot
    auto load_graph = [&]() {
        static_assert(KA <= MM_ROW_KA_PAD, "register window larger than the padding of the device arrays");
        const int nt = 64 * r.NWC;
        const bool mine = wave < r.NWC;
        const auto wp = as_global(r.w);
        const auto ap = as_global(r.addr);
        const int t0 = mine ? tid : 0;
#pragma unroll
        for (int k = 0; k < KA; ++k) wa[k] = d_wpair(mine ? wp[k * nt + t0] : 0.f, mine ? ap[k * nt + t0] : 0u);
    };
    // steps of this launch: (t0, t1]; the vector of step t0 is the starting point
    const int t0 = PHASE ? tA : 1, t1 = PHASE ? tEnd : tA;
    __syncthreads();

    if (service) {
        // ================= service wave =================
        int sl = lane;
        RowNorm norm[2];
        double cum[2] = {0.0, 0.0};
        double zmin[2] = {__builtin_inf(), __builtin_inf()}, zmax[2] = {-__builtin_inf(), -__builtin_inf()};
        float ltmin[2] = {__builtin_inff(), __builtin_inff()};
        auto dma_raw = [&](int t) {  // raw emissions of step t (clamped) -> RAW(t & 3, u)
            const int tt = t < 1 ? 1 : (t > tEnd ? tEnd : t);
#pragma unroll
            for (int u = 0; u < 2; ++u) row_dma_em<NJ>(L::RAW(0, u) + L::RAWS * (unsigned)(t & 3), U[u].Vb, p.vsn, frame_of(tt), p.N, P, sl);
        };
        auto dma_partner = [&](int t) {  // the other agent's vector + offset of step t's frame -> AL(t % NR), POFF(t & (POFFN - 1), u)
            const int tt = t < 1 ? 1 : (t > tEnd ? tEnd : t);
            int f = frame_of(tt);
            f = f > p.N ? p.N : f;  // (frame N+1 is never combined)
            constexpr int NDM = RSH / 1024;
            // (teams: the rows of the own set only -- the other direction's workgroup of the same set stored them, contiguously, at the set's base)
            const mm_f32x4 *src = reinterpret_cast<const mm_f32x4 *>(rowsP + (long long)f * 2 * S1p + 2 * xbase);
            const unsigned dst = L::AL(0) + (unsigned)(tt % L::NR) * (unsigned)RSH;
            dma_row_b128<NDM>(uni(src), (unsigned)sl, dst);  // (no clamping: pair_agent)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (sl < 2) dma_b32(reinterpret_cast<const unsigned *>(U[u].offs + f) + sl, L::POFF(0, u) + 16u * (unsigned)(t & (L::POFFN - 1)));
        };
        constexpr int NDMA = 2 * NJ + (PHASE ? RSH / 1024 + 2 : 0);  // DMAs issued per step
        auto stage = [&](int t, const float (&S)[2]) {
            float E[2];
            if constexpr (LINF) {
                wpair_stage_em2<NJ>(L::EM(t & 1), L::RAW(0, 0) + L::RAWS * (unsigned)(t & 3), L::RAW(0, 1) + L::RAWS * (unsigned)(t & 3), frame_of(t), U[0].len,
                                    U[1].len, P, sl, S[0], S[1], redo0, redo1, E);
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) E[u] = pair_stage_em<NJ>(L::EM(t & 1), L::RAW(0, u) + L::RAWS * (unsigned)(t & 3), u, frame_of(t), U[u].len, P, sl);
            }
            double before[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                before[u] = cum[u];
                cum[u] += (double)S[u] + (double)E[u];
            }
            if (sl == 0) {
                if constexpr (!LINF) ldsw2(L::MS(t & 1), S[0], S[1]);
                // (LINF: the stored vector p carries the step's normaliser and emission in both directions; what is combined with
                // the partner's is s, the sum before either -- pair_agent)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const double off = LINF ? cum[u] : (DIR ? cum[u] - (double)E[u] : cum[u]);
                    *(__attribute__((address_space(3))) double *)(__UINTPTR_TYPE__)(L::OWN(t & 3) + 8u * u) = LINF ? before[u] : off;
                    if (PHASE == 0) U[u].offs[frame_of(t)] = off;
                }
            }
        };
        // ---- prologue: everything step t0 + 1 needs
        for (int t = t0; t <= t0 + 3; ++t) dma_raw(t);
        if constexpr (PHASE == 1) {
            dma_partner(t0 + 1);
            if constexpr (L::NR == 3) dma_partner(t0 + 2);
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
        if (PHASE == 0 || (DIR == 1 && !LINF)) {  // emissions of the starting step
            float E[2];
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
            const float S[2] = {norm[0].s_cur, norm[1].s_cur};
            stage(t0 + 1, S);
        }
        dma_raw(t0 + 4);
        __syncthreads();  // (2) starting vector in LDS, step t0 + 1 prepared
        auto frames_of_step = [&](int ts, unsigned psum) {
            const int f = frame_of(ts);
            const bool live0 = f >= 1 && f <= U[0].len, live1 = f >= 1 && f <= U[1].len;
            float lt[2];
            (void)wpair_finish_frames<NJ>(psum, P1, P, sl, p.gamma + (long long)U[0].b * p.gsb + (long long)(f - 1) * p.gsn,
                                    p.gamma + (long long)U[1].b * p.gsb + (long long)(f - 1) * p.gsn, p.gsp, live0 && U[0].valid, live1 && U[1].valid, lt);
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u ? live1 : live0) {
                    const double own = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(L::OWN(ts & 3) + 8u * u);
                    const double oth = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(L::POFF(0, u) + 16u * (unsigned)(ts & (L::POFFN - 1)));
                    const double z = (double)lt[u] + own + oth;
                    zmin[u] = z < zmin[u] ? z : zmin[u];
                    zmax[u] = z > zmax[u] ? z : zmax[u];
                    if (!(z == z)) zmax[u] = __builtin_inf();  // (an overflow somewhere: inf * 0; mm_dpair_finish_kernel keeps the utterance marked)
                    ltmin[u] = lt[u] < ltmin[u] ? lt[u] : ltmin[u];
                }
        };
        auto step = [&](auto RDc, int t) {
            constexpr int RD = decltype(RDc)::value, WR = 1 - RD;  // RD = parity of steps t - 1 and t + 1
            // (the lane index opaque at the top of every step: nothing derived from it -- the addresses of the emission DMAs, of the
            // posteriors' stores, of the scan -- is hoisted out of the step loop into registers this kernel does not have; every
            // reload of a spilled one is a scratch load whose wait also waits for the LDS-DMAs in flight)
            if constexpr (H > 1 || NJ > 2 || PHASE == 1) asm volatile("" : "+v"(sl));
            constexpr bool ring2 = PHASE == 1 && L::NR == 2;
            if constexpr (ring2) dma_partner(t + 1);
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            MM_STAMP(2);
            float mx[2];
            wpair_scan_max<(RS / 8 + 63) / 64, ((RS / 8 + 63) / 64) % 8 == 0 ? 8 : 4>(L::PP(RD), (S1 + 2) >> 1, sl, mx[0], mx[1]);  // (8 reads in flight: pair_agent)
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
                // gamma of step t - 2: its per-pdf sums were completed in the previous step (teams: the exchange wave's business)
                if constexpr (H == 1)
                    if (t - 2 > t0) frames_of_step(t - 2, L::PSUM(WR));
                MM_STAMP(6);
                if constexpr (ring2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ) : "memory");
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
        // ================= exchange wave (teams): the XCD handshake; phase B: the posteriors of the FIRST workgroup =================
        __syncthreads();  // (1)
        bool dead = (p.x_sleep & 0x200) != 0;
        {   // which XCD is the team on?  (pair_agent: a granule in the unused tail of psum slot PHASE of the own set)
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 15u;
            if (lane == 0) granule_store(xps_send + PHASE * (int)L::XPS, 4u * (L::XPS - 2u), __builtin_bit_cast(float, xcc + 1u), 0.f);
            bool same = true;
            const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
#pragma unroll
            for (int g = 0; g < H; ++g) {
                if (g == hset) continue;
                unsigned other = 0u;
                while (!dead) {
                    other = (unsigned)granule_load(xps_recv[g] + PHASE * (int)L::XPS, 4u * (L::XPS - 2u));
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
        }
        __syncthreads();  // (2)
        double xzmin[2] = {__builtin_inf(), __builtin_inf()}, xzmax[2] = {-__builtin_inf(), -__builtin_inf()};
        float xltmin[2] = {__builtin_inff(), __builtin_inff()};
        auto xframes = [&](int ts, unsigned psum) {
            const int f = frame_of(ts);
            const bool live0 = f >= 1 && f <= U[0].len, live1 = f >= 1 && f <= U[1].len;
            float lt[2];
            const float *xp[H];
#pragma unroll
            for (int g = 0; g < H; ++g) xp[g] = g != hset ? xps_recv[g] + (ts & 3) * (int)L::XPS : nullptr;
            if (!wpair_finish_frames<NJ, H>(psum, P1, P, lane, p.gamma + (long long)U[0].b * p.gsb + (long long)(f - 1) * p.gsn,
                                            p.gamma + (long long)U[1].b * p.gsb + (long long)(f - 1) * p.gsn, p.gsp, live0 && U[0].valid,
                                            live1 && U[1].valid, lt, xp, split_tag(ts, t0, 2), dead ? 0ull : x_tmo)) {
                if (lane == 0) {
                    *redo0 = 2;  // (the team is not running together: the log-domain kernels compute these utterances)
                    *redo1 = 2;
                }
                dead = true;
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
        for (int t = t0 + 1; t <= t1; ++t) {
            if constexpr (PHASE == 1)
                if (t - 2 > t0 && hset == 0) xframes(t - 2, L::PSUM(t & 1));  // (the first workgroup's business: pair_agent)
            MM_STEP_SYNC();
        }
        if constexpr (PHASE == 1) {
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
                const mm_f32x2 e = ldsr2(L::EM(1) + 8u * pdfi);
                const float a0 = as_global(r.init)[i];
                float v0 = a0 + e.x, v1 = a0 + e.y;
                if (LINF && H > 1 && pdfi == (unsigned)P1p) v0 = v1 = MM_NINF;  // (LINF: the slot of lanes without a row holds the linear 0)
                if (row_out_of_range(v0, thr)) *redo0 = 1;
                if (row_out_of_range(v1, thr)) *redo1 = 1;
                ldsw2u(L::PP(1) + 8u * i, w_exp2_hi(v0), w_exp2_hi(v1));
                if constexpr (LINF) *reinterpret_cast<mm_u32x2 *>(rowsP + ((long long)1 * S1p + i) * 2) = mm_u32x2{w_exp2_hi(v0), w_exp2_hi(v1)};
                else *reinterpret_cast<mm_f32x2 *>(rowsP + ((long long)1 * S1p + i) * 2) = mm_f32x2{v0, v1};
            }
        } else if (DIR == 1 && t0 == 1) {  // B[:, N+1] = one at the final state   (src/inference.jl:104)
            if (tid == 0) ldsw2u(L::PP(1) + 8u * r.fpos, 0x3ff00000u, 0x3ff00000u);
        } else {  // phase B: the vector this agent stored at the end of phase A
            const int f = frame_of(t0);
            for (int i = tid; i < S1; i += 64 * NWC) {
                const mm_f32x2 vv = *reinterpret_cast<const mm_f32x2 *>(rowsP + ((long long)f * S1p + i) * 2);
                float v0 = vv.x, v1 = vv.y;
                const unsigned pdfi = as_global(r.rowpdf)[i];
                if constexpr (LINF) {  // the stored vector IS the step's vector of wide values
                    const bool pad = H > 1 && pdfi == 0xffffu;  // (padding: never stored)
                    ldsw2u(L::PP(t0 & 1) + 8u * i, pad ? 0u : __builtin_bit_cast(unsigned, v0), pad ? 0u : __builtin_bit_cast(unsigned, v1));
                    continue;
                }
                if (DIR == 1) {  // beta~ is stored without the frame's emission
                    const mm_f32x2 e = ldsr2(L::EM(t0 & 1) + 8u * (H > 1 && pdfi == 0xffffu ? (unsigned)P1p : pdfi));
                    v0 += e.x;
                    v1 += e.y;
                }
                if (H > 1 && pdfi == 0xffffu) v0 = v1 = MM_NINF;  // (padding: never stored)
                ldsw2u(L::PP(t0 & 1) + 8u * i, w_exp2_hi(v0), w_exp2_hi(v1));
            }
        }
        load_graph();
        __syncthreads();  // (2)
        if constexpr (H > 1) xplain = __builtin_amdgcn_readfirstlane(ldsru(L::XFLAG)) != 0u;
        bool cdead = H > 1 && (p.x_sleep & 0x200) != 0;  // (teams) a poll of this wave timed out: it waits no more
        // the operand pairs of the two utterances: only their high registers are written in the loop
        double tmp[2] = {0.0, 0.0};
        asm volatile("" : "+v"(tmp[0]), "+v"(tmp[1]));
        auto step = [&](auto RDc, int t) {
            constexpr int RD = decltype(RDc)::value, WR = 1 - RD;
            if (nslots > 0) {
                constexpr unsigned rdoff = L::PP(RD);
                mm_u32x2 x[2 * D];
#pragma unroll
                for (int j = 0; j < 2 * D; ++j) x[j] = ldsr2u(d_waddr(wa[j < KA ? j : 0]) + rdoff);  // the first gathers leave before anything else
                // the slot table runs one segment ahead (pair_agent)
                unsigned sa = slot_base;
                unsigned info, info2 = 0u, infoN, info2N = 0u;
                if constexpr (PHASE == 1) {
                    const mm_u32x2 w0 = ldsr2u(sa), w1 = ldsr2u(sa + 512u);
                    info = w0.x;
                    info2 = w0.y;
                    infoN = w1.x;
                    info2N = w1.y;
                } else {
                    info = ldsru(sa);
                    infoN = ldsru(sa + 512u);
                }
                // (the step's normalisers, posted by the service wave, are read in every finish: phase B has no two registers to keep them in)
                mm_f32x2 S0 = {0.f, 0.f};
                if constexpr (PHASE == 0 && !LINF) S0 = ldsr2(L::MS(WR));
                mm_f32x2 e = ldsr2((info >> 16) + L::EM(WR));  // (LINF: the bits of two wide values)
                const int f = frame_of(t);
                const unsigned alb = L::AL(0) + (unsigned)(t % L::NR) * (unsigned)RSH;
                mm_f32x2 al = {0.f, 0.f};
                if constexpr (PHASE == 1) al = ldsr2((info2 & 0xffffu) + alb);
                float *rowP = rowsP + (long long)(f <= p.N ? f : 0) * 2 * S1p;
                // (teams) where the team reads this step's rows, and the step's tag as a sign bit
                float *xw = H > 1 ? xsend + (long long)(t & 1) * p.x_slot - 2 * xbase : nullptr;
                const unsigned xtag = (H > 1 && split_tag(t, t0, 1)) ? 0x80000000u : 0u;
                (void)xw;
                (void)xtag;
                float worst = 0.f;
                unsigned smin = 0xffffffffu;
                double accA0 = 0.0, accA1 = 0.0, accN0 = 0.0, accN1 = 0.0;
                unsigned long long lgw = lgw0;
                auto finish = [&]() {
                    const int lg = (int)(lgw & 15ull);
                    lgw >>= 4;
                    double s0 = accA0, s1 = accA1;
                    if (lg) {
                        s0 = dgrp_sum_last(s0, lg);
                        s1 = dgrp_sum_last(s1, lg);
                    }
                    const unsigned pos8 = info & 0xffffu;
                    if constexpr (LINF) {
                        // range check, deferred to the end of the step: the smallest non-zero sum of the lane by its high dword
                        // (pair_agent); p = s * factor, q = s * partner with the operand pairs of the arcs
                        smin = min3_u32(smin, __builtin_bit_cast(mm_u32x2, s0).y - 1u, __builtin_bit_cast(mm_u32x2, s1).y - 1u);
                        double p0, p1;
                        w_mul2s(p0, p1, s0, s1, __builtin_bit_cast(mm_u32x2, e), tmp);
                        const unsigned h0 = __builtin_bit_cast(mm_u32x2, p0).y, h1 = __builtin_bit_cast(mm_u32x2, p1).y;
                        ldsw2u(pos8 + L::PP(WR), h0, h1);
                        if constexpr (H > 1) {  // the row for the team: its pair of high dwords, the step's tag in their sign bits
                            const mm_u32x2 gv = {h0 | xtag, h1 | xtag};
                            if (xplain) *reinterpret_cast<mm_u32x2 *>(reinterpret_cast<char *>(xw) + pos8) = gv;
                            else granule_store(xw, pos8, __builtin_bit_cast(float, gv.x), __builtin_bit_cast(float, gv.y));
                        }
                        if constexpr (PHASE == 0) {
                            *reinterpret_cast<mm_u32x2 *>(reinterpret_cast<char *>(rowP) + pos8) = mm_u32x2{h0, h1};
                        } else {
                            double q0, q1;
                            w_mul2s(q0, q1, s0, s1, __builtin_bit_cast(mm_u32x2, al), tmp);  // A .* B   (:154)
                            ldsw2u((info2 >> 16) + L::Q(WR), __builtin_bit_cast(mm_u32x2, q0).y, __builtin_bit_cast(mm_u32x2, q1).y);
                        }
                    } else {
                    const mm_f32x2 S = PHASE == 0 ? S0 : ldsr2(L::MS(WR));
                    // forward: (T' alpha) (*) lhs (src/inference.jl:70-71); backward: T (B (*) lhs) (:106-107), the emission is
                    // added for the next step's product only
                    const float b0 = w_log2_acc(s0) - S.x, b1 = w_log2_acc(s1) - S.y;
                    const float y0 = b0 + e.x, y1 = b1 + e.y;
                    worst = __builtin_fmaxf(worst, __builtin_fmaxf(__builtin_fmaf(__builtin_fabsf(y0), 0.f, __builtin_fabsf(y0)),
                                                                   __builtin_fmaf(__builtin_fabsf(y1), 0.f, __builtin_fabsf(y1))));
                    const unsigned h0 = w_exp2_hi(y0), h1 = w_exp2_hi(y1);
                    ldsw2u(pos8 + L::PP(WR), h0, h1);
                    if constexpr (H > 1) {  // the row for the team: its pair of high dwords, the step's tag in their sign bits
                        const mm_u32x2 gv = {h0 | xtag, h1 | xtag};
                        if (xplain) *reinterpret_cast<mm_u32x2 *>(reinterpret_cast<char *>(xw) + pos8) = gv;
                        else granule_store(xw, pos8, __builtin_bit_cast(float, gv.x), __builtin_bit_cast(float, gv.y));
                    }
                    const float st0 = DIR ? b0 : y0, st1 = DIR ? b1 : y1;  // the vector that is stored / combined
                    if constexpr (PHASE == 0) {
                        *reinterpret_cast<mm_f32x2 *>(reinterpret_cast<char *>(rowP) + pos8) = mm_f32x2{st0, st1};
                    } else {
                        ldsw2u((info2 >> 16) + L::Q(WR), w_exp2_hi(st0 + al.x), w_exp2_hi(st1 + al.y));  // A .* B   (:154)
                    }
                    }
                    accA0 = accA1 = 0.0;
                    sa += 512u;
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
                MM_PAIR_CASES(MM_WPAIR_TWO)
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(LINF ? smin < sthr : worst > thr) != 0ull, 0)) {
                    *redo0 = 1;
                    *redo1 = 1;
                }
            }
            if constexpr (PHASE == 1)  // C' * (A .* B) of the previous step (:155)
                if (t - 1 > t0)
                    wpair_pdf_sums<(NJ > 2 ? 2 : 3)>(L::Q(RD), L::PDFSE, L::PSUM(RD), P1, wave, NWC, lane, H > 1 ? xps_send + ((t - 1) & 3) * (int)L::XPS : nullptr,
                                                      H > 1 && split_tag(t - 1, t0, 2) != 0u);
            if constexpr (H > 1) {
                // The rows of the other sets of this step (pair_agent, MM_SPLIT_CWPOLL): chunk j (128 granules) of the q-th other set
                // is item q * NG2 + j, compute wave w receives the items w, w + NWC, ...; a granule is a pair of tagged high dwords.
                constexpr int NG2 = (RSH / 16 + 63) / 64, I = (H - 1) * NG2;
                typedef unsigned mm_u32x4 __attribute__((ext_vector_type(4)));
                const unsigned tg = split_tag(t, t0, 1);
                for (int i = wave; i < I; i += NWC) {
                    const int q = i / NG2, j = i % NG2, g = q < hset ? q : q + 1;
                    const float *src = uni(xrecv[g] + (long long)(t & 1) * p.x_slot);
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
                            mm_u32x4 w;
                            w.x = v.x & 0x7fffffffu;
                            w.y = v.y & 0x7fffffffu;
                            w.z = second ? v.z & 0x7fffffffu : 0u;
                            w.w = second ? v.w & 0x7fffffffu : 0u;
                            *(__attribute__((address_space(3))) mm_u32x4 *)(__UINTPTR_TYPE__)dsta = w;
                            pend = false;
                        }
                        if (__builtin_amdgcn_ballot_w64(pend) == 0ull) break;
                        if (__builtin_amdgcn_s_memrealtime() - tstart > x_tmo) {
                            cdead = true;  // the team is not running together: the log-domain kernels compute these utterances
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
                wpair_pdf_sums<(NJ > 2 ? 2 : 3)>(L::Q(t1 & 1), L::PDFSE, L::PSUM(t1 & 1), P1, wave, NWC, lane, H > 1 ? xps_send + (t1 & 3) * (int)L::XPS : nullptr,
                                                  H > 1 && split_tag(t1, t0, 2) != 0u);
            __syncthreads();  // (a)
        }
    }
#ifdef MM_STAMPS
    if (p.dbg && lane == 0)
        for (int k = 0; k < 2; ++k) p.dbg[(((long long)pair * MM_MAX_WAVES + wave) * 4 + PHASE * 2 + DIR) * 2 + k] = stamp_acc[k];
    if (p.dbg && lane == 0 && service)
        for (int k = 0; k < 8; ++k)
            p.dbg[(long long)((p.B + 1) / 2) * MM_MAX_WAVES * 8 + ((long long)pair * 4 + PHASE * 2 + DIR) * 8 + k] = stamp_acc[k];
#endif
}

}  // namespace mm
