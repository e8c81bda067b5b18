// mm_kernel_dpair.hip -- the EXACT path of the pair kernels (mm_kernel_pairs.hip): the same organisation -- register-resident
// graph in the pair form, one barrier per step, a service wave for all HBM traffic, a forward and a backward agent per
// utterance running at the same time, phase A / phase B -- with ONE utterance per workgroup whose linear values are
// FLOAT64.
//
// Why: the pair kernels keep p = 2^a~ as float32.  Every product of an unmarked utterance must stay above 2^-126 of its
// frame's scale, and a posterior of 1e-30 needs a~ + b~ down to 2^(-100 - L_n); under a sharp acoustic model (a trained
// network: log-softmax of 10 N(0,1) has L_n = -160) no float32 scale can hold both factors, the utterance is marked and
// was recomputed by the quad kernels -- one workgroup per utterance, forward THEN backward: 6.5 ms for a single marked
// utterance of config 3, 18 ms for a batch of them.  A double has 1022 log2 below 1 instead of 126: the same recursion in
// float64 is exact (in the sense of the parity bar) for every input a float32 emission matrix can reasonably hold, and
// gfx950 issues v_fma_f64 at the rate of v_fma_f32 (tools/dev/f64_test.hip: 5.1 against 5.3 cycles per wave instruction).
//
// What changes against pair_agent:
//   * the 8 bytes of a state in the LDS vectors (PP, Q, PSUM) hold one double instead of the two floats of a pair: the
//     pair form of the graph -- addresses in units of 8 bytes, slot tables, bank-aware placement -- is used AS IS;
//   * an arc is ds_read_b64 + v_cvt_f64_f32 (the weight stays a float in its register: two VGPRs per arc as before) +
//     v_fma_f64: the LDS traffic of the pair kernels per workgroup, for one utterance instead of two;
//   * a finish takes log2 of a double as exponent + v_log_f32 of the mantissa and 2^y as v_exp_f32 of the fraction +
//     v_ldexp_f64; the stored vectors stay float32 log2 values ([N + 2][S1p] per utterance: 4 bytes per state and frame);
//   * the service wave has one utterance's emissions, normaliser and posteriors per step instead of two (it is the
//     longest actor of the pair kernels' phase B);
//   * range marks (|normalised log2 value| > thr + 896) go to a second array (redo2) and are decided by
//     mm_dpair_finish_kernel with the same two criteria as mm_pair_finish_kernel, scaled to the double range; what stays
//     marked -- values more than ~1000 log2 below their frame's maximum that carry mass -- goes to the quad / item kernels.
// Workgroups of utterances that are not marked (redo[b] == 0) leave at once.
#pragma once
#include "mm_kernel_pairs.hip"

namespace mm {

__device__ __forceinline__ double ldsr_d(unsigned addr) { return *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)addr; }
__device__ __forceinline__ void ldsw_d(unsigned addr, double v) { *(__attribute__((address_space(3))) double *)(__UINTPTR_TYPE__)addr = v; }

// 2^y as a double for a float y of any magnitude (-inf -> 0; below -1100 -> 0): the fraction through v_exp_f32, the integer
// part through v_ldexp_f64
__device__ __forceinline__ double dexp2(float y) {
    float fl = __builtin_floorf(y);
    fl = __builtin_fmaxf(fl, -1100.f);  // (-inf stays -inf in the difference below: 2^-inf = 0)
    const float m = fast_exp2(y - fl);
    return __builtin_amdgcn_ldexp((double)m, (int)fl);
}
// log2 of a non-negative double as a float: exponent + v_log_f32 of the mantissa (0 -> -inf)
__device__ __forceinline__ float dlog2(double s) {
    const int ex = __builtin_amdgcn_frexp_exp(s);
    const float mf = (float)__builtin_amdgcn_frexp_mant(s);
    return (float)ex + fast_log2(mf);
}

#ifndef MM_DPAIR_LINFIN
#define MM_DPAIR_LINFIN 1
#endif
#define MM_DLINF_EMIN (-500.f)  // log2 of the smallest emission factor of a step that raises no mark (dpair_stage_em)
// pair_stage_em<NJ, LIN> for the float64 kernels: the step's emission factors 2^(v - E - S) as doubles (the pair layout's 8 bytes
// per pdf); returns E
template <int NJ>
__device__ __forceinline__ float dpair_stage_em(unsigned dst, unsigned rawsrc, int n, int len, int P, int lane, float S, int *mark) {
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
            ldsw_d(dst + 8u * q, dexp2(v[j] - E - S));
            tiny = tiny || (v[j] - E - S < MM_DLINF_EMIN && v[j] > MM_NINF);
        }
    }
    if (tiny) *mark = 1;
    return E;
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_add_d(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROWMASK, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROWMASK, 0xF, true);
    return v + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// sum over aligned groups of 1 << lg lanes, valid in the LAST lane of every group (grp_sum_last for doubles)
__device__ __forceinline__ double dgrp_sum_last(double v, int lg) {
    v = dpp_add_d<0x111, 0xF>(v);
    if (lg >= 2) {
        v = dpp_add_d<0x112, 0xF>(v);
        if (lg >= 3) {
            v = dpp_add_d<0x114, 0xF>(v);
            if (lg >= 4) {
                v = dpp_add_d<0x118, 0xF>(v);
                if (lg >= 5) {
                    v = dpp_add_d<0x142, 0xA>(v);
                    if (lg >= 6) v = dpp_add_d<0x143, 0xC>(v);
                }
            }
        }
    }
    return v;
}
// butterfly sum over aligned groups of 8 lanes, valid in every lane (grp_sum(v, 3) for doubles)
__device__ __forceinline__ double dgrp_sum8(double v) {
    v = dpp_add_d<MM_DPP_XOR1, 0xF>(v);
    v = dpp_add_d<MM_DPP_XOR2, 0xF>(v);
    v = dpp_add_d<MM_DPP_HALF_MIRROR, 0xF>(v);
    return v;
}
// wave-wide sum (wave_sum_rl for doubles): 16-lane rows by DPP, the 4 row results through readlane
__device__ __forceinline__ double dwave_sum_rl(double v) {
    v = dgrp_sum8(v);
    v = dpp_add_d<MM_DPP_MIRROR, 0xF>(v);
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 16 * k);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 16 * k);
        r[k] = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
    v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, MM_DPP_XOR1, 0xF, 0xF, true));
    v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, MM_DPP_XOR2, 0xF, 0xF, true));
    v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, MM_DPP_HALF_MIRROR, 0xF, 0xF, true));
    v = mx(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, MM_DPP_MIRROR, 0xF, 0xF, true));
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return mx(mx(r0, r1), mx(r2, r3));
}

// service wave: log2 of the maximum of the linear vector of doubles (n2 float4s = 2 states each, n2 <= 64 * NB), to the
// 20 mantissa bits of the high words (a predictor's input: pair_scan_max for doubles).  Non-negative doubles order like
// their high words; -inf if nothing is alive (or everything is below 2^-1022).
template <int NB>
__device__ __forceinline__ float dpair_scan_max(unsigned pbase, int n2, int lane) {
    typedef unsigned mm_u32x4 __attribute__((ext_vector_type(4)));
    unsigned a = 0u;
#pragma unroll
    for (int j0 = 0; j0 < NB; j0 += 4) {
        mm_u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = lane + 64 * (j0 + j);
            v[j] = *(__attribute__((address_space(3))) const mm_u32x4 *)(__UINTPTR_TYPE__)(pbase + 16u * (q < n2 ? q : n2 - 1));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned m = v[j].y > v[j].w ? v[j].y : v[j].w;
            a = a > m ? a : m;
        }
    }
    const unsigned hi = wave_max_u32(a);
    const int ex = (int)((hi >> 20) & 0x7ffu);
    if (ex == 0) return MM_NINF;
    const float mant = __builtin_bit_cast(float, 0x3f800000u | ((hi & 0xfffffu) << 3));  // [1, 2)
    return (float)(ex - 1023) + fast_log2(mant);
}

// one wave: per-frame sum over the pdfs (psum doubles [pdf]), divide, store gamma (src/inference.jl:156-160); returns
// log2 of the sum (-inf, and gamma = 0, if nothing is alive)
// (teams: xp[g] = the slot of the step in which set g published its partial sums, NULL for the own set; the sum of a pdf is the
// sum of the sets' parts in the order of the sets -- the same bits in every workgroup; *arrived = false if a poll timed out)
template <int NJ, int H = 1>  // NJ * 64 >= P + 1
__device__ __forceinline__ float dpair_finish_frame(unsigned psum, int P1, int P, int lane, float *gp, long long gsp, bool store,
                                                    const float *const *xp = nullptr, unsigned tag = 0u, unsigned long long tmo = MM_SPLIT_TIMEOUT,
                                                    bool *arrived = nullptr) {
    double s[NJ], t = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        s[j] = ldsr_d(psum + 8u * (q < P1 ? q : 0));
    }
    if constexpr (H > 1) {
        // (the other sets' parts requested together, four sets at a time, before the first is looked at: pair_finish_frames)
        constexpr int GB = H < MM_XPS_GB ? H : MM_XPS_GB;
        double tot[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) tot[j] = 0.0;
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
                    for (int j = 0; j < NJ; ++j) ok = ok && ((unsigned)(v[g][j] >> 63) == tag);
                }
                if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                if (!*arrived || __builtin_amdgcn_s_memrealtime() - tstart >= tmo) {
                    *arrived = false;
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
                for (int j = 0; j < NJ; ++j) tot[j] += __builtin_bit_cast(double, v[g][j] & 0x7fffffffffffffffull);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) s[j] = tot[j];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
        if (lane + 64 * j < P1) t += s[j];
    t = dwave_sum_rl(t);
    // gamma = s / t through floats on the scale of t: s 2^-e / (t 2^-e), e = the exponent of t
    const int e = __builtin_amdgcn_frexp_exp(t);
    const float tf = (float)__builtin_amdgcn_ldexp(t, -e);
    const float inv = tf > 0.f ? 1.f / tf : 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = lane + 64 * j;
        if (q < P && store) gp[q * gsp] = (float)__builtin_amdgcn_ldexp(s[j], -e) * inv;
    }
    return dlog2(t);
}

// ---- teams (H > 1: the split kernels of mm_kernel_pairs.hip): a granule is ONE double whose sign bit carries the step's tag
// (linear values are >= 0; 2^-inf = +0 becomes -0)
__device__ __forceinline__ void dgranule_store(float *base, unsigned byte_off, double v) {
    __hip_atomic_store((mm_gu64 *)(__UINTPTR_TYPE__)(reinterpret_cast<char *>(base) + byte_off), __builtin_bit_cast(mm_u64, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double dsigned(double v, bool neg) {  // v with the tag in its sign bit (v >= 0)
    return __builtin_bit_cast(double, __builtin_bit_cast(mm_u64, v) | (neg ? 0x8000000000000000ull : 0ull));
}

// pdf sums (q doubles in pdf-major order): 8 pdfs per wave and pass, 8 lanes per pdf (pair_pdf_sums for doubles)
// (teams: xs = the slot the partial sums are published in, tagged by `neg`)
// (LG: log2 of the lanes per pdf, 3 or -- the instances of more than 256 pdfs -- 1: pair_pdf_sums)
template <int LG = 3>
__device__ __forceinline__ void dpair_pdf_sums(unsigned qbase, unsigned pdfse_base, unsigned psum_base, int P1, int wave, int NWC, int lane,
                                               float *xs = nullptr, bool neg = false) {
    constexpr int LP = 1 << LG, PPW = 64 >> LG;
    constexpr unsigned STR = 8u * LP;
    for (int p0 = wave * PPW; p0 < P1; p0 += NWC * PPW) {
        const int pdf = p0 + (lane >> LG);
        double s0 = 0.0;
        if (pdf < P1) {
            const unsigned se = ldsru(pdfse_base + 4u * pdf);  // first | end << 16
            const unsigned a0 = 8u * ((se & 0xffffu) + (lane & (LP - 1))), a1 = 8u * (se >> 16);
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ldsr_d(qbase + (a0 + STR * k < a1 ? a0 + STR * k : 0u));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (a0 + STR * k < a1) s0 += v[k];
            for (unsigned a = a0 + 4u * STR; a < a1; a += STR) s0 += ldsr_d(qbase + a);
        }
        static_assert(LG >= 1 && LG <= 3, "lanes per pdf: 2, 4 or 8");
        s0 = dpp_add_d<MM_DPP_XOR1, 0xF>(s0);
        if constexpr (LG >= 2) s0 = dpp_add_d<MM_DPP_XOR2, 0xF>(s0);
        if constexpr (LG >= 3) s0 = dpp_add_d<MM_DPP_HALF_MIRROR, 0xF>(s0);
        if (pdf < P1 && (lane & (LP - 1)) == 0) {
            ldsw_d(psum_base + 8u * pdf, s0);
            if (xs) dgranule_store(xs, 8u * (unsigned)pdf, dsigned(s0, neg));
        }
    }
}

// ---- the arcs.  MM_DPAIR_W32 (an experiment of round 5, OFF; its pieces are what mm_kernel_wpair.hip is built from): "wide-exponent 32-bit" operands -- the HIGH dword of a double (sign, 11 exponent bits, 20
// mantissa bits) is all an arc reads or keeps:
//   * the linear vector is gathered with ds_read_b32 of the high dwords (the finish still stores whole doubles: the scans, the
//     teams' exchange and the per-pdf sums read them); the low dword of the FMA's operand pair is whatever its register holds
//     -- up to 2^-20 relative on a term, unbiased enough: the frames are renormalised by their own sums, and the bar is 1e-4 on
//     log gamma;
//   * a weight is the high dword of its double (rounded to nearest on the 20 bits: 2.4e-7 relative, the graph's weights
//     perturbed by less than float32 rounds them in the reference), kept in the odd register of an aligned pair whose even
//     register is the arc's LDS address (its bits sit 2^-32 below the last mantissa bit that counts): the pair IS the v_fma_f64
//     operand.  No conversion, no extra register: an arc is ds_read_b32 + v_fma_f64 where it was ds_read_b64 + v_cvt_f64_f32 +
//     v_fma_f64.
// Measured (config 3, sharp emissions, the whole batch on these kernels): 7.7 ms against 5.5 -- SLOWER.  The vector of doubles
// keeps its 8-byte stride, a 4-byte read of the high dwords touches only the odd banks, and two lanes of a 32-lane pass whose
// positions differ by 16 collide (bank = address / 4 mod 32 for ds_read_b32: tools/dev/lds_test.hip "b32 stride 8 B" 2.0 against
// 1.45 cycles); the saved conversion buys nothing because the step is bound by the LDS, not the vector ALU.  What the format IS
// good for: two utterances in the 8 bytes of a state again -- mm_kernel_wpair.hip.
#ifndef MM_DPAIR_W32
#define MM_DPAIR_W32 0
#endif
typedef unsigned mm_du32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double d_wpair(float w, unsigned addr) {  // {address, high dword of (double)w rounded to 20 mantissa bits}
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, (double)w) + 0x80000000ull;
    mm_du32x2 t;
    t.x = addr;
    t.y = (unsigned)(bits >> 32);
    return __builtin_bit_cast(double, t);
}
__device__ __forceinline__ unsigned d_waddr(const double &wa) { return __builtin_bit_cast(mm_du32x2, wa).x; }
// the gathered operand: the high dword from LDS into the odd register of the pair `x` lives in; the low dword keeps what it
// held (the ring of operands is declared OUTSIDE the time loop for that: a fresh low word per gather is a v_mov per arc)
__device__ __forceinline__ void d_gather_hi(double &x, unsigned addr) {
    mm_du32x2 t = __builtin_bit_cast(mm_du32x2, x);
    t.y = ldsru(addr + 4u);
    x = __builtin_bit_cast(double, t);
}

// One arc: acc += (double)w * x.  (MM_DPAIR_W32 = 0) The conversion sits in the same asm block as the FMA: the weights are loop
// invariant, and a conversion the compiler can see is hoisted out of the time loop -- KA more register pairs than a wave has.
__device__ __forceinline__ void d_fma_w(double &acc, float w, const double &x) {
    double t;
    asm("v_cvt_f64_f32 %1, %2\n\tv_fma_f64 %0, %1, %3, %0" : "+v"(acc), "=&v"(t) : "v"(w), "v"(x));
}
__device__ __forceinline__ void d_mul_w(double &acc, float w, const double &x) {
    double t;
    asm("v_cvt_f64_f32 %1, %2\n\tv_mul_f64 %0, %1, %3" : "=v"(acc), "=&v"(t) : "v"(w), "v"(x));
}
__device__ __forceinline__ void d_fma_ww(double &acc, const double &wa, const double &x) {
    asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(x));
}
__device__ __forceinline__ void d_mul_ww(double &acc, const double &wa, const double &x) {
    asm("v_mul_f64 %0, %1, %2" : "=v"(acc) : "v"(wa), "v"(x));
}

// what a compute wave keeps across the steps: MM_DPAIR_W32 the {address, weight} pairs, else PairRegs (mm_kernel_pairs.hip)
template <int KA>
struct DPairRegs {
#if MM_DPAIR_W32
    double wa[KA];
#else
    mm_f32x2 w2[KA / 2];
    unsigned a[KA];
#endif
};

#if MM_DPAIR_W32
template <int K2, int KA, int D>
__device__ __forceinline__ void dpair_one(const DPairRegs<KA> &rg, double (&x)[2 * D], double &accA, unsigned rdoff) {
    constexpr int s0 = (2 * K2) % (2 * D);
    d_fma_ww(accA, rg.wa[2 * K2], x[s0]);
    d_fma_ww(accA, rg.wa[2 * K2 + 1], x[s0 + 1]);
    if constexpr (2 * (K2 + D) < KA) {
        d_gather_hi(x[s0], d_waddr(rg.wa[2 * (K2 + D)]) + rdoff);
        d_gather_hi(x[s0 + 1], d_waddr(rg.wa[2 * (K2 + D) + 1]) + rdoff);
    }
}
template <int K2, int KA, int D>
__device__ __forceinline__ void dpair_two(const DPairRegs<KA> &rg, double (&x)[2 * D], double &accA, double &accN, unsigned rdoff) {
    constexpr int s0 = (2 * K2) % (2 * D), s1 = (2 * K2 + 2) % (2 * D);
    d_fma_ww(accA, rg.wa[2 * K2], x[s0]);
    d_mul_ww(accN, rg.wa[2 * K2 + 2], x[s1]);
    d_fma_ww(accA, rg.wa[2 * K2 + 1], x[s0 + 1]);
    d_fma_ww(accN, rg.wa[2 * K2 + 3], x[s1 + 1]);
    if constexpr (2 * (K2 + D) < KA) {
        d_gather_hi(x[s0], d_waddr(rg.wa[2 * (K2 + D)]) + rdoff);
        d_gather_hi(x[s0 + 1], d_waddr(rg.wa[2 * (K2 + D) + 1]) + rdoff);
    }
    if constexpr (2 * (K2 + 1 + D) < KA) {
        d_gather_hi(x[s1], d_waddr(rg.wa[2 * (K2 + 1 + D)]) + rdoff);
        d_gather_hi(x[s1 + 1], d_waddr(rg.wa[2 * (K2 + 1 + D) + 1]) + rdoff);
    }
}
#define MM_DPAIR_ARGS rg
#else
template <int K2, int KA, int D>
__device__ __forceinline__ void dpair_one(const mm_f32x2 (&wr)[KA / 2], const unsigned (&ar)[KA], double (&x)[2 * D], double &accA, unsigned rdoff) {
    constexpr int s0 = (2 * K2) % (2 * D);
    d_fma_w(accA, wr[K2].x, x[s0]);
    d_fma_w(accA, wr[K2].y, x[s0 + 1]);
    if constexpr (2 * (K2 + D) < KA) {
        x[s0] = ldsr_d(ar[2 * (K2 + D)] + rdoff);
        x[s0 + 1] = ldsr_d(ar[2 * (K2 + D) + 1] + rdoff);
    }
}
// two pairs in straight-line code: pair K2 to the running sum, pair K2 + 1 to a sum of its own (pair_two)
template <int K2, int KA, int D>
__device__ __forceinline__ void dpair_two(const mm_f32x2 (&wr)[KA / 2], const unsigned (&ar)[KA], double (&x)[2 * D], double &accA, double &accN,
                                          unsigned rdoff) {
    constexpr int s0 = (2 * K2) % (2 * D), s1 = (2 * K2 + 2) % (2 * D);
    d_fma_w(accA, wr[K2].x, x[s0]);
    d_mul_w(accN, wr[K2 + 1].x, x[s1]);
    d_fma_w(accA, wr[K2].y, x[s0 + 1]);
    d_fma_w(accN, wr[K2 + 1].y, x[s1 + 1]);
    if constexpr (2 * (K2 + D) < KA) {
        x[s0] = ldsr_d(ar[2 * (K2 + D)] + rdoff);
        x[s0 + 1] = ldsr_d(ar[2 * (K2 + D) + 1] + rdoff);
    }
    if constexpr (2 * (K2 + 1 + D) < KA) {
        x[s1] = ldsr_d(ar[2 * (K2 + 1 + D)] + rdoff);
        x[s1 + 1] = ldsr_d(ar[2 * (K2 + 1 + D) + 1] + rdoff);
    }
}
#define MM_DPAIR_ARGS rg.w2, rg.a
#endif
#define MM_DPAIR_ONE(k)                                                                   \
    if constexpr (2 * (k) < KA) {                                                         \
        dpair_one<(2 * (k) < KA ? (k) : 0), KA, D>(MM_DPAIR_ARGS, x, accA, rdoff);          \
        if (MM_PAIR_END(k)) finish();                                                     \
    }
#define MM_DPAIR_TWO(k)                                                                   \
    if constexpr (2 * (k) + 2 < KA) {                                                     \
        dpair_two<(2 * (k) + 2 < KA ? (k) : 0), KA, D>(MM_DPAIR_ARGS, x, accA, accN, rdoff); \
        if (__builtin_expect(((((k) < 32 ? em_lo : em_hi) >> ((k) & 31)) & 3u) != 0u, 0)) { \
            if (MM_PAIR_END(k)) finish();                                                 \
            accA += accN;                                                                 \
            accN = 0.0;                                                                   \
            if (MM_PAIR_END((k) + 1)) finish();                                           \
        }                                                                                 \
        accA += accN;                                                                     \
    } else {                                                                              \
        MM_DPAIR_ONE(k)                                                                   \
        MM_DPAIR_ONE((k) + 1)                                                             \
    }

#define MM_DPAIR_THR_EXTRA 896.f  // the double's normal range (1022) over the float's (126)

// One agent: direction rdir (0: forward / alpha, 1: backward / beta) of the utterance of rank `ui` (longest first), phase
// PHASE (0: A, 1: B).  The structure, the step numbering and the LDS layout are pair_agent's (mm_kernel_pairs.hip).
// (the direction is a run-time value, the same for the whole workgroup: see pair_agent)
// H > 1: the agent is a TEAM of H workgroups, this one finishes the rows of set `hset` (the split kernels: graphs beyond the
// registers / LDS of one compute unit); the exchange is pair_agent's with a granule = one tagged double.
template <int KA, int RS, int PHASE, int NJ, int H = 1, int RSH = 2 * RS>
__device__ __forceinline__ void dpair_agent(const RunParams &p, int ui, int rdir, int hset = 0) {
    extern __shared__ float lds[];
    const int DIR = __builtin_amdgcn_readfirstlane(rdir);
    const unsigned long long x_tmo = p.x_timeout;  // (teams) ticks of s_memrealtime a poll waits before it gives the team up
    using L = PairLay<RS, PHASE, RSH, pair_pc(NJ)>;
    constexpr int D = 3;  // gather pairs in flight ahead of the FMAs
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6, NWC = NW - (H > 1 ? 2 : 1);
    const bool service = wave == NWC;
    const bool xwave = H > 1 && wave == NWC + 1;  // the exchange wave of a team's workgroup
    // ---- the utterance
    const int b = uni(p.order ? p.order[ui] : ui);
    if (uni(p.redo[b]) == 0) return;  // (not marked by the float32 kernels: their result stands)
    if (service || xwave) __builtin_amdgcn_s_setprio(3);
    int len = uni(p.lens ? p.lens[b] : p.N);
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const float *Vb = p.V + (long long)b * p.vsb;
    double *offs = p.ws_c + (long long)b * (p.N + 2);  // [N + 2] cumulative offset of the stored vector of every frame
    const int NFp = len + 1;
    const UttDesc &ud = p.utts[b];
    const RowU r = uni(H > 1 ? ud.rps[DIR][hset] : ud.rp[DIR]);
    const int S1 = r.rows, S1p = p.pair_s1p, P1 = uni(ud.P1), P = P1 - 1, P1p = (P1 + 3) & ~3;
    // state vectors of the utterance's frames (alpha~ up to the split, beta~ beyond): float32 log2 values [N + 2][S1p]
    float *rowsP = p.ws_alpha + (long long)b * (long long)(p.N + 2) * S1p;
    const float thr = r.thr + MM_DPAIR_THR_EXTRA;
    // LINF (pair_agent): linear-domain finishes -- p = s * factor (the factor a double in LDS), q = s * partner (the partner's stored
    // value: the high dword of its double, 20 mantissa bits -- the 4 bytes per state of the stored rows as before)
    // (not the one-workgroup instances of more than 250 pdfs: their service wave bounds the step, 8 double exponentials more per lane
    // in it cost more than the compute waves save -- 2000 states / 400 pdfs on sharp emissions: 6.96 against 6.41 ms)
    constexpr bool LINF = MM_DPAIR_LINFIN != 0 && (NJ <= 4 || H > 1);
    unsigned sthr;  // bits - 1 of the high dword of the smallest sum a finish accepts: 2^-(thr + MM_DLINF_EMIN)
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sthr) : "v"(((unsigned)(1023 - (int)(thr + MM_DLINF_EMIN < 1.f ? 1.f : thr + MM_DLINF_EMIN)) << 20) - 1u));
    int m = (int)(((long long)NFp * (p.split_q10 > 0 ? p.split_q10 : 512)) >> 10);
    m = m < 1 ? 1 : (m > NFp - 1 && NFp > 1 ? NFp - 1 : m);
    const int tA = DIR ? NFp - m : m, tEnd = NFp;
    auto frame_of = [&](int t) { return DIR ? NFp + 1 - t : t; };
    PairHand *hand = reinterpret_cast<PairHand *>(p.pair_hand) + ((long long)b * 2 + DIR);
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
    int *redo2 = p.redo2 + b;
    // ---- the team: own set's region, own / others' slots of this launch (the float64 kernels' own exchange area: p.xbuf_d)
    const int xbase = H > 1 ? p.sp_base[hset] : 0, xcnt = H > 1 ? p.sp_cnt[hset] : 0;
    float *xsend = nullptr, *xps_send = nullptr;
    bool xplain = false;
    const float *xrecv[H], *xps_recv[H];
    if constexpr (H > 1) {
        // (indexed by the utterance: mm_pair_finish_kernel zeroes the areas of the utterances it leaves marked)
        float *xb = p.xbuf_d + (long long)PHASE * p.x_phase_d + ((long long)b * 2 + DIR) * H * 2 * p.x_slot;
        float *xq = p.xps_d + ((long long)b * 2 + DIR) * H * 4 * (int)L::XPS;
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
    const int lastp = PHASE ? 63 : (em_hi ? 63 - __builtin_clz(em_hi) : (em_lo ? 31 - __builtin_clz(em_lo) : -1));
    lgw0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(lgw0 >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((unsigned)lgw0);
    nslots = __builtin_amdgcn_readfirstlane(nslots);
    DPairRegs<KA> rg;
    auto load_graph = [&]() {
        static_assert(KA <= MM_ROW_KA_PAD, "register window larger than the padding of the device arrays");
        const int nt = 64 * r.NWC;
        const bool mine = wave < r.NWC;
        const auto wp = as_global(r.w);
        const auto ap = as_global(r.addr);
        const int t0 = mine ? tid : 0;
#if MM_DPAIR_W32
#pragma unroll
        for (int k = 0; k < KA; ++k) rg.wa[k] = d_wpair(mine ? wp[k * nt + t0] : 0.f, mine ? ap[k * nt + t0] : 0u);
#else
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
#endif
    };
    // steps of this launch: (t0, t1]; the vector of step t0 is the starting point
    const int t0 = PHASE ? tA : 1, t1 = PHASE ? tEnd : tA;
    // (teams) the rows of the own set of a frame, as the partner stored them: floats at rowsP + f * S1p + xbase; the LDS-DMA
    // moves aligned float4s, so the copy starts at the float4 that holds the first row and a row sits xoff bytes further on
    const int xal = xbase & ~3;
    const unsigned xoff = 4u * (unsigned)(xbase - xal);
    __syncthreads();

    if (service) {
        // ================= service wave =================
        int sl = lane;
        RowNorm norm;
        double cum = 0.0, zmin = __builtin_inf(), zmax = -__builtin_inf();
        float ltmin = __builtin_inff();  // smallest log2 of a frame's sum of 2^(a~ + b~) (see mm_dpair_finish_kernel)
        auto dma_raw = [&](int t) {  // raw emissions of step t (clamped) -> RAW(t & 3, 0)
            const int tt = t < 1 ? 1 : (t > tEnd ? tEnd : t);
            row_dma_em<NJ>(L::RAW(0, 0) + L::RAWS * (unsigned)(t & 3), Vb, p.vsn, frame_of(tt), p.N, P, sl);
        };
        constexpr int NDM = (RSH / 2 + 16 + 1023) / 1024;  // 1 KB DMAs of a row of floats (teams: + the alignment shift)
        auto dma_partner = [&](int t) {  // the other agent's vector + offset of step t's frame -> AL(t % 3), POFF(t & 7, 0)
            const int tt = t < 1 ? 1 : (t > tEnd ? tEnd : t);
            int f = frame_of(tt);
            f = f > p.N ? p.N : f;  // (frame N+1 is never combined)
            const int n4 = H > 1 ? (int)((xoff / 4u + (unsigned)xcnt + 3u) >> 2) : S1p >> 2;  // float4s of the row
            const mm_f32x4 *src = reinterpret_cast<const mm_f32x4 *>(rowsP + (long long)f * S1p + xal);
            const unsigned dst = L::AL(0) + (unsigned)(tt % L::NR) * (unsigned)RSH;
            (void)n4;
            dma_row_b128<NDM>(uni(src), (unsigned)sl, dst);  // (no clamping: see pair_agent)
            if (sl < 2) dma_b32(reinterpret_cast<const unsigned *>(offs + f) + sl, L::POFF(0, 0) + 16u * (unsigned)(t & (L::POFFN - 1)));
        };
        constexpr int NDMA = NJ + (PHASE ? NDM + 1 : 0);  // DMAs issued per step
        // stage the emissions of step t into EM(t & 1) and account its offset; S = the normaliser the step subtracts
        auto stage = [&](int t, float S) {
            const float E = LINF ? dpair_stage_em<NJ>(L::EM(t & 1), L::RAW(0, 0) + L::RAWS * (unsigned)(t & 3), frame_of(t), len, P, sl, S, redo2)
                                 : pair_stage_em<NJ>(L::EM(t & 1), L::RAW(0, 0) + L::RAWS * (unsigned)(t & 3), 0, frame_of(t), len, P, sl);
            const double before = cum;
            cum += (double)S + (double)E;
            if (sl == 0) {
                if constexpr (!LINF) ldsw(L::MS(t & 1), S);
                // the offset that turns the step's stored vector into log2 values: forward alpha~ includes the emission
                // (LINF: the stored vector carries the step's normaliser and emission in both directions; what is combined with the
                // partner's is s, the sum before either -- pair_agent)
                const double off = LINF ? cum : (DIR ? cum - (double)E : cum);
                *(__attribute__((address_space(3))) double *)(__UINTPTR_TYPE__)(L::OWN(t & 3)) = LINF ? before : off;
                if (PHASE == 0 && (H == 1 || hset == 0)) offs[frame_of(t)] = off;
            }
        };
        // ---- prologue: everything step t0 + 1 needs
        for (int t = t0; t <= t0 + 3; ++t) dma_raw(t);
        if constexpr (PHASE == 1) {
            dma_partner(t0 + 1);
            if constexpr (L::NR == 3) dma_partner(t0 + 2);  // (a ring of 2: pair_agent)
            const PairHand h = hand[0];
            norm.m_prev = h.m_prev;
            norm.s_cur = h.s_cur;
            norm.s_prev = h.s_prev;
            norm.cbar = h.cbar;
            norm.seen = h.seen;
            cum = h.cum;
        }
        MM_ROW_VMCNT(0);
        if (PHASE == 0 || (DIR == 1 && !LINF)) {  // emissions of the starting step
            const float E = pair_stage_em<NJ>(L::EM(t0 & 1), L::RAW(0, 0) + L::RAWS * (unsigned)(t0 & 3), 0, frame_of(t0), len, P, sl);
            if (PHASE == 0) {  // step 1 subtracts nothing but E
                cum = (double)E;
                if (DIR == 0 && sl == 0 && (H == 1 || hset == 0)) offs[1] = cum;
            }
        }
        __syncthreads();  // (1) emissions of step t0 staged
        if (t0 + 1 <= t1) stage(t0 + 1, norm.s_cur);  // (phase A: 0 -- step 2 subtracts nothing but E)
        dma_raw(t0 + 4);
        __syncthreads();  // (2) starting vector in LDS, step t0 + 1 prepared
        // posteriors and per-frame log Z of step ts (its per-pdf sums are complete); H > 1: the exchange wave's job
        auto frames_of_step = [&](int ts, unsigned psum) {
            const int f = frame_of(ts);
            const bool live = f >= 1 && f <= len;
            const float lt = dpair_finish_frame<NJ>(psum, P1, P, sl, p.gamma + (long long)b * p.gsb + (long long)(f - 1) * p.gsn, p.gsp, live);
            if (live) {
                const double own = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(L::OWN(ts & 3));
                const double oth = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(L::POFF(0, 0) + 16u * (unsigned)(ts & (L::POFFN - 1)));
                const double z = (double)lt + own + oth;
                zmin = z < zmin ? z : zmin;
                zmax = z > zmax ? z : zmax;
                if (!(z == z)) zmax = __builtin_inf();  // (an overflow somewhere: inf * 0; mm_dpair_finish_kernel keeps the utterance marked)
                ltmin = lt < ltmin ? lt : ltmin;
            }
        };
        auto step = [&](auto RDc, int t) {
            constexpr int RD = decltype(RDc)::value, WR = 1 - RD;  // RD = parity of steps t - 1 and t + 1
            if constexpr (H > 1 || NJ > 2) asm volatile("" : "+v"(sl));
            constexpr bool ring2 = PHASE == 1 && L::NR == 2;  // (the partner row of step t + 1 requested at the top of step t: pair_agent)
            if constexpr (ring2) dma_partner(t + 1);
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            MM_STAMP(2);
            // the normaliser of step t + 1 from the maximum of step t - 1 (complete since the last barrier)
            const float mx = dpair_scan_max<(RS / 8 + 63) / 64>(L::PP(RD), (S1 + 2) >> 1, sl);
            MM_STAMP(3);
            if (t + 1 <= tEnd) {
                const float S = norm.next(mx);
                if (t + 1 <= t1) stage(t + 1, S);
            }
            MM_STAMP(4);
            dma_raw(t + 4);
            if constexpr (PHASE == 1) {
                if constexpr (!ring2) dma_partner(t + 2);
                MM_STAMP(5);
                if constexpr (H == 1)
                    if (t - 2 > t0) frames_of_step(t - 2, L::PSUM(WR));  // gamma of step t - 2: its per-pdf sums were completed in the previous step
                MM_STAMP(6);
                if constexpr (ring2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");  // the partner vector of step t + 1 (requested at step t - 1)
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
            if (sl == 0 && (H == 1 || hset == 0)) {
                PairHand h;
                h.m_prev = norm.m_prev;
                h.s_cur = norm.s_cur;
                h.s_prev = norm.s_prev;
                h.cbar = norm.cbar;
                h.seen = norm.seen;
                h.pad = 0;
                h.cum = cum;
                hand[0] = h;
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
                p.pair_zmin[(long long)b * 6 + DIR] = zmin;
                p.pair_zmin[(long long)b * 6 + 2 + DIR] = zmax;
                p.pair_zmin[(long long)b * 6 + 4 + DIR] = (double)ltmin;
            }
        }
    } else if (xwave) {
        // ================= exchange wave (teams) =================
        __syncthreads();  // (1)
        bool dead = (p.x_sleep & 0x200) != 0;
        {   // which XCD is the team on?  (pair_agent: a granule in the unused tail of psum slot PHASE of the own set)
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
            if (dead && lane == 0) *redo2 = 2;
            if (lane == 0) ldswu(L::XFLAG, (same && !dead && !(p.x_sleep & 0x800)) ? 1u : 0u);
        }
        __syncthreads();  // (2)
        double xzmin = __builtin_inf(), xzmax = -__builtin_inf();
        float xltmin = __builtin_inff();
        auto xframes = [&](int ts, unsigned psum) {
            const int f = frame_of(ts);
            const bool live = f >= 1 && f <= len;
            const float *xp[H];
#pragma unroll
            for (int g = 0; g < H; ++g) xp[g] = g != hset ? xps_recv[g] + (ts & 3) * (int)L::XPS : nullptr;
            bool arrived = true;
            const float lt = dpair_finish_frame<NJ, H>(psum, P1, P, lane, p.gamma + (long long)b * p.gsb + (long long)(f - 1) * p.gsn, p.gsp,
                                                       live && hset == 0, xp, split_tag(ts, t0, 2), dead ? 0ull : x_tmo, &arrived);
            if (!arrived) {
                if (lane == 0) *redo2 = 2;  // (the team is not running together: the log-domain kernels compute the utterance)
                dead = true;
            }
            if (live) {
                const double own = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(L::OWN(ts & 3));
                const double oth = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(L::POFF(0, 0) + 16u * (unsigned)(ts & (L::POFFN - 1)));
                const double z = (double)lt + own + oth;
                xzmin = z < xzmin ? z : xzmin;
                xzmax = z > xzmax ? z : xzmax;
                if (!(z == z)) xzmax = __builtin_inf();
                xltmin = lt < xltmin ? lt : xltmin;
            }
        };
        MM_STAMP_RESET;
        for (int t = t0 + 1; t <= t1; ++t) {
            if constexpr (PHASE == 1)
                if (t - 2 > t0 && hset == 0) xframes(t - 2, L::PSUM(t & 1));  // (the first workgroup's business: pair_agent)
            MM_STAMP(0);
            MM_STEP_SYNC();
            MM_STAMP(1);
        }
        if constexpr (PHASE == 1) {
            for (int k = 1; k >= 0; --k) {
                const int t = t1 - k;
                if (k == 0) __syncthreads();  // (a)
                if (t > t0 && hset == 0) xframes(t, L::PSUM(t & 1));
            }
            if (lane == 0 && hset == 0) {
                p.pair_zmin[(long long)b * 6 + DIR] = xzmin;
                p.pair_zmin[(long long)b * 6 + 2 + DIR] = xzmax;
                p.pair_zmin[(long long)b * 6 + 4 + DIR] = (double)xltmin;
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
                float v0 = as_global(r.init)[i] + ldsr(L::EM(1) + 8u * pdfi);
                if (LINF && H > 1 && pdfi == (unsigned)P1p) v0 = MM_NINF;  // (LINF: the slot of lanes without a row holds the linear 0)
                if (row_out_of_range(v0, thr)) *redo2 = 1;
                const double p0 = dexp2(v0);
                ldsw_d(L::PP(1) + 8u * i, p0);
                if constexpr (LINF) reinterpret_cast<unsigned *>(rowsP)[(long long)1 * S1p + i] = __builtin_bit_cast(mm_u32x2, p0).y;
                else rowsP[(long long)1 * S1p + i] = v0;
            }
        } else if (DIR == 1 && t0 == 1) {  // B[:, N+1] = one at the final state   (src/inference.jl:104)
            if (tid == 0) ldsw_d(L::PP(1) + 8u * r.fpos, 1.0);
        } else {  // phase B: the vector this agent stored at the end of phase A
            const int f = frame_of(t0);
            for (int i = tid; i < S1; i += 64 * NWC) {
                float v0 = rowsP[(long long)f * S1p + i];
                const unsigned pdfi = as_global(r.rowpdf)[i];
                if constexpr (LINF) {  // the stored value IS the step's (the high dword of its double)
                    const bool pad = H > 1 && pdfi == 0xffffu;  // (padding: never stored)
                    ldsw_d(L::PP(t0 & 1) + 8u * i, pad ? 0.0 : __builtin_bit_cast(double, mm_u32x2{0u, __builtin_bit_cast(unsigned, v0)}));
                    continue;
                }
                if (DIR == 1) v0 += ldsr(L::EM(t0 & 1) + 8u * (H > 1 && pdfi == 0xffffu ? (unsigned)P1p : pdfi));  // beta~ is stored without the frame's emission
                if (H > 1 && pdfi == 0xffffu) v0 = MM_NINF;  // (padding: never stored)
                ldsw_d(L::PP(t0 & 1) + 8u * i, dexp2(v0));
            }
        }
        load_graph();
        __syncthreads();  // (2)
        if constexpr (H > 1) xplain = __builtin_amdgcn_readfirstlane(ldsru(L::XFLAG)) != 0u;
        bool cdead = H > 1 && (p.x_sleep & 0x200) != 0;  // (teams) a poll of this wave timed out: it waits no more
#if MM_DPAIR_W32
        double x[2 * D];  // the ring of gathered operands: register pairs whose low words nothing writes after this (d_gather_hi)
#pragma unroll
        for (int j = 0; j < 2 * D; ++j) {
            x[j] = 0.0;
            asm volatile("" : "+v"(x[j]));
        }
#endif
        auto step = [&](auto RDc, int t) {
            constexpr int RD = decltype(RDc)::value, WR = 1 - RD;
            if (nslots > 0) {
                constexpr unsigned rdoff = L::PP(RD);
#if !MM_DPAIR_W32
                double x[2 * D];
#endif
#pragma unroll
                for (int j = 0; j < D; ++j) {  // the first gathers leave before anything else
#if MM_DPAIR_W32
                    d_gather_hi(x[2 * j], d_waddr(rg.wa[(2 * j < KA) ? 2 * j : 0]) + rdoff);
                    d_gather_hi(x[2 * j + 1], d_waddr(rg.wa[(2 * j + 1 < KA) ? 2 * j + 1 : 0]) + rdoff);
#else
                    x[2 * j] = ldsr_d(rg.a[(2 * j < KA) ? 2 * j : 0] + rdoff);
                    x[2 * j + 1] = ldsr_d(rg.a[(2 * j + 1 < KA) ? 2 * j + 1 : 0] + rdoff);
#endif
                }
                // the slot table runs one segment ahead (infoN / info2N): see pair_agent
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
                float S = 0.f;
                if constexpr (!LINF) S = ldsr(L::MS(WR));  // the step's normaliser, posted by the service wave
                float e = 0.f;
                double ed = 0.0;  // (LINF) the emission factor of the segment's rows
                if constexpr (LINF) ed = ldsr_d((info >> 16) + L::EM(WR));
                else e = ldsr((info >> 16) + L::EM(WR));
                const int f = frame_of(t);
                const unsigned alb = L::AL(0) + (unsigned)(t % L::NR) * (unsigned)RSH + xoff;
                float al = 0.f;
                if constexpr (PHASE == 1) al = ldsr(((info2 & 0xffffu) >> 1) + alb);
                float *rowP = rowsP + (long long)(f <= p.N ? f : 0) * S1p;
                // (teams) where the team reads this step's rows, and the step's tag
                float *xw = H > 1 ? xsend + (long long)(t & 1) * p.x_slot - 2 * xbase : nullptr;
                const bool xneg = H > 1 && split_tag(t, t0, 1) != 0u;
                float worst = 0.f;
                unsigned smin = 0xffffffffu;
                double alz = 0.0;  // (LINF) the operand pair of the partner's value: only its high register is written
                asm volatile("" : "+v"(alz));
                double accA = 0.0, accN = 0.0;
                unsigned long long lgw = lgw0;
                auto finish = [&]() {
                    const int lg = (int)(lgw & 15ull);
                    lgw >>= 4;
                    double s0 = accA;
                    if (lg) s0 = dgrp_sum_last(s0, lg);
                    const unsigned pos8 = info & 0xffffu;
                    if constexpr (LINF) {
                        // range check, deferred to the end of the step: the smallest non-zero sum of the lane by its high dword (pair_agent)
                        const unsigned sh = __builtin_bit_cast(mm_u32x2, s0).y - 1u;
                        smin = sh < smin ? sh : smin;
                        const double p0 = s0 * ed;
                        ldsw_d(pos8 + L::PP(WR), p0);
                        if constexpr (H > 1) {
                            if (xplain) *reinterpret_cast<double *>(reinterpret_cast<char *>(xw) + pos8) = dsigned(p0, xneg);
                            else dgranule_store(xw, pos8, dsigned(p0, xneg));
                        }
                        if constexpr (PHASE == 0) {
                            *reinterpret_cast<unsigned *>(reinterpret_cast<char *>(rowP) + (pos8 >> 1)) = __builtin_bit_cast(mm_u32x2, p0).y;
                        } else {
                            mm_u32x2 t = __builtin_bit_cast(mm_u32x2, alz);
                            t.y = __builtin_bit_cast(unsigned, al);
                            alz = __builtin_bit_cast(double, t);
                            double q0;
                            asm("v_mul_f64 %0, %1, %2" : "=v"(q0) : "v"(s0), "v"(alz));
                            ldsw_d((info2 >> 16) + L::Q(WR), q0);  // A .* B   (:154)
                        }
                    } else {
                    // forward: (T' alpha) (*) lhs (src/inference.jl:70-71); backward: T (B (*) lhs) (:106-107), the emission is
                    // added for the next step's product only
                    const float b0 = dlog2(s0) - S;
                    const float y0 = b0 + e;
                    // range check, deferred to the end of the step: the largest finite |y| of the lane (NaN for -inf: ignored)
                    worst = __builtin_fmaxf(worst, __builtin_fmaf(__builtin_fabsf(y0), 0.f, __builtin_fabsf(y0)));
                    const double p0 = dexp2(y0);
                    ldsw_d(pos8 + L::PP(WR), p0);
                    if constexpr (H > 1) {
                        if (xplain) *reinterpret_cast<double *>(reinterpret_cast<char *>(xw) + pos8) = dsigned(p0, xneg);
                        else dgranule_store(xw, pos8, dsigned(p0, xneg));
                    }
                    const float st0 = DIR ? b0 : y0;  // the vector that is stored / combined
                    if constexpr (PHASE == 0) {
                        *reinterpret_cast<float *>(reinterpret_cast<char *>(rowP) + (pos8 >> 1)) = st0;
                    } else {
                        ldsw_d((info2 >> 16) + L::Q(WR), dexp2(st0 + al));  // A .* B   (:154)
                    }
                    }
                    accA = 0.0;
                    sa += 512u;
                    asm volatile("v_mov_b32 %0, %1" : "=v"(info) : "v"(infoN));
                    if constexpr (LINF) ed = ldsr_d((info >> 16) + L::EM(WR));
                    else e = ldsr((info >> 16) + L::EM(WR));
                    if constexpr (PHASE == 1) {
                        asm volatile("v_mov_b32 %0, %1" : "=v"(info2) : "v"(info2N));
                        al = ldsr(((info2 & 0xffffu) >> 1) + alb);
                        const mm_u32x2 w1 = ldsr2u(sa + 512u);
                        infoN = w1.x;
                        info2N = w1.y;
                    } else {
                        infoN = ldsru(sa + 512u);
                    }
                };
                asm volatile("" : "+s"(em_lo), "+s"(em_hi));
                __builtin_amdgcn_s_setprio(2);
                MM_PAIR_CASES(MM_DPAIR_TWO)
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(LINF ? smin < sthr : worst > thr) != 0ull, 0)) *redo2 = 1;
            }
            if constexpr (PHASE == 1)  // C' * (A .* B) of the previous step (:155)
                if (t - 1 > t0)
                    dpair_pdf_sums<(NJ > 4 ? 1 : (NJ > 2 ? 2 : 3))>(L::Q(RD), L::PDFSE, L::PSUM(RD), P1, wave, NWC, lane, H > 1 ? xps_send + ((t - 1) & 3) * (int)L::XPS : nullptr,
                                   H > 1 && split_tag(t - 1, t0, 2) != 0u);
            if constexpr (H > 1) {
                // The rows of the other sets of this step (pair_agent, MM_SPLIT_CWPOLL): chunk j (128 granules) of the q-th other
                // set is item q * NG2 + j, compute wave w receives the items w, w + NWC, ...; a granule is one tagged double.
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
                        if (pend && (v.y >> 31) == tg && (!second || (v.w >> 31) == tg)) {
                            mm_u32x4 w;
                            w.x = v.x;
                            w.y = v.y & 0x7fffffffu;
                            w.z = second ? v.z : 0u;
                            w.w = second ? v.w & 0x7fffffffu : 0u;
                            *(__attribute__((address_space(3))) mm_u32x4 *)(__UINTPTR_TYPE__)dsta = w;
                            pend = false;
                        }
                        if (__builtin_amdgcn_ballot_w64(pend) == 0ull) break;
                        if (__builtin_amdgcn_s_memrealtime() - tstart > x_tmo) {
                            cdead = true;  // the team is not running together: the log-domain kernels compute the utterance
                            if (lane == 0) *redo2 = 2;
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
                dpair_pdf_sums<(NJ > 4 ? 1 : (NJ > 2 ? 2 : 3))>(L::Q(t1 & 1), L::PDFSE, L::PSUM(t1 & 1), P1, wave, NWC, lane, H > 1 ? xps_send + (t1 & 3) * (int)L::XPS : nullptr,
                               H > 1 && split_tag(t1, t0, 2) != 0u);
            __syncthreads();  // (a)
        }
    }
#ifdef MM_STAMPS
    if (p.dbg && lane == 0 && H == 1)  // [utterance][wave][phase * 2 + dir][work, barrier]
        for (int k = 0; k < 2; ++k) p.dbg[(((long long)ui * MM_MAX_WAVES + wave) * 4 + PHASE * 2 + DIR) * 2 + k] = stamp_acc[k];
#endif
}

// mm_pair_finish_kernel for the utterances the double kernels computed (redo[b] != 0 on entry): ttl = min over both agents'
// frames, zeros beyond len_b, and what a range mark of the double kernels (redo2[b]) means -- the same two criteria on the
// double's range: the per-frame normalisers must agree (nothing that matters for log Z was flushed), and with the smallest
// overlap term L_n >= floor_d no posterior above the floor can have been lost (a flushed term was below 2^-1022 of its
// frame's scale).  redo2[b] becomes 0 (the result stands) or stays (the exact log-domain kernels compute the utterance);
// redo[b] keeps saying which utterances the float64 kernels computed (mm_batch_last_redo_count).
static __global__ void mm_dpair_finish_kernel(RunParams p) {
    const int b = blockIdx.x;
    if (p.redo[b] == 0) {
        if (p.stat_mode == 1 && threadIdx.x == 0) report_hard(p, 0);
        return;
    }
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
        const bool agree = z > -__builtin_inf() && zM < __builtin_inf() && zM - z <= MM_Z_SPREAD_TOL;
        // (no mark: the result stands, like an unmarked utterance of the float32 kernels); redo2[b] != 0 afterwards: the
        // log-domain kernels compute the utterance
        // (a mark of value 2 -- a team that did not run together -- stays)
        if (p.redo2[b] == 1 && agree && lm >= (double)p.lt_floor - (double)MM_DPAIR_THR_EXTRA) p.redo2[b] = 0;
        // (the wide kernels' linear finishes raise no mark on an overflow: it ends as a frame sum that is not a number, zM = inf)
        if (p.redo2[b] == 0 && len >= 1 && !(zM < __builtin_inf())) p.redo2[b] = 1;
        // (... and the overlap condition holds for unmarked utterances too: a product of the combine below 2^-1022 is flushed without
        // a mark -- mm_pair_finish_kernel; all frames without mass: an utterance without a path)
        if (p.redo2[b] == 0 && len >= 1 && !(z == -__builtin_inf() && zM == -__builtin_inf()) && !(lm >= (double)p.lt_floor - (double)MM_DPAIR_THR_EXTRA)) p.redo2[b] = 1;
        // (a call that skipped the float32 kernels: would they have coped?  Not with an overlap term below their floor)
        if (p.stat_mode == 1) report_hard(p, !(lm >= (double)p.lt_floor));
    }
    const long long gbase = (long long)b * p.gsb;
    for (long long q = threadIdx.x; q < (long long)(p.N - len) * P; q += blockDim.x)
        p.gamma[gbase + (len + q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
}

}  // namespace mm
