// mm_kernel_quad.hip -- the fast pdfposteriors kernel for gfx950 ("quad" kernel).
// Included by mm_engine.hip after mm_kernels.hip.
//
// Same contract and numerics as mm_log_kernel<MODE_FB> (pdfposteriors,
// src/inference.jl:145-161: alpha-recursion :62-74, beta-recursion :99-110, combine
// :154-160), different evaluation of the log-semiring row products.  Measured on
// the item kernel (profiles/r01_v1_*): ~70 VALU + ~60 SALU instructions and
// three dependent LDS round trips per 4x64 arcs, 61 % of wave time waiting.  Here:
//
//   * log-sum-exp with a FRAME-level shift instead of a row-level one.  The
//     recursion keeps, next to the normalised log2 vector a~ (max over states
//     ~ 0 by the lagged normaliser), its linear image p = 2^a~ in LDS.  A row is
//     then  log2( sum_k 2^w_k * p[col_k] )  -- one LDS gather and ONE FMA per arc,
//     no per-arc transcendental, no max pass, no cross-lane reduction.
//   * the graph lives in REGISTERS: every row is cut into quads of 4 arcs; a lane
//     owns KQ quads for the whole time loop (linear weights + LDS byte offsets:
//     6.5 VGPRs per quad).  The per-frame code is branch-free and statically
//     unrolled, so all of a lane's gathers are in flight together.
//   * a lane adds the quads of one row that sit next to each other in its
//     registers (static per-lane bit mask), stores the running sums to LDS (plain
//     ds_write; LDS float atomics measured ~80 cycles per wave instruction), and
//     after the barrier one thread per row adds the few lane-partials of its row
//     in a fixed order (deterministic) and finishes the row: log2, emission,
//     normaliser, 2^x for the next frame.
//   * states are renumbered per direction (mm_pack.h): rows by decreasing size,
//     so the lanes of a wave run loops of similar length and every per-row LDS
//     access is contiguous; in the backward direction grouped by pdf, so the
//     reference's C' * (A .* B) (src/inference.jl:154-155) is a sum over
//     contiguous positions done by one rotating wave -- no atomics anywhere.
//     The alpha store in HBM is written coalesced in the forward numbering; the
//     backward pass prefetches each thread's own rows through a register-held map.
//   * inside a row the host places the arcs on the (quad, slot) grid so that the
//     gathers of a half-wave hit distinct LDS banks where possible.
//   * EXACT fallback per row: when the linear sum leaves the range where it is
//     trustworthy (sum < 2^-90: the row lies > 62 nats below the frame maximum or
//     is unreachable; or overflow) the row is recomputed as a two-pass
//     log-sum-exp straight from the log2 vector and the CSR row.  Results are the
//     log-semiring's for every input; only the speed depends on the data.
//     (Graphs whose weights do not fit the linear range are refused by the host
//     for this kernel and run on mm_log_kernel.)
#pragma once
#include "mm_kernels.hip"

namespace mm {

// Pointers read out of the utterance descriptor are generic to the compiler; dereferencing them
// as FLAT loads makes it wait vmcnt(0)/lgkmcnt(0) conservatively all over the frame loop.  They
// always point to device global memory: say so.
template <class T>
__device__ __forceinline__ const __attribute__((address_space(1))) T *as_global(const T *p) {
    return (const __attribute__((address_space(1))) T *)p;
}

typedef unsigned mm_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned mm_u32x2 __attribute__((ext_vector_type(2)));

#define MM_Q_THR 8.0779357e-28f  // 2^-90
#define MM_Q_BIG 1.2676506e30f   // 2^100

struct LdsPlanQ {
    int abuf, pbuf, qs, qrow, em, part, recs, dist, pdfse, psum, total;
};
// nqcap = quad slots: threads * KQ, or more when the graph overflows the register window
__host__ __device__ inline LdsPlanQ lds_plan_q(int S1p, int P1p, int nqcap) {
    LdsPlanQ l;
    // p first: the gather offsets held in registers are then absolute LDS addresses (no base add)
    l.pbuf = 0;
    l.abuf = l.pbuf + 2 * (((S1p + 31) & ~31) + 16);  // up to two bank-rotated copies of p (mm_pack.h quad_pstride)
    l.qs = l.abuf + 2 * S1p;
    l.qrow = l.qs + ((nqcap + MM_QS_PAD + 3) & ~3);  // two always-zero slots, then the quad sums
    l.em = l.qrow + S1p;
    l.part = l.em + 2 * P1p;
    l.recs = l.part + 2 * MM_MAX_WAVES;
    l.dist = l.recs + 2 * S1p;  // u16 per row
    l.pdfse = l.dist + ((S1p + 7) & ~7) / 2;
    l.psum = l.pdfse + P1p;
    l.total = l.psum + 2 * P1p;  // two frames: one being summed, one being normalised and written out
    return l;
}

template <int KQ>
struct QuadRegs {
    float wl[KQ][4];
    unsigned off[KQ][2];  // off0 | off1 << 16, off2 | off3 << 16  (LDS byte offsets)
};

__device__ __forceinline__ void load_quad(const Quad *q, float (&wl)[4], unsigned (&off)[2]) {
    const mm_u32x4 a = *as_global(reinterpret_cast<const mm_u32x4 *>(q));
    const mm_u32x2 b = *as_global(reinterpret_cast<const mm_u32x2 *>(reinterpret_cast<const char *>(q) + 16));
    wl[0] = __uint_as_float(a.x);
    wl[1] = __uint_as_float(a.y);
    wl[2] = __uint_as_float(a.z);
    wl[3] = __uint_as_float(a.w);
    off[0] = b.x;
    off[1] = b.y;
}

template <int KQ>
__device__ __forceinline__ void load_quad_regs(QuadRegs<KQ> &rg, const QuadDev &g, int tid) {
    static_for<0, KQ>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const int q = tid * KQ + j;
        rg.wl[j][0] = rg.wl[j][1] = rg.wl[j][2] = rg.wl[j][3] = 0.f;
        rg.off[j][0] = rg.off[j][1] = 0u;
        if (q < g.nq) load_quad(g.quads + q, rg.wl[j], rg.off[j]);
    });
}

// sum of one quad on top of `run`, the running sum of the lane: kept if the quad continues its
// predecessor's row (sign bit of the first weight, mm_pack.h Quad), dropped if it starts a row.
// The unpacking of the 16-bit LDS offsets and the sign test are loop invariant; hoisted out of the time
// loop they would cost 2 VGPRs + 2 SGPRs per quad, which do not exist.  They are therefore written as asm
// that also reads `vz`, a zero the compiler cannot see through (one instruction per address, no copies).
__device__ __forceinline__ float lds_abs(unsigned addr) { return *(lds_cfptr)(__UINTPTR_TYPE__)addr; }
__device__ __forceinline__ unsigned unpack_lo(unsigned packed, unsigned vz) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
        : "=v"(r) : "v"(vz), "v"(packed));
    return r;
}
__device__ __forceinline__ unsigned unpack_hi(unsigned packed, unsigned vz) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(r) : "v"(vz), "v"(packed));
    return r;
}
__device__ __forceinline__ float quad_sum(const float (&wl)[4], unsigned off01, unsigned off23, const float *pbuf,
                                          float run, unsigned vz) {
    // vz = LDS address of pbuf (0 in practice): the sums are absolute LDS addresses
    const float x0 = lds_abs(unpack_lo(off01, vz)), x1 = lds_abs(unpack_hi(off01, vz));
    const float x2 = lds_abs(unpack_lo(off23, vz)), x3 = lds_abs(unpack_hi(off23, vz));
    float acc;  // run, or 0 when the sign bit of wl[0] is clear
    asm("v_ashrrev_i32 %0, 31, %1\n\tv_and_b32 %0, %0, %2" : "=&v"(acc) : "v"(wl[0]), "v"(run));
    acc = fmaf(__builtin_fabsf(wl[0]), x0, acc);
    acc = fmaf(wl[1], x1, acc);
    acc = fmaf(wl[2], x2, acc);
    acc = fmaf(wl[3], x3, acc);
    return acc;
}

// phase A.  qs[q] = running sum, inside lane q / KQ, of the quads of q's row up to q
// (quad sum = sum_k 2^w_k * p[col_k] over its 4 arcs).  So the partial sums of a row are
// found at the last quad of the row and at every lane end (q % KQ == KQ - 1) before it.
template <int KQ>
__device__ __forceinline__ void quad_phase(const QuadRegs<KQ> &rg, const QuadDev &g, int nq, int tid, int NT,
                                           const float *__restrict__ pbuf, float *__restrict__ qs) {
    // pbuf and qs never overlap (__restrict__): the stores of one quad do not hold back the gathers of
    // the next, and no array of quad sums has to stay live
    float run = 0.f;
    unsigned vz = lds_addr_of(pbuf);
    asm volatile("" : "+v"(vz));
    static_for<0, KQ>([&](auto J) {
        constexpr int j = decltype(J)::value;
        run = quad_sum(rg.wl[j], rg.off[j][0], rg.off[j][1], pbuf, run, vz);
        qs[tid * KQ + j] = run;
    });
    // lanes beyond the register window ("virtual lanes"): the same, streamed from L2
    for (int v = NT + tid; v * KQ < nq; v += NT) {
        float r = 0.f;
        for (int j = 0; j < KQ && v * KQ + j < nq; ++j) {
            float wl[4];
            unsigned off[2];
            load_quad(g.quads + v * KQ + j, wl, off);
            r = quad_sum(wl, off[0], off[1], pbuf, r, vz);
            qs[v * KQ + j] = r;
        }
    }
}

// phase B: the sum of a row from its record (mm_pack.h RowRec as two words: x = qe | i1 << 16, y = pdf |
// i2 << 16; qs2 = the two zero slots followed by the quad sums): the running sum in its last quad plus the
// ends of the lanes it started in.  Rows that span at most three lanes -- nearly all -- take three
// unconditional, independent loads (absent terms read a zero slot).
__device__ __forceinline__ float row_short(const float *__restrict__ qs2, unsigned x, unsigned y) {
    const float a = qs2[x & 0xffffu], b = qs2[x >> 16], c = qs2[y >> 16];
    return a + (b + c);
}
__device__ __forceinline__ bool row_is_long(unsigned y) { return (y >> 16) == (unsigned)MM_ROW_LONG; }
template <int KQ>
__device__ __forceinline__ float row_long(const float *__restrict__ qs2, unsigned x) {
    const int qe = x & 0xffffu;
    float acc = qs2[qe];
    for (int q = x >> 16; q < qe; q += 4 * KQ) {
        const float d0 = qs2[q];
        const float d1 = (q + KQ < qe) ? qs2[q + KQ] : 0.f;
        const float d2 = (q + 2 * KQ < qe) ? qs2[q + 2 * KQ] : 0.f;
        const float d3 = (q + 3 * KQ < qe) ? qs2[q + 3 * KQ] : 0.f;
        acc += (d0 + d1) + (d2 + d3);
    }
    return acc;
}
template <int KQ>
__device__ __forceinline__ float row_total(const float *__restrict__ qs2, unsigned x, unsigned y) {
    return row_is_long(y) ? row_long<KQ>(qs2, x) : row_short(qs2, x, y);
}
// is the linear-domain sum in the range where log2 of it is accurate?  (one subtract + one unsigned
// compare; negative, NaN and zero fall outside)
__device__ __forceinline__ bool sum_in_range(float acc) {
    return (__float_as_uint(acc) - 0x12800000u) <= (0x71800000u - 0x12800000u);  // bits of 2^-90, 2^100
}

// exact log-semiring row product from the log2 vector: two-pass log-sum-exp over the CSR row.  The arcs
// are fetched four at a time with independent loads (index and weight, then the gathered values), and the
// first four stay in registers for the second pass -- rows of left-to-right graphs rarely have more.
template <class IP, class FP>
__device__ __forceinline__ float exact_row_walk(IP rowptr, IP col, FP w, int r, const float *a) {
    const int b = rowptr[r], e = rowptr[r + 1];
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = b + i < e ? b + i : b;  // clamped: a valid arc, masked below
        const int c = col[k];
        const float wk = w[k];
        t[i] = (b + i < e) ? wk + a[c] : MM_NINF;
    }
    float m = fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3]));
    for (int k0 = b + 4; k0 < e; k0 += 4) {
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + i < e ? k0 + i : k0;
            const int c = col[k];
            const float wk = w[k];
            u[i] = (k0 + i < e) ? wk + a[c] : MM_NINF;
        }
        m = fmaxf(m, fmaxf(fmaxf(u[0], u[1]), fmaxf(u[2], u[3])));
    }
    if (!(m > MM_NINF) || b >= e) return MM_NINF;
    if (!(m < __builtin_inff())) return m;
    float s = (fast_exp2(t[0] - m) + fast_exp2(t[1] - m)) + (fast_exp2(t[2] - m) + fast_exp2(t[3] - m));
    for (int k0 = b + 4; k0 < e; k0 += 4) {
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + i < e ? k0 + i : k0;
            const int c = col[k];
            const float wk = w[k];
            u[i] = (k0 + i < e) ? wk + a[c] : MM_NINF;
        }
        s += (fast_exp2(u[0] - m) + fast_exp2(u[1] - m)) + (fast_exp2(u[2] - m) + fast_exp2(u[3] - m));
    }
    return m + fast_log2(s);
}
// FAST: the pipelined walk (small-graph geometries, where most rows come here and registers are plentiful);
// otherwise the plain loop, whose few registers do not disturb the allocation of the large geometries
template <bool FAST>
__device__ __forceinline__ float exact_row(const QuadDev &g, int r, const float *a) {
    const auto rowptr = as_global(g.rowptr);
    const auto col = as_global(g.col);
    const auto w = as_global(g.w);
    if constexpr (FAST) return exact_row_walk(rowptr, col, w, r, a);
    const int b = rowptr[r], e = rowptr[r + 1];
    float m = MM_NINF;
    for (int k = b; k < e; ++k) m = fmaxf(m, w[k] + a[col[k]]);
    if (!(m > MM_NINF)) return MM_NINF;
    if (!(m < __builtin_inff())) return m;
    float s = 0.f;
    for (int k = b; k < e; ++k) s += fast_exp2(w[k] + a[col[k]] - m);
    return m + fast_log2(s);
}
// the same walk over a copy of the CSR in LDS (small graphs, RunParams::xcsr)
__device__ __forceinline__ float exact_row_lds(const int *rowptr, const int *col, const float *w, int r, const float *a) {
    return exact_row_walk(rowptr, col, w, r, a);
}
// copy one direction's CSR into LDS: rowptr[S1 + 1], col[nnz], w[nnz]
__device__ __forceinline__ void stage_xcsr(float *xc, const QuadDev &g, int S1, int tid, int NT) {
    const int nnz = as_global(g.rowptr)[S1];
    int *xr = reinterpret_cast<int *>(xc), *xcol = xr + S1 + 1;
    float *xw = reinterpret_cast<float *>(xcol + nnz);
    for (int s = tid; s <= S1; s += NT) xr[s] = as_global(g.rowptr)[s];
    for (int k = tid; k < nnz; k += NT) {
        xcol[k] = as_global(g.col)[k];
        xw[k] = as_global(g.w)[k];
    }
}

struct RowRecU {
    unsigned x, y;  // RowRec as two words: qe | i1 << 16,  pdf | i2 << 16
    __device__ __forceinline__ unsigned pdf() const { return y & 0xffffu; }
    __device__ __forceinline__ bool has_arcs() const { return (x & 0xffffu) != 0u; }
};
__device__ __forceinline__ RowRecU load_rec(const float *recs, int i) {
    const uint2 r = reinterpret_cast<const uint2 *>(recs)[i];
    return RowRecU{r.x, r.y};
}
// A row whose sum came out of range is usually simply dead: no path of this many arcs from an initial
// state (forward) / to the final state (backward) exists (mm_pack.h reach_distance).  Those are zero(K)
// without a walk; only the others go through exact_row().
__device__ __forceinline__ bool row_is_dead(const unsigned short *distl, int i, int steps) {
    const unsigned d = distl[i];
    return d == 0xffffu || (unsigned)steps < d;
}

// The records of this thread's rows tid, tid + NT, ... are read from LDS BEFORE the barrier that ends the
// quad phase (they do not depend on it; the registers of the gathers are free by then), so that after the
// barrier the emissions and partial sums of all rows are fetched by independent loads at once.
template <int RPT>
__device__ __forceinline__ void load_row_recs(RowRecU (&rr)[RPT], const float *recs, int tid, int NT, int S1) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int i = tid + k * NT;
        rr[k] = load_rec(recs, i < S1 ? i : S1 - 1);  // no row: a valid record, the result is not stored
    }
}
// does this thread own a row that needs the long walk?  (decided once per direction)
template <int RPT>
__device__ __forceinline__ bool any_long_row(const QuadDev &g, int tid, int NT, int S1) {
    bool anylong = false;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int i = tid + k * NT;
        if (i < S1) anylong |= row_is_long(as_global(reinterpret_cast<const mm_u32x2 *>(g.recs))[i].y);
    }
    return anylong;
}

// C' * (A .* B) (src/inference.jl:154-155) without atomics: states of one pdf are contiguous in the
// backward numbering.  Every wave sums 8 pdfs per pass, 8 lanes per pdf (+ a 3-step DPP reduction).
__device__ __forceinline__ void pdf_sums(const float *qrow, const unsigned short *pdfse, float *psum, int P1, int wave,
                                         int NW, int lane) {
    for (int p0 = wave * 8; p0 < P1; p0 += NW * 8) {
        const int pdf = p0 + (lane >> 3);
        float s = 0.f;
        if (pdf < P1) {
            const unsigned se = reinterpret_cast<const unsigned *>(pdfse)[pdf];  // first | end << 16
            for (int t = (int)(se & 0xffffu) + (lane & 7); t < (int)(se >> 16); t += 8) s += qrow[t];
        }
        s = grp_sum(s, 3);
        if (pdf < P1 && (lane & 7) == 0) psum[pdf] = s;
    }
}

// one wave: per-frame sum over the pdfs, divide, store gamma (src/inference.jl:156-160); returns the sum
__device__ __forceinline__ float finish_frame(const float *psum, int P1, int P, int lane, float *gp, long long gsp) {
    const float s0 = lane < P1 ? psum[lane] : 0.f, s1 = lane + 64 < P1 ? psum[lane + 64] : 0.f;
    float tot = s0 + s1;
    for (int q = lane + 128; q < P1; q += 64) tot += psum[q];
    tot = wave_sum(tot);
    const float inv = 1.f / tot;
    if (lane < P) gp[lane * gsp] = s0 * inv;
    if (lane + 64 < P) gp[(lane + 64) * gsp] = s1 * inv;
    for (int q = lane + 128; q < P; q += 64) gp[q * gsp] = psum[q] * inv;
    return tot;
}

// KQ: quads per lane held in registers.  RPT: rows per thread handled by the unrolled row-finishing code
// (and whose alpha prefetch is carried in registers); more rows per thread go through a generic loop.
// PASS 0: the forward kernel (alpha-recursion; leaves alpha, the per-frame normalisers and log Z in the
// workspace).  PASS 1: the backward kernel (beta-recursion fused with the combine).  Two kernels rather than
// one: each keeps only its own direction's pointers, masks and loop state in registers, and each direction
// gets the number of quads per lane (KQ) its own matrix needs.
template <int KQ, int RPT, int PASS>
__global__ void __launch_bounds__(KQ > 13 ? 512 : 1024) mm_fbq_kernel(RunParams p) {
    extern __shared__ float lds[];
    const int b = p.order ? p.order[blockIdx.x] : blockIdx.x;
    if (p.redo && !p.redo[b]) return;  // launched behind the row kernels: only the utterances they marked
    const UttDesc &u = p.utts[b];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6;
    const int S1 = u.S1, S1p = u.S1p, P1 = u.P1, P = P1 - 1, P1p = (P1 + 3) & ~3;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int NF = len + 1;
    // Backward kernel: the waves of the upper half get a higher issue priority.  Measured, config 3: 4.29 ->
    // 4.0 ms (the later waves reach the barriers last; the same in the forward kernel gains nothing).
    if constexpr (PASS == 1 && KQ <= 13) {  // (the 16-wave geometries; 8-wave and small graphs lose a little)
        if (NW < 12) {
        } else if (wave * 4 >= NW * 3) __builtin_amdgcn_s_setprio(2);
        else if (wave * 2 >= NW) __builtin_amdgcn_s_setprio(1);
    }
    // this kernel's direction.  A reference on purpose: the pointers only the set-up and the cold exact path
    // need are re-read from memory there instead of occupying scalar registers through the time loop
    const QuadDev &qf = u.q[0], &qb = u.q[1];
    const int nqd = u.q[PASS].nq, fpos = u.q[PASS].fpos;
    const int vl = (nqd + KQ - 1) / KQ;  // lanes (real + virtual) that hold quads
    const LdsPlanQ L = lds_plan_q(S1p, P1p, (vl > NT ? vl : NT) * KQ);
    float *abuf = lds + L.abuf, *pbuf = lds + L.pbuf, *qs2 = lds + L.qs, *qs = qs2 + MM_QS_PAD, *qrow = lds + L.qrow;
    float *em = lds + L.em, *part = lds + L.part, *recs = lds + L.recs;
    unsigned short *pdfse = reinterpret_cast<unsigned short *>(lds + L.pdfse);
    unsigned short *distl = reinterpret_cast<unsigned short *>(lds + L.dist);
    float *psum = lds + L.psum;
    float *xc = lds + L.total;  // exact-fallback CSR, when the batch reserved LDS for it
    // (only the geometries of small graphs carry this code: it must not cost the large ones registers)
    const bool xres = KQ <= 3 && p.xcsr > 0;
    auto exact = [&](const QuadDev &g, int i, const float *a) -> float {
        if (xres) {
            const int *xr = reinterpret_cast<const int *>(xc), *xcol = xr + S1 + 1;
            return exact_row_lds(xr, xcol, reinterpret_cast<const float *>(xcol + xr[S1]), i, a);
        }
        return exact_row<(KQ <= 3)>(g, i, a);
    };
    const float *Vb = p.V + (long long)b * p.vsb;
    float *wsA = p.ws_alpha + u.s1p_prefix * (long long)(p.N + 1);
    // per-frame forward normalisers M_k (floats; the double workspace row of this utterance is reused)
    float *wsM = reinterpret_cast<float *>(p.ws_c + (long long)b * (p.N + 2));
    // hand-over between the two kernels, in the utterance's workspace row: [N] = normalised log2 value of the
    // final state in the last frame, [N + 1] = log2 Z
    double *hand = p.ws_c + (long long)b * (p.N + 2) + p.N;
    QuadRegs<KQ> rg;
    const int ncopy = u.q[PASS].ncopy, pstride = quad_pstride(S1p, ncopy);
    auto put_p = [&](int i, float v) {  // every copy of the linear vector
        pbuf[i] = v;
        if (ncopy > 1) pbuf[pstride + i] = v;
    };

    MM_STAMP_DECL;
    if constexpr (PASS == 0) {
    // ---------------- forward: alpha-recursion (src/inference.jl:62-74), forward numbering ----------------
    stage_em(em + 1 * P1p, Vb, p.vsn, 1, len, P, tid, NT, MM_LOG2E);
    if (tid < MM_QS_PAD) qs2[tid] = 0.f;  // absent terms of the row sums read these
    for (int q = tid; q < 2 * S1p; q += NT) abuf[q] = MM_NINF;
    for (int q = tid; q < 2 * (((S1p + 31) & ~31) + 16); q += NT) pbuf[q] = 0.f;
    for (int s = tid; s < S1; s += NT)
        reinterpret_cast<mm_u32x2 *>(recs)[s] = as_global(reinterpret_cast<const mm_u32x2 *>(qf.recs))[s];
    for (int s = tid; s < S1; s += NT) distl[s] = as_global(qf.dist)[s];
    if (xres) stage_xcsr(xc, qf, S1, tid, NT);
    __syncthreads();
    {   // frame 1: alpha_hat (*) lhs[:,1]   (src/inference.jl:68)
        float wm = MM_NINF;
        float *a1 = abuf + 1 * S1p;
        const float *e1 = em + 1 * P1p;
        for (int i = tid; i < S1; i += NT) {
            const float v = as_global(u.init_f)[i] + e1[load_rec(recs, i).pdf()];
            a1[i] = v;
            put_p(i, fast_exp2(v));
            wm = max_nc(wm, v);
        }
        part_put(part + 1 * MM_MAX_WAVES, wave, lane, wm);
        if (NF >= 2) stage_em(em + 0 * P1p, Vb, p.vsn, 2, len, P, tid, NT, MM_LOG2E);
    }
    load_quad_regs<KQ>(rg, qf, tid);
    bool anylong = any_long_row<RPT>(qf, tid, NT, S1);
    // where the geometry leaves registers, the forward kernel keeps its rows' records (and their unpacked
    // addresses) for good; otherwise they are re-read from LDS every step, before the barrier
    constexpr bool RR = KQ <= 10 || (KQ > 13 && KQ <= 25);
    RowRecU rr[RPT];
    if constexpr (RR) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = tid + k * NT;
            const mm_u32x2 r = as_global(reinterpret_cast<const mm_u32x2 *>(qf.recs))[i < S1 ? i : S1 - 1];
            rr[k] = RowRecU{r.x, r.y};
        }
    }
    __syncthreads();
    double C = 0.0;
    float ev = 0.f;
    MM_STAMP_RESET;
    for (int n = 2; n <= NF; ++n) {
        const float *ap = abuf + ((n - 1) & 1) * S1p;
        float *an = abuf + (n & 1) * S1p;
        const float *emn = em + (n & 1) * P1p;
        // emissions: frame n's raw values were loaded during step n-1 and are stored to LDS now (read
        // after the next barrier); storing them here rather than at the end of the previous step keeps
        // that step from waiting on its own alpha-store writes (vmcnt counts loads and stores in order)
        if (n > 2) {
            if (tid <= P) em[(n & 1) * P1p + tid] = em_value(ev, n, len, P, tid);
            if (P >= NT) stage_em(em + (n & 1) * P1p + NT, Vb + NT, p.vsn, n, len, P - NT, tid, NT, MM_LOG2E);
        }
        if (KQ > 13 || wave * 64 <= P) ev = em_load_raw(Vb, p.vsn, n + 1, p.N, P, tid);  // (16-wave geometries: only the waves that hold a pdf; measured)
        MM_STAMP(0);
        quad_phase<KQ>(rg, qf, nqd, tid, NT, pbuf, qs);
        {   // frame n-1 leaves the chip once (coalesced, forward numbering) while frame n is computed
            float4 *dst = reinterpret_cast<float4 *>(wsA + (long long)(n - 1) * S1p);
            const float4 *src = reinterpret_cast<const float4 *>(ap);
            const int n4 = S1p >> 2;
            if (tid < n4) dst[tid] = src[tid];
            if (n4 > NT)
                for (int q = tid + NT; q < n4; q += NT) dst[q] = src[q];
        }
        // everything phase B needs that does not depend on this frame's sums is read BEFORE the
        // barrier: the normaliser, the row records and the emissions of this thread's rows
        const float M = part_max_dpp(part + ((n - 1) & 1) * MM_MAX_WAVES, NW, lane);
        C += (double)M;
        if (tid == 0) wsM[n - 1] = M;  // M_{n-1}: C_n = sum_{k<n} M_k
        if constexpr (!RR) load_row_recs<RPT>(rr, recs, tid, NT, S1);
        MM_STAMP(1);
        __syncthreads();
        MM_STAMP(2);
        float wm = MM_NINF;
        {   // this thread's rows tid, tid + NT, ...: emissions and partial sums, all loads independent
            float e[RPT], acc[RPT];
#pragma unroll
            for (int k = 0; k < RPT; ++k) e[k] = emn[rr[k].pdf()];
#pragma unroll
            for (int k = 0; k < RPT; ++k) acc[k] = row_short(qs2, rr[k].x, rr[k].y);
            if (__builtin_expect(anylong, 0)) {
                // a long row whose emission is zero(K) this frame is zero whatever its sum is (the phony
                // final state, whose row is by far the longest, for every frame but the last): not read
#pragma unroll
                for (int k = 0; k < RPT; ++k)
                    if (row_is_long(rr[k].y)) acc[k] = e[k] > MM_NINF ? row_long<KQ>(qs2, rr[k].x) : 0.f;
            }
            MM_STAMP(5);
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int i = tid + k * NT;
                if (i < S1) {
                    const bool ok = sum_in_range(acc[k]) || !(e[k] > MM_NINF);
#ifdef MM_STAMPS
                    stamp_acc[7] += __popcll(__ballot(!ok && rr[k].has_arcs()));
#endif
                    float v = fast_log2(acc[k]);
                    // exact fallback, unless the sum is exactly 0 and no source can be alive-but-underflowed
                    if (__builtin_expect(!ok, 0))
                        v = (rr[k].has_arcs() && !row_is_dead(distl, i, n - 1)) ? exact(qf, i, ap) : MM_NINF;
                    v = v + e[k] - M;  // (T' alpha_{n-1}) (*) lhs[:,n]   (src/inference.jl:70-71)
                    const float pv = fast_exp2(v);
                    an[i] = v;
                    put_p(i, pv);
                    wm = max_nc(wm, v);
                }
            }
        }
        MM_STAMP(6);
        for (int i = tid + RPT * NT; i < S1; i += NT) {
            const RowRecU rec = load_rec(recs, i);
            const float acc = row_total<KQ>(qs2, rec.x, rec.y);
            const bool ok = sum_in_range(acc);
            float v = fast_log2(acc);
            if (__builtin_expect(!ok, 0)) v = (rec.has_arcs() && !row_is_dead(distl, i, n - 1)) ? exact(qf, i, ap) : MM_NINF;
            v = v + emn[rec.pdf()] - M;
            const float pv = fast_exp2(v);
            an[i] = v;
            put_p(i, pv);
            wm = max_nc(wm, v);
        }
        part_put(part + (n & 1) * MM_MAX_WAVES, wave, lane, wm);
        MM_STAMP(3);
        __syncthreads();
        MM_STAMP(4);
    }
#ifdef MM_STAMPS
    if (p.dbg && lane == 0)
        for (int k = 0; k < 8; ++k) p.dbg[((long long)b * MM_MAX_WAVES + wave) * 16 + k] = stamp_acc[k];
    for (int k = 0; k < 8; ++k) stamp_acc[k] = 0;
#endif
    if (tid == 0) {
        const float afin = abuf[(NF & 1) * S1p + fpos];  // normalised log2 value of the final state, last frame
        hand[0] = (double)afin;
        hand[1] = (double)afin + C;
    }
    } else {
    // ---------------- backward: beta-recursion fused with the combine, backward numbering ----------------
    const float afin = (float)hand[0];
    const double logZ2 = hand[1];
    const long long gbase = (long long)b * p.gsb;
    if (!(logZ2 > -1e300)) {  // no accepting path: gamma = 0, ttl = -inf
        for (long long q = tid; q < (long long)p.N * P; q += NT)
            p.gamma[gbase + (q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
        if (tid == 0) p.ttl[b] = MM_NINF;
        return;
    }
    for (int q = tid; q < 2 * S1p; q += NT) abuf[q] = MM_NINF;
    for (int q = tid; q < 2 * (((S1p + 31) & ~31) + 16); q += NT) pbuf[q] = 0.f;
    if (tid < MM_QS_PAD) qs2[tid] = 0.f;  // absent terms of the row sums read these
    for (int q = tid; q < S1p; q += NT) qrow[q] = 0.f;
    for (int s = tid; s < S1; s += NT) reinterpret_cast<mm_u32x2 *>(recs)[s] = as_global(reinterpret_cast<const mm_u32x2 *>(qb.recs))[s];
    for (int s = tid; s < S1; s += NT) distl[s] = as_global(qb.dist)[s];
    if (xres) stage_xcsr(xc, qb, S1, tid, NT);
    for (int s = tid; s < 2 * P1; s += NT) pdfse[s] = as_global(qb.pdfse)[s];
    // this thread's rows tid, tid + NT, ...: where their alpha sits in the (forward-numbered) store
    int amap[RPT];
    float acur[RPT], anxt[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int i = tid + k * NT;
        amap[k] = (i < S1) ? (int)as_global(u.map_bf)[i] : 0;
        acur[k] = anxt[k] = 0.f;
        if (len >= 1 && i < S1) acur[k] = wsA[(long long)len * S1p + amap[k]];
    }
    __syncthreads();
    if (tid == 0) {  // frame len+1: B (*) lhs = one for the final state only
        abuf[(NF & 1) * S1p + fpos] = 0.f;
        put_p(fpos, 1.f);
    }
    if (len >= 1) stage_em(em + (len & 1) * P1p, Vb, p.vsn, len, len, P, tid, NT, MM_LOG2E);
    load_quad_regs<KQ>(rg, qb, tid);
    __syncthreads();
    // kappa_n = log2 Z - C_n - D_n = afin + G_n with G_n = sum_{k>=n} M_k(forward) - sum M(backward):
    // accumulated incrementally (double; its magnitude stays small), no per-frame double loads
    double G = 0.0;
    const int fin_wave = NW > 2 ? 2 : 0;
    float tmin = 0.f, evb = 0.f;  // min over frames of log2 of the per-frame sum (relative to log2 Z)
    float mfn = len >= 1 ? wsM[len] : 0.f;  // forward normaliser M_len, then prefetched one step ahead
    double *zslot = hand + 1;  // log2 Z
    if (tid == 0) wsM[0] = 0.f;
    MM_STAMP_RESET;
    for (int n = len; n >= 1; --n) {
        const float *yp = abuf + ((n + 1) & 1) * S1p;
        float *yn = abuf + (n & 1) * S1p;
        const float *emn = em + (n & 1) * P1p;
        const float mf = mfn;                 // M_n (forward)
        mfn = wsM[n - 1];                     // M_{n-1} for the next step (wsM[0] = 0)
        // prefetch of frame n-1 (alpha of this thread's rows, emissions): consumed one step later
        if (n < len) {  // emissions of frame n, loaded during the previous step
            if (tid <= P) em[(n & 1) * P1p + tid] = em_value(evb, n, len, P, tid);
            if (P >= NT) stage_em(em + (n & 1) * P1p + NT, Vb + NT, p.vsn, n, len, P - NT, tid, NT, MM_LOG2E);
        }
        evb = em_load_raw(Vb, p.vsn, n - 1, p.N, P, tid);  // (every wave: restricting it to the waves that hold a pdf was measured slower here)
        if (n - 1 >= 1) {
            const float *src = wsA + (long long)(n - 1) * S1p;
#pragma unroll
            for (int k = 0; k < RPT; ++k)
                if (tid + k * NT < S1) anxt[k] = src[amap[k]];
        }
        MM_STAMP(0);
        // gamma of frame n+2: its per-pdf sums were completed in the previous step.  One wave normalises
        // and writes them now, ahead of its quad phase -- a wave of the first quarter, whose rows (sorted
        // by length within the pdf groups) leave it waiting at the next barrier anyway
        if (n + 2 <= len && wave == fin_wave) {
            const float s = finish_frame(psum + ((n + 2) & 1) * P1p, P1, P, lane, p.gamma + gbase + (long long)(n + 1) * p.gsn, p.gsp);
            tmin = fminf(tmin, fast_log2(s));
        }
        if (n < len) pdf_sums(qrow, pdfse, psum + ((n + 1) & 1) * P1p, P1, wave, NW, lane);  // frame n+1, per pdf
        quad_phase<KQ>(rg, qb, nqd, tid, NT, pbuf, qs);
        // read before the barrier what phase B needs and does not depend on this frame's sums
        const float M = (n == len) ? 0.f : part_max_dpp(part + ((n + 1) & 1) * MM_MAX_WAVES, NW, lane);
        G += (double)mf - (double)M;
        RowRecU rrb[RPT];
        load_row_recs<RPT>(rrb, recs, tid, NT, S1);
        MM_STAMP(1);
        __syncthreads();
        MM_STAMP(2);
        const float kappa = afin + (float)G;
        float wm = MM_NINF;
        // (one row after the other here: fetching all rows' terms at once, as the forward pass does, was
        // measured 30 % slower for this pass)
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = tid + k * NT;
            if (i < S1) {
                const RowRecU rec = rrb[k];
                const float acc = row_total<KQ>(qs2, rec.x, rec.y);
                const bool ok = sum_in_range(acc);
#ifdef MM_STAMPS
                stamp_acc[7] += __popcll(__ballot(!ok && rec.has_arcs()));
#endif
                float v = fast_log2(acc);
                // (the phony final state is exactly zero(K) or one(K) in the backward pass: no walk for it)
                if (__builtin_expect(!ok, 0))
                    v = (rec.has_arcs() && i != fpos && !row_is_dead(distl, i, NF - n)) ? exact(qb, i, yp) : MM_NINF;
                const float beta = v - M;  // T (B[:,n+1] (*) lhs[:,n+1])  (src/inference.jl:106-107)
                qrow[i] = fast_exp2(acur[k] + beta - kappa);
                const float y = beta + emn[rec.pdf()];
                const float py = fast_exp2(y);
                yn[i] = y;
                put_p(i, py);
                wm = max_nc(wm, y);
            }
        }
        for (int i = tid + RPT * NT; i < S1; i += NT) {  // more rows per thread than register slots
            const RowRecU rec = load_rec(recs, i);
            const float acc = row_total<KQ>(qs2, rec.x, rec.y);
            const bool ok = sum_in_range(acc);
            float v = fast_log2(acc);
            if (__builtin_expect(!ok, 0))
                    v = (rec.has_arcs() && i != fpos && !row_is_dead(distl, i, NF - n)) ? exact(qb, i, yp) : MM_NINF;
            const float beta = v - M;
            qrow[i] = fast_exp2(wsA[(long long)n * S1p + as_global(u.map_bf)[i]] + beta - kappa);
            const float y = beta + emn[rec.pdf()];
            const float py = fast_exp2(y);
            yn[i] = y;
            put_p(i, py);
            wm = max_nc(wm, y);
        }
        part_put(part + (n & 1) * MM_MAX_WAVES, wave, lane, wm);
#pragma unroll
        for (int k = 0; k < RPT; ++k) acur[k] = anxt[k];
        MM_STAMP(3);
        __syncthreads();
        MM_STAMP(4);
    }
#ifdef MM_STAMPS
    if (p.dbg && lane == 0)
        for (int k = 0; k < 8; ++k) p.dbg[((long long)b * MM_MAX_WAVES + wave) * 16 + 8 + k] = stamp_acc[k];
#endif
    // gamma of frame 1, zeros beyond len, ttl
    if (len >= 2 && wave == fin_wave) {  // frame 2: summed in the last step
        const float s = finish_frame(psum + 0 * P1p, P1, P, lane, p.gamma + gbase + p.gsn, p.gsp);
        tmin = fminf(tmin, fast_log2(s));
    }
    if (len >= 1) {
        pdf_sums(qrow, pdfse, psum + 1 * P1p, P1, wave, NW, lane);
        __syncthreads();
        if (wave == 0) {
            const float s = finish_frame(psum + 1 * P1p, P1, P, lane, p.gamma + gbase, p.gsp);
            tmin = fminf(tmin, fast_log2(s));
        }
    }
    for (long long q = tid; q < (long long)(p.N - len) * P; q += NT)
        p.gamma[gbase + (len + q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
    __syncthreads();
    if (lane == 0) part[wave] = tmin;  // ttl = log Z + min over frames of log(per-frame sum)
    __syncthreads();
    if (tid == 0) {
        float t = part[0];
        for (int w = 1; w < NW; ++w) t = fminf(t, part[w]);
        p.ttl[b] = (float)((*zslot + (double)t) * (double)MM_LN2);
    }
    }  // PASS
}

}  // namespace mm
