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
//   * every quad sum is stored to LDS (plain, conflict-free ds_write); after the
//     barrier one thread per row adds up the row's quads in a fixed order
//     (deterministic; LDS float atomics measured ~80 cycles per wave instruction
//     and made the kernel LDS-bound) and finishes the row: log2, emission,
//     normaliser, 2^x for the next frame, and in the backward pass the posterior
//     accumulation.  Rows are visited in decreasing-size order so the lanes of a
//     wave run loops of similar length.
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

#define MM_Q_THR 8.0779357e-28f  // 2^-90
#define MM_Q_BIG 1.2676506e30f   // 2^100
#define MM_Q_SINK 64             // dummy accumulator slots for padding quads (one per lane: no conflicts)

struct LdsPlanQ {
    int abuf, pbuf, qs, stage, em, bins, part, s2p, qstart, rord, total;
};
// nqcap = max(number of quads, threads * KQ): one float per quad sum
__host__ __device__ inline LdsPlanQ lds_plan_q(int S1p, int P1p, int nqcap) {
    LdsPlanQ l;
    l.abuf = 0;
    l.pbuf = l.abuf + 2 * S1p;
    l.qs = l.pbuf + S1p;
    l.stage = l.qs + ((nqcap + 3) & ~3);
    l.em = l.stage + 2 * S1p;
    l.bins = l.em + 2 * P1p;
    l.part = l.bins + 2 * P1p;
    l.s2p = l.part + 2 * MM_MAX_WAVES;
    l.qstart = l.s2p + (S1p + 1) / 2;
    l.rord = l.qstart + (S1p + 4) / 2;
    l.total = l.rord + (S1p + 1) / 2;
    return l;
}

template <int KQ>
struct QuadRegs {
    float wl[KQ][4];
    unsigned off[KQ][2];  // off0 | off1 << 16, off2 | off3 << 16  (LDS byte offsets)
};

__device__ __forceinline__ void load_quad(const Quad *q, float (&wl)[4], unsigned (&off)[2]) {
    const uint4 a = *reinterpret_cast<const uint4 *>(q);
    const uint4 b = *(reinterpret_cast<const uint4 *>(q) + 1);
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

__device__ __forceinline__ float lds_f32(const float *base, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}

__device__ __forceinline__ float quad_sum(const float (&wl)[4], unsigned off01, unsigned off23, const float *pbuf) {
    // keep the offsets packed: unpacking is loop invariant and hoisting it would cost 2 VGPRs per quad
    asm volatile("" : "+v"(off01), "+v"(off23));
    const float x0 = lds_f32(pbuf, off01 & 0xffffu), x1 = lds_f32(pbuf, off01 >> 16);
    const float x2 = lds_f32(pbuf, off23 & 0xffffu), x3 = lds_f32(pbuf, off23 >> 16);
    float acc = wl[0] * x0;
    acc = fmaf(wl[1], x1, acc);
    acc = fmaf(wl[2], x2, acc);
    acc = fmaf(wl[3], x3, acc);
    return acc;
}

// phase A: qs[q] = sum_k 2^w_k * p[col_k] over the 4 arcs of quad q.  Lane tid owns the
// quads tid * KQ .. tid * KQ + KQ - 1 (KQ odd: the stores of a wave hit distinct banks);
// a row's quads are contiguous in qs.
template <int KQ>
__device__ __forceinline__ void quad_phase(const QuadRegs<KQ> &rg, const QuadDev &g, int tid, int NT, const float *pbuf,
                                           float *qs) {
    static_for<0, KQ>([&](auto J) {
        constexpr int j = decltype(J)::value;
        qs[tid * KQ + j] = quad_sum(rg.wl[j], rg.off[j][0], rg.off[j][1], pbuf);
    });
    for (int q = NT * KQ + tid; q < g.nq; q += NT) {  // quads beyond the register window: streamed from L2
        float wl[4];
        unsigned off[2];
        load_quad(g.quads + q, wl, off);
        qs[q] = quad_sum(wl, off[0], off[1], pbuf);
    }
}

// phase B helper: add up the quads [q0, q1) of one row in a fixed order; the loads of
// a group of 8 are independent, so a long row costs few LDS round trips
__device__ __forceinline__ float row_sum(const float *qs, int q0, int q1) {
    float acc = 0.f;
    int q = q0;
    for (; q + 8 <= q1; q += 8) {
        const float a0 = qs[q], a1 = qs[q + 1], a2 = qs[q + 2], a3 = qs[q + 3];
        const float a4 = qs[q + 4], a5 = qs[q + 5], a6 = qs[q + 6], a7 = qs[q + 7];
        acc += ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    }
    if (q + 4 <= q1) {
        const float a0 = qs[q], a1 = qs[q + 1], a2 = qs[q + 2], a3 = qs[q + 3];
        acc += (a0 + a1) + (a2 + a3);
        q += 4;
    }
    if (q + 2 <= q1) {
        acc += qs[q] + qs[q + 1];
        q += 2;
    }
    if (q < q1) acc += qs[q];
    return acc;
}

// exact log-semiring row product from the log2 vector (two-pass log-sum-exp over the CSR row)
__device__ __forceinline__ float exact_row(const QuadDev &g, int r, const float *a) {
    const int b = g.rowptr[r], e = g.rowptr[r + 1];
    float m = MM_NINF;
    for (int k = b; k < e; ++k) m = fmaxf(m, g.w[k] + a[g.col[k]]);
    if (!(m > MM_NINF)) return MM_NINF;
    if (!(m < __builtin_inff())) return m;
    float s = 0.f;
    for (int k = b; k < e; ++k) s += fast_exp2(g.w[k] + a[g.col[k]] - m);
    return m + fast_log2(s);
}

template <int KQ>
__global__ void __launch_bounds__(1024) mm_fbq_kernel(RunParams p) {
    extern __shared__ float lds[];
    const int b = blockIdx.x;
    const UttDesc &u = p.utts[b];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6;
    const int S1 = u.S1, S1p = u.S1p, P1 = u.P1, P = P1 - 1, P1p = (P1 + 3) & ~3;
    const int fstate = S1 - 1;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int NF = len + 1;
    const QuadDev qf = u.q[0], qb = u.q[1];
    const int nqmax = qf.nq > qb.nq ? qf.nq : qb.nq;
    const LdsPlanQ L = lds_plan_q(S1p, P1p, nqmax > NT * KQ ? nqmax : NT * KQ);
    float *abuf = lds + L.abuf, *pbuf = lds + L.pbuf, *qs = lds + L.qs, *stage = lds + L.stage;
    float *em = lds + L.em, *bins = lds + L.bins, *part = lds + L.part;
    unsigned short *s2p = reinterpret_cast<unsigned short *>(lds + L.s2p);
    unsigned short *qstart = reinterpret_cast<unsigned short *>(lds + L.qstart);
    unsigned short *rord = reinterpret_cast<unsigned short *>(lds + L.rord);
    const float *Vb = p.V + (long long)b * p.vsb;
    float *wsA = p.ws_alpha + u.s1p_prefix * (long long)(p.N + 1);
    double *wsC = p.ws_c + (long long)b * (p.N + 2);
    QuadRegs<KQ> rg;

    // ---------------- forward: alpha-recursion (src/inference.jl:62-74) ----------------
    stage_em(em + 1 * P1p, Vb, p.vsn, 1, len, P, tid, NT, MM_LOG2E);
    for (int q = tid; q < 2 * S1p; q += NT) abuf[q] = MM_NINF;
    for (int q = tid; q < S1p; q += NT) pbuf[q] = 0.f;
    for (int s = tid; s < S1; s += NT) {
        s2p[s] = (unsigned short)u.s2p[s];
        rord[s] = qf.rord[s];
    }
    for (int s = tid; s <= S1; s += NT) qstart[s] = qf.qstart[s];
    __syncthreads();
    {   // frame 1: alpha_hat (*) lhs[:,1]   (src/inference.jl:68)
        float wm = MM_NINF;
        float *a1 = abuf + 1 * S1p;
        const float *e1 = em + 1 * P1p;
        for (int s = tid; s < S1; s += NT) {
            const float v = u.init[s] + e1[s2p[s]];
            a1[s] = v;
            pbuf[s] = fast_exp2(v);
            wm = fmaxf(wm, v);
        }
        wm = wave_max(wm);
        if (lane == 0) part[1 * MM_MAX_WAVES + wave] = wm;
        if (NF >= 2) stage_em(em + 0 * P1p, Vb, p.vsn, 2, len, P, tid, NT, MM_LOG2E);
        if (tid == 0) wsC[1] = 0.0;
    }
    load_quad_regs<KQ>(rg, qf, tid);
    __syncthreads();
    double C = 0.0;
    for (int n = 2; n <= NF; ++n) {
        const float *ap = abuf + ((n - 1) & 1) * S1p;
        float *an = abuf + (n & 1) * S1p;
        const float *emn = em + (n & 1) * P1p;
        // frame 1's p = 2^a is un-normalised (C_1 = 0): the lagged normaliser enters from frame 2 on
        const float M = part_max(part + ((n - 1) & 1) * MM_MAX_WAVES, NW);
        C += (double)M;
        if (tid == 0) wsC[n] = C;
        if (n + 1 <= NF) stage_em(em + ((n + 1) & 1) * P1p, Vb, p.vsn, n + 1, len, P, tid, NT, MM_LOG2E);
        {   // frame n-1 leaves the chip once (coalesced), while frame n is computed
            float4 *dst = reinterpret_cast<float4 *>(wsA + (long long)(n - 1) * S1p);
            const float4 *src = reinterpret_cast<const float4 *>(ap);
            for (int q = tid; q < (S1p >> 2); q += NT) dst[q] = src[q];
        }
        quad_phase<KQ>(rg, qf, tid, NT, pbuf, qs);
        __syncthreads();
        float wm = MM_NINF;
        for (int i = tid; i < S1; i += NT) {
            const int r = rord[i];
            const float acc = row_sum(qs, qstart[r], qstart[r + 1]);
            const bool ok = acc >= MM_Q_THR && acc <= MM_Q_BIG;
            float v = fast_log2(acc);
            if (__builtin_expect(!ok, 0)) v = exact_row(qf, r, ap);
            v = v + emn[s2p[r]] - M;  // (T' alpha_{n-1}) (*) lhs[:,n]   (src/inference.jl:70-71)
            an[r] = v;
            pbuf[r] = fast_exp2(v);
            wm = fmaxf(wm, v);
        }
        wm = wave_max(wm);
        if (lane == 0) part[(n & 1) * MM_MAX_WAVES + wave] = wm;
        __syncthreads();
    }
    const double logZ2 = (double)abuf[(NF & 1) * S1p + fstate] + C;
    __syncthreads();

    // ---------------- backward: beta-recursion fused with the combine ----------------
    const long long gbase = (long long)b * p.gsb;
    if (!(logZ2 > -1e300)) {  // no accepting path: gamma = 0, ttl = -inf
        for (long long q = tid; q < (long long)p.N * P; q += NT)
            p.gamma[gbase + (q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
        if (tid == 0) p.ttl[b] = MM_NINF;
        return;
    }
    for (int q = tid; q < 2 * S1p; q += NT) abuf[q] = MM_NINF;
    for (int q = tid; q < S1p; q += NT) pbuf[q] = 0.f;
    for (int q = tid; q < 2 * P1p; q += NT) bins[q] = 0.f;
    __syncthreads();
    if (tid == 0) {  // frame len+1: B (*) lhs = one for the final state only
        abuf[(NF & 1) * S1p + fstate] = 0.f;
        pbuf[fstate] = 1.f;
    }
    if (len >= 1) {
        stage_em(em + (len & 1) * P1p, Vb, p.vsn, len, len, P, tid, NT, MM_LOG2E);
        const float4 *src = reinterpret_cast<const float4 *>(wsA + (long long)len * S1p);
        float4 *dst = reinterpret_cast<float4 *>(stage + (len & 1) * S1p);
        for (int q = tid; q < (S1p >> 2); q += NT) dst[q] = src[q];
    }
    load_quad_regs<KQ>(rg, qb, tid);
    for (int s = tid; s < S1; s += NT) rord[s] = qb.rord[s];
    for (int s = tid; s <= S1; s += NT) qstart[s] = qb.qstart[s];
    __syncthreads();
    double D = 0.0;
    float tmin = (float)logZ2;
    for (int n = len; n >= 1; --n) {
        const float *yp = abuf + ((n + 1) & 1) * S1p;
        float *yn = abuf + (n & 1) * S1p;
        const float *ast = stage + (n & 1) * S1p;
        const float *emn = em + (n & 1) * P1p;
        float *bn = bins + (n & 1) * P1p;
        const float M = (n == len) ? 0.f : part_max(part + ((n + 1) & 1) * MM_MAX_WAVES, NW);
        D += (double)M;
        const double Cn = __hip_atomic_load(&wsC[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float kappa = (float)(logZ2 - Cn - D);
        // finalise frame n+1 (one rotating wave): C' * AB, per-frame sum, divide (src/inference.jl:155-160)
        if (n < len && wave == ((n + 1) % NW)) {
            float *bf = bins + ((n + 1) & 1) * P1p;
            float s = 0.f;
            for (int q = lane; q < P1; q += 64) s += bf[q];
            s = wave_sum(s);
            const float inv = 1.f / s;
            float *gp = p.gamma + gbase + (long long)n * p.gsn;
            for (int q = lane; q < P1; q += 64) {
                if (q < P) gp[q * p.gsp] = bf[q] * inv;
                bf[q] = 0.f;
            }
            tmin = fminf(tmin, (float)(logZ2 + (double)fast_log2(s)));
        }
        if (n - 1 >= 1) {
            stage_em(em + ((n - 1) & 1) * P1p, Vb, p.vsn, n - 1, len, P, tid, NT, MM_LOG2E);
            const float4 *src = reinterpret_cast<const float4 *>(wsA + (long long)(n - 1) * S1p);
            float4 *dst = reinterpret_cast<float4 *>(stage + ((n - 1) & 1) * S1p);
            for (int q = tid; q < (S1p >> 2); q += NT) dst[q] = src[q];
        }
        quad_phase<KQ>(rg, qb, tid, NT, pbuf, qs);
        __syncthreads();
        float wm = MM_NINF;
        for (int i = tid; i < S1; i += NT) {
            const int r = rord[i];
            const float acc = row_sum(qs, qstart[r], qstart[r + 1]);
            const int pdf = s2p[r];
            const bool ok = acc >= MM_Q_THR && acc <= MM_Q_BIG;
            float v = fast_log2(acc);
            if (__builtin_expect(!ok, 0)) v = exact_row(qb, r, yp);
            const float beta = v - M;  // T (B[:,n+1] (*) lhs[:,n+1])  (src/inference.jl:106-107)
            const float q = fast_exp2(ast[r] + beta - kappa);   // state_A .* state_B / Z
            if (q > 0.f) atomicAdd(&bn[pdf], q);
            const float y = beta + emn[pdf];
            yn[r] = y;
            pbuf[r] = fast_exp2(y);
            wm = fmaxf(wm, y);
        }
        wm = wave_max(wm);
        if (lane == 0) part[(n & 1) * MM_MAX_WAVES + wave] = wm;
        __syncthreads();
    }
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
    __syncthreads();
    if (lane == 0) part[wave] = tmin;
    __syncthreads();
    if (tid == 0) {
        float t = part[0];
        for (int w = 1; w < NW; ++w) t = fminf(t, part[w]);
        p.ttl[b] = t * MM_LN2;
    }
}

}  // namespace mm
