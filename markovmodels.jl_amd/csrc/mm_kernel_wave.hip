// mm_kernel_wave.hip -- the "wave" pdfposteriors kernel for gfx950: ONE WAVE computes a whole direction of one utterance,
// in the log domain, with no workgroup barrier in the time loop.  The path of small deep graphs -- LF-MMI numerators
// (examples/test_cuda.jl:78,128: one graph per utterance, ~450 states, left-to-right, degree ~2.3) -- whose values within
// one frame span far more than the float range, so that the linear-domain kernels do not apply, and for which a
// 16-wave workgroup with a barrier per frame (the item kernel, mm_kernels.hip) is 1.3 us of pure latency per frame.
//
//   * a workgroup = the two AGENTS of one utterance, NWD waves each: the forward agent runs the alpha-recursion
//     (src/inference.jl:62-74) from frame 1 upwards, the backward agent the beta-recursion (:99-110) from frame len+1 downwards,
//     at the same time.  Each stores its normalised log2 vectors for the first half of ITS frames (phase A); after ONE
//     workgroup barrier each continues through the other half and combines its fresh vector with what the other stored for
//     that frame (:154-160).  Serial depth of a call: len+1 steps, as in the pair kernels (mm_kernel_pairs.hip);
//   * the NWD waves of an agent share its state vector and split its segments; they meet once per step at a barrier of
//     their own -- an LDS counter (s_barrier would tie the two agents together, whose phases differ by a step when the
//     number of frames is odd): ~250 cycles, against the ~200 cycles of a segment;
//   * the graph of a direction sits in the wave's registers in the row-lane form of mm_rows.h with NWC = 1 (RowPackOpts::
//     acap_force / seg_stride / log_weights): up to NSEG segments of 64 / g rows, a lane holds at most 4 arcs of a segment
//     (log2 weight + LDS byte address of the source), so a segment is straight-line code: <= 4 gathers, a two-pass
//     log-sum-exp in registers, for rows of more than 4 arcs a lane-group maximum and sum by DPP;
//   * the state vector of a step lives in the wave's own LDS slice, double buffered by the parity of the step; a wave's LDS
//     writes are ordered before its own later reads, so no barrier is needed;
//   * normalisation by the lagged frame maximum (a~_t = a_t - C_t, C_t = sum_{k<t} max_j a~_k[j], C in double) as in the item
//     kernel; the semiring's zero is the finite sentinel MM_WAVE_NEG (-1e30: -inf - -inf never occurs);
//   * every reduction has a fixed order: the posteriors of a pdf are summed by ONE lane over the pdf's states in pdf-major
//     order (no atomics: the kernel is deterministic, unlike the item kernel's default mode).
#pragma once
#include "mm_kernel_rows.hip"

namespace mm {

#define MM_WAVE_STRIDE 4         // arc slots of a segment
#define MM_WAVE_NEG (-1.0e30f)   // zero(K) in the log2 domain
#define MM_WAVE_VSZ 4352u        // bytes of one state vector (1024 states + the "no row" position, padded)
#define MM_WAVE_ESZ 1056u        // bytes of one emission / pdf-sum buffer (256 pdfs + the "no row" slot)
// LDS slice of one wave (bytes, relative to the slice)
#define MM_WAVE_VEC(par) ((unsigned)(par) * MM_WAVE_VSZ)
#define MM_WAVE_EM(par) (2u * MM_WAVE_VSZ + (unsigned)(par) * MM_WAVE_ESZ)
#define MM_WAVE_NWD 4            // waves per agent
// (buf = step & 3: the sums of a step are added during the next step and read during the one after)
#define MM_WAVE_PS(buf, w) (2u * MM_WAVE_VSZ + 2u * MM_WAVE_ESZ + ((unsigned)(buf) * MM_WAVE_NWD + (unsigned)(w)) * MM_WAVE_ESZ)  // per-pdf sums of a wave
#define MM_WAVE_YM(par) (2u * MM_WAVE_VSZ + (2u + 4u * MM_WAVE_NWD) * MM_WAVE_ESZ + (unsigned)(par) * 32u)        // per-wave maxima of the vector
#define MM_WAVE_UM(buf) (MM_WAVE_YM(2) + (unsigned)(buf) * 32u)                                                  // ... of a~ + b~
#define MM_WAVE_OFF(buf) (MM_WAVE_UM(4) + (unsigned)(buf) * 16u)                                                 // {own, partner} offsets of a step (doubles)
#define MM_WAVE_SYNC MM_WAVE_OFF(4)
#define MM_WAVE_ZZ (MM_WAVE_SYNC + 16u)   // the waves' minima of the per-frame log2 normalisers (doubles)
#define MM_WAVE_RAW(k) (MM_WAVE_ZZ + 8u * MM_WAVE_NWD + 16u + (unsigned)(k) * 1024u)  // raw emissions of 4 frames in flight (LDS-DMA)
#define MM_WAVE_SLICE MM_WAVE_RAW(4)

__device__ __forceinline__ float wave_sum_fixed(float v) {  // the same tree in every run and in every lane's view
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

template <int NSEG, int NJ>  // NSEG: segments the registers of a wave hold; NJ * 64 >= P + 1
__global__ void __launch_bounds__(128 * (MM_WAVE_NWD + 1)) mm_wave_kernel(RunParams p) {
    extern __shared__ float lds[];
    constexpr int KA = MM_WAVE_STRIDE * NSEG, NWD = MM_WAVE_NWD, NWA = NWD + 1, NT = 128 * NWA;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // 0: forward agent, 1: backward agent; the wave of the agent: NWD compute waves and a SERVICE wave, which stages the
    // emissions and puts out the posteriors of a frame a step after the compute waves have summed them -- none of that is
    // on the compute waves' path from one barrier to the next
    const int DIR = wv / NWA, sub = wv % NWA;
    const bool service = sub == NWD;
    const int b = uni(p.order ? p.order[blockIdx.x] : (int)blockIdx.x);
    if (p.redo && !uni(p.redo[b])) return;
    const UttDesc &u = p.utts[b];
    const RowU r = uni(u.rw[DIR]);
    const int S1 = r.rows, S1p = uni(u.S1p), P1 = uni(u.P1), P = P1 - 1;
    int len = uni(p.lens ? p.lens[b] : p.N);
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int NF = len + 1;
    const float *Vb = p.V + (long long)b * p.vsb;
    const long long gbase = (long long)b * p.gsb;
    const long long s1p_prefix = ((long long)uni((int)(u.s1p_prefix >> 32)) << 32) | (unsigned)uni((int)u.s1p_prefix);
    float *rows = p.ws_alpha + s1p_prefix * (long long)(p.N + 1);  // [frame - 1][S1p]: a~ up to the split, b~ beyond
    double *offs = p.ws_c + (long long)b * (p.N + 2);               // [frame]: the offset of the stored vector
    if (lds_addr_of(lds) != 0u) __builtin_trap();
    MM_STAMP_DECL;
    const unsigned base = (unsigned)DIR * MM_WAVE_SLICE;
    if (len == 0) {  // no frame: gamma = 0, no path of length 0
        for (long long q = threadIdx.x; q < (long long)p.N * P; q += NT) p.gamma[gbase + (q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
        if (threadIdx.x == 0) p.ttl[b] = MM_NINF;
        return;
    }

    // ---- the graph of this direction: registers
    float w[KA];
    unsigned a[KA], s0[NSEG], s1[NSEG];
    const bool mine_w = sub < r.NWC && !service;  // (a graph of fewer segments than waves: the others only keep the step)
    const RowSched &sc = r.sched[mine_w ? sub : 0];
    const unsigned long long lgw =
        mine_w ? ((unsigned long long)(unsigned)uni((int)(sc.lg >> 32)) << 32) | (unsigned)uni((int)sc.lg) : 0ull;
    const int nseg = mine_w ? uni((int)(sc.nslots & 0xffffu)) : 0, slot0 = mine_w ? uni((int)sc.slot0) : 0;
    {
        const auto wp = as_global(r.w);
        const auto ap = as_global(r.addr);
        const auto sp = as_global(r.slots);
        const int nt = 64 * r.NWC, col = (mine_w ? sub : 0) * 64 + lane;
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            const bool have = mine_w && k < r.KA;
            w[k] = have ? wp[k * nt + col] : MM_NINF;
            a[k] = (have ? ap[k * nt + col] : 0u) + base;
            if (!(w[k] > MM_NINF)) w[k] = MM_WAVE_NEG;
        }
#pragma unroll
        for (int i = 0; i < NSEG; ++i) {
            const bool have = i < nseg;
            // (segments the wave does not have: no row -- the trash position, the emission slot that holds zero(K))
            s0[i] = have ? sp[((slot0 + i) * 64 + lane) * 2] : (4u * (unsigned)S1) | ((4u * (unsigned)((P1 + 3) & ~3)) << 16);
            s1[i] = have ? sp[((slot0 + i) * 64 + lane) * 2 + 1] : 0u;
        }
    }
    // ---- LDS set-up (the agent's slice, by its waves)
    for (unsigned q = 4u * (unsigned)(sub * 64 + lane); q < MM_WAVE_SYNC; q += 256u * NWA) ldsw(base + q, MM_WAVE_NEG);
    if (sub == 0 && lane < 4) ldswu(base + MM_WAVE_SYNC + 4u * lane, 0u);
    const unsigned trash4 = 4u * (unsigned)S1;  // position written by lanes that finish no row
    __syncthreads();
    // the agent's own barrier: every wave adds 1 to the counter when its LDS writes of the step are done and waits for
    // all NWD (LDS operations of a wave complete in order)
    unsigned epoch = 0;
    auto agent_sync = [&]() __attribute__((always_inline)) {
        epoch += NWA;
        // (ONE lane adds: 64 lanes adding to one address are 64 serialised atomics, ~500 cycles)
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(base + MM_WAVE_SYNC), "v"(1u) : "memory");
        unsigned seen;
        asm volatile("1: ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_cmp_lt_u32 vcc, %0, %2\n\ts_cbranch_vccnz 1b"
                     : "=&v"(seen)
                     : "v"(base + MM_WAVE_SYNC), "v"(epoch)
                     : "vcc", "memory");
    };
    // maximum of the vector of a step: this wave's part to LDS before the barrier, all parts after it
    auto vec_max = [&](unsigned ym) __attribute__((always_inline)) {
        float M = ldsr(base + ym);
#pragma unroll
        for (int k = 1; k < NWA; ++k) M = fmaxf(M, ldsr(base + ym + 4u * k));
        M = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, M)));
        return M > 0.5f * MM_WAVE_NEG ? M : 0.f;
    };

    // steps: step t handles frame t (forward) or NF + 1 - t (backward); steps 1 .. tA are phase A
    const int m = NF / 2;  // (len >= 1: NF >= 2, both agents have at least one step of phase A)
    const int tA = DIR ? NF - m : m;
    auto frame_of = [&](int t) __attribute__((always_inline)) { return DIR ? NF + 1 - t : t; };
    // emissions of frame f into EM(par): expand() (src/inference.jl:54-60) in the log2 domain, zero(K) = MM_WAVE_NEG
    // (service wave) raw values by LDS-DMA FOUR steps ahead -- a step is ~1 us, a load from HBM up to 2: fetched one step
    // ahead, the agent waited for the service wave's load every step -- staged a step ahead
    auto em_fetch = [&](int f) __attribute__((always_inline)) { row_dma_em<NJ>(base + MM_WAVE_RAW(f & 3), Vb, p.vsn, f, p.N, P, lane); };
    auto em_stage = [&](int f, int par) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            float v = em_value(ldsr(base + MM_WAVE_RAW(f & 3) + 256u * j + 4u * lane), f, len, P, q);
            v = v > MM_WAVE_NEG ? v : MM_WAVE_NEG;
            if (q <= P) ldsw(base + MM_WAVE_EM(par) + 4u * q, v);
        }
    };
    // ---- step 1: the initial vector (the emissions are staged by the agent's first wave)
    if (service) {
        for (int t = 1; t <= 4; ++t) em_fetch(frame_of(t));
        MM_ROW_VMCNT(0);
        em_stage(frame_of(1), 1);
        em_fetch(frame_of(5));
    }
    agent_sync();
    double cum = 0.0;     // C_t (forward) / D_t (backward): what the vector of step t lacks to its log2 value
    float Mprev = 0.f;    // maximum of the vector of the previous step (what this step subtracts)
    double zmin = __builtin_inf();
    {
        float ymax = MM_WAVE_NEG;
        if (DIR == 0) {  // alpha_hat (*) lhs[:,1]   (src/inference.jl:68)
            for (int i = sub * 64 + lane; i < S1; i += 64 * NWA) {
                const float ai = as_global(r.init)[i];
                float v = (ai > MM_WAVE_NEG ? ai : MM_WAVE_NEG) + ldsr(base + MM_WAVE_EM(1) + 4u * as_global(r.rowpdf)[i]);
                v = v > MM_WAVE_NEG ? v : MM_WAVE_NEG;
                ldsw(base + MM_WAVE_VEC(1) + 4u * i, v);
                rows[i] = v;  // frame 1
                ymax = fmaxf(ymax, v);
            }
            if (sub == 0 && lane == 0) offs[1] = 0.0;
        } else {  // B[:, len+1] = one at the final state   (src/inference.jl:104): y = b~ + lhs, the phony pdf emits one
            if (sub == 0 && lane == 0) ldsw(base + MM_WAVE_VEC(1) + 4u * r.fpos, 0.f);
            for (int i = sub * 64 + lane; i < S1; i += 64 * NWA) rows[(long long)(NF - 1) * S1p + i] = i == r.fpos ? 0.f : MM_WAVE_NEG;
            if (sub == 0 && lane == 0) offs[NF] = 0.0;
            ymax = sub == 0 ? 0.f : MM_WAVE_NEG;
        }
        ymax = wave_max_rl(ymax);
        if (lane == 0) ldsw(base + MM_WAVE_YM(1) + 4u * sub, ymax);
    }
    if (service) {
        em_stage(frame_of(2), 0);
        em_fetch(frame_of(6));
    }
    agent_sync();
    Mprev = vec_max(MM_WAVE_YM(1));

    // partner values of the rows this lane finishes, for the step after the one being computed (phase B)
    struct uq_t {
        float v[NSEG];
    };
    float pv[NSEG], pvn[NSEG];
    double po = 0.0, pon = 0.0;  // ... and the offset the partner stored with that frame
#pragma unroll
    for (int i = 0; i < NSEG; ++i) pv[i] = pvn[i] = MM_WAVE_NEG;
    auto partner_fetch = [&](int t) __attribute__((always_inline)) {  // (clamped: an always valid row)
        int f = frame_of(t);
        f = f < 1 ? 1 : (f > NF ? NF : f);
        const float *row = rows + (long long)(f - 1) * S1p;
        pon = offs[f];
        if (!service) {
#pragma unroll
            for (int i = 0; i < NSEG; ++i) pvn[i] = row[(s1[i] & 0xffffu) >> 2];
        }
    };
    // Posteriors of the frame of step ts (its per-pdf sums are complete: the agent's barrier of that step has been
    // passed), by ONE wave of the agent: gamma = sum over the waves of their sums, each on its own scale 2^(max u of the
    // wave), normalised by the frame's own sum (:155-158); log Z of the frame = log2(sum) + max u + the two offsets (:159).
    auto frame_out = [&](int ts) __attribute__((always_inline)) {
        const int f = frame_of(ts);
        const unsigned par = (unsigned)(ts & 3);
        const double own_off = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(base + MM_WAVE_OFF(par));
        const double partner_off = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(base + MM_WAVE_OFF(par) + 8u);
        float um[NWD], mq = MM_WAVE_NEG;
#pragma unroll
        for (int k = 0; k < NWD; ++k) {
            um[k] = ldsr(base + MM_WAVE_UM(par) + 4u * k);
            mq = fmaxf(mq, um[k]);
        }
        const bool alive = mq > 0.5f * MM_WAVE_NEG;  // (nothing alive: no accepting path, gamma = 0)
        float ps[NJ], tot = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            ps[j] = 0.f;
#pragma unroll
            for (int k = 0; k < NWD; ++k)  // (fixed order)
                ps[j] += ldsr(base + MM_WAVE_PS(par, k) + 4u * (unsigned)(lane + 64 * j)) * (alive ? fast_exp2(um[k] - mq) : 0.f);
            tot += (lane + 64 * j) < P1 ? ps[j] : 0.f;
        }
        tot = wave_sum_fixed(tot);
        const float inv = tot > 0.f ? 1.f / tot : 0.f;
        float *gp = p.gamma + gbase + (long long)(f - 1) * p.gsn;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            if (q < P) gp[q * p.gsp] = ps[j] * inv;
        }
        const double z = alive ? (double)fast_log2(tot) + (double)mq + own_off + partner_off : -__builtin_inf();
        zmin = z < zmin ? z : zmin;
    };
    // (compute waves) the frame's posterior terms of the step before, on the scale of the wave's maximum: added into the
    // wave's per-pdf sums at the start of the next step, BEHIND that step's gathers -- LDS float adds with conflicts take
    // hundreds of cycles, and the LDS serves a wave's operations in order: issued before the barrier they were the barrier
    uq_t uq;
    auto flush_q = [&](int ts) __attribute__((always_inline)) {  // the adds of step ts (> tA)
        const unsigned bufq = (unsigned)(ts & 3);
#pragma unroll
        for (int j = 0; j < NJ; ++j) ldsw(base + MM_WAVE_PS(bufq, sub) + 4u * (unsigned)(lane + 64 * j), 0.f);
        // one LDS float add per segment into the wave's OWN array: conflicting lanes of one instruction are served in the
        // hardware's fixed order and the instructions in program order -- the same sums on every run (the item kernel's
        // atomics race between waves)
#pragma unroll
        for (int i = 0; i < NSEG; ++i) {
            const unsigned at = base + MM_WAVE_PS(bufq, sub) + (s0[i] >> 16);
            asm volatile("ds_add_f32 %0, %1" ::"v"(at), "v"(uq.v[i]) : "memory");
        }
    };

    auto step = [&](auto RDc, auto PHc, int t) __attribute__((always_inline)) {
        constexpr int RD = decltype(RDc)::value, WR = 1 - RD, PHASE = decltype(PHc)::value;
        const int f = frame_of(t);
        float *rowf = rows + (long long)(f - 1) * S1p;
        float uu[NSEG];
        float ymax = MM_WAVE_NEG, umax = MM_WAVE_NEG;
        if constexpr (PHASE == 1) {
#pragma unroll
            for (int i = 0; i < NSEG; ++i) pv[i] = pvn[i];
            po = pon;
        }
        cum += (double)Mprev;
        if (service) {
            // (what was fetched during the previous step is consumed BEFORE anything new goes to memory: the compiler waits
            // for vmcnt(0) at the first use in a loop iteration, which would include stores issued just before)
            // (the frame of step t + 1 was requested at step t - 3: of what is in flight only the youngest 3 NJ operations
            // may be newer -- the other memory operations of this wave in between only make the wait stricter)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NJ) : "memory");
            em_stage(frame_of(t + 1), RD);  // emissions of step t + 1
            em_fetch(frame_of(t + 5));
            if constexpr (PHASE == 1) partner_fetch(t + 1);
            if (lane == 0) {
                ldsw(base + MM_WAVE_YM(WR) + 4u * sub, MM_WAVE_NEG);
                if constexpr (PHASE == 0) {
                    offs[f] = cum;
                } else {
                    *(__attribute__((address_space(3))) double *)(__UINTPTR_TYPE__)(base + MM_WAVE_OFF(t & 3)) = cum;
                    *(__attribute__((address_space(3))) double *)(__UINTPTR_TYPE__)(base + MM_WAVE_OFF(t & 3) + 8u) = po;
                }
            }
            if constexpr (PHASE == 1) {
                // the frame of the step before last: the compute waves added its sums during the last step
                if (t - 2 > tA) {
                    const int fp = frame_of(t - 2);
                    if (fp >= 1 && fp <= len) frame_out(t - 2);
                }
            }
            MM_STAMP(0);
            agent_sync();
            MM_STAMP(1);
            Mprev = vec_max(MM_WAVE_YM(WR));
            return;
        }
        if constexpr (PHASE == 1) partner_fetch(t + 1);
        // (1) every segment's lane-local log-sum-exp: ONE basic block -- all gathers and emission reads of the step are in
        // flight together; a branch per segment (fewer than NSEG segments, two arcs instead of four) made every segment
        // wait for its own LDS round trips, 8 x ~400 cycles per step.  Unused slots hold weight MM_WAVE_NEG: 2^(-1e30) = 0.
        float mxs[NSEG], sums[NSEG], ems[NSEG], xs[4 * NSEG];
#pragma unroll
        for (int i = 0; i < NSEG; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) xs[4 * i + q] = ldsr(a[4 * i + q] + MM_WAVE_VEC(RD));
            ems[i] = ldsr(base + MM_WAVE_EM(WR) + (s0[i] >> 16));
        }
        if constexpr (PHASE == 1)
            if (t - 1 > tA) flush_q(t - 1);  // (behind the gathers)
        static_for<0, NSEG>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const float x0 = xs[4 * i] + w[4 * i], x1 = xs[4 * i + 1] + w[4 * i + 1];
            const float x2 = xs[4 * i + 2] + w[4 * i + 2], x3 = xs[4 * i + 3] + w[4 * i + 3];
            const float mx = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
            mxs[i] = mx;
            sums[i] = (fast_exp2(x0 - mx) + fast_exp2(x1 - mx)) + (fast_exp2(x2 - mx) + fast_exp2(x3 - mx));
        });
        MM_STAMP(2);
        // (2) rows of more than 4 arcs (rare in the graphs of this kernel): lane-group maximum and sum
        if (lgw != 0ull) {
            static_for<0, NSEG>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                const int lg = (int)((lgw >> (4 * i)) & 15ull);
                if (lg) {
                    const float M = grp_max(mxs[i], lg);
                    sums[i] = grp_sum(sums[i] * fast_exp2(mxs[i] - M), lg);
                    mxs[i] = M;
                }
            });
        }
        // (3) finish the rows, again in one block
        static_for<0, NSEG>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const unsigned pos4 = s0[i] & 0xffffu;
            // forward: (T' alpha) (*) lhs (src/inference.jl:70-71); backward: T (B (*) lhs) (:106-107), the emission is
            // added for the next step's product only
            float bq = mxs[i] + fast_log2(sums[i]) - Mprev;
            float y = bq + ems[i];
            y = y > MM_WAVE_NEG ? y : MM_WAVE_NEG;
            bq = bq > MM_WAVE_NEG ? bq : MM_WAVE_NEG;
            ldsw(base + MM_WAVE_VEC(WR) + pos4, y);
            const bool mine = pos4 != trash4;
            ymax = fmaxf(ymax, mine ? y : MM_WAVE_NEG);
            const float st = DIR ? bq : y;  // the vector that is stored / combined
            if constexpr (PHASE == 0) {
                *reinterpret_cast<float *>(reinterpret_cast<char *>(rowf) + pos4) = st;
            } else {
                uu[i] = mine ? st + pv[i] : MM_WAVE_NEG;  // log2 (alpha beta) up to the two offsets   (:154)
                umax = fmaxf(umax, uu[i]);
            }
        });
        MM_STAMP(3);
        ymax = wave_max_rl(ymax);
        if (lane == 0) ldsw(base + MM_WAVE_YM(WR) + 4u * sub, ymax);
        MM_STAMP(4);
        if constexpr (PHASE == 1) {
            // this wave's share of the per-pdf sums of the frame, on the scale of its own maximum: q = 2^(u - max u of the
            // wave); added to the sums at the start of the next step (flush_q)
            umax = wave_max_rl(umax);
            if (lane == 0) ldsw(base + MM_WAVE_UM(t & 3) + 4u * sub, umax);
#pragma unroll
            for (int i = 0; i < NSEG; ++i) uq.v[i] = umax > 0.5f * MM_WAVE_NEG ? fast_exp2(uu[i] - umax) : 0.f;
        }
        MM_STAMP(0);
        agent_sync();
        MM_STAMP(1);
        Mprev = vec_max(MM_WAVE_YM(WR));
        MM_STAMP(5);
    };
    auto run = [&](auto PHc, int tfirst, int tlast) __attribute__((always_inline)) {
        for (int t = tfirst; t <= tlast; t += 2) {
            if (t & 1) step(std::integral_constant<int, 0>{}, PHc, t);
            else step(std::integral_constant<int, 1>{}, PHc, t);
            if (t + 1 <= tlast) {
                if ((t + 1) & 1) step(std::integral_constant<int, 0>{}, PHc, t + 1);
                else step(std::integral_constant<int, 1>{}, PHc, t + 1);
            }
        }
    };
    MM_STAMP_RESET;
    run(std::integral_constant<int, 0>{}, 2, tA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the ONE workgroup barrier: phase A of both agents is stored
    partner_fetch(tA + 1);
    run(std::integral_constant<int, 1>{}, tA + 1, NF);
    // the last two steps' frames (NF is not live for the forward agent: frame len + 1; frame 1 for the backward agent)
    if (NF > tA) {
        if (service) {
            const int fp = frame_of(NF - 1);
            if (NF - 1 > tA && fp >= 1 && fp <= len) frame_out(NF - 1);
        } else {
            flush_q(NF);
        }
        agent_sync();
        if (service) {
            const int fp = frame_of(NF);
            if (fp >= 1 && fp <= len) frame_out(NF);
        }
    }
#ifdef MM_STAMPS
    if (p.dbg && lane == 0)
        for (int q = 0; q < 8; ++q) p.dbg[((long long)b * 16 + wv) * 8 + q] = stamp_acc[q];
#endif
    // ---- ttl, zeros beyond the sequence length
    if (lane == 0 && service) *(__attribute__((address_space(3))) double *)(__UINTPTR_TYPE__)(base + MM_WAVE_ZZ) = zmin;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        double z = __builtin_inf();
        for (int d = 0; d < 2; ++d) {
            const double zk = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)((unsigned)d * MM_WAVE_SLICE + MM_WAVE_ZZ);
            z = zk < z ? zk : z;
        }
        p.ttl[b] = (z < __builtin_inf() && z > -__builtin_inf()) ? (float)(z * (double)MM_LN2) : MM_NINF;
    }
    for (long long q = threadIdx.x; q < (long long)(p.N - len) * P; q += NT)
        p.gamma[gbase + (len + q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
}

}  // namespace mm
