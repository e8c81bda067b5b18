// mm_kernel_lane.hip -- pdfposteriors (src/inference.jl:145-161) for TINY graphs: FSMs of up to 64 states (+ the phony final
// state) and up to 64 pdfs -- the 3-state HMM of the reference's demo and tests (BASELINE config 1), dense ergodic HMMs
// (config 2: 64 states, every arc present).
//
// Such a batch is a chain of dependent frames and nothing else: config 2 spent 0.60 ms on the pair kernels -- 16-wave
// workgroups, a barrier, a service wave and LDS-DMA rings per step for a product of 64 x 65 numbers.  Here ONE WAVE is an
// agent: lane s holds state s, the 64 weights of its row of the product sit in its registers as doubles, and a step is
//     acc_s = sum_i w_s[i] * p_i        p_i broadcast from lane i by v_readlane_b32 (x 2: a double) into v_fma_f64's scalar operand
// -- no LDS, no barrier, no other wave on the path.  A workgroup is two waves, the forward and the backward agent of one
// utterance (the bidirectional time split of the pair / wave kernels: each stores its normalised log2 vectors for its half
// of the frames, ONE workgroup barrier, then each continues through the other half and combines).
//
// Numerics: linear domain in FLOAT64, renormalised every step by the exponent of the frame's maximum (an exact power of two;
// the cumulative offsets are integers + the frames' emission maxima, kept in double): 1022 log2 of range below every frame's
// maximum, no range marks, no second pass -- the float32 linear kernels' flag-and-redo does not exist here.  The phony final
// state (src/fsm.jl:19-28) is not a lane: before frame len + 1 it holds zero(K), so it enters as two boundary conditions --
// the backward recursion starts from beta_len(i) = omega_i, and log Z = the per-frame normaliser like everywhere else.
#pragma once
#include "mm_kernel_dpair.hip"

namespace mm {

struct LaneDev {          // one FSM of up to 64 states for the lane kernel (device memory, built at mm_batch_create)
    const double *w[2];   // [dir][k * 64 + lane]: forward, lane j: the weight of the arc k -> j; backward, lane i: of the arc i -> k (linear; 0: none)
    const float *init;    // [64] log2 alpha_hat(s)  (-inf: not initial)
    const float *fin;     // [64] log2 omega(s): the arc s -> final  (-inf: none)
    const int *s2p;       // [64] pdf of state s (0 beyond S)
    const int *pdf_ptr;   // [P + 1] CSR pdf -> states
    const int *pdf_states;
    int S, P, ident, pad;  // ident: pdf p is state p's and only its (the identity map: no sums over states)
};

#define MM_LANE_D 8  // frames the emissions (and, in phase B, the partner rows) are requested ahead

// The product of a step: acc = sum_i w[i] * p_i, p_i = lane i's value.  The vector goes through LDS (the agent wrote it at the
// end of the last step: one ds_write_b64 per lane): every lane reads ALL values with ds_read_b128 of the same address in every lane (a broadcast: one pass of the
// LDS, two values per instruction) -- 32 reads + 64 v_fma_f64 for 64 states.  (v_readlane_b32 into the FMA's scalar operand
// needs two readlanes per value: 192 instructions, 1430 cycles of a 2400-cycle step, cycle stamps.)  LDS operations of one
// wave execute in order: no barrier between the write and the reads, nor before the next step's write.
typedef double mm_f64x2 __attribute__((ext_vector_type(2)));
template <int NS>
__device__ __forceinline__ double lane_product(const double (&w)[NS], unsigned vec_addr) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NS; i += 2) {
        const mm_f64x2 s = *(__attribute__((address_space(3))) const mm_f64x2 *)(__UINTPTR_TYPE__)(vec_addr + 8u * (unsigned)i);
        acc[(i >> 1) & 3] = __builtin_fma(w[i], s.x, acc[(i >> 1) & 3]);
        acc[((i >> 1) + 2) & 3] = __builtin_fma(w[i + 1], s.y, acc[((i >> 1) + 2) & 3]);
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// the exponent of the wave's largest value (lanes beyond S hold 0; 0 if everything is 0): the NEXT step divides by 2^k -- the
// maximum of a step is found while the next step's product runs, never on the path from one vector to the next
__device__ __forceinline__ int lane_exponent(double x) {
    const unsigned hi = wave_max_u32((unsigned)(__builtin_bit_cast(unsigned long long, x) >> 32));
    const int ex = (int)((hi >> 20) & 0x7ffu);
    return ex == 0 ? 0 : ex - 1023;
}

// volatile LDS word (the waves of a workgroup hand steps to each other through counters: LDS operations of one wave are
// carried out in order, so a counter written after a vector is seen after it)
__device__ __forceinline__ int lds_peek(const int *q) { return __builtin_amdgcn_readfirstlane(*(const volatile int *)q); }
__device__ __forceinline__ void lds_post(int *q, int v) {
    if ((threadIdx.x & 63) == 0) *(volatile int *)q = v;
}

// A workgroup = one utterance, four waves: the forward AGENT, the backward AGENT, and a SERVICE wave for each.  An agent's
// step is the product, the emissions and the power of two; its service wave, behind it, stores the vector of a step (phase A)
// or combines it with the partner's stored row into the frame's posteriors (phase B), from the copy of the vector the agent
// leaves in LDS for its own next product anyway.  (One wave doing all of it: 1660 / 2580 cycles per step of which 740 / 1020
// the product.  A service wave that also staged the emissions: 1800 per step, that wave the longest.  Two service waves per
// agent: six waves share four SIMDs, the 256 registers left to a wave do not hold the 64 weights and the reads in flight,
// and the spills cost more than the split saved -- cycle stamps, tools/stamps_lane.py.)
template <int NS>
__global__ void __launch_bounds__(256) mm_lane_kernel(RunParams p) {
    constexpr int D = MM_LANE_D;
    __shared__ float em_ring[2][D][64];   // raw emissions of a frame, per state (LDS-DMA gather)
    __shared__ float pr_ring[2][D][64];   // the partner's stored row of a frame (phase B)
    __shared__ double pr_off[2][D][32];   // ... and its offset (a DMA writes 4 bytes for every lane: a slot is 256 bytes, the double its first 8)
    __shared__ double qbuf[2][64];        // a .* b per state, for the sums over the states of a pdf
    __shared__ double zstat[2][3];        // per agent: min, max over its frames of the per-frame log2 normaliser; min of log2 sum_s 2^(a~ + b~)
    __shared__ int pdfp[66], pdfs[64];    // pdf -> states (CSR), for maps that are not the identity
    __shared__ mm_f64x2 pvec[2][2][32];   // the agent's vector of a step (by the step's parity): what its next product reads
    __shared__ double bvec[2][64];        // backward agent: beta~ of a step without the frame's emission (what is stored / combined)
    __shared__ double cumv[2][2];         // ... the offset of that vector
    __shared__ int a_done[2], s_done[2], s_stored[2];  // counters: steps the agent published / its service wave consumed; phase A stored
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int DIR = wv & 1;
    const bool svcF = wv >= 2;  // (waves 0, 1: the agents; 2, 3: their service waves)
    const int b = uni(p.order ? p.order[blockIdx.x] : (int)blockIdx.x);
    const UttDesc &u = p.utts[b];
    const LaneDev *ldp = uni(u.lane);
    const int S = uni(ldp->S), P = uni(ldp->P), ident = uni(ldp->ident), S1p = uni(u.S1p);
    int len = uni(p.lens ? p.lens[b] : p.N);
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const float *Vb = p.V + (long long)b * p.vsb;
    const long long gbase = (long long)b * p.gsb;
    // beyond the sequence: exact zeros (src/inference.jl:54-60: expand() leaves the real pdfs zero(K) there)
    for (long long q = threadIdx.x; q < (long long)(p.N - len) * P; q += 256) p.gamma[gbase + (len + q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
    if (len == 0) {  // no frame: no path of length 0
        if (threadIdx.x == 0) {
            p.ttl[b] = MM_NINF;
            if (p.redo) p.redo[b] = 0;
        }
        return;
    }
    const long long s1p_prefix = ((long long)uni((int)(u.s1p_prefix >> 32)) << 32) | (unsigned)uni((int)u.s1p_prefix);
    float *rows = p.ws_alpha + s1p_prefix * (long long)(p.N + 1);  // [frame - 1][S1p]: a~ up to the split, b~ beyond
    double *offs = p.ws_c + (long long)b * (p.N + 2);               // [frame]: the offset of the stored vector
    const bool real = lane < S;
    if (threadIdx.x < 2) {
        a_done[threadIdx.x] = 0;
        s_done[threadIdx.x] = 0;
        s_stored[threadIdx.x] = 0;
    }
    if (!ident && wv == 0) {
        if (lane <= P) pdfp[lane] = uni(ldp->pdf_ptr)[lane];
        if (lane < S) pdfs[lane] = uni(ldp->pdf_states)[lane];
    }
    // the frames: the forward agent walks 1, 2, ..., the backward agent len, len - 1, ...; each stores its first nA frames
    const int m = len / 2, nA = DIR ? len - m : m;
    auto frame_of = [&](int t) { return DIR ? len + 1 - t : t; };
    MM_STAMP_DECL;
    __syncthreads();

    if (!svcF) {
        // ================= agent =================
        double w[NS];
        {
            const double *wp = uni(ldp->w[DIR]);
#pragma unroll
            for (int k = 0; k < NS; ++k) w[k] = wp[k * 64 + lane];
        }
        const float start_v = real ? uni(DIR ? ldp->fin : ldp->init)[lane] : MM_NINF;
        const unsigned vec0 = lds_addr_of(reinterpret_cast<const float *>(&pvec[DIR][0][0]));
        // the emissions: requested D steps ahead by LDS-DMA (the only VMEM operation of this wave: one per step), turned into
        // 2^(e - E) -- E the frame's maximum over the states -- ONE step ahead, in registers, while the product's reads are on
        // their way
        const int mypdf = real ? uni(ldp->s2p)[lane] : 0;
        const float *vlane = Vb + mypdf;  // this state's emission of frame f: vlane[(f - 1) * vsn]
        const unsigned em_base = lds_addr_of(&em_ring[DIR][0][0]);
        auto dma_em = [&](int t) {  // raw emissions of step t (clamped) -> slot t % D
            const int tt = t < 1 ? 1 : (t > len ? len : t);
            dma_b32(vlane + (long long)(frame_of(tt) - 1) * p.vsn, em_base + 256u * (unsigned)(t & (D - 1)));
        };
        auto emission = [&](float raw, float *Eout) {
            const float e2 = real ? raw * MM_LOG2E : MM_NINF;
            float E = wave_max_rl(e2);
            if (!(E > MM_NINF)) E = 0.f;
            *Eout = E;
            // (dexp2: the fraction through v_exp_f32, the integer part through v_ldexp_f64 -- a float 2^(e - E) is 0 below 2^-149 of the
            // frame's best pdf, and a left-to-right graph's only path may go through such a state: the double's 1022 log2 are the
            // kernel's contract)
            return dexp2(e2 - E);  // (<= 1; 0 beyond S and for zero(K))
        };
        for (int t = 1; t <= D; ++t) dma_em(t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the counted wait below holds from then on)
        // (the address of the next request walks by one frame per step: no 64-bit multiply in the loop)
        const long long em_stride = DIR ? -(long long)p.vsn : (long long)p.vsn;
        const float *em_ptr = vlane + (long long)(frame_of(D + 1 > len ? len : D + 1) - 1) * p.vsn;
        float E_next;
        double em_next = emission(em_ring[DIR][1][lane], &E_next);
        double cum = 0.0, pv = 0.0;
        MM_STAMP_RESET;
        for (int t = 1; t <= len; ++t) {
            // Everything the step reads from LDS is REQUESTED first -- the counter, the raw emission of the next step, the broadcast
            // reads of the last vector -- and the exponent of the last vector's maximum and the next step's emissions (chains of DPP
            // steps and readlanes on registers) are found while those are on their way.
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 2) : "memory");  // (the request of step t + 1 was made at step t + 1 - D)
            const int done = *(const volatile int *)&s_done[DIR];
            const float raw1 = em_ring[DIR][(t + 1) & (D - 1)][lane];
            const double em = em_next;
            const float E = E_next;
            // (the reads in two batches: with all 32 in flight -- 128 registers -- next to the 128 of the weights the wave spilled)
            constexpr int HB = NS / 2 > 16 ? 16 : NS / 2;  // reads of a batch
            mm_f64x2 v[HB];
            const unsigned va = vec0 + 512u * (unsigned)((t - 1) & 1);
#pragma unroll
            for (int i = 0; i < HB; ++i) v[i] = *(__attribute__((address_space(3))) const mm_f64x2 *)(__UINTPTR_TYPE__)(va + 16u * (unsigned)i);
            const int K = t == 1 ? 0 : lane_exponent(pv);  // the exponent this step divides by: of the last vector's maximum
            em_next = emission(raw1, &E_next);
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                acc[i & 3] = __builtin_fma(w[2 * i], v[i].x, acc[i & 3]);
                acc[(i + 2) & 3] = __builtin_fma(w[2 * i + 1], v[i].y, acc[(i + 2) & 3]);
            }
            if constexpr (NS / 2 > HB) {
                __builtin_amdgcn_sched_barrier(0);  // (the second batch's reads stay behind the first batch's products)
#pragma unroll
                for (int i = 0; i < HB; ++i) v[i] = *(__attribute__((address_space(3))) const mm_f64x2 *)(__UINTPTR_TYPE__)(va + 16u * (unsigned)(HB + i));
#pragma unroll
                for (int i = 0; i < HB; ++i) {
                    acc[i & 3] = __builtin_fma(w[2 * (HB + i)], v[i].x, acc[i & 3]);
                    acc[(i + 2) & 3] = __builtin_fma(w[2 * (HB + i) + 1], v[i].y, acc[(i + 2) & 3]);
                }
            }
            // forward: alpha_hat (*) lhs[:,1] (src/inference.jl:68) / (T' alpha_{t-1}) (*) lhs[:,t] (:70-71);
            // backward: B[:, len + 1] = one at the final state (:104): beta_len(i) = omega_i, else T (B[:,t+1] (*) lhs[:,t+1]) (:106-107)
            const double x = t == 1 ? dexp2(start_v) : (acc[0] + acc[1]) + (acc[2] + acc[3]);
            MM_STAMP(0);
            double keep;  // the vector that is stored / combined
            if (DIR == 0) {
                pv = __builtin_amdgcn_ldexp(x * em, -K);
                cum += (double)E + (double)K;
                keep = pv;
            } else {
                keep = __builtin_amdgcn_ldexp(x, -K);  // (beta~: without the frame's emission)
                cum += (double)K;
                pv = keep * em;  // what the next step's product reads
            }
            if (__builtin_expect(__builtin_amdgcn_readfirstlane(done) < t - 2, 0))  // (the slot of step t - 2 must have been consumed)
                while (lds_peek(&s_done[DIR]) < t - 2) __builtin_amdgcn_s_sleep(1);
            MM_STAMP(1);
            ldsw_d(vec0 + 512u * (unsigned)(t & 1) + 8u * (unsigned)lane, pv);
            if (DIR == 1) bvec[t & 1][lane] = keep;
            if (lane == 0) {  // (one lane, one exec-mask change: the offset, then the counter -- LDS writes of a wave land in order)
                cumv[DIR][t & 1] = cum;
                *(volatile int *)&a_done[DIR] = t;
            }
            if (DIR == 1) cum += (double)E;
            // the request of step t + D (frames beyond the agent's last: its last frame again, never used)
            dma_b32(em_ptr, em_base + 256u * (unsigned)((t + D) & (D - 1)));
            if (t + D < len) em_ptr += em_stride;
            MM_STAMP(2);
        }
#ifdef MM_STAMPS
        if (p.dbg && lane == 0)
            for (int k = 0; k < 8; ++k) p.dbg[((long long)blockIdx.x * 4 + wv) * 8 + k] = stamp_acc[k];
#endif
    } else {
        // ================= service wave F: behind the agent -- stores its vectors (phase A), the posteriors (phase B) =================
        const unsigned pr_base = lds_addr_of(&pr_ring[DIR][0][0]);
        const unsigned po_base = lds_addr_of(reinterpret_cast<const float *>(&pr_off[DIR][0][0]));
        auto dma_partner = [&](int t) {  // the other agent's row and offset of step t's frame (phase B)
            const int tt = t < 1 ? 1 : (t > len ? len : t);
            const int f = frame_of(tt);
            dma_b32(rows + (long long)(f - 1) * S1p + (real ? lane : 0), pr_base + 256u * (unsigned)(t & (D - 1)));
            dma_b32(reinterpret_cast<const unsigned *>(offs + f) + (lane & 1), po_base + 256u * (unsigned)(t & (D - 1)));
        };
        double zmin = __builtin_inf(), zmax = -__builtin_inf();
        float ltmin = __builtin_inff();  // smallest log2 of a frame's sum of 2^(a~ + b~): how far below their maxima the two masses overlap
        // posteriors of frame f from the agent's vector v and the partner's stored row (slot of step t)
        auto combine = [&](int t, int f, double v, double own_off) {
            const float pr = pr_ring[DIR][t & (D - 1)][lane];
            const double q = real ? v * dexp2(pr) : 0.0;  // A .* B   (src/inference.jl:154)
            double g = q;
            if (!ident) {  // C' * (A .* B)   (:155): lane p sums the states of pdf p
                qbuf[DIR][lane] = q;
                g = 0.0;
                if (lane < P)
                    for (int k = pdfp[lane]; k < pdfp[lane + 1]; ++k) g += qbuf[DIR][pdfs[k]];
            } else if (lane >= P) {
                g = 0.0;
            }
            const double tot = dwave_sum_rl(g);  // the frame's normaliser   (:157)
            const int e = __builtin_amdgcn_frexp_exp(tot);
            const float tf = (float)__builtin_amdgcn_ldexp(tot, -e);
            const float inv = tf > 0.f ? 1.f / tf : 0.f;
            if (lane < P) p.gamma[gbase + (long long)(f - 1) * p.gsn + (long long)lane * p.gsp] = (float)__builtin_amdgcn_ldexp(g, -e) * inv;  // (:158, :160)
            const float lt = dlog2(tot);
            ltmin = lt < ltmin ? lt : ltmin;
            const double z = (double)lt + own_off + pr_off[DIR][t & (D - 1)][0];
            zmin = z < zmin ? z : zmin;
            zmax = z > zmax ? z : zmax;
        };
        auto run_phase = [&](auto PHc, int ta, int tb) {  // steps ta .. tb of this agent
            constexpr int PHASE = decltype(PHc)::value;
            for (int t = ta; t <= tb; ++t) {
                const int f = frame_of(t);
                // (phase B) the partner's row and offset of step t were requested at step t - D; per step the posteriors' store and
                // the two requests follow, in that order
                if constexpr (PHASE == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (D - 1)) : "memory");
                MM_STAMP(0);
                while (lds_peek(&a_done[DIR]) < t) __builtin_amdgcn_s_sleep(1);  // the agent has published step t
                MM_STAMP(1);
                const double v = DIR ? bvec[t & 1][lane] : ldsr_d(lds_addr_of(reinterpret_cast<const float *>(&pvec[DIR][t & 1][0])) + 8u * (unsigned)lane);
                const double cum = cumv[DIR][t & 1];
                if constexpr (PHASE == 0) {
                    if (real) rows[(long long)(f - 1) * S1p + lane] = dlog2(v);
                    if (lane == 0) offs[f] = cum;
                    lds_post(&s_done[DIR], t);
                } else {
                    // (the vector is in registers: the agent may have its slot back before the posteriors are out)
                    double vr = v, cr = cum;
                    asm volatile("" : "+v"(vr), "+v"(cr)::"memory");
                    lds_post(&s_done[DIR], t);
                    combine(t, f, vr, cr);
                    dma_partner(t + D);
                }
                MM_STAMP(2);
            }
        };
        MM_STAMP_RESET;
        run_phase(std::integral_constant<int, 0>{}, 1, nA);
        // both service waves have stored their halves before either reads the other's
        __threadfence();
        lds_post(&s_stored[DIR], 1);
        while (lds_peek(&s_stored[1 - DIR]) == 0) __builtin_amdgcn_s_sleep(1);
        for (int t = nA + 1; t <= nA + D; ++t) dma_partner(t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        run_phase(std::integral_constant<int, 1>{}, nA + 1, len);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            zstat[DIR][0] = zmin;
            zstat[DIR][1] = zmax;
            zstat[DIR][2] = (double)ltmin;
        }
#ifdef MM_STAMPS
        if (p.dbg && lane == 0)
            for (int k = 0; k < 8; ++k) p.dbg[((long long)blockIdx.x * 4 + wv) * 8 + k] = stamp_acc[k];
#endif
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // ttl = min over the frames of the per-frame log-normaliser (src/inference.jl:159)
        const double z = zstat[0][0] < zstat[1][0] ? zstat[0][0] : zstat[1][0];
        p.ttl[b] = (z < __builtin_inf()) ? (float)(z * (double)MM_LN2) : MM_NINF;
        // The double's range is wide, not unbounded: a value more than ~1000 log2 below its frame's maximum is flushed, and on a
        // left-to-right graph with very sharp emissions the ONE path may run through such values (64 states in 64 frames,
        // emissions 250 nats apart).  The same two tests as mm_dpair_finish_kernel decide whether anything that matters can have
        // been lost -- the frames' normalisers agree, and the forward and the backward mass overlap within the double's range less
        // the posterior floor --; otherwise redo[b] hands the utterance to the log-domain kernel behind this launch.
        if (p.redo) {
            const double zM = zstat[0][1] > zstat[1][1] ? zstat[0][1] : zstat[1][1];
            const double lm = zstat[0][2] < zstat[1][2] ? zstat[0][2] : zstat[1][2];
            const bool agree = z > -__builtin_inf() && zM < __builtin_inf() && zM - z <= MM_Z_SPREAD_TOL;
            p.redo[b] = (len >= 1 && !(agree && lm >= (double)p.lt_floor - (double)MM_DPAIR_THR_EXTRA)) ? 1 : 0;
        }
    }
}

}  // namespace mm
