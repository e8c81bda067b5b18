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
//   * the state vector of a step lives in the agent's LDS slice, double buffered by the parity of the step;
//   * three SERVICE waves per agent do everything that is not on the path from one barrier to the next: wave E the emissions
//     (LDS-DMA four steps ahead, staged relative to the frame's level E_t) and the offsets; wave P the frame normaliser; wave
//     F the partner's offsets (LDS-DMA) and the posteriors of a frame two steps after the compute waves produced their terms.
//     Normalisation: the vector of step t is the log2 vector minus C_t = sum_{k<=t} (M_{k-2} + E_k), M_k = the maximum of
//     the vector of step k, which wave P finds by scanning that vector during step k + 1 -- two steps of lag, so that no wave
//     waits for a maximum
//     (C in double); the semiring's zero is the finite sentinel MM_WAVE_NEG (-1e30: -inf - -inf never occurs);
//   * every reduction has a fixed order, and all of it stays in the log domain: the rows of phase B write
//     u = log2(alpha~ beta~) to their pdf-major positions (plain LDS stores), ONE lane per pdf forms the log-sum-exp over
//     the pdf's states during the next step, the service wave normalises over the pdfs a step later.  No atomics: the
//     kernel is deterministic (round 3's first version added the rows' terms with LDS float atomics: lanes that share a pdf
//     serialise, and the LDS serves one wave's conflicts before anybody else's gathers -- phase B took 3x phase A).
#pragma once
#include "mm_kernel_rows.hip"

namespace mm {

#define MM_WAVE_STRIDE 4         // arc slots of a segment
#define MM_WAVE_NEG (-1.0e30f)   // zero(K) in the log2 domain
#define MM_WAVE_VSZ 4352u        // bytes of one state vector (1024 states + the "no row" position, padded)
#define MM_WAVE_ESZ 1056u        // bytes of one emission / per-pdf buffer (256 pdfs + the "no row" slot)
#define MM_WAVE_NWD 4            // compute waves per agent
// LDS slice of one agent (bytes, relative to the slice)
#define MM_WAVE_VEC(par) ((unsigned)(par) * MM_WAVE_VSZ)
#define MM_WAVE_QV(par) (2u * MM_WAVE_VSZ + (unsigned)(par) * MM_WAVE_VSZ)                     // u of a step, pdf-major
#define MM_WAVE_EM(par) (4u * MM_WAVE_VSZ + (unsigned)(par) * MM_WAVE_ESZ)
#define MM_WAVE_PL(par) (4u * MM_WAVE_VSZ + 2u * MM_WAVE_ESZ + (unsigned)(par) * MM_WAVE_ESZ)  // per-pdf log2 sums of a step
#define MM_WAVE_FILL (4u * MM_WAVE_VSZ + 4u * MM_WAVE_ESZ)                                     // (everything below starts as zero(K))
#define MM_WAVE_MS(par) (MM_WAVE_FILL + (unsigned)(par) * 16u)                                 // maximum of the vector of step t, t & 1 == par
#define MM_WAVE_OFF(buf) (MM_WAVE_MS(2) + (unsigned)(buf) * 8u)                                // own offset of a step (double), t & 3
#define MM_WAVE_SYNC MM_WAVE_OFF(4)
#define MM_WAVE_ZZ (MM_WAVE_SYNC + 16u)   // the agent's minimum of the per-frame log2 normalisers (double)
#define MM_WAVE_RAW(k) (MM_WAVE_ZZ + 16u + (unsigned)(k) * 1024u)   // raw emissions of 4 frames in flight (LDS-DMA)
#define MM_WAVE_POFF(k) (MM_WAVE_RAW(4) + (unsigned)(k) * 256u)     // the partner's offsets of 8 steps in flight (LDS-DMA)
#define MM_WAVE_SLICE MM_WAVE_POFF(8)

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

// (TWO, NSEG = 2 only: 8 waves per SIMD -- at most 64 VGPRs and 96 SGPRs -- so that two workgroups of 14 waves fit a compute unit
// (4 + 4 + 3 + 3 waves per SIMD each): with the 106 SGPRs the compiler took by itself the second workgroup of a compute
// unit waited for the first, whatever hipOccupancyMaxActiveBlocksPerMultiprocessor says, and a batch of 512 utterances
// took twice the time of 256 -- tools/dev/resident_test.hip)
// TWO: the instance for batches of more utterances than compute units; the 41 spilled SGPRs cost a batch that has a
// compute unit per utterance 6 % (the WSJ numerators, B = 128: 0.456 against 0.432 ms), two workgroups per compute unit
// give B = 512 0.45 instead of 0.58 ms.
template <int NSEG, int NJ, bool TWO = false>  // NSEG: segments the registers of a wave hold; NJ * 64 >= P + 1
__global__ void __launch_bounds__(128 * (MM_WAVE_NWD + 3)) __attribute__((amdgpu_waves_per_eu(TWO ? 8 : 4, TWO ? 8 : (NSEG <= 2 ? 7 : 4))))
mm_wave_kernel(RunParams p) {
    extern __shared__ float lds[];
    constexpr int KA = MM_WAVE_STRIDE * NSEG, NWD = MM_WAVE_NWD, NWA = NWD + 3, NT = 128 * NWA;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // 0: forward agent, 1: backward agent; the wave of the agent: NWD compute waves and three SERVICE waves -- E stages the
    // emissions and keeps the offsets, P finds the frame normalisers, F (phase B) puts out the posteriors of a frame two
    // steps after the compute waves produced their terms: none of that is on the compute waves' path from one barrier to
    // the next (one service wave for all of it was the longest wave of every step: ~200 instructions of a single wave,
    // 1600 cycles against the compute waves' 900; scan + posteriors in one wave were the longest of phase B)
    const int DIR = wv / NWA, sub = wv % NWA;
    const bool svcE = sub == NWD, svcP = sub == NWD + 1, svcF = sub == NWD + 2, service = svcE || svcP || svcF;
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
        if (!p.g_acc)
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
    int lgmax = 0;  // largest log2 of the lanes of a row group in this wave's segments
    for (int i = 0; i < 16; ++i) {
        const int lgi = (int)((lgw >> (4 * i)) & 15ull);
        lgmax = lgi > lgmax ? lgi : lgmax;
    }
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
    for (unsigned q = 4u * (unsigned)(sub * 64 + lane); q < MM_WAVE_FILL; q += 256u * NWA) ldsw(base + q, MM_WAVE_NEG);
    if (sub == 0 && lane < 8) ldswu(base + MM_WAVE_MS(0) + 4u * lane, 0u);    // (the normalisers of steps 2 and 3 start from 0)
    if (sub == 1 && lane < 4) ldswu(base + MM_WAVE_SYNC + 4u * lane, 0u);
    // (compute waves) the per-pdf sums of phase B as NPS packed segments of this wave (mm_engine.hip wave_pdf_table):
    // 4 source addresses in the vector u per lane and segment, the pdf the lane's group stores, log2 of the group's lanes
    constexpr int NPS = NSEG / 2;
    unsigned aq[4 * NPS], iq[NPS];
    int qlvmax[NPS];
    {
        const auto tp = as_global(uni(u.rw[DIR].ptab));
#pragma unroll
        for (int j = 0; j < NPS; ++j) {
            const int row = ((service ? 0 : sub) * 2 + j) * 5;
#pragma unroll
            for (int q = 0; q < 4; ++q) aq[4 * j + q] = tp[(row + q) * 64 + lane] + base;
            iq[j] = tp[(row + 4) * 64 + lane];
            qlvmax[j] = __builtin_amdgcn_readfirstlane((int)wave_max_rl((float)(iq[j] >> 16)));
        }
    }
    __syncthreads();
    // the agent's own barrier: every wave adds 1 to the counter when its LDS writes of the step are done and waits for
    // all of them (LDS operations of a wave complete in order)
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

    // steps: step t handles frame t (forward) or NF + 1 - t (backward); steps 1 .. tA are phase A
    const int m = NF / 2;  // (len >= 1: NF >= 2, both agents have at least one step of phase A)
    const int tA = DIR ? NF - m : m;
    auto frame_of = [&](int t) __attribute__((always_inline)) { return DIR ? NF + 1 - t : t; };
    // ---- service wave: emissions of frame f into EM(par): expand() (src/inference.jl:54-60) in the log2 domain relative to
    // the frame's maximum E (returned; zero(K) = MM_WAVE_NEG).  Raw values by LDS-DMA FOUR steps ahead -- a step is ~0.5 us,
    // a load from HBM up to 2: fetched one step ahead, the agent waited for the service wave's load every step
    auto em_fetch = [&](int f) __attribute__((always_inline)) { row_dma_em<NJ>(base + MM_WAVE_RAW(f & 3), Vb, p.vsn, f, p.N, P, lane); };
    auto em_stage = [&](int f, int par) __attribute__((always_inline)) {
        float v[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            v[j] = em_value(ldsr(base + MM_WAVE_RAW(f & 3) + 256u * j + 4u * lane), f, len, P, q);
            v[j] = v[j] > MM_WAVE_NEG ? v[j] : MM_WAVE_NEG;  // (also NaN -> zero(K))
        }
        // E: the level of the frame's emissions.  Any finite number serves (it is accounted in the offsets): the first
        // pdf's value stands for the frame -- what matters is that a common shift of a frame's scores (GMM-style values
        // around -300) stays out of the vectors, a wave-wide maximum is ~100 cycles of this wave
        float E = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v[0])));
        if (!(E > 0.5f * MM_WAVE_NEG)) E = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            if (q <= P) ldsw(base + MM_WAVE_EM(par) + 4u * q, v[j] > 0.5f * MM_WAVE_NEG ? v[j] - E : MM_WAVE_NEG);
        }
        return E;
    };
    // ... the offset the partner stored with the frame of step t (a double: lanes 0 and 1 fetch its halves) -> POFF(t & 7)
    auto poff_fetch = [&](int t) __attribute__((always_inline)) {
        int f = frame_of(t);
        f = f < 1 ? 1 : (f > NF ? NF : f);
        dma_b32(reinterpret_cast<const unsigned *>(offs + f) + (lane & 1), base + MM_WAVE_POFF(t & 7));
    };
    // ... the maximum of the vector in VEC(par) (complete: the barrier of its step has been passed): 1024 states = 4 float4 per
    // lane, all loads issued before the first maximum; clamped indices (a duplicate changes no maximum).  The positions from
    // S1 to the end of the last float4 hold zero(K): the trash position only ever receives zero(K) (the lanes that write it
    // have weights of zero(K) and the emission slot of zero(K)), the ones behind it are never written.
    auto scan_max = [&](int par) __attribute__((always_inline)) {
        const int n4 = (S1 + 3) >> 2;
        const unsigned vb = base + MM_WAVE_VEC(par);
        mm_f32x4 v0 = *(__attribute__((address_space(3))) const mm_f32x4 *)(__UINTPTR_TYPE__)(vb + 16u * (unsigned)(lane < n4 ? lane : n4 - 1));
        mm_f32x4 v1 = *(__attribute__((address_space(3))) const mm_f32x4 *)(__UINTPTR_TYPE__)(vb + 16u * (unsigned)(lane + 64 < n4 ? lane + 64 : n4 - 1));
        float M = fmaxf(fmaxf(fmaxf(v0.x, v0.y), fmaxf(v0.z, v0.w)), fmaxf(fmaxf(v1.x, v1.y), fmaxf(v1.z, v1.w)));
        if (n4 > 128) {
            v0 = *(__attribute__((address_space(3))) const mm_f32x4 *)(__UINTPTR_TYPE__)(vb + 16u * (unsigned)(lane + 128 < n4 ? lane + 128 : n4 - 1));
            v1 = *(__attribute__((address_space(3))) const mm_f32x4 *)(__UINTPTR_TYPE__)(vb + 16u * (unsigned)(lane + 192 < n4 ? lane + 192 : n4 - 1));
            M = fmaxf(M, fmaxf(fmaxf(fmaxf(v0.x, v0.y), fmaxf(v0.z, v0.w)), fmaxf(fmaxf(v1.x, v1.y), fmaxf(v1.z, v1.w))));
        }
        M = wave_max_rl(M);
        return M > 0.5f * MM_WAVE_NEG ? M : 0.f;
    };
    double cum = 0.0;     // C_t (forward) / D_t (backward): what the vector of step t lacks to its log2 value
    float Ecur = 0.f, Enext = 0.f;  // (service wave E) emission levels of this step's and the next step's frame
    double zmin = __builtin_inf();
    // ---- step 1: the initial vector (the emissions are staged by the agent's service wave)
    if (svcE) {
        for (int t = 1; t <= 4; ++t) em_fetch(frame_of(t));
        MM_ROW_VMCNT(0);
        Ecur = em_stage(frame_of(1), 1);
        em_fetch(frame_of(5));
        cum = (double)Ecur;
        if (lane == 0) offs[frame_of(1)] = DIR ? 0.0 : cum;  // (backward: the stored vector is without the frame's emission)
    }
    agent_sync();
    if (!service) {
        if (DIR == 0) {  // alpha_hat (*) lhs[:,1]   (src/inference.jl:68)
            for (int i = sub * 64 + lane; i < S1; i += 64 * NWD) {
                const float ai = as_global(r.init)[i];
                float v = (ai > MM_WAVE_NEG ? ai : MM_WAVE_NEG) + ldsr(base + MM_WAVE_EM(1) + 4u * as_global(r.rowpdf)[i]);
                v = v > MM_WAVE_NEG ? v : MM_WAVE_NEG;
                ldsw(base + MM_WAVE_VEC(1) + 4u * i, v);
                rows[i] = v;  // frame 1
            }
        } else {  // B[:, len+1] = one at the final state   (src/inference.jl:104): y = b~ + lhs, the phony pdf emits one
            if (sub == 0 && lane == 0) ldsw(base + MM_WAVE_VEC(1) + 4u * r.fpos, 0.f);
            for (int i = sub * 64 + lane; i < S1; i += 64 * NWD) rows[(long long)(NF - 1) * S1p + i] = i == r.fpos ? 0.f : MM_WAVE_NEG;
        }
    } else if (svcE) {
        Enext = em_stage(frame_of(2), 0);
        em_fetch(frame_of(6));
    }
    agent_sync();

    // partner values of the rows this lane finishes, for the step after the one being computed (phase B)
    float pv[NSEG], pvn[NSEG];
#pragma unroll
    for (int i = 0; i < NSEG; ++i) pv[i] = pvn[i] = MM_WAVE_NEG;
    auto partner_fetch = [&](int t) __attribute__((always_inline)) {  // (clamped: an always valid row)
        int f = frame_of(t);
        f = f < 1 ? 1 : (f > NF ? NF : f);
        const float *row = rows + (long long)(f - 1) * S1p;
#pragma unroll
        for (int i = 0; i < NSEG; ++i) pvn[i] = row[(s1[i] & 0xffffu) >> 2];
    };
    // (service wave) posteriors of the frame of step ts: its per-pdf log2 sums are complete (the compute waves formed them
    // during step ts + 1).  gamma = 2^(pl - max) normalised by the frame's own sum (:155-158); log Z of the frame =
    // log2(sum) + max + the two offsets (:159)
    auto frame_out = [&](int ts) __attribute__((always_inline)) {
        const int f = frame_of(ts);
        const double own_off = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(base + MM_WAVE_OFF(ts & 3));
        const double partner_off = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(base + MM_WAVE_POFF(ts & 7));
        float pl[NJ], M = MM_WAVE_NEG;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            pl[j] = ldsr(base + MM_WAVE_PL(ts & 1) + 4u * (unsigned)(q < P1 ? q : 0));
            if (q < P1) M = fmaxf(M, pl[j]);
        }
        M = wave_max_rl(M);
        const bool alive = M > 0.5f * MM_WAVE_NEG;  // (nothing alive: no accepting path, gamma = 0)
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            pl[j] = (alive && q < P1) ? fast_exp2(pl[j] - M) : 0.f;
            tot += pl[j];
        }
        tot = wave_sum_fixed(tot);
        const float inv = (tot > 0.f ? 1.f / tot : 0.f) * p.g_scale;  // (mm_batch_set_gamma_mode: 1 by default)
        float *gp = p.gamma + gbase + (long long)(f - 1) * p.gsn;
        if (p.g_acc) {  // gamma_out += scale * gamma: fire-and-forget float atomics (nothing for this wave to wait for)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = lane + 64 * j;
                if (q < P) (void)__hip_atomic_fetch_add(gp + q * p.gsp, pl[j] * inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = lane + 64 * j;
                if (q < P) gp[q * p.gsp] = pl[j] * inv;
            }
        }
        const double z = alive ? (double)fast_log2(tot) + (double)M + own_off + partner_off : -__builtin_inf();
        zmin = z < zmin ? z : zmin;
    };
    // log-sum-exp over lane groups: (mx, s) of every lane -> the group's, in all its lanes.  lvl = log2 of the lanes of
    // this lane's group (per lane or wave-uniform), lvmax = the wave's largest.  Straight-line for groups of up to 16
    // lanes (a select per level: a branch per level and value costs a wave ~30 cycles each, 300 for a segment).
#define MM_WAVE_DPP3(op, ctrl, dst, src) asm("s_nop 1\n\t" op " %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=&v"(dst) : "v"(src))
    auto group_lse = [&](float &mx, float &s, int lvl, int lvmax) __attribute__((always_inline)) {
        float M = mx, t;
        MM_WAVE_DPP3("v_max_f32_dpp", "quad_perm:[1,0,3,2]", t, M);
        M = lvl >= 1 ? t : M;
        if (lvmax >= 2) {
            MM_WAVE_DPP3("v_max_f32_dpp", "quad_perm:[2,3,0,1]", t, M);
            M = lvl >= 2 ? t : M;
            if (lvmax >= 3) {
                MM_WAVE_DPP3("v_max_f32_dpp", "row_half_mirror", t, M);
                M = lvl >= 3 ? t : M;
                MM_WAVE_DPP3("v_max_f32_dpp", "row_mirror", t, M);
                M = lvl >= 4 ? t : M;
                if (lvmax >= 5) {
                    t = fmaxf(M, __shfl_xor(M, 16));
                    M = lvl >= 5 ? t : M;
                    t = fmaxf(M, __shfl_xor(M, 32));
                    M = lvl >= 6 ? t : M;
                }
            }
        }
        float v = s * fast_exp2(mx - M);
        MM_WAVE_DPP3("v_add_f32_dpp", "quad_perm:[1,0,3,2]", t, v);
        v = lvl >= 1 ? t : v;
        if (lvmax >= 2) {
            MM_WAVE_DPP3("v_add_f32_dpp", "quad_perm:[2,3,0,1]", t, v);
            v = lvl >= 2 ? t : v;
            if (lvmax >= 3) {
                MM_WAVE_DPP3("v_add_f32_dpp", "row_half_mirror", t, v);
                v = lvl >= 3 ? t : v;
                MM_WAVE_DPP3("v_add_f32_dpp", "row_mirror", t, v);
                v = lvl >= 4 ? t : v;
                if (lvmax >= 5) {
                    t = v + __shfl_xor(v, 16);
                    v = lvl >= 5 ? t : v;
                    t = v + __shfl_xor(v, 32);
                    v = lvl >= 6 ? t : v;
                }
            }
        }
        mx = M;
        s = v;
    };
    // ... the same for a wave-uniform lvl (the state segments: all rows of a segment have groups of one size): no selects, the
    // combine of a level is ONE instruction (v_max_f32_dpp / v_add_f32_dpp), the 16-lane rows meet in scalar registers
#define MM_WAVE_DPP(op, ctrl) asm("s_nop 1\n\t" op " %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(v))
    auto group_lse_uniform = [&](float &mx, float &s, int lvl) __attribute__((always_inline)) {
        if (lvl == 0) return;
        float v = mx;
        MM_WAVE_DPP("v_max_f32_dpp", "quad_perm:[1,0,3,2]");
        if (lvl >= 2) {
            MM_WAVE_DPP("v_max_f32_dpp", "quad_perm:[2,3,0,1]");
            if (lvl >= 3) {
                MM_WAVE_DPP("v_max_f32_dpp", "row_half_mirror");
                if (lvl >= 4) {
                    MM_WAVE_DPP("v_max_f32_dpp", "row_mirror");
                    if (lvl >= 5) {
                        const int iv = __builtin_bit_cast(int, v);
                        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
                        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
                        float lo = fmaxf(r0, r1), hi = fmaxf(r2, r3);
                        if (lvl >= 6) lo = hi = fmaxf(lo, hi);
                        v = lane < 32 ? lo : hi;
                    }
                }
            }
        }
        const float M = v;
        v = s * fast_exp2(mx - M);
        MM_WAVE_DPP("v_add_f32_dpp", "quad_perm:[1,0,3,2]");
        if (lvl >= 2) {
            MM_WAVE_DPP("v_add_f32_dpp", "quad_perm:[2,3,0,1]");
            if (lvl >= 3) {
                MM_WAVE_DPP("v_add_f32_dpp", "row_half_mirror");
                if (lvl >= 4) {
                    MM_WAVE_DPP("v_add_f32_dpp", "row_mirror");
                    if (lvl >= 5) {
                        const int iv = __builtin_bit_cast(int, v);
                        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
                        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
                        float lo = r0 + r1, hi = r2 + r3;
                        if (lvl >= 6) lo = hi = lo + hi;
                        v = lane < 32 ? lo : hi;
                    }
                }
            }
        }
        mx = M;
        s = v;
    };
    // (compute waves) the per-pdf log2 sums of the frame of step ts, whose rows wrote u = log2(alpha~ beta~) to QV during that
    // step: the wave's pdf segments -- gathers, a lane-local log-sum-exp, the groups' butterflies, one store per pdf; every
    // order is fixed.  Issued BEHIND the next step's gathers.
    auto pdf_reduce = [&](int ts) __attribute__((always_inline)) {
        float x[4 * NPS];
#pragma unroll
        for (int q = 0; q < 4 * NPS; ++q) x[q] = ldsr(aq[q] + MM_WAVE_QV(ts & 1));
#pragma unroll
        for (int j = 0; j < NPS; ++j) {
            float mx = fmaxf(fmaxf(x[4 * j], x[4 * j + 1]), fmaxf(x[4 * j + 2], x[4 * j + 3]));
            float s = (fast_exp2(x[4 * j] - mx) + fast_exp2(x[4 * j + 1] - mx)) + (fast_exp2(x[4 * j + 2] - mx) + fast_exp2(x[4 * j + 3] - mx));
            if (qlvmax[j] > 0) group_lse(mx, s, (int)(iq[j] >> 16), qlvmax[j]);
            const float pl = mx > 0.5f * MM_WAVE_NEG ? mx + fast_log2(s) : MM_WAVE_NEG;
            ldsw(base + MM_WAVE_PL(ts & 1) + (iq[j] & 0xffffu), pl);
        }
    };

    // ROLE: 0 compute wave, 1 service wave E, 2 service wave P -- one loop per role (run()): what only one role needs (the
    // pointers of the emissions and posteriors, the graph registers) does not stay live in the other roles' loops
    auto step = [&](auto RDc, auto PHc, auto ROLEc, int t) __attribute__((always_inline)) {
        constexpr int RD = decltype(RDc)::value, WR = 1 - RD, PHASE = decltype(PHc)::value, ROLE = decltype(ROLEc)::value;
        const int f = frame_of(t);
        if constexpr (ROLE == 1) {
            // (the frame of step t + 1 was requested at step t - 4: of what is in flight only the youngest 3 NJ operations may be
            // newer -- the other memory operations of this wave in between only make the wait stricter)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NJ) : "memory");
            MM_STAMP(2);
            // the maximum of the vector of step t - 2 (found by wave P during the last step): what this step subtracts
            const float Muse = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ldsr(base + MM_WAVE_MS(WR)))));
            cum += (double)Muse + (double)Enext;
            Ecur = Enext;
            Enext = em_stage(frame_of(t + 1), RD);  // emissions of step t + 1
            em_fetch(frame_of(t + 5));
            MM_STAMP(3);
            if (lane == 0) {
                const double off = DIR ? cum - (double)Ecur : cum;  // (backward: the stored / combined vector is without the frame's emission)
                if constexpr (PHASE == 0) offs[f] = off;
                else *(__attribute__((address_space(3))) double *)(__UINTPTR_TYPE__)(base + MM_WAVE_OFF(t & 3)) = off;
            }
            MM_STAMP(4);
            agent_sync();
            MM_STAMP(1);
        }
        if constexpr (ROLE == 2) {
            // the maximum of the vector of step t - 1: what step t + 1 subtracts
            const float Mnew = scan_max(RD);
            if (lane == 0) ldsw(base + MM_WAVE_MS(RD), Mnew);
            MM_STAMP(2);
            agent_sync();
            MM_STAMP(1);
        }
        if constexpr (ROLE == 3) {
            if constexpr (PHASE == 1) {
                // (the partner offset of step t - 2 was requested at step t - 4: at most the 3 requests since are newer)
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                poff_fetch(t + 2);
                MM_STAMP(3);
                // the frame of the step before last: the compute waves formed its per-pdf sums during the last step
                if (t - 2 > tA) {
                    const int fp = frame_of(t - 2);
                    if (fp >= 1 && fp <= len) frame_out(t - 2);
                }
            }
            MM_STAMP(4);
            agent_sync();
            MM_STAMP(1);
        }
        if constexpr (ROLE == 0) {
        float *rowf = rows + (long long)(f - 1) * S1p;
        if constexpr (PHASE == 1) {
#pragma unroll
            for (int i = 0; i < NSEG; ++i) pv[i] = pvn[i];
            partner_fetch(t + 1);
        }
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
        // the maximum of the vector of step t - 2 (found by the service wave during the last step)
        const float Muse = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ldsr(base + MM_WAVE_MS(WR)))));
        if constexpr (PHASE == 1)
            if (t - 1 > tA) pdf_reduce(t - 1);  // (behind the gathers)
        static_for<0, NSEG>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const float x0 = xs[4 * i] + w[4 * i], x1 = xs[4 * i + 1] + w[4 * i + 1];
            const float x2 = xs[4 * i + 2] + w[4 * i + 2], x3 = xs[4 * i + 3] + w[4 * i + 3];
            const float mx = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
            mxs[i] = mx;
            sums[i] = (fast_exp2(x0 - mx) + fast_exp2(x1 - mx)) + (fast_exp2(x2 - mx) + fast_exp2(x3 - mx));
        });
        MM_STAMP(2);
        // (2) rows of more than 4 arcs (few in the graphs of this kernel; the waves that have none skip this)
        if (lgw != 0ull) {
            static_for<0, NSEG>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                group_lse_uniform(mxs[i], sums[i], (int)((lgw >> (4 * i)) & 15ull));
            });
        }
        // (3) finish the rows, again in one block
        static_for<0, NSEG>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const unsigned pos4 = s0[i] & 0xffffu;
            // forward: (T' alpha) (*) lhs (src/inference.jl:70-71); backward: T (B (*) lhs) (:106-107), the emission is
            // added for the next step's product only
            float bq = mxs[i] + fast_log2(sums[i]) - Muse;
            float y = bq + ems[i];
            y = y > MM_WAVE_NEG ? y : MM_WAVE_NEG;
            bq = bq > MM_WAVE_NEG ? bq : MM_WAVE_NEG;
            ldsw(base + MM_WAVE_VEC(WR) + pos4, y);
            const float st = DIR ? bq : y;  // the vector that is stored / combined
            if constexpr (PHASE == 0) {
                *reinterpret_cast<float *>(reinterpret_cast<char *>(rowf) + pos4) = st;
            } else {
                float uu = st + pv[i];  // log2 (alpha beta) up to the two offsets   (:154)
                uu = uu > MM_WAVE_NEG ? uu : MM_WAVE_NEG;
                ldsw(base + MM_WAVE_QV(WR) + pos4, uu);
            }
        });
        MM_STAMP(3);
        agent_sync();
        MM_STAMP(1);
        }
    };
    auto run_role = [&](auto PHc, auto ROLEc, int tfirst, int tlast) __attribute__((always_inline)) {
        for (int t = tfirst; t <= tlast; t += 2) {
            if (t & 1) step(std::integral_constant<int, 0>{}, PHc, ROLEc, t);
            else step(std::integral_constant<int, 1>{}, PHc, ROLEc, t);
            if (t + 1 <= tlast) {
                if ((t + 1) & 1) step(std::integral_constant<int, 0>{}, PHc, ROLEc, t + 1);
                else step(std::integral_constant<int, 1>{}, PHc, ROLEc, t + 1);
            }
        }
    };
    auto run = [&](auto PHc, int tfirst, int tlast) __attribute__((always_inline)) {
        if (svcE) run_role(PHc, std::integral_constant<int, 1>{}, tfirst, tlast);
        else if (svcP) run_role(PHc, std::integral_constant<int, 2>{}, tfirst, tlast);
        else if (svcF) run_role(PHc, std::integral_constant<int, 3>{}, tfirst, tlast);
        else run_role(PHc, std::integral_constant<int, 0>{}, tfirst, tlast);
    };
    MM_STAMP_RESET;
    run(std::integral_constant<int, 0>{}, 2, tA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MM_STAMPS
    if (p.x_sleep & 0x2000) {  // (report phase A only)
        if (p.dbg && lane == 0)
            for (int q = 0; q < 8; ++q) p.dbg[((long long)b * 16 + wv) * 8 + q] = stamp_acc[q];
        p.dbg = nullptr;
    }
    if (p.x_sleep & 0x1000)  // (report phase B only)
        for (int q = 0; q < 8; ++q) stamp_acc[q] = 0;
#endif
    __syncthreads();  // the ONE workgroup barrier: phase A of both agents is stored
    MM_STAMP_RESET;
    if (svcF) {
        poff_fetch(tA + 1);
        poff_fetch(tA + 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (!service) {
        partner_fetch(tA + 1);
    }
    run(std::integral_constant<int, 1>{}, tA + 1, NF);
    // the last two steps' frames (NF is not live for the forward agent: frame len + 1; frame 1 for the backward agent)
    if (NF > tA) {
        if (svcF) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int fp = frame_of(NF - 1);
            if (NF - 1 > tA && fp >= 1 && fp <= len) frame_out(NF - 1);
        } else if (!service) {
            pdf_reduce(NF);
        }
        agent_sync();
        if (svcF) {
            const int fp = frame_of(NF);
            if (fp >= 1 && fp <= len) frame_out(NF);
        }
    }
#ifdef MM_STAMPS
    if (p.dbg && lane == 0)
        for (int q = 0; q < 8; ++q) p.dbg[((long long)b * 16 + wv) * 8 + q] = stamp_acc[q];
#endif
    // ---- ttl, zeros beyond the sequence length
    if (lane == 0 && svcF) *(__attribute__((address_space(3))) double *)(__UINTPTR_TYPE__)(base + MM_WAVE_ZZ) = zmin;
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
    if (!p.g_acc)
        for (long long q = threadIdx.x; q < (long long)(p.N - len) * P; q += NT)
            p.gamma[gbase + (len + q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
}

}  // namespace mm
