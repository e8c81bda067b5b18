// mm_kernel_vit.hip -- Viterbi for gfx950 on the row-lane form: the tropical alpha-recursion (src/inference.jl:62-74 with
// K = TropicalSemiring) with compact back-pointers, and a back-trace that streams them through LDS.  bestpath is
// documented (docs/src/inference.md:6) and absent from src/ at this commit (src/MarkovModels.jl:56-57; historical use
// examples/demo.ipynb cell 23); the specification of the back-pointers is the oracle's: the lowest source state among
// the maximisers, none (-1) if the maximum is zero(K).
//
// Against the item kernel (mm_tropical_kernel, mm_kernels.hip: 2.66 ms on BASELINE config 5 + 0.61 ms for a back-trace in
// which one lane chased 1000 pointers through HBM):
//   * the graph sits in registers in the row-lane form of mm_rows.h (RowPackOpts::acap_force / seg_stride / keep_order:
//     at most 4 arcs of a row per lane and segment, in the order of the row), 15 compute waves + a service wave for the
//     emissions, ONE workgroup barrier per frame; a max-plus row is 4 gathers, 4 adds, 3 compare-selects -- no
//     transcendental, no normaliser (the reference's plain float adds: the scores are bit-identical to the CPU
//     restatement), for rows of more than 4 arcs a lane-group reduction by DPP under the same tie rule;
//   * a back-pointer is the NUMBER OF THE ARC in its row (rows have their arcs by ascending source state, so the first
//     maximum is the one the tie rule wants): one byte per state and frame -- a quarter of the int32 rows,
//     which were the algorithmic traffic of this path;
//   * back-trace: one workgroup per utterance streams the byte rows through a double-buffered LDS ring, 8 KB per hop of the
//     chase hidden behind the copy of the next chunk; the hop itself is three LDS reads (arc number, row pointer, source).
// Graphs with a row of more than 255 arcs, more than 6 segments per wave (15 waves x 6 x 64 lanes = 5760 rows at most) stay on the item kernel,
// and so does a call that asks for the int32 back-pointer table (mm_viterbi_f32 with bp != NULL).
#pragma once
#include "mm_kernel_rows.hip"

namespace mm {

typedef float mm_vf32x2 __attribute__((ext_vector_type(2)));

// (max, lowest arg-max) over an aligned group of 1 << lg lanes (lg wave-uniform), in all its lanes -- a row of 5 to 256
// arcs: the hub states of a lexicon graph.  Two butterflies of ONE instruction per level instead of one of a dozen (two
// moves, four compares, two selects per level, as trop_grp_reduce of the item kernel has it): the maximum of the lanes'
// values, then the minimum of the arc numbers of the lanes whose own value IS that maximum (a lane's own pair already
// follows the tie rule: strict '>' over ascending arcs; arc numbers grow with the source state).  Across the 16-lane rows
// through scalar registers, not ds_bpermute (an LDS round trip per value and level).  The four waves that own a hub row
// were the slowest of every frame by 40 %.
#define MM_VIT_DPP(op, ctrl) asm("s_nop 1\n\t" op " %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(v))
__device__ __forceinline__ float vit_grp_max(float v, int lg, int lane) {
    MM_VIT_DPP("v_max_f32_dpp", "quad_perm:[1,0,3,2]");
    if (lg >= 2) {
        MM_VIT_DPP("v_max_f32_dpp", "quad_perm:[2,3,0,1]");
        if (lg >= 3) {
            MM_VIT_DPP("v_max_f32_dpp", "row_half_mirror");
            if (lg >= 4) {
                MM_VIT_DPP("v_max_f32_dpp", "row_mirror");
                if (lg >= 5) {
                    const int iv = __builtin_bit_cast(int, v);
                    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
                    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
                    float lo = max_nc(r0, r1), hi = max_nc(r2, r3);
                    if (lg >= 6) lo = hi = max_nc(lo, hi);
                    v = lane < 32 ? lo : hi;
                }
            }
        }
    }
    return v;
}
__device__ __forceinline__ int vit_grp_min(int v, int lg, int lane) {
    MM_VIT_DPP("v_min_i32_dpp", "quad_perm:[1,0,3,2]");
    if (lg >= 2) {
        MM_VIT_DPP("v_min_i32_dpp", "quad_perm:[2,3,0,1]");
        if (lg >= 3) {
            MM_VIT_DPP("v_min_i32_dpp", "row_half_mirror");
            if (lg >= 4) {
                MM_VIT_DPP("v_min_i32_dpp", "row_mirror");
                if (lg >= 5) {
                    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
                    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
                    int lo = r0 < r1 ? r0 : r1, hi = r2 < r3 ? r2 : r3;
                    if (lg >= 6) lo = hi = lo < hi ? lo : hi;
                    v = lane < 32 ? lo : hi;
                }
            }
        }
    }
    return v;
}
__device__ __forceinline__ void vit_grp_reduce(float &best, int &arg, int lg, int lane) {
    const float M = vit_grp_max(best, lg, lane);
    arg = vit_grp_min(best == M ? arg : 0x7fffffff, lg, lane);
    best = M;
}

#define MM_VIT_NWC 15
#define MM_VIT_ESZ 1056u  // bytes of an emission buffer (256 pdfs + the "no row" slot)

// N4 positions of 4 arc slots + N2 positions of 2 per wave (RowPackOpts::mix_n4 / mix_n2: most rows of a lexicon or an
// HMM chain have 2 arcs, and a 4-slot position costs them twice the gathers and compares); NJ * 64 >= P + 1; VSZ bytes of
// a state vector
template <int N4, int N2, int NJ, int VSZ>
__global__ void __launch_bounds__(1024) mm_vit_kernel(RunParams p) {
    extern __shared__ float lds[];
    constexpr int KA = 4 * N4 + 2 * N2, NSEG = N4 + N2;
    constexpr unsigned EMB = 2u * VSZ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = wave == MM_VIT_NWC;
    const int b = blockIdx.x;
    const UttDesc &u = p.utts[b];
    const RowU r = uni(u.rv);
    const int S1 = r.rows, P1 = uni(u.P1), P = P1 - 1;
    int len = uni(p.lens ? p.lens[b] : p.N);
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int NF = len + 1;
    const float *Vb = p.V + (long long)b * p.vsb;
    // [utterance][frame - 1][bp_stride_n bytes] arc numbers (rows padded to 256 bytes: a chunk of frames is one aligned block)
    unsigned char *bpk = reinterpret_cast<unsigned char *>(p.bp) + (long long)b * (p.N + 1) * p.bp_stride_n;
    if (lds_addr_of(lds) != 0u) __builtin_trap();
    MM_STAMP_DECL;

    // ---- the wave's rows: registers
    mm_vf32x2 w2[KA / 2];  // (weights in aligned register pairs: the two arcs of a pair are added by ONE v_pk_add_f32)
    unsigned a[KA], s0[NSEG];
    const bool mine_w = !service && wave < r.NWC;
    const RowSched &sc = r.sched[mine_w ? wave : 0];
    const unsigned long long lgw =
        mine_w ? ((unsigned long long)(unsigned)uni((int)(sc.lg >> 32)) << 32) | (unsigned)uni((int)sc.lg) : 0ull;
    const int nseg = mine_w ? uni((int)(sc.nslots & 0xffffu)) : 0, slot0 = mine_w ? uni((int)sc.slot0) : 0;
    const int n4w = mine_w ? uni((int)(sc.nslots >> 16)) : 0, n2w = nseg - n4w;  // segments in wide / narrow positions
    {
        const auto wp = as_global(r.w);
        const auto ap = as_global(r.addr);
        const auto sp = as_global(r.slots);
        const int nt = 64 * r.NWC, col = (mine_w ? wave : 0) * 64 + lane;
#pragma unroll
        for (int k = 0; k < KA; ++k) {
            const bool have = mine_w && k < r.KA;
            const float wk = have ? wp[k * nt + col] : MM_NINF;
            if (k & 1) w2[k / 2].y = wk;
            else w2[k / 2].x = wk;
            a[k] = have ? ap[k * nt + col] : 0u;
        }
        // position i < N4 (wide) holds the wave's segment i, position N4 + j (narrow) its segment n4w + j; positions without
        // a segment: the "no row" position and emission slot
        const unsigned none = (4u * (unsigned)S1) | ((4u * (unsigned)((P1 + 3) & ~3)) << 16);
#pragma unroll
        for (int i = 0; i < N4; ++i) s0[i] = i < n4w ? sp[(slot0 + i) * 64 + lane] : none;
#pragma unroll
        for (int j = 0; j < N2; ++j) s0[N4 + j] = j < n2w ? sp[(slot0 + n4w + j) * 64 + lane] : none;
    }
    // log2 of the lanes per row of every position (4 bits each, in the order of the positions)
    unsigned long long lgp = 0ull;
#pragma unroll
    for (int i = 0; i < N4; ++i) lgp |= (i < n4w ? ((lgw >> (4 * i)) & 15ull) : 0ull) << (4 * i);
#pragma unroll
    for (int j = 0; j < N2; ++j) lgp |= (j < n2w ? ((lgw >> (4 * (n4w + j))) & 15ull) : 0ull) << (4 * (N4 + j));
    const unsigned sub = (unsigned)lane & ((1u << 6) - 1u);
    for (unsigned q = 4u * tid; q < 2u * VSZ + 2u * MM_VIT_ESZ + 4u * NJ * 256u; q += 4096u) ldsw(q, MM_NINF);
    // emissions (expand(), src/inference.jl:54-60; natural log like the reference's tropical values), by the service wave:
    // raw values by LDS-DMA FOUR frames ahead (a frame is ~0.5 us, a load from HBM 1-2 us: fetched one frame ahead the
    // whole workgroup waited for the service wave's load every frame), staged a frame ahead
    constexpr unsigned RAWB = EMB + 2u * MM_VIT_ESZ;  // [4][NJ * 256 bytes]
    auto em_fetch = [&](int f) __attribute__((always_inline)) { row_dma_em<NJ>(RAWB + (unsigned)(f & 3) * (NJ * 256u), Vb, p.vsn, f, p.N, P, lane); };
    auto em_stage = [&](int f, int par) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = lane + 64 * j;
            const float rawv = ldsr(RAWB + (unsigned)(f & 3) * (NJ * 256u) + 256u * j + 4u * lane);
            float v;
            if (q < P) v = f <= len ? rawv : MM_NINF;
            else v = f <= len ? MM_NINF : 0.f;
            if (q <= P) ldsw(EMB + (unsigned)par * MM_VIT_ESZ + 4u * q, v);
        }
    };
    __syncthreads();
    if (service) {
        for (int f = 1; f <= 4; ++f) em_fetch(f);
        MM_ROW_VMCNT(0);
        em_stage(1, 1);
        em_fetch(5);
    }
    __syncthreads();
    // frame 1: alpha_hat (*) lhs[:,1]   (src/inference.jl:68)
    for (int i = tid; i < S1; i += 1024)
        ldsw(VSZ + 4u * i, as_global(r.init)[i] + ldsr(EMB + MM_VIT_ESZ + 4u * as_global(r.rowpdf)[i]));
    if (service) {
        em_stage(2, 0);
        em_fetch(6);
    }
    __syncthreads();

    // (SVC: the service wave's loop and the compute waves' loop are two loops: what only one role needs -- the emission
    // pointers, the graph registers -- does not stay live in the other's)
    auto step = [&](auto RDc, auto SVCc, int n) __attribute__((always_inline)) {
        constexpr int RD = decltype(RDc)::value, WR = 1 - RD;  // RD = parity of frame n - 1
        constexpr bool SVC = decltype(SVCc)::value;
        if constexpr (SVC) {
            // frame n + 1 was requested at step n - 3: at most the NJ DMAs of each of the 3 later requests are in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NJ) : "memory");
            MM_STAMP(2);
            em_stage(n + 1, RD);
            em_fetch(n + 5);  // (its buffer, frame (n + 1) & 3, has just been read)
        } else {
            // the frame's row of back-pointers: one byte per position, the lanes of a segment store adjacent bytes.  Every
            // lane stores -- the ones without a row to the "no row" position, a byte of the padded row like any other -- and
            // dead states store whatever arc won the comparison of -inf values: the back-trace starts from a state that
            // is alive (or not at all: score = -inf) and a live state's best arc comes from a live state.  (An exec mask
            // for the lanes without a row and a compare-select per position for the dead ones were a sixth of the vector
            // instructions of a 2-arc position, in a kernel in which the SIMDs' issue slots and the LDS are both ~90 % busy.)
            unsigned char *row = bpk + (long long)(n - 1) * p.bp_stride_n;
            // a block of positions at a time: its gathers and emission reads in flight together, then straight-line code.
            // W arc slots per position, NB positions from position h0 on, their slots from k0 on.
            auto block = [&](auto Wc, auto NBc, auto H0c, auto K0c) __attribute__((always_inline)) {
                constexpr int W = decltype(Wc)::value, NB = decltype(NBc)::value, h0 = decltype(H0c)::value, k0 = decltype(K0c)::value;
                static_assert(W % 2 == 0 && k0 % 2 == 0, "arc slots come in pairs");
                mm_vf32x2 xs[W / 2 * NB];
                float es[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
#pragma unroll
                    for (int q = 0; q < W / 2; ++q) {
                        xs[W / 2 * i + q].x = ldsr(a[k0 + W * i + 2 * q] + (unsigned)RD * VSZ);
                        xs[W / 2 * i + q].y = ldsr(a[k0 + W * i + 2 * q + 1] + (unsigned)RD * VSZ);
                    }
                    es[i] = ldsr(EMB + (unsigned)WR * MM_VIT_ESZ + (s0[h0 + i] >> 16));
                }
                float best[NB];
                int arg[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    // T_hat[i, j] (*) A[i, n-1], strict '>' over ascending source states: the lowest source among the maximisers
                    // (two arcs per v_pk_add_f32: the same IEEE additions, half the instructions)
                    float bv = 0.f;
                    int bk = 0;
#pragma unroll
                    for (int q = 0; q < W / 2; ++q) {
                        mm_vf32x2 v;
                        asm("v_pk_add_f32 %0, %1, %2" : "=v"(v) : "v"(xs[W / 2 * i + q]), "v"(w2[k0 / 2 + W / 2 * i + q]));
                        if (q == 0) {
                            bv = v.x;
                        } else {
                            bk = v.x > bv ? 2 * q : bk;
                            bv = v.x > bv ? v.x : bv;
                        }
                        bk = v.y > bv ? 2 * q + 1 : bk;
                        bv = v.y > bv ? v.y : bv;
                    }
                    best[i] = bv;
                    arg[i] = bk;
                }
                if (((lgp >> (4 * h0)) & ((1ull << (4 * NB)) - 1ull)) != 0ull) {  // rows split over 1 << lg lanes: lane `s` of the group holds arcs s, s + g, ...
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        const int lg = (int)((lgp >> (4 * (h0 + i))) & 15ull);
                        if (lg) {
                            arg[i] = (int)(sub & ((1u << lg) - 1u)) + (arg[i] << lg);
                            vit_grp_reduce(best[i], arg[i], lg, lane);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const unsigned pos4 = s0[h0 + i] & 0xffffu;
                    ldsw(pos4 + (unsigned)WR * VSZ, best[i] + es[i]);  // (*) lhs[:, n]   (:70-71)
                    row[pos4 >> 2] = (unsigned char)arg[i];
                }
            };
            using std::integral_constant;
            // wide positions in blocks of <= 3, narrow ones in blocks of <= 4
            if constexpr (N4 > 0) block(integral_constant<int, 4>{}, integral_constant<int, (N4 < 3 ? N4 : 3)>{}, integral_constant<int, 0>{}, integral_constant<int, 0>{});
            if constexpr (N4 > 3) block(integral_constant<int, 4>{}, integral_constant<int, N4 - 3>{}, integral_constant<int, 3>{}, integral_constant<int, 12>{});
            if constexpr (N2 > 0) block(integral_constant<int, 2>{}, integral_constant<int, (N2 < 4 ? N2 : 4)>{}, integral_constant<int, N4>{}, integral_constant<int, 4 * N4>{});
            if constexpr (N2 > 4) block(integral_constant<int, 2>{}, integral_constant<int, N2 - 4>{}, integral_constant<int, N4 + 4>{}, integral_constant<int, 4 * N4 + 8>{});
        }
        MM_STAMP(0);
        __syncthreads();
        MM_STAMP(1);
    };
    MM_STAMP_RESET;
    auto run = [&](auto SVCc) __attribute__((always_inline)) {
        for (int n = 2; n <= NF; n += 2) {
            step(std::integral_constant<int, 1>{}, SVCc, n);
            if (n + 1 <= NF) step(std::integral_constant<int, 0>{}, SVCc, n + 1);
        }
    };
    if (service) run(std::integral_constant<bool, true>{});
    else run(std::integral_constant<bool, false>{});
#ifdef MM_STAMPS
    if (p.dbg && lane == 0)
        for (int q = 0; q < 8; ++q) p.dbg[((long long)b * 16 + wave) * 8 + q] = stamp_acc[q];
#endif
    if (tid == 0) p.score[b] = ldsr((unsigned)(NF & 1) * VSZ + 4u * (unsigned)r.fpos);
}

// Back-trace: from the phony final state at frame len + 1 (historical bestpath, examples/demo.ipynb cell 23).  The byte
// rows of the frames travel to LDS in chunks of R frames, double buffered, by LDS-DMA of waves 1..7 (everything of a chunk
// in flight at once: the copy is bound by bandwidth, not by the latency of a load); wave 0 chases the chunk that is
// there.  CSRL: the graph's row pointers and sources (positions) are in LDS as well, a hop is three LDS reads.
// LDS: ring [2][R * RSB + 1024] bytes (RSB = the row padded to 256 bytes), then [rowptr (S1 + 1) u32][col (arcs) u16].
template <bool CSRL>
__global__ void __launch_bounds__(512) mm_vit_backtrace_kernel(RunParams p, int R, int RSB) {
    extern __shared__ float lds[];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = NT >> 6;
    const UttDesc &u = p.utts[b];
    const RowDev &r = u.rv;
    const int S1 = r.rows;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    int *path = p.path + (long long)b * p.path_stride_b;
    for (int n = len + tid; n < p.N; n += NT) path[n] = -1;
    if (!(p.score[b] > MM_NINF)) {
        for (int n = tid; n < len; n += NT) path[n] = -1;
        return;
    }
    if (lds_addr_of(lds) != 0u) __builtin_trap();
    const unsigned char *bpk = reinterpret_cast<const unsigned char *>(p.bp) + (long long)b * (p.N + 1) * p.bp_stride_n;  // (bp_stride_n == RSB)
    const int arcs = r.rowptr[S1];
    // (a chunk is copied by 1 KB DMAs and need not end on one -- RSB is a multiple of 256: every buffer of the ring has 1 KB to spare)
    const unsigned RB = (unsigned)R * (unsigned)RSB + 1024u;
    const unsigned ringb = 2u * RB;
    unsigned *lrp = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(lds) + ringb);
    unsigned short *lcol = reinterpret_cast<unsigned short *>(lrp + S1 + 1);
    if (CSRL) {
        for (int i = tid; i <= S1; i += NT) lrp[i] = (unsigned)r.rowptr[i];
        for (int i = tid; i < arcs; i += NT) lcol[i] = (unsigned short)r.col[i];
    }
    // chunk `hi` holds frames hi - R + 1 .. hi (frame f at row f - 1 of bpk: ONE block of R * RSB bytes, a multiple of 1 KB)
    // in slot buf of the ring, frame f at ring row f - (hi - R + 1); rows of frames < 1 are not copied
    auto copy_chunk = [&](int hi, int buf) {
        if (wave == 0) return;
        const int lo = hi - R + 1 < 1 ? 1 : hi - R + 1;  // first frame whose row exists
        const unsigned char *src0 = bpk + (long long)(lo - 1) * RSB;
        const unsigned dst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)buf * RB + (unsigned)(lo - (hi - R + 1)) * (unsigned)RSB));
        const int nd = ((hi - lo + 1) * RSB + 1023) >> 10;  // 1 KB DMAs (the last may run up to 768 bytes past the chunk: the spare KB here, the
                                                            // next rows of the workspace -- which ends with a spare KB too -- there)
        for (int d = wave - 1; d < nd; d += NW - 1) dma_b128(src0 + 1024ll * d + 16 * lane, dst0 + 1024u * (unsigned)d);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    int s = r.fpos;  // position (internal numbering) of the current state; the chase starts at frame len + 1
    copy_chunk(len + 1, 0);
    __syncthreads();
    int cidx = 0;
    for (int hi = len + 1; hi >= 2; hi -= R, ++cidx) {
        if (hi - R >= 2) copy_chunk(hi - R, (cidx + 1) & 1);
        if (tid == 0) {
            const unsigned char *cur = reinterpret_cast<const unsigned char *>(lds) + (size_t)(cidx & 1) * RB;
            for (int k = 0; k < R; ++k) {
                const int f = hi - k;  // the frame whose back-pointer is followed: state at f -> state at f - 1
                if (f < 2) break;
                const unsigned kk = cur[(R - 1 - k) * RSB + s];
                const unsigned rp = CSRL ? lrp[s] : (unsigned)r.rowptr[s];
                s = CSRL ? (int)lcol[rp + kk] : r.col[rp + kk];
                path[f - 2] = s;  // (the POSITION: a store the chase does not wait for; a load of order[s] here stalled every hop)
            }
        }
        __syncthreads();
    }
    // positions -> original states, by everybody
    for (int n = tid; n < len; n += NT) path[n] = r.order[path[n]];
}

}  // namespace mm
