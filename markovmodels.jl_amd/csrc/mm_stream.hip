// mm_stream.hip -- the stream kernels: pdfposteriors for graphs beyond every register-resident form (more than ~6000 states,
// more than 506 pdfs), where the reference has no size limit (src/linalg.jl:170-181 launches over any number of rows).
//
// Until round 5 such graphs ran on the item kernel (mm_kernels.hip): one workgroup per utterance, log domain, state vectors in
// global memory once they outgrow the LDS (16 bytes per state) -- 10 000 states / 1000 pdfs, B = 64, T = 700: 162 ms, bound by
// instruction issue (~150 instructions per 256 arc slots) and by gathers that are L2 accesses.  Here:
//   * the ARCS are streamed, the VECTOR stays on chip.  A workgroup (15 compute waves + a service wave) owns one utterance; the
//     linear state vector sits in LDS as "wide-exponent 32-bit" values -- the high dword of a double, 4 bytes per state
//     (mm_kernel_wpair.hip): two buffers of up to 59 KB, graphs of up to ~15 000 states -- and every frame the waves read their
//     share of the arcs from L2 as 8-byte records {LDS address of the source, high dword of the weight's double}: the record
//     IS the v_fma_f64 operand pair of the weight, its low dword the gather's address.  An arc is a coalesced 8-byte load (16 in
//     flight per wave), a ds_read_b32 into the high register of a persistent operand pair, and one v_fma_f64.  The graph
//     (1.4 MB for 170 k arcs) is shared by the workgroups of an XCD through its L2;
//   * rows are sorted by length and cut into SEGMENTS of 64 (one lane per row, padded to the segment's longest), rows of more
//     than 128 arcs get a wave to themselves; segments are dealt to the 15 waves longest-processing-time first; the internal
//     numbering is the finishing order, so a finish stores to consecutive LDS / HBM positions;
//   * float64 accumulation and the wide format's 1022 log2 of range: no float32 marks; the range marks of the float64 kernels
//     (decided by the same two criteria, mm_stream_finish_kernel) hand an utterance to the item kernel -- normally none;
//   * emissions per ROW: a finish reads its pdf's staged emission from LDS (1000 pdfs: 4 KB per frame, staged by the service
//     wave a step ahead); the state -> pdf sums of the combine are LDS float64 atomics (ds_add_f64) into per-pdf sums the service
//     wave normalises a step later -- any number of pdfs up to 1024, no per-pdf ranges;
//   * forward launch (alpha~ stored once per frame, in the BACKWARD numbering: the scattered stores cost nothing, the backward
//     launch then reads its rows' alpha~ coalesced and a segment ahead), then backward launch (beta never materialised).
// Per frame a workgroup streams its whole graph: ~1.4 MB at the ~130 GB/s a compute unit gets from its L2 is ~10 us; 700 + 700
// frames = 15 ms at B = 64 (64 workgroups; the bidirectional split of the pair kernels would halve it and is not built).
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_wpair.hip"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <numeric>

namespace mm {

struct StreamDev {  // one direction, device pointers
    const unsigned long long *arcs;  // [slots of all waves][64]: {4 * position of the source, high dword of the weight}
    const unsigned *seg;             // [segments][4]: arc slots, log2 lanes per row (0 or 6), first position, rows
    const unsigned *rinfo;           // [rows] by position: pdf | position in the OTHER direction's numbering << 12
    const float *init;               // [rows] by position: alpha_hat, log2 (forward)
    int wave_seg0[16], wave_slot0[16];  // per compute wave: first segment / first arc slot (entry 15: the totals)
    int rows, fpos, vb, pad;         // vb: bytes of one LDS vector
    float thr;
    int pad2;
};
struct StreamPairDev {
    StreamDev d[2];
};

struct StreamForm {
    StreamPairDev host;   // with DEVICE pointers once uploaded
    void *blob = nullptr;
    StreamPairDev *dev = nullptr;  // the descriptor on the device (start of the blob)
    int S1 = 0, P1 = 0;
    // host copies for mm_stream_eval (test aid)
    std::vector<unsigned long long> h_arcs[2];
    std::vector<unsigned> h_seg[2], h_rinfo[2];
    std::vector<float> h_init;
    std::vector<int32_t> pos[2];
};

constexpr int kStreamWaves = 15;
constexpr int kStreamWide = 128;  // rows of more arcs get a whole wave

static unsigned w_hi_of(double v) {  // high dword, 20 mantissa bits rounded to nearest
    unsigned long long b;
    memcpy(&b, &v, 8);
    return unsigned((b + 0x80000000ull) >> 32);
}

static int stream_nj(int P1) { return P1 <= 128 ? 2 : (P1 <= 256 ? 4 : (P1 <= 512 ? 8 : 16)); }  // 64-lane passes over the pdfs
size_t mm_stream_lds_bytes(int S1, int P1) {
    const size_t vb = (size_t(4) * (size_t(S1) + 1) + 255) & ~size_t(255);
    const size_t pc = size_t(64) * size_t(stream_nj(P1));
    return 2 * vb + 24 * pc /* EM 2 x 4, PSUM 2 x 8 per pdf */ + 256;
}

static bool stream_pack(int d, int64_t S1, const int64_t *rowptr, const int32_t *col, const float *val, std::vector<int32_t> &pos,
                        std::vector<unsigned> &seg, int (&wave_seg0)[16], int (&wave_slot0)[16], float *wmin_out) {
    // rows by length, longest first
    std::vector<int32_t> rows(static_cast<size_t>(S1));
    std::iota(rows.begin(), rows.end(), 0);
    auto nnz = [&](int32_t r) { return int(rowptr[r + 1] - rowptr[r]); };
    std::stable_sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return nnz(a) > nnz(b); });
    struct Seg {
        int nsl, lg, first, n;  // first: index into `rows`
    };
    std::vector<Seg> segs;
    size_t i = 0;
    while (i < rows.size() && nnz(rows[i]) > kStreamWide) {
        segs.push_back({(nnz(rows[i]) + 63) / 64, 6, int(i), 1});
        ++i;
    }
    for (; i < rows.size(); i += 64) {
        const int n = int(std::min<size_t>(64, rows.size() - i));
        segs.push_back({nnz(rows[i]), 0, int(i), n});
    }
    // longest-processing-time first over the waves (a finish costs about a dozen arc slots)
    std::vector<int> order(segs.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return segs[a].nsl > segs[b].nsl; });
    std::vector<std::vector<int>> per(kStreamWaves);
    std::vector<long long> load(kStreamWaves, 0);
    for (int s : order) {
        const int w = int(std::min_element(load.begin(), load.end()) - load.begin());
        per[w].push_back(s);
        load[w] += segs[s].nsl + 12;
    }
    pos.assign(static_cast<size_t>(S1), -1);
    seg.clear();
    int p0 = 0, nseg = 0, nslot = 0;
    for (int w = 0; w < kStreamWaves; ++w) {
        wave_seg0[w] = nseg;
        wave_slot0[w] = nslot;
        if (per[w].size() > 0xffffu) return false;
        for (int s : per[w]) {
            const Seg &sg = segs[size_t(s)];
            seg.push_back(unsigned(sg.nsl));
            seg.push_back(unsigned(sg.lg));
            seg.push_back(unsigned(p0));
            seg.push_back(unsigned(sg.n));
            for (int k = 0; k < sg.n; ++k) pos[size_t(rows[size_t(sg.first + k)])] = p0 + k;
            p0 += sg.n;
            nslot += sg.nsl;
            ++nseg;
        }
    }
    wave_seg0[kStreamWaves] = nseg;
    wave_slot0[kStreamWaves] = nslot;
    float wmin = 0.f;
    for (int64_t k = 0; k < rowptr[S1]; ++k)
        if (val[k] > -INFINITY) wmin = std::min(wmin, val[k]);
    *wmin_out = wmin;
    (void)d;
    (void)col;
    return p0 == S1;
}

static void stream_fill(int64_t S1, const int64_t *rowptr, const int32_t *col, const float *val, const std::vector<int32_t> &pos,
                        const std::vector<unsigned> &seg, const int (&wave_slot0)[16], const int (&wave_seg0)[16],
                        std::vector<unsigned long long> &arcs) {
    std::vector<int32_t> order(static_cast<size_t>(S1));
    for (int64_t r = 0; r < S1; ++r) order[size_t(pos[size_t(r)])] = int32_t(r);
    arcs.assign(size_t(wave_slot0[kStreamWaves]) * 64, 0ull);
    for (int w = 0; w < kStreamWaves; ++w) {
        size_t slot = size_t(wave_slot0[w]);
        for (int s = wave_seg0[w]; s < wave_seg0[w + 1]; ++s) {
            const unsigned nsl = seg[size_t(4 * s)], lg = seg[size_t(4 * s + 1)], p0 = seg[size_t(4 * s + 2)], n = seg[size_t(4 * s + 3)];
            for (unsigned l = 0; l < 64; ++l) {
                const int32_t r = lg ? order[p0] : (l < n ? order[p0 + l] : -1);
                if (r < 0) continue;
                const int64_t b = rowptr[r], e = rowptr[r + 1];
                for (unsigned k = 0; k < nsl; ++k) {
                    const int64_t a = lg ? b + int64_t(k) * 64 + l : b + k;
                    if (a >= e) break;
                    const double wl = std::exp2(double(val[a]));
                    arcs[(slot + k) * 64 + l] = (static_cast<unsigned long long>(w_hi_of(wl)) << 32) | (4u * unsigned(pos[size_t(col[a])]));
                }
            }
            slot += nsl;
        }
    }
}

// host: pack both directions and upload.  rowptr / col / val: [0] T_hat' (forward: row j = the arcs INTO j), [1] T_hat
// (backward), weights log2; init: dense alpha_hat (log2); s2p: state -> pdf.  *out = NULL (and MM_OK) if the graph does not fit.
int mm_stream_build(int64_t S1, int32_t P1, const int64_t *const rowptr[2], const int32_t *const col[2], const float *const val[2],
                    const float *init, const int32_t *s2p, bool upload, StreamForm **out) {
    *out = nullptr;
    if (S1 < 2 || S1 > 16383 || P1 > 1024 || mm_stream_lds_bytes(int(S1), P1) > 160 * 1024) return MM_OK;
    auto f = std::make_unique<StreamForm>();
    f->S1 = int(S1);
    f->P1 = P1;
    float wmin[2] = {0.f, 0.f};
    for (int d = 0; d < 2; ++d) {
        if (!stream_pack(d, S1, rowptr[d], col[d], val[d], f->pos[d], f->h_seg[d], f->host.d[d].wave_seg0, f->host.d[d].wave_slot0, &wmin[d]))
            return MM_OK;
        stream_fill(S1, rowptr[d], col[d], val[d], f->pos[d], f->h_seg[d], f->host.d[d].wave_slot0, f->host.d[d].wave_seg0, f->h_arcs[d]);
    }
    f->h_init.assign(size_t(S1), -INFINITY);
    for (int d = 0; d < 2; ++d) {
        f->h_rinfo[d].assign(size_t(S1), 0u);
        for (int64_t r = 0; r < S1; ++r)
            f->h_rinfo[d][size_t(f->pos[d][size_t(r)])] = unsigned(s2p[r]) | (unsigned(f->pos[1 - d][size_t(r)]) << 12);
        StreamDev &sd = f->host.d[d];
        sd.rows = int(S1);
        sd.fpos = f->pos[d][size_t(S1 - 1)];
        sd.vb = int((4 * (S1 + 1) + 255) & ~int64_t(255));
        sd.thr = 125.f + wmin[d] + 896.f;
        sd.pad = sd.pad2 = 0;
    }
    for (int64_t r = 0; r < S1; ++r) f->h_init[size_t(f->pos[0][size_t(r)])] = init[r];
    if (upload) {
        size_t off = (sizeof(StreamPairDev) + 255) & ~size_t(255);
        size_t o_arcs[2], o_seg[2], o_rinfo[2], o_init;
        auto place = [&](size_t bytes) {
            const size_t o = off;
            off = (off + bytes + 255) & ~size_t(255);
            return o;
        };
        for (int d = 0; d < 2; ++d) {
            o_arcs[d] = place(f->h_arcs[d].size() * 8);
            o_seg[d] = place(f->h_seg[d].size() * 4);
            o_rinfo[d] = place(f->h_rinfo[d].size() * 4);
        }
        o_init = place(f->h_init.size() * 4);
        std::vector<char> img(off, 0);
        HIP_TRY(hipMalloc(&f->blob, off));
        char *base = static_cast<char *>(f->blob);
        StreamPairDev dv = f->host;
        for (int d = 0; d < 2; ++d) {
            memcpy(img.data() + o_arcs[d], f->h_arcs[d].data(), f->h_arcs[d].size() * 8);
            memcpy(img.data() + o_seg[d], f->h_seg[d].data(), f->h_seg[d].size() * 4);
            memcpy(img.data() + o_rinfo[d], f->h_rinfo[d].data(), f->h_rinfo[d].size() * 4);
            dv.d[d].arcs = reinterpret_cast<const unsigned long long *>(base + o_arcs[d]);
            dv.d[d].seg = reinterpret_cast<const unsigned *>(base + o_seg[d]);
            dv.d[d].rinfo = reinterpret_cast<const unsigned *>(base + o_rinfo[d]);
            dv.d[d].init = reinterpret_cast<const float *>(base + o_init);
        }
        memcpy(img.data() + o_init, f->h_init.data(), f->h_init.size() * 4);
        memcpy(img.data(), &dv, sizeof(dv));
        if (hipMemcpy(f->blob, img.data(), off, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(f->blob);
            return mm_fail(MM_ERR_HIP, "stream form: upload failed");
        }
        f->dev = static_cast<StreamPairDev *>(f->blob);
    }
    *out = f.release();
    return MM_OK;
}
void mm_stream_free(StreamForm *f) {
    if (!f) return;
    if (f->blob) (void)hipFree(f->blob);
    delete f;
}
const void *mm_stream_dev(const StreamForm *f) { return f ? f->dev : nullptr; }

// test aid (host): out = M (x) in through the stream form of direction d exactly as a workgroup walks it -- wave by wave, segment by
// segment, the records' 20-bit weights, float64 accumulation, wave-wide sums for the rows that own a wave; natural log in / out
void mm_stream_eval(const StreamForm *f, int d, const float *in, float *out, double stats[4]) {
    const int S1 = f->S1;
    std::vector<double> lin(static_cast<size_t>(S1) + 1, 0.0);
    for (int r = 0; r < S1; ++r) lin[size_t(f->pos[d][size_t(r)])] = std::exp(double(in[r]));
    std::vector<double> res(static_cast<size_t>(S1), 0.0);
    const StreamDev &sd = f->host.d[d];
    long long real = 0;
    for (int w = 0; w < kStreamWaves; ++w) {
        size_t slot = size_t(sd.wave_slot0[w]);
        for (int s = sd.wave_seg0[w]; s < sd.wave_seg0[w + 1]; ++s) {
            const unsigned nsl = f->h_seg[d][size_t(4 * s)], lg = f->h_seg[d][size_t(4 * s + 1)], p0 = f->h_seg[d][size_t(4 * s + 2)],
                           n = f->h_seg[d][size_t(4 * s + 3)];
            double acc[64] = {0};
            for (unsigned k = 0; k < nsl; ++k)
                for (unsigned l = 0; l < 64; ++l) {
                    const unsigned long long a = f->h_arcs[d][(slot + k) * 64 + l];
                    const unsigned long long wb = (a >> 32) << 32;
                    double wv;
                    memcpy(&wv, &wb, 8);
                    real += (a >> 32) != 0;
                    acc[l] += wv * lin[size_t(unsigned(a) / 4u)];
                }
            if (lg) {
                double t = 0;
                for (double v : acc) t += v;
                res[p0] = t;
            } else {
                for (unsigned l = 0; l < n; ++l) res[p0 + l] = acc[l];
            }
            slot += nsl;
        }
    }
    for (int r = 0; r < S1; ++r) {
        const double v = res[size_t(f->pos[d][size_t(r)])];
        out[r] = v > 0 ? float(std::log(v)) : -INFINITY;
    }
    if (stats) {
        stats[0] = double(sd.wave_slot0[kStreamWaves]);                      // arc slots per lane, all waves
        stats[1] = double(sd.wave_seg0[kStreamWaves]);                       // segments
        stats[2] = double(real) / (double(sd.wave_slot0[kStreamWaves]) * 64);  // real arcs / arc slots
        int mx = 0;
        for (int w = 0; w < kStreamWaves; ++w) mx = std::max(mx, sd.wave_slot0[w + 1] - sd.wave_slot0[w]);
        stats[3] = double(mx);                                               // slots of the most loaded wave
    }
}

// ---------------------------------------------------------------------------------------------------------------- device
template <int NJ>
struct StreamLay {  // LDS bytes behind the two vectors (vb each)
    static constexpr unsigned PC = 64u * NJ;
    static constexpr unsigned EM(int par) { return unsigned(par) * 4u * PC; }              // staged emissions [pdf] (+ the phony pdf)
    static constexpr unsigned PSUM(int par) { return 8u * PC + unsigned(par) * 8u * PC; }  // doubles [pdf]
    static constexpr unsigned MS(int par) { return 24u * PC + 8u * unsigned(par); }        // the step's normaliser
    static constexpr unsigned MX(int par) { return 24u * PC + 16u + 8u * unsigned(par); }  // maximum (high dword) of the vector a step writes
    static constexpr unsigned OWN(int k) { return 24u * PC + 32u + 8u * unsigned(k); }     // own offsets of the last 4 steps (doubles)
    static constexpr unsigned TOTAL = 24u * PC + 64u;
};

__device__ __forceinline__ void lds_atomic_add_f64(unsigned addr, double v) {
    typedef __attribute__((address_space(3))) double lds_f64;
    (void)__hip_atomic_fetch_add((lds_f64 *)(__UINTPTR_TYPE__)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_atomic_max_u32(unsigned addr, unsigned v) {
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    (void)__hip_atomic_fetch_max((lds_u32 *)(__UINTPTR_TYPE__)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// PASS 0: forward (alpha~ of every frame stored in the backward numbering), 1: backward (+ combine, per-pdf sums, gamma)
template <int PASS, int NJ>
__global__ void __launch_bounds__(1024) mm_stream_kernel(RunParams p) {
    extern __shared__ float lds[];
    using L = StreamLay<NJ>;
    constexpr int C = 8;  // arc records per chunk (two chunks in flight per wave)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = wave == kStreamWaves;
    const int b = uni(p.order ? p.order[blockIdx.x] : (int)blockIdx.x);
    const UttDesc &ud = p.utts[b];
    const StreamPairDev *spd = uni(reinterpret_cast<const StreamPairDev *>(ud.stream));
    const StreamDev &sd = spd->d[PASS];
    const int S1 = uni(sd.rows), P1 = uni(ud.P1), P = P1 - 1, S1p = uni(ud.S1p);
    const unsigned VB = (unsigned)uni(sd.vb), FIX = 2u * VB;
    int len = uni(p.lens ? p.lens[b] : p.N);
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int NF = len + 1;
    const float *Vb = p.V + (long long)b * p.vsb;
    const long long s1p_prefix = ((long long)uni((int)(ud.s1p_prefix >> 32)) << 32) | (unsigned)uni((int)ud.s1p_prefix);
    float *rows = p.ws_alpha + s1p_prefix * (long long)(p.N + 1);  // [frame - 1][S1p]: alpha~ by BACKWARD position
    double *offs = p.ws_c + (long long)b * (p.N + 2);               // [frame]: the offset of the stored alpha~
    const float thr = sd.thr;
    auto frame_of = [&](int t) { return PASS ? NF + 1 - t : t; };
    if (lds_addr_of(lds) != 0u) __builtin_trap();
    if (len == 0) {  // no frame: no path of length 0 (the finish kernel writes ttl = -inf and the zeros)
        if (PASS == 1 && tid == 0) {
            p.pair_zmin[(long long)b * 6 + 0] = p.pair_zmin[(long long)b * 6 + 1] = __builtin_inf();
            p.pair_zmin[(long long)b * 6 + 2] = p.pair_zmin[(long long)b * 6 + 3] = -__builtin_inf();
            p.pair_zmin[(long long)b * 6 + 4] = p.pair_zmin[(long long)b * 6 + 5] = __builtin_inf();
        }
        return;
    }
    for (unsigned q = tid * 4u; q < FIX + L::TOTAL; q += 4096u) ldsw(q, 0.f);
    __syncthreads();

    if (service) {
        // ================= service wave: emissions, normalisers, offsets; backward: the posteriors =================
        RowNorm norm;
        double cum = 0.0, zmin = __builtin_inf(), zmax = -__builtin_inf();
        float ltmin = __builtin_inff();
        // stage the emissions of step t (frame f) into EM(t & 1): log2 values relative to the frame's maximum E
        auto stage = [&](int t, float S) {
            const int f = frame_of(t);
            float v[NJ], E = MM_NINF;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = lane + 64 * j;
                v[j] = em_value(em_load_raw(Vb, p.vsn, f, p.N, P, q), f, len, P, q);
                if (q < P) E = max_nc(E, v[j]);
            }
            E = wave_max_rl(E);
            if (!(E > MM_NINF)) E = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = lane + 64 * j;
                if (q <= P) ldsw(FIX + L::EM(t & 1) + 4u * q, v[j] - E);
            }
            cum += (double)S + (double)E;
            if (lane == 0) {
                ldsw(FIX + L::MS(t & 1), S);
                // the offset that turns the step's stored / combined vector into log2 values: alpha~ includes the frame's emission
                const double off = PASS ? cum - (double)E : cum;
                *(__attribute__((address_space(3))) double *)(__UINTPTR_TYPE__)(FIX + L::OWN(t & 3)) = off;
                if (PASS == 0) offs[f] = off;
            }
        };
        // posteriors and per-frame log Z of step ts (its per-pdf sums are complete); then the sums are zeroed for step ts + 2
        auto frames_of_step = [&](int ts) {
            const int f = frame_of(ts);
            const unsigned psum = FIX + L::PSUM(ts & 1);
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {  // (the sums are read twice: 16 passes of doubles kept in registers spilled)
                const int q = lane + 64 * j;
                const double v = ldsr_d(psum + 8u * (unsigned)(q < P1 ? q : 0));
                if (q < P1) t += v;
            }
            t = dwave_sum_rl(t);
            const int e = __builtin_amdgcn_frexp_exp(t);
            const float tf = (float)__builtin_amdgcn_ldexp(t, -e);
            const float inv = tf > 0.f ? 1.f / tf : 0.f;
            float *gp = p.gamma + (long long)b * p.gsb + (long long)(f - 1) * p.gsn;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = lane + 64 * j;
                if (q < P1) {
                    const double v = ldsr_d(psum + 8u * (unsigned)q);
                    if (q < P && f >= 1 && f <= len) gp[q * p.gsp] = (float)__builtin_amdgcn_ldexp(v, -e) * inv;
                    ldsw_d(psum + 8u * (unsigned)q, 0.0);
                }
            }
            if (f >= 1 && f <= len) {
                const float lt = dlog2(t);
                const double own = *(__attribute__((address_space(3))) const double *)(__UINTPTR_TYPE__)(FIX + L::OWN(ts & 3));
                const double z = (double)lt + own + offs[f];
                zmin = z < zmin ? z : zmin;
                zmax = z > zmax ? z : zmax;
                ltmin = lt < ltmin ? lt : ltmin;
            }
        };
        if (PASS == 0) {  // step 1: alpha_hat (*) lhs[:, 1] needs the emissions of frame 1 (nothing but E is subtracted)
            stage(1, 0.f);
        }
        __syncthreads();  // (1)
        if (2 <= NF) stage(2, 0.f);
        __syncthreads();  // (2) the starting vector is in LDS
        for (int t = 2; t <= NF; ++t) {
            // the normaliser of step t + 1 from the maximum of the vector of step t - 1 (complete since the last barrier)
            const unsigned mxa = FIX + L::MX((t - 1) & 1);
            const float mx = w_log2_hi(ldsru(mxa));
            if (lane == 0) ldswu(mxa, 0u);
            if (t + 1 <= NF) stage(t + 1, norm.next(mx));
            if (PASS == 1 && t - 1 >= 2) frames_of_step(t - 1);
            __syncthreads();
        }
        if (PASS == 1) {
            if (NF >= 2) frames_of_step(NF);
            if (lane == 0) {
                p.pair_zmin[(long long)b * 6 + 0] = p.pair_zmin[(long long)b * 6 + 1] = zmin;
                p.pair_zmin[(long long)b * 6 + 2] = p.pair_zmin[(long long)b * 6 + 3] = zmax;
                p.pair_zmin[(long long)b * 6 + 4] = p.pair_zmin[(long long)b * 6 + 5] = (double)ltmin;
            }
        }
    } else {
        // ================= compute waves =================
        __syncthreads();  // (1)
        unsigned vmax = 0u;
        float worst = 0.f;
        if (PASS == 0) {  // alpha_hat (*) lhs[:, 1]   (src/inference.jl:68)
            const unsigned *ri = uni(sd.rinfo);
            const float *ini = uni(sd.init);
            for (int i = tid; i < S1; i += 64 * kStreamWaves) {
                const unsigned info = ri[i];
                const float v0 = ini[i] + ldsr(FIX + L::EM(1) + 4u * (info & 0xfffu));
                worst = __builtin_fmaxf(worst, __builtin_fmaf(__builtin_fabsf(v0), 0.f, __builtin_fabsf(v0)));
                const unsigned hi = w_exp2_hi(v0);
                vmax = vmax > hi ? vmax : hi;
                ldswu(VB * 1u + 4u * (unsigned)i, hi);
                rows[(long long)0 * S1p + (info >> 12)] = v0;
            }
        } else {  // B[:, N+1] = one at the final state   (src/inference.jl:104)
            if (tid == 0) {
                ldswu(VB * 1u + 4u * (unsigned)uni(sd.fpos), 0x3ff00000u);
                vmax = 0x3ff00000u;
            }
        }
        {
            const unsigned m = wave_max_u32(vmax);
            if (lane == 0 && m) lds_atomic_max_u32(FIX + L::MX(1), m);
            vmax = 0u;
        }
        __syncthreads();  // (2)
        const int seg0 = uni(sd.wave_seg0[wave]), seg1 = uni(sd.wave_seg0[wave + 1]);
        const int slot0 = uni(sd.wave_slot0[wave]), nslots = uni(sd.wave_slot0[wave + 1]) - slot0;
        const unsigned long long *ap = uni(sd.arcs) + (long long)slot0 * 64 + lane;
        const unsigned *segt = uni(sd.seg);
        const unsigned *rinfo = uni(sd.rinfo);
        const int nchunks = (nslots + C - 1) / C;
        // the operand pair of the gathered value: only its high register is written in the loops below
        double xop[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            xop[j] = 0.0;
            asm volatile("" : "+v"(xop[j]));
        }
        for (int t = 2; t <= NF; ++t) {
            const unsigned rd = VB * (unsigned)((t - 1) & 1), wr = VB * (unsigned)(t & 1);
            const int f = frame_of(t);
            const float S = ldsr(FIX + L::MS(t & 1));
            const unsigned emb = FIX + L::EM(t & 1), psb = FIX + L::PSUM(t & 1);
            float *rowf = rows + (long long)(f - 1) * S1p;
            // the segment the wave is in, and what its finish needs (a segment ahead: the loads are long done when it ends)
            int sg = seg0;
            unsigned s_nsl = 0, s_lg = 0, s_p0 = 0, s_n = 0, info = 0;
            float al = 0.f;
            int remaining = 0;
            auto load_seg = [&](int s) {  // (wave-uniform table entries; the per-row words of the lanes)
                if (s < seg1) {
                    s_nsl = (unsigned)uni((int)segt[4 * s]);
                    s_lg = (unsigned)uni((int)segt[4 * s + 1]);
                    s_p0 = (unsigned)uni((int)segt[4 * s + 2]);
                    s_n = (unsigned)uni((int)segt[4 * s + 3]);
                    const unsigned i = s_p0 + (s_lg ? 0u : ((unsigned)lane < s_n ? (unsigned)lane : 0u));
                    info = rinfo[i];
                    if (PASS == 1) al = rowf[i];
                    remaining = (int)s_nsl;
                }
            };
            double acc = 0.0;
            auto finish = [&]() {
                double s0 = acc;
                if (s_lg) s0 = dwave_sum_rl(s0);
                const bool mine = s_lg ? lane == 0 : (unsigned)lane < s_n;
                const unsigned posi = s_p0 + (s_lg ? 0u : (unsigned)lane);
                // forward: (T' alpha) (*) lhs (src/inference.jl:70-71); backward: T (B (*) lhs) (:106-107), the emission is
                // added for the next step's product only
                const float bb = w_log2_acc(s0) - S;
                const float y = bb + ldsr(emb + 4u * (info & 0xfffu));
                if (mine) {
                    worst = __builtin_fmaxf(worst, __builtin_fmaf(__builtin_fabsf(y), 0.f, __builtin_fabsf(y)));
                    const unsigned hi = w_exp2_hi(y);
                    vmax = vmax > hi ? vmax : hi;
                    ldswu(wr + 4u * posi, hi);
                    if (PASS == 0) {
                        rowf[info >> 12] = y;  // alpha~ in the backward numbering
                    } else if (f <= len) {
                        lds_atomic_add_f64(psb + 8u * (info & 0xfffu), dexp2(bb + al));  // C' * (A .* B)   (:154-155)
                    }
                }
                acc = 0.0;
                ++sg;
                load_seg(sg);
            };
            load_seg(sg);
            while (sg < seg1 && remaining == 0) finish();  // (rows without arcs)
            unsigned long long cur[C], nxt[C];
#pragma unroll
            for (int j = 0; j < C; ++j) cur[j] = j < nslots ? ap[(long long)j * 64] : 0ull;
            for (int c = 0; c < nchunks; ++c) {
                const int base = c * C;
#pragma unroll
                for (int j = 0; j < C; ++j) nxt[j] = base + C + j < nslots ? ap[(long long)(base + C + j) * 64] : 0ull;
#pragma unroll
                for (int j = 0; j < C; ++j) {  // the gathers of the chunk leave together
                    mm_u32x2 o = __builtin_bit_cast(mm_u32x2, xop[j]);
                    o.y = ldsru(rd + (unsigned)cur[j]);
                    xop[j] = __builtin_bit_cast(double, o);
                }
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    if (base + j < nslots) {
                        asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(__builtin_bit_cast(double, cur[j])), "v"(xop[j]));
                        if (--remaining == 0) {
                            finish();
                            while (sg < seg1 && remaining == 0) finish();
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < C; ++j) cur[j] = nxt[j];
            }
            {
                const unsigned m = wave_max_u32(vmax);
                if (lane == 0 && m) lds_atomic_max_u32(FIX + L::MX(t & 1), m);
                vmax = 0u;
            }
            __syncthreads();
        }
        if (__builtin_amdgcn_ballot_w64(worst > thr) != 0ull && lane == 0) p.redo[b] = 1;
    }
}

// ttl = min over the frames of the per-frame log-normaliser (src/inference.jl:159); zeros beyond the sequence lengths; and what
// a range mark means (mm_dpair_finish_kernel's two criteria on the double's range): cleared, or the item kernel computes the
// utterance again
static __global__ void mm_stream_finish_kernel(RunParams p) {
    const int b = blockIdx.x;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int P = p.utts[b].P1 - 1;
    if (threadIdx.x == 0) {
        const double z = p.pair_zmin[6 * b], zM = p.pair_zmin[6 * b + 2], lm = p.pair_zmin[6 * b + 4];
        p.ttl[b] = (z < __builtin_inf()) ? (float)(z * (double)MM_LN2) : MM_NINF;
        const bool agree = z > -__builtin_inf() && zM < __builtin_inf() && zM - z <= MM_Z_SPREAD_TOL;
        if (p.redo[b] == 1 && agree && lm >= (double)p.lt_floor - (double)MM_DPAIR_THR_EXTRA) p.redo[b] = 0;
    }
    const long long gbase = (long long)b * p.gsb;
    for (long long q = threadIdx.x; q < (long long)(p.N - len) * P; q += blockDim.x)
        p.gamma[gbase + (len + q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
}

template <int NJ>
static int launch_stream_nj(int64_t B, size_t lds, const RunParams &p, hipStream_t st) {
    auto k0 = mm_stream_kernel<0, NJ>;
    auto k1 = mm_stream_kernel<1, NJ>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(k0, dim3(unsigned(B)), dim3(1024), lds, st, p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k1, dim3(unsigned(B)), dim3(1024), lds, st, p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(mm_stream_finish_kernel, dim3(unsigned(B)), dim3(256), 0, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
int mm_launch_stream(int64_t B, int max_S1, int max_P1, const RunParams &p, hipStream_t st) {
    const size_t lds = mm_stream_lds_bytes(max_S1, max_P1);
    if (lds > 160 * 1024 || max_P1 > 1024) return mm_fail(MM_ERR_UNSUPPORTED, "stream kernel: LDS");
    switch (stream_nj(max_P1)) {
        case 2: return launch_stream_nj<2>(B, lds, p, st);
        case 4: return launch_stream_nj<4>(B, lds, p, st);
        case 8: return launch_stream_nj<8>(B, lds, p, st);
        default: return launch_stream_nj<16>(B, lds, p, st);
    }
}

}  // namespace mm
