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
//     wave a step ahead) -- any number of pdfs up to 1024; the state -> pdf sums are the separate combine kernel's (below);
//   * BOTH recursions at once: the forward and the backward workgroup of an utterance are workgroups of ONE grid (2 B workgroups:
//     B = 64 uses 128 compute units; two grids one after the other when 2 B exceeds the chip); each stores its normalised log2
//     vector of every frame -- alpha~ with, beta~ without the frame's emission -- in a numbering both share (states sorted by
//     pdf: the scattered 4-byte stores of a finish cost nothing), and mm_stream_combine_kernel then reads both rows of a frame
//     coalesced, sums 2^(alpha~ + beta~) over each pdf's contiguous range in float64 -- no atomics, a fixed order: deterministic
//     --, normalises the frame like the reference (src/inference.jl:154-160) and leaves the frame's log Z for the finish kernel.
//     (The first version ran a forward launch, then a backward launch that combined on the fly with LDS float64 atomics: 41 ms for
//     10 000 states / 1000 pdfs at B = 64 on 64 compute units; this one: see DESIGN.md.)
// What bounds a frame (10 000 states, 162 k arcs: 1.3 MB of records): the vector memory path of the compute unit -- the records
// alone cost ~14 us per frame (2500 wave loads of 512 bytes; the same kernel with the ring loaded once per frame takes half the
// time) -- next to ~14 us of gathers, FMAs, finishes and barriers that a wave's serial chain (wait for records, gather, wait, FMA)
// overlaps only partly; the gathers' bank conflicts (no bank-aware placement here) do not show: without the gathers the time is
// the same (-DMM_STREAM_NOGATHER / -DMM_STREAM_NOLOAD, timing experiments).
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_wpair.hip"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <numeric>

namespace mm {

constexpr int kStreamHMax = 4;  // workgroups of a team (1: the whole direction in one workgroup)
struct StreamDev {  // one direction and SET of rows (a team of H workgroups computes a direction: workgroup h finishes the rows of set h)
    // the record stream, 6 bytes per arc slot (round 6; 8 until then: the stream's bytes through ONE compute unit's vector memory path are
    // what bounds a frame): per chunk of 4 slots [64 lanes][4] high dwords of the weights (16 bytes per lane: one dwordx4 load), then
    // [64 lanes][4] 16-bit LDS byte offsets of the sources (8 bytes per lane: one dwordx2 load) -- 384 dwords per chunk
    const unsigned *arcs;
    const unsigned *seg;             // [segments][4]: arc slots, log2 lanes per row (0 or 6), first position, rows
    const unsigned *rinfo;           // [positions] of the WHOLE vector: pdf | position in the pdf-major numbering both directions store in << 12 (padding: ~0)
    const float *init;               // [positions]: alpha_hat, log2 (forward; padding: -inf)
    int wave_seg0[16], wave_slot0[16];  // per compute wave: first segment / first arc slot (entry 15: the totals)
    int rows, fpos, vb;              // positions of the whole vector (the sets' regions, each padded to a multiple of 4); the final state's; bytes of one LDS vector
    int base, cnt;                   // this set's region of the vector: first position, rows
    float thr;
};
struct StreamPairDev {
    StreamDev d[2][kStreamHMax];
    const int *pdf_ptr;  // [P1 + 1]: the states of pdf q are the pdf-major positions pdf_ptr[q] .. pdf_ptr[q + 1] - 1
    int P1, H;
};

struct StreamForm {
    StreamPairDev host;   // with DEVICE pointers once uploaded
    void *blob = nullptr;
    StreamPairDev *dev = nullptr;  // the descriptor on the device (start of the blob)
    int S1 = 0, P1 = 0, H = 1;
    int npos[2] = {0, 0};          // positions of a direction's vector (regions padded to multiples of 4)
    // host copies for mm_stream_eval (test aid)
    std::vector<unsigned long long> h_arcs[2][kStreamHMax];
    std::vector<unsigned> h_seg[2][kStreamHMax], h_rinfo[2];
    std::vector<float> h_init;
    std::vector<int32_t> pos[2], qpos, h_pdf_ptr;
};

constexpr int kStreamWaves = 15;
constexpr int kStreamWide = 128;  // rows of more arcs get a whole wave
constexpr int kStreamChunk = 4;   // records per chunk: a segment is a whole number of chunks, its info record the last of them

static unsigned w_hi_of(double v) {  // high dword, 20 mantissa bits rounded to nearest
    unsigned long long b;
    memcpy(&b, &v, 8);
    return unsigned((b + 0x80000000ull) >> 32);
}

static int stream_nj(int P1) { return P1 <= 128 ? 2 : (P1 <= 256 ? 4 : (P1 <= 512 ? 8 : 16)); }  // 64-lane passes over the pdfs
size_t mm_stream_lds_bytes(int S1, int P1) {
    // (+ 12: the regions of up to four sets are padded to multiples of 4 positions)
    const size_t vb = (size_t(4) * (size_t(S1) + 12 + 1) + 255) & ~size_t(255);
    const size_t pc = size_t(64) * size_t(stream_nj(P1));
    return 2 * vb + 8 * pc /* EM: 2 x 4 bytes per pdf */ + 256;
}

// the rows `sub` (sorted longest first) of one set: segments of 64 rows (a whole wave for a row of more than kStreamWide arcs), dealt
// to the waves longest-processing-time first; positions base .. base + |sub| - 1 in finishing order
// (groups: rows of 65 .. 128 arcs get 4 lanes each, rows of 33 .. 64 two -- no segment is longer than 32 arc slots.  For the teams'
// forms: a workgroup of a team has 1 / H of the arc slots, and a segment of 64 rows of ~128 arcs -- ONE wave's -- was then the longest
// thing of every step: 144 of an average 105 slots per wave on a 10 000-state graph of config 3's family with teams of 2)
static bool stream_pack(const std::vector<int32_t> &sub, int base, const int64_t *rowptr, std::vector<int32_t> &pos, std::vector<unsigned> &seg,
                        int (&wave_seg0)[16], int (&wave_slot0)[16], bool groups) {
    auto nnz = [&](int32_t r) { return int(rowptr[r + 1] - rowptr[r]); };
    struct Seg {
        int nsl, lg, first, n;  // first: index into `sub`
    };
    std::vector<Seg> segs;
    size_t i = 0;
    while (i < sub.size() && nnz(sub[i]) > kStreamWide) {
        segs.push_back({(nnz(sub[i]) + 63) / 64, 6, int(i), 1});
        ++i;
    }
    while (i < sub.size()) {
        const int len = nnz(sub[i]);
        const int lg = !groups ? 0 : (len > 64 ? 2 : (len > 32 ? 1 : 0));
        const int n = int(std::min<size_t>(size_t(64 >> lg), sub.size() - i));
        segs.push_back({(len + (1 << lg) - 1) >> lg, lg, int(i), n});
        i += size_t(n);
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
    seg.clear();
    int p0 = base, nseg = 0, nslot = 0;
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
            for (int k = 0; k < sg.n; ++k) pos[size_t(sub[size_t(sg.first + k)])] = p0 + k;
            p0 += sg.n;
            nslot += kStreamChunk * ((sg.nsl + 1 + kStreamChunk - 1) / kStreamChunk);  // (its arcs + the info record, in whole chunks)
            ++nseg;
        }
    }
    wave_seg0[kStreamWaves] = nseg;
    wave_slot0[kStreamWaves] = nslot;
    return p0 == base + int(sub.size());
}

// the records of one set: `order` = position -> row (-1: padding) of the whole vector, `pos` = row -> position
static void stream_fill(const int64_t *rowptr, const int32_t *col, const float *val, const std::vector<int32_t> &pos, const std::vector<int32_t> &order,
                        const std::vector<int32_t> &qpos, const int32_t *s2p, const std::vector<unsigned> &seg,
                        const int (&wave_slot0)[16], const int (&wave_seg0)[16], std::vector<unsigned long long> &arcs) {
    arcs.assign((size_t(wave_slot0[kStreamWaves]) + 32) * 64, 0ull);  // (+ padding: the kernels' ring reads 24 records ahead)
    for (int w = 0; w < kStreamWaves; ++w) {
        size_t slot = size_t(wave_slot0[w]);
        for (int s = wave_seg0[w]; s < wave_seg0[w + 1]; ++s) {
            const unsigned nsl = seg[size_t(4 * s)], lg = seg[size_t(4 * s + 1)], p0 = seg[size_t(4 * s + 2)], n = seg[size_t(4 * s + 3)];
            const unsigned nrec = unsigned(kStreamChunk) * ((nsl + 1 + kStreamChunk - 1) / kStreamChunk);
            const unsigned G = 1u << lg;  // lanes per row: lane l holds the arcs k * G + l % G of row l / G
            for (unsigned l = 0; l < 64; ++l) {
                const int32_t r = (l >> lg) < n ? order[p0 + (l >> lg)] : -1;
                if (r < 0) continue;
                const int64_t b = rowptr[r], e = rowptr[r + 1];
                for (unsigned k = 0; k < nsl; ++k) {
                    const int64_t a = b + int64_t(k) * G + (l & (G - 1));
                    if (a >= e) break;
                    const double wl = std::exp2(double(val[a]));
                    arcs[(slot + k) * 64 + l] = (static_cast<unsigned long long>(w_hi_of(wl)) << 32) | (4u * unsigned(pos[size_t(col[a])]));
                }
                // the info record of the segment: {0, pdf | pdf-major position << 12} of the lane's row
                arcs[(slot + nrec - 1) * 64 + l] = static_cast<unsigned long long>(unsigned(s2p[r]) | (unsigned(qpos[size_t(r)]) << 12)) << 32;
            }
            slot += nrec;
        }
    }
}

// host: pack both directions for teams of H workgroups (1, 2 or 4) and upload.  rowptr / col / val: [0] T_hat' (forward: row j = the
// arcs INTO j), [1] T_hat (backward), weights log2; init: dense alpha_hat (log2); s2p: state -> pdf.  *out = NULL (and MM_OK) if the
// graph does not fit.  The rows of a direction are dealt to the H sets round-robin by decreasing length (every set gets the same mix
// of long and short rows: the workgroups of a team finish a step together); the sets' regions of the vector follow each other,
// each padded to a multiple of 4 positions (the 16-byte granules of the exchange).  The two directions partition independently:
// what they share is the pdf-major numbering they STORE in (the combine kernel's).
int mm_stream_build(int64_t S1, int32_t P1, const int64_t *const rowptr[2], const int32_t *const col[2], const float *const val[2],
                    const float *init, const int32_t *s2p, bool upload, int H, StreamForm **out) {
    *out = nullptr;
    if (H != 1 && H != 2 && H != 4) return MM_OK;
    // (16-bit LDS byte offsets in the records: 4 * position <= 65 532, positions include up to 12 of padding)
    if (S1 < 2 || S1 > 16370 || P1 > 1024 || mm_stream_lds_bytes(int(S1), P1) > 160 * 1024) return MM_OK;
    auto f = std::make_unique<StreamForm>();
    f->S1 = int(S1);
    f->P1 = P1;
    f->H = H;
    memset(&f->host, 0, sizeof(f->host));
    float wmin[2] = {0.f, 0.f};
    std::vector<int32_t> order[2];  // position -> row (-1: padding)
    for (int d = 0; d < 2; ++d) {
        std::vector<int32_t> rows(static_cast<size_t>(S1));
        std::iota(rows.begin(), rows.end(), 0);
        auto nnz = [&](int32_t r) { return int(rowptr[d][r + 1] - rowptr[d][r]); };
        std::stable_sort(rows.begin(), rows.end(), [&](int32_t a, int32_t b) { return nnz(a) > nnz(b); });
        f->pos[d].assign(static_cast<size_t>(S1), -1);
        int base = 0;
        for (int h = 0; h < H; ++h) {
            std::vector<int32_t> sub;
            for (size_t i = size_t(h); i < rows.size(); i += size_t(H)) sub.push_back(rows[i]);
            StreamDev &sd = f->host.d[d][h];
            if (!stream_pack(sub, base, rowptr[d], f->pos[d], f->h_seg[d][h], sd.wave_seg0, sd.wave_slot0, H > 1)) return MM_OK;
            sd.base = base;
            sd.cnt = int(sub.size());
            base += (int(sub.size()) + 3) & ~3;
        }
        f->npos[d] = base;
        order[d].assign(size_t(base), -1);
        for (int64_t r = 0; r < S1; ++r) order[d][size_t(f->pos[d][size_t(r)])] = int32_t(r);
        for (int64_t k = 0; k < rowptr[d][S1]; ++k)
            if (val[d][k] > -INFINITY) wmin[d] = std::min(wmin[d], val[d][k]);
    }
    // the numbering both directions STORE in: states sorted by pdf (the combine kernel sums a pdf's contiguous range)
    {
        std::vector<int32_t> by(static_cast<size_t>(S1));
        std::iota(by.begin(), by.end(), 0);
        std::stable_sort(by.begin(), by.end(), [&](int32_t a, int32_t b) { return s2p[a] < s2p[b]; });
        f->qpos.assign(size_t(S1), 0);
        f->h_pdf_ptr.assign(size_t(P1) + 1, 0);
        for (int64_t i = 0; i < S1; ++i) {
            f->qpos[size_t(by[size_t(i)])] = int32_t(i);
            if (s2p[by[size_t(i)]] < 0 || s2p[by[size_t(i)]] >= P1) return MM_OK;
            ++f->h_pdf_ptr[size_t(s2p[by[size_t(i)]]) + 1];
        }
        for (int q = 0; q < P1; ++q) f->h_pdf_ptr[size_t(q) + 1] += f->h_pdf_ptr[size_t(q)];
    }
    for (int d = 0; d < 2; ++d) {
        for (int h = 0; h < H; ++h)
            stream_fill(rowptr[d], col[d], val[d], f->pos[d], order[d], f->qpos, s2p, f->h_seg[d][h], f->host.d[d][h].wave_slot0, f->host.d[d][h].wave_seg0,
                        f->h_arcs[d][h]);
        f->h_rinfo[d].assign(size_t(f->npos[d]), 0xffffffffu);
        for (int64_t r = 0; r < S1; ++r)
            f->h_rinfo[d][size_t(f->pos[d][size_t(r)])] = unsigned(s2p[r]) | (unsigned(f->qpos[size_t(r)]) << 12);
        for (int h = 0; h < H; ++h) {
            StreamDev &sd = f->host.d[d][h];
            sd.rows = f->npos[d];
            sd.fpos = f->pos[d][size_t(S1 - 1)];
            sd.vb = int((4 * (int64_t(f->npos[d]) + 1) + 255) & ~int64_t(255));
            sd.thr = 125.f + wmin[d] + 896.f;
        }
    }
    f->h_init.assign(size_t(f->npos[0]), -INFINITY);
    for (int64_t r = 0; r < S1; ++r) f->h_init[size_t(f->pos[0][size_t(r)])] = init[r];
    if (upload) {
        size_t off = (sizeof(StreamPairDev) + 255) & ~size_t(255);
        size_t o_arcs[2][kStreamHMax], o_seg[2][kStreamHMax], o_rinfo[2], o_init, o_pp;
        auto place = [&](size_t bytes) {
            const size_t o = off;
            off = (off + bytes + 255) & ~size_t(255);
            return o;
        };
        for (int d = 0; d < 2; ++d) {
            for (int h = 0; h < H; ++h) {
                o_arcs[d][h] = place(f->h_arcs[d][h].size() / 4 * 24);  // (6 bytes per arc slot and lane)
                o_seg[d][h] = place(f->h_seg[d][h].size() * 4);
            }
            o_rinfo[d] = place(f->h_rinfo[d].size() * 4);
        }
        o_init = place(f->h_init.size() * 4);
        o_pp = place(f->h_pdf_ptr.size() * 4);
        std::vector<char> img(off, 0);
        HIP_TRY(hipMalloc(&f->blob, off));
        char *base = static_cast<char *>(f->blob);
        StreamPairDev dv = f->host;
        for (int d = 0; d < 2; ++d) {
            memcpy(img.data() + o_rinfo[d], f->h_rinfo[d].data(), f->h_rinfo[d].size() * 4);
            for (int h = 0; h < H; ++h) {
                {   // the device's 6-byte form of the records: per chunk [64][4] weights' high dwords, [64][4] 16-bit byte offsets
                    const std::vector<unsigned long long> &a = f->h_arcs[d][h];
                    unsigned *dst = reinterpret_cast<unsigned *>(img.data() + o_arcs[d][h]);
                    const size_t nch = a.size() / (size_t(kStreamChunk) * 64);
                    for (size_t c = 0; c < nch; ++c)
                        for (unsigned l = 0; l < 64; ++l) {
                            unsigned short *po = reinterpret_cast<unsigned short *>(dst + c * 384 + 256) + 4 * l;
                            for (int j = 0; j < kStreamChunk; ++j) {
                                const unsigned long long r = a[(c * kStreamChunk + size_t(j)) * 64 + l];
                                dst[c * 384 + 4 * l + unsigned(j)] = unsigned(r >> 32);
                                po[j] = static_cast<unsigned short>(unsigned(r));  // (4 * position: at most 65 532)
                            }
                        }
                }
                memcpy(img.data() + o_seg[d][h], f->h_seg[d][h].data(), f->h_seg[d][h].size() * 4);
                dv.d[d][h].arcs = reinterpret_cast<const unsigned *>(base + o_arcs[d][h]);
                dv.d[d][h].seg = reinterpret_cast<const unsigned *>(base + o_seg[d][h]);
                dv.d[d][h].rinfo = reinterpret_cast<const unsigned *>(base + o_rinfo[d]);
                dv.d[d][h].init = reinterpret_cast<const float *>(base + o_init);
            }
        }
        memcpy(img.data() + o_init, f->h_init.data(), f->h_init.size() * 4);
        memcpy(img.data() + o_pp, f->h_pdf_ptr.data(), f->h_pdf_ptr.size() * 4);
        dv.pdf_ptr = reinterpret_cast<const int *>(base + o_pp);
        dv.P1 = P1;
        dv.H = H;
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
int mm_stream_positions(const StreamForm *f) { return f ? std::max(f->npos[0], f->npos[1]) : 0; }

// test aid (host): out = M (x) in through the stream form of direction d exactly as the workgroups of a team walk it -- set by set,
// wave by wave, segment by segment, the records' 20-bit weights, float64 accumulation, wave-wide sums for the rows that own a wave;
// natural log in / out
void mm_stream_eval(const StreamForm *f, int d, const float *in, float *out, double stats[4]) {
    const int S1 = f->S1;
    std::vector<double> lin(static_cast<size_t>(f->npos[d]) + 1, 0.0);
    for (int r = 0; r < S1; ++r) lin[size_t(f->pos[d][size_t(r)])] = std::exp(double(in[r]));
    std::vector<double> res(static_cast<size_t>(f->npos[d]), 0.0);
    long long real = 0, slots = 0, segsum = 0;
    int mx = 0;
    for (int h = 0; h < f->H; ++h) {
        const StreamDev &sd = f->host.d[d][h];
        for (int w = 0; w < kStreamWaves; ++w) {
            size_t slot = size_t(sd.wave_slot0[w]);
            for (int s = sd.wave_seg0[w]; s < sd.wave_seg0[w + 1]; ++s) {
                const unsigned nsl = f->h_seg[d][h][size_t(4 * s)], lg = f->h_seg[d][h][size_t(4 * s + 1)], p0 = f->h_seg[d][h][size_t(4 * s + 2)],
                               n = f->h_seg[d][h][size_t(4 * s + 3)];
                double acc[64] = {0};
                for (unsigned k = 0; k < nsl; ++k)
                    for (unsigned l = 0; l < 64; ++l) {
                        const unsigned long long a = f->h_arcs[d][h][(slot + k) * 64 + l];
                        const unsigned long long wb = (a >> 32) << 32;
                        double wv;
                        memcpy(&wv, &wb, 8);
                        real += (a >> 32) != 0;
                        acc[l] += wv * lin[size_t(unsigned(a) / 4u)];
                    }
                for (unsigned l = 0; l < 64; ++l)  // (the lanes of a row's group add up)
                    if ((l >> lg) < n) res[p0 + (l >> lg)] += acc[l];
                slot += size_t(kStreamChunk) * ((nsl + 1 + kStreamChunk - 1) / kStreamChunk);  // (whole chunks: arcs, padding, the info record)
            }
            mx = std::max(mx, sd.wave_slot0[w + 1] - sd.wave_slot0[w]);
        }
        slots += sd.wave_slot0[kStreamWaves];
        segsum += sd.wave_seg0[kStreamWaves];
    }
    for (int r = 0; r < S1; ++r) {
        const double v = res[size_t(f->pos[d][size_t(r)])];
        out[r] = v > 0 ? float(std::log(v)) : -INFINITY;
    }
    if (stats) {
        stats[0] = double(slots);                         // arc slots per lane, all waves of all sets
        stats[1] = double(segsum);                        // segments
        stats[2] = double(real) / (double(slots) * 64);   // real arcs / arc slots
        stats[3] = double(mx);                            // slots of the most loaded wave (of any set)
    }
}

// ---------------------------------------------------------------------------------------------------------------- device
template <int NJ>
struct StreamLay {  // LDS bytes behind the two vectors (vb each)
    static constexpr unsigned PC = 64u * NJ;
    static constexpr unsigned EM(int par) { return unsigned(par) * 4u * PC; }               // staged emissions [pdf] (+ the phony pdf)
    static constexpr unsigned MS(int par) { return 8u * PC + 8u * unsigned(par); }          // the step's normaliser
    static constexpr unsigned MX(int par) { return 8u * PC + 16u + 8u * unsigned(par); }    // maximum (high dword) of the vector a step writes
    static constexpr unsigned TOTAL = 8u * PC + 64u;
};

__device__ __forceinline__ void lds_atomic_max_u32(unsigned addr, unsigned v) {
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    (void)__hip_atomic_fetch_max((lds_u32 *)(__UINTPTR_TYPE__)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// where the two recursions leave their vectors and offsets (p.ws_alpha / p.ws_c hold the forward direction's; the backward
// direction's follow in the stream batch's extra workspace: RunParams::xbuf / xbuf_d reused as plain pointers)
__device__ __forceinline__ float *stream_rows(const RunParams &p, int dir) { return dir ? p.xbuf : p.ws_alpha; }
__device__ __forceinline__ double *stream_offs(const RunParams &p, int dir) { return dir ? reinterpret_cast<double *>(p.xbuf_d) : p.ws_c; }

// One workgroup = one direction (dir: 0 forward / alpha, 1 backward / beta; a run-time value, uniform) of one utterance.
// The record stream of a wave: per segment its arc records, then ONE info record {0, pdf | pdf-major position << 12} of the rows
// the lanes finish -- everything a finish needs arrives in the ring, nothing is loaded inside it: a load issued in the middle of
// the stream can only be waited for by draining the ring (the counter is in order), 10 drains per wave and frame.
// H > 1 (round 6): a direction of an utterance is a TEAM of H workgroups -- when the batch leaves compute units idle (2 B H <= their
// number).  Workgroup h streams the arcs INTO the rows of set h only (a direction's records are what bounds a frame: 1.3 MB through one
// compute unit's vector memory path for 10 000 states) and finishes them; every finish also stores its value -- the step's tag in the
// sign bit: the values are >= 0 -- into the team's exchange slot of the step (write-through), and after its own arcs every compute
// wave polls its share of the OTHER sets' regions (16-byte sc1 loads: 4 positions) and writes them into the workgroup's LDS vector;
// the step's barrier then releases everybody with the complete vector.  The protocol is the split pair kernels' (mm_kernel_pairs.hip):
// two slots by the parity of the step, the tag flips with every reuse, the areas are zeroed before every call, a poll that outlasts
// RunParams::x_timeout marks the utterance (redo = 2: the item kernel computes it) and waits no more.  Service waves: every workgroup
// of a team stages the emissions and predicts the normalisers for itself -- from the same complete vector: the same bits.
template <int NJ, int H>
__global__ void __launch_bounds__(1024) mm_stream_kernel(RunParams p, int dir_base) {
    extern __shared__ float lds[];
    using L = StreamLay<NJ>;
    constexpr int C = kStreamChunk, K = 6;  // records per chunk, chunks in flight per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = wave == kStreamWaves;
    // (one grid holds both directions -- the first B teams forward, the second B backward -- or dir_base says which; the H workgroups
    // of a team are 8 blocks apart: the dispatcher deals consecutive workgroups to the XCDs in turn, so these share one and its L2)
    const int grp = (int)blockIdx.x / (8 * H), rem = (int)blockIdx.x % (8 * H);
    const int hset = H > 1 ? uni(rem / 8) : 0, team = H > 1 ? grp * 8 + rem % 8 : (int)blockIdx.x;
    if (team >= (dir_base >= 0 ? p.B : 2 * p.B)) return;
    const int DIR = uni(dir_base >= 0 ? dir_base : (team >= p.B ? 1 : 0));
    const int ui = team - (dir_base < 0 && DIR ? p.B : 0);
    const int b = uni(p.order ? p.order[ui] : ui);
    const UttDesc &ud = p.utts[b];
    const StreamPairDev *spd = uni(reinterpret_cast<const StreamPairDev *>(ud.stream));
    const StreamDev &sd = spd->d[DIR][hset];
    const int S1 = uni(sd.rows), P1 = uni(ud.P1), P = P1 - 1, S1p = uni(ud.S1p);
    const unsigned VB = (unsigned)uni(sd.vb), FIX = 2u * VB;
    int len = uni(p.lens ? p.lens[b] : p.N);
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int NF = len + 1;
    const float *Vb = p.V + (long long)b * p.vsb;
    const long long s1p_prefix = ((long long)uni((int)(ud.s1p_prefix >> 32)) << 32) | (unsigned)uni((int)ud.s1p_prefix);
    float *rows = stream_rows(p, DIR) + s1p_prefix * (long long)(p.N + 1);  // [frame - 1][S1p]: the direction's log2 vectors, pdf-major
    double *offs = stream_offs(p, DIR) + (long long)b * (p.N + 2);           // [frame]: their offsets
    const float thr = sd.thr;
    auto frame_of = [&](int t) { return DIR ? NF + 1 - t : t; };
    if (lds_addr_of(lds) != 0u) __builtin_trap();
    if (len == 0) return;  // no frame: no path of length 0 (the combine / finish kernels write ttl = -inf and the zeros)
    for (unsigned q = tid * 4u; q < FIX + L::TOTAL; q += 4096u) ldsw(q, 0.f);
    __syncthreads();
    MM_STAMP_DECL;  // (diagnostic build: [0] arcs + finishes / staging, [1] waiting for the team's rows, [2] at the barrier)

    if (service) {
        // ================= service wave: emissions, normalisers, offsets =================
        RowNorm norm;
        double cum = 0.0;
        // stage the emissions of step t (frame f) into EM(t & 1): log2 values relative to the frame's maximum E for the initial
        // vector (lin = false); for the steps their linear factors 2^(v - E - S) as wide values -- a finish is then p = s * factor,
        // one v_mul_f64, where it went through a logarithm and an exponential until round 5 (pair_agent, LINF)
        auto stage = [&](int t, float S, bool lin) {
            const int f = frame_of(t);
            float v[NJ], E = MM_NINF;
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[j] = em_load_raw(Vb, p.vsn, f, p.N, P, lane + 64 * j);  // (all loads first: one round trip, not NJ)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = lane + 64 * j;
                v[j] = em_value(v[j], f, len, P, q);
                if (q < P) E = max_nc(E, v[j]);
            }
            E = wave_max_rl(E);
            if (!(E > MM_NINF)) E = 0.f;
            bool tiny = false;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int q = lane + 64 * j;
                if (q <= P) {
                    if (lin) {
                        ldswu(FIX + L::EM(t & 1) + 4u * q, w_exp2_hi(v[j] - E - S));
                        tiny = tiny || (v[j] - E - S < MM_WLINF_EMIN && v[j] > MM_NINF);
                    } else {
                        ldsw(FIX + L::EM(t & 1) + 4u * q, v[j] - E);
                    }
                }
            }
            if (tiny) p.redo[b] = 1;
            const double before = cum;
            cum += (double)S + (double)E;
            // the offset that turns the step's stored vector into log2 values: forward, p = s * factor, normaliser and emission
            // included; backward, the sum s itself -- beta~ without the frame's emission -- before either
            if (lane == 0 && hset == 0) offs[f] = DIR ? before : cum;  // (every workgroup of a team has the same numbers: the first stores them)
        };
        if (DIR == 0) stage(1, 0.f, false);  // step 1: alpha_hat (*) lhs[:, 1] needs the emissions of frame 1 (nothing but E is subtracted)
        __syncthreads();  // (1)
        if (2 <= NF) stage(2, 0.f, true);
        __syncthreads();  // (2) the starting vector is in LDS
        MM_STAMP_RESET;
        for (int t = 2; t <= NF; ++t) {
            // the normaliser of step t + 1 from the maximum of the vector of step t - 1 (complete since the last barrier)
            const unsigned mxa = FIX + L::MX((t - 1) & 1);
            const float mx = w_log2_hi(ldsru(mxa));
            if (lane == 0) ldswu(mxa, 0u);
            if (t + 1 <= NF) stage(t + 1, norm.next(mx), true);
            MM_STAMP(0);
            __syncthreads();
            MM_STAMP(2);
        }
    } else {
        // ================= compute waves =================
        __syncthreads();  // (1)
        unsigned vmax = 0u;
        float worst = 0.f;
        typedef __attribute__((address_space(1))) float *gfptr;
        if (DIR == 0) {  // alpha_hat (*) lhs[:, 1]   (src/inference.jl:68)
            const auto rinfo = as_global(uni(sd.rinfo));
            const auto ini = as_global(uni(sd.init));
            const gfptr rows_g = (gfptr)(__UINTPTR_TYPE__)rows;
            for (int i = tid; i < S1; i += 64 * kStreamWaves) {  // (every workgroup of a team: the whole vector)
                const unsigned info = rinfo[i];
                if (info == 0xffffffffu) continue;  // (padding between the sets' regions: stays 0)
                const float v0 = ini[i] + ldsr(FIX + L::EM(1) + 4u * (info & 0xfffu));
                worst = __builtin_fmaxf(worst, __builtin_fmaf(__builtin_fabsf(v0), 0.f, __builtin_fabsf(v0)));
                const unsigned hi = w_exp2_hi(v0);
                vmax = vmax > hi ? vmax : hi;
                ldswu(VB * 1u + 4u * (unsigned)i, hi);
                if (hset == 0) rows_g[(long long)0 * S1p + (info >> 12)] = __builtin_bit_cast(float, hi);  // (the stored vectors: wide values)
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
        // (a scalar base in the GLOBAL address space: global_load, counted by vmcnt alone -- a flat load also counts as an LDS
        // operation, and every wait became a wait for everything)
        typedef unsigned mm_u32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) mm_u32x4 *g_u32x4;
        typedef const __attribute__((address_space(1))) mm_u32x2 *g_u32x2;
        const unsigned *const apw = uni(sd.arcs) + (long long)(slot0 / C) * 384;  // (the wave's first chunk)
        // (the segment table through the scalar cache: constant address space, s_load)
        typedef const __attribute__((address_space(4))) unsigned *seg_cptr;
        const seg_cptr segt = (seg_cptr)(__UINTPTR_TYPE__)uni(sd.seg);
        const int nchunks = nslots / C;  // (whole chunks by construction)
        // the operand pairs of the gathered values: only their high registers are written in the loops below
        double xop[2 * C];
#pragma unroll
        for (int j = 0; j < 2 * C; ++j) {
            xop[j] = 0.0;
            asm volatile("" : "+v"(xop[j]));
        }
        double eop = 0.0;  // ... and of a finish's emission factor
        asm volatile("" : "+v"(eop));
        double wop[2] = {0.0, 0.0};  // ... and of the 1st and the 3rd weight of a chunk
        asm volatile("" : "+v"(wop[0]), "+v"(wop[1]));
        unsigned smin = 0xffffffffu;
        bool cdead = false;  // (teams) a poll of this wave timed out: it waits no more
        MM_STAMP_RESET;
        for (int t = 2; t <= NF; ++t) {
            const unsigned rd = VB * (unsigned)((t - 1) & 1), wr = VB * (unsigned)(t & 1);
            const int f = frame_of(t);
            const unsigned emb = FIX + L::EM(t & 1);
            const gfptr rowf = (gfptr)(__UINTPTR_TYPE__)(rows + (long long)(f - 1) * S1p);
            // (teams) the slot of the step in the team's exchange area, and the step's tag as a sign bit
            typedef __attribute__((address_space(1))) unsigned gu32;
            gu32 *const xslot = H > 1 ? (gu32 *)(__UINTPTR_TYPE__)(p.sx + ((long long)(2 * b + DIR) * 2 + (t & 1)) * p.sx_slot) : nullptr;
            const unsigned xtag = (H > 1 && split_tag(t, 1, 1)) ? 0x80000000u : 0u;
            // the segment the wave is in
            int sg = seg0;
            unsigned s_lg = 0, s_p0 = 0, s_n = 0;
            int remaining = 0;
            auto load_seg = [&](int s) {
                if (s < seg1) {
                    remaining = ((int)segt[4 * s] + C) / C;  // chunks: its arcs + the info record, rounded up
                    s_lg = segt[4 * s + 1];
                    s_p0 = segt[4 * s + 2];
                    s_n = segt[4 * s + 3];
                }
            };
            double acc = 0.0;
            auto finish = [&](unsigned info) {
                double s0 = acc;
                // (the lanes of a row's group: the whole wave, or -- the teams' forms -- 4 or 2 neighbours; s_lg is a scalar)
                if (s_lg == 6) s0 = dwave_sum_rl(s0);
                else if (s_lg) {
                    s0 = dpp_add_d<MM_DPP_XOR1, 0xF>(s0);
                    if (s_lg == 2) s0 = dpp_add_d<MM_DPP_XOR2, 0xF>(s0);
                }
                const bool mine = (lane & ((1 << s_lg) - 1)) == 0 && ((unsigned)lane >> s_lg) < s_n;
                const unsigned posi = s_p0 + ((unsigned)lane >> s_lg);
                // forward: (T' alpha) (*) lhs (src/inference.jl:70-71); backward: T (B (*) lhs) (:106-107), the emission is
                // multiplied in for the next step's product only
                mm_u32x2 eo = __builtin_bit_cast(mm_u32x2, eop);
                eo.y = ldsru(emb + 4u * (info & 0xfffu));
                eop = __builtin_bit_cast(double, eo);
                double p0;
                asm("v_mul_f64 %0, %1, %2" : "=v"(p0) : "v"(s0), "v"(eop));
                if (mine) {
                    // (range check, at the end of the launch: the smallest non-zero sum of the lane by its high dword -- pair_agent)
                    const unsigned sh = __builtin_bit_cast(mm_u32x2, s0).y;
                    smin = sh - 1u < smin ? sh - 1u : smin;
                    const unsigned hi = __builtin_bit_cast(mm_u32x2, p0).y;
                    vmax = vmax > hi ? vmax : hi;
                    ldswu(wr + 4u * posi, hi);
                    if constexpr (H > 1) __hip_atomic_store(xslot + posi, hi | xtag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the row for the team)
                    rowf[info >> 12] = __builtin_bit_cast(float, DIR ? sh : hi);  // the vector that is combined, in the pdf-major numbering
                }
                acc = 0.0;
                ++sg;
                load_seg(sg);
            };
            load_seg(sg);
            static_assert(C == 4, "a chunk is one dwordx4 of weights and one dwordx2 of offsets");
            // K chunks of C records in flight per wave (the ring's slots are static registers: the chunk loop is unrolled K times;
            // the loads are unconditional -- the array ends in K * C records of padding -- so that their number in flight is a
            // constant of the code).  A segment is a whole number of chunks and its info record the LAST record of its last chunk:
            // ONE scalar test per chunk -- with a test per record (is it in the stream? is it an arc?) the frame was bound by the
            // scalar unit.  Chunks are taken in pairs: the gathers of both leave before the first FMA (half the LDS round trips on
            // a wave's chain).
            mm_u32x4 wbuf[K];  // a chunk's 4 weights (high dwords; an info record: {.., .., .., pdf | pdf-major position << 12})
            mm_u32x2 pbuf[K];  // ... and its 4 LDS byte offsets, 16 bits each
            auto load_chunk = [&](int slot, long long q) __attribute__((always_inline)) {
                const unsigned *cb = apw + q * 384;
                wbuf[slot] = ((g_u32x4)(__UINTPTR_TYPE__)cb)[lane];
                pbuf[slot] = ((g_u32x2)(__UINTPTR_TYPE__)(cb + 256))[lane];
            };
#pragma unroll
            for (int kk = 0; kk < K; ++kk) load_chunk(kk, kk);
            for (int c = 0; c < nchunks; c += K) {
#pragma unroll
                for (int k2 = 0; k2 < K; k2 += 2) {
                    if (c + k2 < nchunks) {
                        const bool two = c + k2 + 1 < nchunks;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const unsigned off[C] = {pbuf[k2 + h].x & 0xffffu, pbuf[k2 + h].x >> 16, pbuf[k2 + h].y & 0xffffu, pbuf[k2 + h].y >> 16};
#pragma unroll
                            for (int j = 0; j < C; ++j) {  // (padding and info records gather address 0)
                                mm_u32x2 o = __builtin_bit_cast(mm_u32x2, xop[h * C + j]);
#ifdef MM_STREAM_NOGATHER  // (timing experiment: no LDS gathers)
                                o.y = off[j] | 0x3ff00000u;
#else
                                o.y = ldsru(rd + off[j]);
#endif
                                xop[h * C + j] = __builtin_bit_cast(double, o);
                            }
                        }
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            if (h == 0 || two) {
                                // the weights as float64 operands: {low register: whatever lies there, high: the weight's dword} -- the pairs
                                // (x, y) and (z, w) of the load as they landed give the 2nd and the 4th (their low dwords: another weight's bits,
                                // a fixed perturbation of the graph below 2^-20 relative), the 1st and the 3rd take a move into a pair of their own
                                const mm_u32x4 w = wbuf[k2 + h];
                                mm_u32x2 t0 = __builtin_bit_cast(mm_u32x2, wop[0]), t2 = __builtin_bit_cast(mm_u32x2, wop[1]);
                                t0.y = w.x;
                                t2.y = w.z;
                                wop[0] = __builtin_bit_cast(double, t0);
                                wop[1] = __builtin_bit_cast(double, t2);
                                asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(wop[0]), "v"(xop[h * C + 0]));
                                asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(__builtin_bit_cast(double, mm_u32x2{w.x, w.y})), "v"(xop[h * C + 1]));
                                asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(wop[1]), "v"(xop[h * C + 2]));
                                if (--remaining == 0) {
                                    finish(w.w);
                                } else {
                                    asm("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(__builtin_bit_cast(double, mm_u32x2{w.z, w.w})), "v"(xop[h * C + 3]));
                                }
                            }
                        }
#ifndef MM_STREAM_NOLOAD  // (timing experiment: the ring is loaded once per frame)
#pragma unroll
                        for (int h = 0; h < 2; ++h) load_chunk(k2 + h, (long long)(c + k2 + h + K));  // the slots' next chunks
#endif
                    }
                }
            }
            MM_STAMP(0);
            if constexpr (H > 1) {
                // the rows of the other sets of this step: chunk j (64 granules of 4 positions) of the q-th other set, dealt to the
                // compute waves in turn -- all chunks are polled at the same time by waves that have nothing else to do until the barrier
                typedef unsigned mm_u32x4 __attribute__((ext_vector_type(4)));
                const unsigned tg = xtag >> 31;
                int item = 0;
#pragma unroll
                for (int q = 0; q < H - 1; ++q) {
                    const int g = q < hset ? q : q + 1;
                    const int gb = uni(spd->d[DIR][g].base), gn = uni(spd->d[DIR][g].cnt);
                    for (int j = 0; j * 256 < gn; ++j, ++item) {
                        if (item % kStreamWaves != wave) continue;
                        const unsigned at = (unsigned)gb + 256u * (unsigned)j + 4u * (unsigned)lane;
                        const int left = gn - (256 * j + 4 * lane);  // rows of the set from this granule on (the last granule may hold padding, which nobody stores)
                        bool pend = left > 0;
                        if (cdead || __builtin_amdgcn_ballot_w64(pend) == 0ull) continue;
                        const unsigned off = pend ? 4u * at : 4u * (unsigned)gb;
                        const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
                        const gu32 *const xsrc = uni(xslot);
                        for (;;) {
                            mm_u32x4 v;
                            asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(xsrc) : "memory");
                            if (pend && (v.x >> 31) == tg && (left < 2 || (v.y >> 31) == tg) && (left < 3 || (v.z >> 31) == tg) && (left < 4 || (v.w >> 31) == tg)) {
                                mm_u32x4 w = {v.x & 0x7fffffffu, left < 2 ? 0u : v.y & 0x7fffffffu, left < 3 ? 0u : v.z & 0x7fffffffu, left < 4 ? 0u : v.w & 0x7fffffffu};
                                *(__attribute__((address_space(3))) mm_u32x4 *)(__UINTPTR_TYPE__)(wr + 4u * at) = w;
                                const unsigned m01 = w.x > w.y ? w.x : w.y, m23 = w.z > w.w ? w.z : w.w;
                                vmax = vmax > m01 ? vmax : m01;
                                vmax = vmax > m23 ? vmax : m23;
                                pend = false;
                            }
                            if (__builtin_amdgcn_ballot_w64(pend) == 0ull) break;
                            if (__builtin_amdgcn_s_memrealtime() - tstart > p.x_timeout) {
                                cdead = true;  // the team is not running together: the item kernel computes the utterance
                                if (lane == 0) p.redo[b] = 2;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                }
            }
            {
                const unsigned m = wave_max_u32(vmax);
                if (lane == 0 && m) lds_atomic_max_u32(FIX + L::MX(t & 1), m);
                vmax = 0u;
            }
            MM_STAMP(1);
            __syncthreads();
            MM_STAMP(2);
        }
        // (sums down to 2^-(thr + MM_WLINF_EMIN), factors down to 2^MM_WLINF_EMIN: every non-zero value of a vector >= 2^-thr)
        const unsigned sthr = ((unsigned)(1023 - (int)(thr + MM_WLINF_EMIN < 1.f ? 1.f : thr + MM_WLINF_EMIN)) << 20) - 1u;
        if (__builtin_amdgcn_ballot_w64(worst > thr || smin < sthr) != 0ull && lane == 0) p.redo[b] = 1;
    }
#ifdef MM_STAMPS
    if (p.dbg && lane == 0 && hset == 0)  // [utterance][direction][wave][section]: the first workgroup of every team
        for (int k = 0; k < 8; ++k) p.dbg[(((long long)b * 2 + DIR) * 16 + wave) * 8 + k] = stamp_acc[k];
#endif
}

// C' * (A .* B), the per-frame sums, the division and exp (src/inference.jl:154-160) from the two directions' stored vectors:
// one workgroup per utterance and group of frames, a thread per pdf (its states are a contiguous range of the pdf-major
// numbering both directions stored in).  float64 sums in a fixed order: deterministic.  Leaves log2 Z of every frame (and the
// log2 of the frame's sum of 2^(a~ + b~): the overlap term of mm_pair_finish_kernel's second criterion) for the finish kernel.
constexpr int kStreamCombineFrames = 8;
static __global__ void __launch_bounds__(1024) mm_stream_combine_kernel(RunParams p) {
    __shared__ double part[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const UttDesc &ud = p.utts[b];
    const StreamPairDev *spd = reinterpret_cast<const StreamPairDev *>(ud.stream);
    const int P1 = ud.P1, P = P1 - 1, S1p = ud.S1p;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const long long rbase = ud.s1p_prefix * (long long)(p.N + 1);
    const float *ra = p.ws_alpha + rbase, *rb = p.xbuf + rbase;
    const double *oa = p.ws_c + (long long)b * (p.N + 2), *ob = reinterpret_cast<const double *>(p.xbuf_d) + (long long)b * (p.N + 2);
    double *zf = reinterpret_cast<double *>(p.xps) + (long long)b * 2 * (p.N + 2);  // [frame][2]: log2 Z, overlap term
    const int q0 = tid < P1 ? spd->pdf_ptr[tid] : 0, q1 = tid < P1 ? spd->pdf_ptr[tid + 1] : 0;
    const int f0 = 1 + (int)blockIdx.y * kStreamCombineFrames;
    for (int f = f0; f < f0 + kStreamCombineFrames && f <= len; ++f) {
        const float *a = ra + (long long)(f - 1) * S1p, *bt = rb + (long long)(f - 1) * S1p;
        double s = 0.0;
        for (int q = q0; q < q1; ++q) s += w_from_hi(__builtin_bit_cast(unsigned, a[q])) * w_from_hi(__builtin_bit_cast(unsigned, bt[q]));
        double t = dwave_sum_rl(s);
        __syncthreads();
        if (lane == 0) part[wave] = t;
        __syncthreads();
        t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w];
        const int e = __builtin_amdgcn_frexp_exp(t);
        const float tf = (float)__builtin_amdgcn_ldexp(t, -e);
        const float inv = tf > 0.f ? 1.f / tf : 0.f;
        if (tid < P) p.gamma[(long long)b * p.gsb + (long long)(f - 1) * p.gsn + (long long)tid * p.gsp] = (float)__builtin_amdgcn_ldexp(s, -e) * inv;
        if (tid == 0) {
            const float lt = dlog2(t);
            zf[2 * f] = (double)lt + oa[f] + ob[f];
            zf[2 * f + 1] = (double)lt;
        }
    }
}

// ttl = min over the frames of the per-frame log-normaliser (src/inference.jl:159); zeros beyond the sequence lengths; and what
// a range mark means (mm_dpair_finish_kernel's two criteria on the double's range: the frames' log Z agree, the forward and the
// backward mass overlap within the range less the posterior floor): cleared, or the item kernel computes the utterance again
static __global__ void __launch_bounds__(256) mm_stream_finish_kernel(RunParams p) {
    __shared__ double red[3][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    int len = p.lens ? p.lens[b] : p.N;
    len = len < 0 ? 0 : (len > p.N ? p.N : len);
    const int P = p.utts[b].P1 - 1;
    const double *zf = reinterpret_cast<const double *>(p.xps) + (long long)b * 2 * (p.N + 2);
    double zmin = __builtin_inf(), zmax = -__builtin_inf(), lmin = __builtin_inf();
    for (int f = 1 + tid; f <= len; f += 256) {
        const double z = zf[2 * f], l = zf[2 * f + 1];
        zmin = z < zmin ? z : zmin;
        zmax = z > zmax ? z : zmax;
        if (!(z == z)) zmax = __builtin_inf();  // (an overflow somewhere: inf * 0)
        lmin = l < lmin ? l : lmin;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double a = __shfl_xor(zmin, off), c = __shfl_xor(zmax, off), d = __shfl_xor(lmin, off);
        zmin = a < zmin ? a : zmin;
        zmax = c > zmax ? c : zmax;
        lmin = d < lmin ? d : lmin;
    }
    if ((tid & 63) == 0) {
        red[0][tid >> 6] = zmin;
        red[1][tid >> 6] = zmax;
        red[2][tid >> 6] = lmin;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            zmin = red[0][w] < zmin ? red[0][w] : zmin;
            zmax = red[1][w] > zmax ? red[1][w] : zmax;
            lmin = red[2][w] < lmin ? red[2][w] : lmin;
        }
        p.ttl[b] = (zmin < __builtin_inf()) ? (float)(zmin * (double)MM_LN2) : MM_NINF;  // (no frame: no path of length 0)
        p.pair_zmin[6 * b] = p.pair_zmin[6 * b + 1] = zmin;
        p.pair_zmin[6 * b + 2] = p.pair_zmin[6 * b + 3] = zmax;
        p.pair_zmin[6 * b + 4] = p.pair_zmin[6 * b + 5] = lmin;
        const bool agree = zmin > -__builtin_inf() && zmax < __builtin_inf() && zmax - zmin <= MM_Z_SPREAD_TOL;
        if (p.redo[b] == 1 && agree && lmin >= (double)p.lt_floor - (double)MM_DPAIR_THR_EXTRA) p.redo[b] = 0;
        if (p.redo[b] == 0 && len >= 1 && !(zmax < __builtin_inf())) p.redo[b] = 1;  // (the linear finishes raise no mark on an overflow)
        // (the overlap condition for unmarked utterances too: mm_pair_finish_kernel; all frames without mass: no path)
        if (p.redo[b] == 0 && len >= 1 && !(zmin == -__builtin_inf() && zmax == -__builtin_inf()) && !(lmin >= (double)p.lt_floor - (double)MM_DPAIR_THR_EXTRA)) p.redo[b] = 1;
    }
    const long long gbase = (long long)b * p.gsb;
    for (long long q = tid; q < (long long)(p.N - len) * P; q += 256)
        p.gamma[gbase + (len + q / P) * p.gsn + (q % P) * p.gsp] = 0.f;
}

// both recursions in ONE grid when the chip holds them (2 B H workgroups of one per compute unit), else one grid per direction
template <int NJ, int H>
static int launch_stream_nj(int64_t B, int n_cus, size_t lds, const RunParams &p, hipStream_t st) {
    auto k = mm_stream_kernel<NJ, H>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    auto grid = [&](int64_t teams) { return dim3(unsigned(H > 1 ? (teams + 7) / 8 * 8 * H : teams)); };  // (teams: groups of 8, their workgroups 8 apart)
    if (2 * B * H <= int64_t(n_cus) || H > 1) {  // (a team's workgroups must run together: H > 1 is chosen only when everything fits, mm_stream_pick_h)
        hipLaunchKernelGGL(k, grid(2 * B), dim3(1024), lds, st, p, -1);
        HIP_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k, grid(B), dim3(1024), lds, st, p, 0);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k, grid(B), dim3(1024), lds, st, p, 1);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(mm_stream_combine_kernel, dim3(unsigned(B), unsigned((p.N + kStreamCombineFrames - 1) / kStreamCombineFrames)), dim3(1024), 0, st, p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(mm_stream_finish_kernel, dim3(unsigned(B)), dim3(256), 0, st, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
// workgroups per team for a batch of B utterances on n_cus compute units: as many as leave every workgroup of both directions its own
// compute unit (B = 64 on 256: 2; B <= 32: 4)
int mm_stream_pick_h(int64_t B, int n_cus) { return 2 * B * 4 <= int64_t(n_cus) ? 4 : (2 * B * 2 <= int64_t(n_cus) ? 2 : 1); }
// extra workspace of a stream batch behind the common one: the backward direction's vectors [sum S1p][N + 1] floats and offsets
// [B][N + 2] doubles, the frames' {log2 Z, overlap term} [B][N + 2][2] doubles, (teams) the exchange area [B][2][2 slots][slot] dwords
size_t mm_stream_extra_bytes(int64_t B, int64_t total_s1p, int64_t N, size_t off[4], int H, int max_S1) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    off[0] = 0;
    off[1] = up(size_t(total_s1p) * size_t(N + 1) * 4);
    off[2] = off[1] + up(size_t(B) * size_t(N + 2) * 8);
    off[3] = off[2] + up(size_t(B) * size_t(N + 2) * 16);
    return off[3] + (H > 1 ? up(size_t(B) * 4 * mm_stream_slot(max_S1) * 4) : 0);
}
size_t mm_stream_slot(int max_S1) { return (size_t(max_S1) + 16 + 63) & ~size_t(63); }  // dwords of one exchange slot (positions incl. padding)
size_t mm_stream_exchange_bytes(int64_t B, int H, int max_S1) { return H > 1 ? ((size_t(B) * 4 * mm_stream_slot(max_S1) * 4 + 255) & ~size_t(255)) : 0; }
int mm_launch_stream(int64_t B, int n_cus, int max_S1, int max_P1, int H, const RunParams &p, hipStream_t st) {
    const size_t lds = mm_stream_lds_bytes(max_S1, max_P1);
    if (lds > 160 * 1024 || max_P1 > 1024) return mm_fail(MM_ERR_UNSUPPORTED, "stream kernel: LDS");
    if ((p.N + kStreamCombineFrames - 1) / kStreamCombineFrames > 65535) return mm_fail(MM_ERR_UNSUPPORTED, "stream kernel: more than 524 280 frames");
#define MM_STREAM_CASE(NJ_)                                                          \
    case NJ_:                                                                        \
        if (H == 4) return launch_stream_nj<NJ_, 4>(B, n_cus, lds, p, st);           \
        if (H == 2) return launch_stream_nj<NJ_, 2>(B, n_cus, lds, p, st);           \
        return launch_stream_nj<NJ_, 1>(B, n_cus, lds, p, st);
    switch (stream_nj(max_P1)) {
        MM_STREAM_CASE(2)
        MM_STREAM_CASE(4)
        MM_STREAM_CASE(8)
        default:
            if (H == 4) return launch_stream_nj<16, 4>(B, n_cus, lds, p, st);
            if (H == 2) return launch_stream_nj<16, 2>(B, n_cus, lds, p, st);
            return launch_stream_nj<16, 1>(B, n_cus, lds, p, st);
    }
#undef MM_STREAM_CASE
}

}  // namespace mm
