// mm_engine.hip -- C ABI (include/markovmodels_amd.h) over the gfx950 kernels.
// Handles, host-side compile (packing), workspace management, launches.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <new>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/markovmodels_amd.h"
#include "mm_kernels.hip"
#include "mm_kernel_quad.hip"
#include "mm_kernel_rows.hip"
#include "mm_internal.h"
#include "mm_pack.h"
#include "mm_rows.h"

using namespace mm;

namespace {
thread_local std::string g_err_store;
}
namespace mm {
int mm_fail(int code, const std::string &msg) {
    g_err_store = msg;
    return code;
}
}  // namespace mm

namespace {

int fail(int code, const std::string &msg) { return mm::mm_fail(code, msg); }

int64_t rd_index(const void *p, int bytes, int64_t i) {
    return bytes == 4 ? int64_t(static_cast<const int32_t *>(p)[i]) : static_cast<const int64_t *>(p)[i];
}
float rd_val(const void *p, int bytes, int64_t i) {
    return bytes == 4 ? static_cast<const float *>(p)[i] : float(static_cast<const double *>(p)[i]);
}

struct Csr {
    std::vector<int64_t> rowptr;
    std::vector<int32_t> col;
    std::vector<float> val;
};

Csr transpose(const Csr &a, int64_t n) {
    Csr t;
    t.rowptr.assign(n + 1, 0);
    const int64_t nnz = a.rowptr[n];
    t.col.resize(nnz);
    t.val.resize(nnz);
    for (int64_t k = 0; k < nnz; ++k) t.rowptr[a.col[k] + 1]++;
    for (int64_t i = 0; i < n; ++i) t.rowptr[i + 1] += t.rowptr[i];
    std::vector<int64_t> cur(t.rowptr.begin(), t.rowptr.end() - 1);
    for (int64_t r = 0; r < n; ++r)
        for (int64_t k = a.rowptr[r]; k < a.rowptr[r + 1]; ++k) {
            int64_t d = cur[a.col[k]]++;
            t.col[d] = int32_t(r);
            t.val[d] = a.val[k];
        }
    return t;
}

void sort_rows(Csr &a, int64_t n) {
    std::vector<std::pair<int32_t, float>> tmp;
    for (int64_t r = 0; r < n; ++r) {
        int64_t b = a.rowptr[r], e = a.rowptr[r + 1];
        bool sorted = true;
        for (int64_t k = b + 1; k < e; ++k) sorted &= a.col[k - 1] <= a.col[k];
        if (sorted) continue;
        tmp.clear();
        for (int64_t k = b; k < e; ++k) tmp.push_back({a.col[k], a.val[k]});
        std::stable_sort(tmp.begin(), tmp.end(), [](auto &x, auto &y) { return x.first < y.first; });
        for (int64_t k = b; k < e; ++k) {
            a.col[k] = tmp[k - b].first;
            a.val[k] = tmp[k - b].second;
        }
    }
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

#ifdef MM_STAMPS
static unsigned long long *g_dbg = nullptr;
static size_t g_dbg_n = 0;
#endif

// The quad form of one FSM laid out for KQ quads per lane (mm_pack.h), resident on the device.
struct QuadVariant {  // the quad form of ONE direction of an FSM for one KQ (quads per lane)
    int KQ = 0, dir = 0;
    QuadGraph g;
    std::vector<float> init_f;      // dir 0: alpha_hat in forward numbering
    std::vector<uint16_t> map_bf;   // dir 1: backward position -> forward position (the alpha store's order)
    std::vector<uint16_t> dist;     // internal numbering
    void *blob = nullptr;
    QuadDev qdev;
    const float *d_init_f = nullptr;
    const unsigned short *d_map_bf = nullptr;
};

// The row-lane form of ONE direction of an FSM (mm_rows.h), resident on the device.
struct RowVariant {
    RowGraph g;
    std::vector<float> init;     // forward: alpha_hat by position
    std::vector<uint32_t> ptab;  // wave form: [4 waves][2 segments][4 addresses + info][64 lanes] (wave_pdf_table)
    int pdf_nps = 0;             // ... segments per wave it uses (1 or 2)
    void *blob = nullptr;
    std::shared_ptr<void> arena; // ... or a share of one device allocation for many forms (mm_fsm_create_many): freed with its last user
    RowDev rdev;
};

struct mm_fsm_s {
    int semiring;
    int64_t S1, nnz;
    int32_t P1;
    int S1p;
    Csr mat[2];   // 0: T_hat' (forward), 1: T_hat (backward); engine-domain weights
    Csr qmat[2];  // the same restricted to the useful states (able to reach the final state AND reachable from an
                  // initial state -- or one of a FEW that are not: mm_fsm_create): the kernels compute posteriors,
                  // to which the other states contribute exactly nothing
    Packed packed[2];      // the item forms: packed when first needed (ensure_packed) -- a numerator graph of the wave kernel never needs them
    std::once_flag packed_once, gen_once;
    // what the generic path's copies are built from when first asked for (gen_build): the matrix as it was handed over
    int gen_layout = 0;
    std::vector<int64_t> raw_ptr;
    std::vector<int32_t> raw_col;
    std::vector<double> raw_val;
    bool fast_ok = false;  // the quad kernel's linear path is valid for this FSM
    // ProbSemiring FSMs in float32 (round 6): the SAME graph in the log semiring -- weights = log of the probabilities -- with all the
    // kernel forms of a log FSM.  mm_pdfposteriors_f32 on a batch of such FSMs takes the logarithm of the likelihoods in one pass, runs
    // the fast kernels on the twins and returns ttl as a probability; the generic entry (mm_pdfposteriors_ex) keeps the FSM as given.
    mm_fsm_s *log_twin = nullptr;
    bool export_ok[2] = {false, false};  // the pruned forms (qmat) give the reference's alpha (0) / beta (1) recursion: see mm_fsm_create
    int depth = 0;         // most arcs from an initial state to any (useful) state
    int64_t nquads[2] = {0, 0};
    std::map<int, QuadVariant *> variants;  // by 2 * KQ + direction
    RowVariant *rows[2] = {nullptr, nullptr};  // row-lane forms (built on first use; rows_tried: do not retry)
    bool rows_tried = false;
    bool pairs_tried = false;
    RowVariant *prows[2] = {nullptr, nullptr};  // ... and their pair variants (mm_kernel_pairs.hip)
    // split pair forms (mm_rows.h make_rows_split): [direction][set], for FSMs beyond the registers / LDS of one compute unit
    RowVariant *srows[2][MM_SPLIT_HMAX] = {};
    SplitInfo split;
    int split_tried = 0;  // bit H: the split forms for teams of H have been tried
    RowVariant *wrows[2] = {nullptr, nullptr};  // wave forms (mm_kernel_wave.hip)
    RowVariant *wpend[2] = {nullptr, nullptr};  // ... packed, not yet uploaded (wave_pack)
    bool wave_tried = false, wave_packed = false;
    void *lane_blob = nullptr;                  // lane form (mm_kernel_lane.hip: up to 64 states): the device image, the LaneDev at its start
    bool lane_tried = false;
    StreamForm *stream_h[3] = {nullptr, nullptr, nullptr};  // stream forms (mm_stream.hip: graphs beyond the register-resident forms) for teams of 1, 2, 4 workgroups
    bool stream_tried[3] = {false, false, false};
    RowVariant *vrow = nullptr;                 // Viterbi form (mm_kernel_vit.hip)
    bool vit_tried = false;
    int vit_n4 = 0, vit_n2 = 0;                 // its layout: positions of 4 / of 2 arc slots per wave
    // what the generic path (mm_generic.hip: any semiring, float32 or float64) works on: both matrices and alpha_hat as
    // they were handed over, in double, natural units (log weights for Log / Tropical, probabilities for Prob)
    FsmGenView gen;
    std::vector<int64_t> gen_ptr[2];
    std::vector<int32_t> gen_col[2];
    std::vector<double> gen_val[2], gen_init;
    std::vector<float> init;  // dense alpha_hat, engine domain
    std::vector<int32_t> s2p;
    int device = -1;
    void *dev_blob = nullptr;
    size_t dev_bytes = 0;
    GraphDev gdev[2];
    const float *d_init = nullptr;
    const int *d_s2p = nullptr;
    const int *d_pdf_ptr = nullptr, *d_pdf_rows = nullptr;  // pdf -> states (CSR)
};

// Test/diagnostic switches.  Read from the environment at mm_batch_create (and once per process for the entries that have no
// batch: process_debug_opts), and only when MM_DEBUG is set: the run entry points never call getenv, and no environment variable
// changes what a production process computes or packs.  MM_KERNEL = item | quad | row | pair forces a pdfposteriors kernel,
// MM_KQ / MM_NWAVES / MM_NITEMS force a geometry, MM_NO_XCSR keeps the exact-fallback CSR out of LDS,
// MM_VERBOSE prints the packing statistics.
struct DebugOpts {
    enum { K_AUTO = 0, K_ITEM, K_QUAD, K_ROW, K_PAIR, K_WAVE, K_SPLIT, K_LANE, K_STREAM };
    int kernel = K_AUTO;
    int kq = 0, nwaves = 0, nitems = -1;
    bool no_xcsr = false, verbose = false;
    bool no_redo = false;           // MM_NO_REDO: the exact kernels do not run after the fast ones (what the fast path alone computes)
    bool no_dpair = false;          // MM_NO_DPAIR: no float64 pair kernels (marked utterances go straight to the quad / item kernels)
    bool bankopt = false;           // MM_BANKOPT: the pair forms choose the banks of their rows (RowPackOpts::bank_opt; an experiment, off by default)
    bool no_wpair = false;          // MM_NO_WPAIR: a whole batch on the exact kernels runs the one-utterance float64 kernels, not the wide pair kernels
    bool no_fallback = false;       // MM_NO_FALLBACK: the float64 pair kernels run, the log-domain kernels behind them do not
    int stream_h = 0;               // MM_STREAM_H: workgroups of a team of the stream kernels (0: by the batch size)
    bool no_pdf_halves = false;     // MM_NO_PDF_HALVES: the packer does not deal a segment's rows to its half-waves by the bank pair of their pdf
    int wave_place = -1;            // MM_WAVE_PLACE: placement mode of the wave forms (-1: the packer's default)
    int exact_first = -1;           // MM_EXACT_FIRST=0/1: never / always skip the float32 pair kernels (default: by the last call's marks)
    bool bigv = false;              // MM_BIGV: item / tropical kernels with the state vectors in global memory whatever the size
    int finish_cost = 0;            // MM_FINISH_COST: cost model of the pair forms (0: default)
    int split_q10 = 0;              // MM_SPLIT_Q10: where the pair kernels cut the frames between the agents (1024ths; 0: the engine's choice)
    int x_sleep = 8;                // MM_SPLIT_SLEEP: split kernels, 64-clock units the exchange wave sleeps before a step's first poll
    float group_speed[4] = {0, 0, 0, 0};  // MM_GROUP_SPEED=a,b,c,d
};
static DebugOpts read_debug_opts() {
    DebugOpts d;
    const char *on = getenv("MM_DEBUG");
    if (!on || !*on || !strcmp(on, "0")) return d;
    if (const char *e = getenv("MM_KERNEL"))
        d.kernel = !strcmp(e, "item") ? DebugOpts::K_ITEM : !strcmp(e, "quad") ? DebugOpts::K_QUAD
                 : !strcmp(e, "row") ? DebugOpts::K_ROW : !strcmp(e, "pair") ? DebugOpts::K_PAIR : !strcmp(e, "wave") ? DebugOpts::K_WAVE : !strcmp(e, "split") ? DebugOpts::K_SPLIT : !strcmp(e, "lane") ? DebugOpts::K_LANE : !strcmp(e, "stream") ? DebugOpts::K_STREAM : DebugOpts::K_AUTO;
    if (const char *e = getenv("MM_KQ")) d.kq = atoi(e);
    if (const char *e = getenv("MM_NWAVES")) d.nwaves = atoi(e);
    if (const char *e = getenv("MM_NITEMS")) d.nitems = atoi(e);
    d.no_xcsr = getenv("MM_NO_XCSR") != nullptr;
    d.verbose = getenv("MM_VERBOSE") != nullptr;
    d.bigv = getenv("MM_BIGV") != nullptr;
    d.no_redo = getenv("MM_NO_REDO") != nullptr;
    d.no_dpair = getenv("MM_NO_DPAIR") != nullptr;
    d.no_wpair = getenv("MM_NO_WPAIR") != nullptr;
    d.bankopt = getenv("MM_BANKOPT") != nullptr;
    d.no_fallback = getenv("MM_NO_FALLBACK") != nullptr;
    d.no_pdf_halves = getenv("MM_NO_PDF_HALVES") != nullptr;
    if (const char *e = getenv("MM_STREAM_H")) d.stream_h = atoi(e);
    if (const char *e = getenv("MM_WAVE_PLACE")) d.wave_place = atoi(e);
    if (const char *e = getenv("MM_EXACT_FIRST")) d.exact_first = atoi(e) != 0;
    if (const char *e = getenv("MM_FINISH_COST")) d.finish_cost = atoi(e);
    if (const char *e = getenv("MM_SPLIT_SLEEP")) d.x_sleep = atoi(e);
    if (const char *e = getenv("MM_SPLIT_Q10")) d.split_q10 = atoi(e);
    if (const char *e = getenv("MM_GROUP_SPEED")) sscanf(e, "%f,%f,%f,%f", &d.group_speed[0], &d.group_speed[1], &d.group_speed[2], &d.group_speed[3]);
    return d;
}

// (entries without a batch -- mm_fsm_create_many's packing, the host-only test aids: the switches as the process started with them)
static const DebugOpts &process_debug_opts() {
    static const DebugOpts d = read_debug_opts();
    return d;
}

struct mm_batch_s {
    DebugOpts dbg;
    std::vector<mm_fsm_t> fsms;
    int semiring;
    int64_t B;
    int64_t total_states = 0;
    int64_t total_s1p = 0;
    int max_S1p = 0, max_P1 = 0, max_items = 0;
    int max_quads[2] = {0, 0};  // per direction (0 forward, 1 backward)
    int max_depth = 0;     // deepest FSM of the batch (mm_fsm_s::depth)
    int64_t max_xcsr = 0;  // floats of the largest exact-fallback CSR (rowptr + col + w) of the batch
    int xcsr = 0;          // floats of LDS reserved for it (0: it stays in global memory)
    bool fast_ok = true;
    int geo_kq[2] = {0, 0}, geo_nw[2] = {1, 1};  // quad kernel geometry of the forward and the backward kernel
    bool rows_ok = false;                        // every FSM has its row-lane forms: the row kernels can run
    int row_ka[2] = {0, 0}, row_nwc[2] = {1, 1}, row_slotrows[2] = {0, 0};
    bool pairs_ok = false;                       // one FSM shared by all utterances, in pair form: the pair kernels can run
    int pair_ka = 0, pair_nwc = 1, pair_slotrows = 0;
    bool vit_ok = false;   // every FSM has its Viterbi form: mm_vit_kernel + mm_vit_backtrace_kernel can run
    int vit_n4 = 0, vit_n2 = 0, vit_arcs = 0;
    bool lane_ok = false;  // every FSM has at most 64 states and 64 pdfs: the lane kernel runs (one wave per utterance and direction)
    int lane_S = 0;        // ... the most states of one
    bool lane_redo_wave = false;  // ... and every FSM has its wave forms too: what the lane kernel marks goes to the wave kernel (else: the item kernel)
    bool wave_ok = false;  // every FSM has its wave forms: the wave kernel can run (small graphs that are off the linear paths)
    int wave_nseg = 0;
    int pair_H = 1;        // workgroups per team: 1 = the pair kernels proper, > 1 = the split pair kernels
    int split_s1p = 0;     // floats / 2 of a stored vector of the split kernels (positions of the team's vector, padded)
    bool deterministic = false;  // mm_batch_set_deterministic(): no float atomics in the item kernel
    float lt_floor = -20.f;      // mm_batch_set_posterior_floor(): smallest accepted log2 overlap of a frame (mm_pair_finish_kernel)
    float g_scale = 1.f;         // mm_batch_set_gamma_mode(): gamma_out = g_scale * gamma, or (g_acc) gamma_out += g_scale * gamma
    bool g_acc = false;
    bool keep_marks = false;     // mm_batch_set_mark_policy(MM_MARKS_KEEP): a range mark is never cleared by the finish kernels' two criteria
    float *ws_big = nullptr;  // [B][4 * max_S1p]: state vectors of FSMs beyond the LDS (launch())
    int device = -1;
    int n_cus = 256;  // compute units of the device
    UttDesc *d_utts = nullptr;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    const int *last_redo = nullptr;  // redo marks of the last pdfposteriors call (inside ws; mm_batch_last_redo_count)
    const double *last_z = nullptr;  // ... and the pair kernels' per-utterance normaliser statistics
    GenScratch gen;                  // the generic entry's workspace and descriptors (mm_generic.hip)
    // The float64 exact pair kernels (mm_kernel_dpair.hip) take the utterances the float32 pair kernels mark -- and the whole
    // batch, the float32 kernels skipped, while the inputs are "hard": more than a quarter of the last finished call's
    // utterances were beyond the float32 kernels (stat_host, written by the finish kernels; read without synchronising).
    std::vector<UttDesc> utts_host;     // what d_utts holds
    bool items_resident = true;         // the FSMs' item forms are on the device (a batch of the wave kernel uploads them on first need)
    bool dpair_ok = false;
    int stream_S1 = 0, stream_H = 1;    // ... the most states of one; workgroups of a team (mm_stream_pick_h)
    bool stream_ok = false;             // every FSM has a stream form and nothing faster takes the batch (mm_stream.hip)
    bool wpair_ok = false;              // a whole batch on the exact kernels fits the wide pair kernels (mm_kernel_wpair.hip: two utterances per workgroup)
    bool quad_built = false;            // the FSMs' quad forms exist (not built for batches whose exact path is the float64 kernels)
    int *stat_dev = nullptr;            // {count, ticket, team workgroups whose team sits on ONE XCD, team workgroups} (the last two: mm_batch_team_xcd_stats)
    bool xcd_counting = false;          // ... counted by the team kernels since mm_batch_team_xcd_stats was first called (a measurement aid: off until asked for)
    volatile int *stat_host = nullptr;  // pinned: {count of hard utterances, sequence number of the call that counted, marks of the last alpha / beta export}
    unsigned export_calls[2] = {0, 0};  // alpha / beta exports on this batch (every 32nd tries the linear-domain kernels again)
    int stat_seq = 0;
    int exact_first = -1;               // MM_EXACT_FIRST (under MM_DEBUG): 0 / 1 force the choice, -1: by the statistics
    bool last_exact_first = false;      // what the last call did
    const int *last_redo2 = nullptr;    // marks the float64 kernels left for the log-domain kernels (mm_batch_last_fallback_count)
    // a batch of ProbSemiring FSMs that all have their log twins: the batch of the twins (what mm_pdfposteriors_f32 runs), and the
    // logarithms of the call's likelihoods [B][N][P]
    mm_batch_s *log_twin = nullptr;
    float *prob_logv = nullptr;
    size_t prob_logv_bytes = 0;
};

// Launch geometry of the item kernels: NW waves per workgroup, NI register-resident items per wave
// (mm_kernels.hip, "Register-resident graph"; 8 items keep 16 waves per CU inside the 128-VGPR
// budget), the remaining items are streamed from L2.
struct Geometry {
    int NW, NI;
};

static Geometry pick_geometry(mm_batch_t h) {
    Geometry g{16, 8};
    const int it = h->max_items;
    // one workgroup per CU whatever its size: spread the items over as many waves as there are items
    g.NW = std::max(1, std::min(MM_MAX_WAVES, it + 1));  // + one wave without items: it normalises the frames (FB)
    if (h->dbg.nwaves >= 1 && h->dbg.nwaves <= MM_MAX_WAVES) g.NW = h->dbg.nwaves;
    if (h->dbg.nitems == 0 || h->dbg.nitems == 8) g.NI = h->dbg.nitems;
    if (h->max_S1p > 65534) g.NI = 0;  // (resident items hold 16-bit state indices)
    return g;
}

// item / tropical kernels: `kernel` keeps the state vectors in LDS; `big` is the same kernel with the vectors in global
// memory, for FSMs beyond the LDS (the reference has no size limit: src/linalg.jl:170-181)
static int fsm_to_device(mm_fsm_t f);
// The item forms of a batch that was created without them (a batch of the wave kernel): upload them and refresh the
// utterance descriptors, once, when an entry that runs the item / tropical kernels is first called on the batch.
static int ensure_item_forms(mm_batch_t h, void *stream) {
    if (h->items_resident) return MM_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(MM_ERR_INVALID, "the item forms of this batch are not on the device yet: run the entry once outside a stream capture");
    for (int64_t b = 0; b < h->B; ++b) {
        mm_fsm_t f = h->fsms[size_t(b)];
        int rc = fsm_to_device(f);
        if (rc) return rc;
        UttDesc &u = h->utts_host[size_t(b)];
        u.g[0] = f->gdev[0];
        u.g[1] = f->gdev[1];
        u.init = f->d_init;
        u.s2p = f->d_s2p;
        u.pdf_ptr = f->d_pdf_ptr;
        u.pdf_rows = f->d_pdf_rows;
        h->max_items = std::max(h->max_items, int(std::max(f->packed[0].items.size(), f->packed[1].items.size())));
    }
    // (every other field keeps its value: a kernel still reading the descriptors sees the same bytes)
    HIP_TRY(hipMemcpy(h->d_utts, h->utts_host.data(), sizeof(UttDesc) * size_t(h->B), hipMemcpyHostToDevice));
    h->items_resident = true;
    return MM_OK;
}

template <typename K>
static int launch(K kernel, K big, mm_batch_t h, const RunParams &p0, bool with_stage, int NW, void *stream) {
    {
        const int rc = ensure_item_forms(h, stream);
        if (rc) return rc;
    }
    const int P1p = (h->max_P1 + 3) & ~3;
    RunParams p = p0;
    p.deterministic = h->deterministic ? 1 : 0;
    LdsPlan L = lds_plan(h->max_S1p, P1p, with_stage);
    if (size_t(L.total) * 4 > 160 * 1024 || h->dbg.bigv) {
        if (!h->ws_big) return fail(MM_ERR_UNSUPPORTED, "FSM too large for the LDS and no global-memory vectors were allocated");
        L = lds_plan(0, P1p, with_stage);
        if (size_t(L.total) * 4 > 160 * 1024) return fail(MM_ERR_UNSUPPORTED, "too many pdfs for the LDS: " + std::to_string(h->max_P1));
        kernel = big;
        p.ws_big = h->ws_big;
        p.big_stride = 4ll * h->max_S1p;
    }
    const size_t lds = size_t(L.total) * 4;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(lds)));
    hipLaunchKernelGGL(kernel, dim3(unsigned(h->B)), dim3(64 * NW), lds, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

template <int MODE, bool TROP = false>
static int launch_log(mm_batch_t h, const RunParams &p, void *stream) {
    const Geometry g = pick_geometry(h);
    const bool st = MODE == MODE_FB;
    if (MODE == MODE_FB && g.NI != 0) {  // forward kernel, then backward kernel on the same stream
        int rc = launch(mm_log_kernel<MODE, 8, 1, TROP, false>, mm_log_kernel<MODE, 8, 1, TROP, true>, h, p, st, g.NW, stream);
        if (rc) return rc;
        return launch(mm_log_kernel<MODE, 8, 2, TROP, false>, mm_log_kernel<MODE, 8, 2, TROP, true>, h, p, st, g.NW, stream);
    }
    switch (g.NI) {
        case 0: return launch(mm_log_kernel<MODE, 0, 0, TROP, false>, mm_log_kernel<MODE, 0, 0, TROP, true>, h, p, st, g.NW, stream);
        default: return launch(mm_log_kernel<MODE, 8, 0, TROP, false>, mm_log_kernel<MODE, 8, 0, TROP, true>, h, p, st, g.NW, stream);
    }
}

// The quad kernels (mm_kernel_quad.hip): KQ register-resident quads per lane, NW waves; one kernel per
// direction (PASS 0: forward, 1: backward), each with the geometry its own matrix needs.
static size_t quad_lds_bytes(mm_batch_t h, int dir) {
    const int P1p = (h->max_P1 + 3) & ~3, KQ = h->geo_kq[dir], NW = h->geo_nw[dir];
    const int vl = (h->max_quads[dir] + KQ - 1) / KQ;
    return (size_t(lds_plan_q(h->max_S1p, P1p, std::max(vl, 64 * NW) * KQ).total) + size_t(h->xcsr)) * 4;
}

static bool quad_kernel_usable(mm_batch_t h) {
    if (h->dbg.kernel == DebugOpts::K_ITEM || h->wave_ok || h->lane_ok || !h->quad_built) return false;
    if (!h->fast_ok || h->geo_kq[0] < 1 || h->geo_kq[1] < 1) return false;
    // small deep graphs (numerators): measured on the reference's WSJ numerator graph (depth 165),
    // item kernel 2.3 ms against 2.7 ms; shallow graphs of the same size are 1.6x faster on the quad kernels
    if (h->max_depth >= 64 && h->geo_kq[0] <= 3 && h->geo_kq[1] <= 3 && h->dbg.kernel == DebugOpts::K_AUTO) return false;
    return quad_lds_bytes(h, 0) <= 160 * 1024 && quad_lds_bytes(h, 1) <= 160 * 1024;
}

// (the quad kernels' instances live in mm_quad_tu.hip)
template <int PASS>
static int launch_quad_pass(mm_batch_t h, const RunParams &p, void *stream) {
    QuadLaunch ql;
    ql.B = h->B;
    ql.kq = h->geo_kq[PASS];
    ql.nw = h->geo_nw[PASS];
    ql.max_S1p = h->max_S1p;
    ql.lds = quad_lds_bytes(h, PASS);
    return mm_launch_quad_pass(PASS, ql, p, static_cast<hipStream_t>(stream));
}

// forward kernel, then backward kernel on the same stream: alpha, the per-frame normalisers and log Z travel
// through the workspace
static int launch_quad(mm_batch_t h, const RunParams &p, void *stream) {
    int rc = launch_quad_pass<0>(h, p, stream);
    if (rc) return rc;
    return launch_quad_pass<1>(h, p, stream);
}

static int launch_tropical(mm_batch_t h, const RunParams &p, void *stream) {
    // register-resident items when the whole graph fits 8 items per wave, else streamed
    const Geometry g = pick_geometry(h);
    if (g.NI == 8 && h->max_items <= 8 * MM_MAX_WAVES) {  // as many waves as there is work for (latency), at most 8 items each
        int NW = std::min(MM_MAX_WAVES, std::max(g.NW, (h->max_items + 3) / 4));
        if (h->dbg.nwaves >= g.NW && h->dbg.nwaves <= MM_MAX_WAVES) NW = h->dbg.nwaves;
        return launch(mm_tropical_kernel<8, false>, mm_tropical_kernel<8, true>, h, p, true, NW, stream);
    }
    return launch(mm_tropical_kernel<0, false>, mm_tropical_kernel<0, true>, h, p, true, 16, stream);
}

// The row kernels (mm_kernel_rows.hip): KA register-resident arcs per lane, NWC compute waves + 1 service wave.
static const int kRowKA[] = {24, 40, 42, 44};  // instantiated register windows (48 arcs per lane spill)

template <int KA, int PASS>
static int launch_row_ka(mm_batch_t h, const RunParams &p, void *stream) {
    const size_t lds = row_lds_bytes(MM_ROW_RS, PASS, h->row_slotrows[PASS]);
    if (lds > 160 * 1024) return fail(MM_ERR_UNSUPPORTED, "row kernel: LDS");
    auto kernel = mm_fbr_kernel<KA, MM_ROW_RS, PASS>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(kernel, dim3(unsigned(h->B)), dim3(64 * (h->row_nwc[PASS] + 1)), lds, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int PASS>
static int launch_row_pass(mm_batch_t h, const RunParams &p, void *stream) {
    const int ka = h->row_ka[PASS];
    if (ka <= 24) return launch_row_ka<24, PASS>(h, p, stream);
    if (ka <= 40) return launch_row_ka<40, PASS>(h, p, stream);
    if (ka <= 42) return launch_row_ka<42, PASS>(h, p, stream);
    if (ka <= 44) return launch_row_ka<44, PASS>(h, p, stream);
    return MM_ERR_UNSUPPORTED;
}
static int launch_rows(mm_batch_t h, const RunParams &p, void *stream) {
    int rc = launch_row_pass<0>(h, p, stream);
    if (rc) return rc;
    return launch_row_pass<1>(h, p, stream);
}

// The pair kernels live in a translation unit of their own (mm_pairs_tu.hip)
static int launch_pairs(mm_batch_t h, const RunParams &p, void *stream) {
    PairLaunch pl;
    pl.B = h->B;
    pl.nwc = h->pair_nwc;
    pl.slotrows = h->pair_slotrows;
    pl.max_P1 = h->max_P1;
    pl.pair_ka = h->pair_ka;
    pl.H = h->pair_H;
    pl.small = h->pair_H == 1 && h->max_S1p <= 128;
    return h->pair_H > 1 ? mm_launch_split(pl, p, static_cast<hipStream_t>(stream)) : mm_launch_pairs(pl, p, static_cast<hipStream_t>(stream));
}

namespace {
// one device allocation per FSM: sections appended with 256-byte alignment
struct Blob {
    std::vector<char> host;
    // (mm_fsm_create_many) sizes only (dry), or written straight into a staging buffer shared by many forms (ext)
    bool dry = false;
    char *ext = nullptr;
    size_t ext_size = 0;
    size_t size() const { return (dry || ext) ? ext_size : host.size(); }
    template <class T>
    size_t add(const std::vector<T> &v) {
        const size_t bytes = v.size() * sizeof(T);
        if (dry || ext) {
            const size_t off = align_up(ext_size, 256);
            ext_size = off + bytes;
            if (ext && bytes) memcpy(ext + off, v.data(), bytes);
            return off;
        }
        const size_t off = align_up(host.size(), 256);
        host.resize(off + bytes);
        if (bytes) memcpy(host.data() + off, v.data(), bytes);
        return off;
    }
};
}  // namespace

// (no C++ exception crosses the C boundary: the functions that build host structures in proportion to their input catch
// what the standard library throws -- an allocation that fails, a size beyond max_size -- and report it as a status)
template <class F>
static int no_throw(const char *what, F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(MM_ERR_NOMEM, std::string(what) + ": out of host memory");
    } catch (const std::exception &ex) {
        return fail(MM_ERR_INVALID, std::string(what) + ": " + ex.what());
    } catch (...) {
        return fail(MM_ERR_INVALID, std::string(what) + ": unknown exception");
    }
}

extern "C" {

int mm_abi_version(void) { return MM_ABI_VERSION; }
const char *mm_last_error(void) { return g_err_store.c_str(); }

// the generic path's copies of an FSM (mm_generic.hip): both matrices in double, natural units, rows sorted by column
static void gen_build(mm_fsm_s *f) {
    const int64_t S1 = f->S1, nnz = f->nnz;
    std::vector<int64_t> gptr(f->raw_ptr);
    std::vector<int32_t> gcol(std::move(f->raw_col));
    std::vector<double> gval(std::move(f->raw_val));
    std::vector<std::pair<int32_t, double>> tmp;
    for (int64_t r = 0; r < S1; ++r) {
        tmp.clear();
        for (int64_t k = gptr[r]; k < gptr[r + 1]; ++k) tmp.push_back({gcol[size_t(k)], gval[size_t(k)]});
        std::stable_sort(tmp.begin(), tmp.end(), [](auto &x, auto &y) { return x.first < y.first; });
        for (int64_t k = gptr[r]; k < gptr[r + 1]; ++k) {
            gcol[size_t(k)] = tmp[size_t(k - gptr[r])].first;
            gval[size_t(k)] = tmp[size_t(k - gptr[r])].second;
        }
    }
    std::vector<int64_t> tptr(size_t(S1) + 1, 0);
    std::vector<int32_t> tcol(static_cast<size_t>(nnz));
    std::vector<double> tval(static_cast<size_t>(nnz));
    for (int64_t k = 0; k < nnz; ++k) tptr[size_t(gcol[size_t(k)]) + 1]++;
    for (int64_t i = 0; i < S1; ++i) tptr[size_t(i) + 1] += tptr[size_t(i)];
    std::vector<int64_t> cur(tptr.begin(), tptr.end() - 1);
    for (int64_t r = 0; r < S1; ++r)
        for (int64_t k = gptr[r]; k < gptr[r + 1]; ++k) {
            const int64_t d = cur[size_t(gcol[size_t(k)])]++;
            tcol[size_t(d)] = int32_t(r);
            tval[size_t(d)] = gval[size_t(k)];
        }
    const int gi = f->gen_layout == MM_CSC ? 0 : 1;  // (the given matrix is the forward one when it came as CSC(T_hat))
    f->gen_ptr[gi] = std::move(gptr);
    f->gen_col[gi] = std::move(gcol);
    f->gen_val[gi] = std::move(gval);
    f->gen_ptr[1 - gi] = std::move(tptr);
    f->gen_col[1 - gi] = std::move(tcol);
    f->gen_val[1 - gi] = std::move(tval);
    for (int d = 0; d < 2; ++d) {
        f->gen.ptr[d] = f->gen_ptr[d].data();
        f->gen.col[d] = f->gen_col[d].data();
        f->gen.val[d] = f->gen_val[d].data();
    }
    f->raw_ptr.clear();
    f->raw_ptr.shrink_to_fit();
}
// the item forms of an FSM (mm_pack.h), packed when first needed
static void ensure_packed(mm_fsm_s *f) {
    if (f->semiring == MM_PROB) return;  // (the generic path only)
    std::call_once(f->packed_once, [&]() {
        const float NINF = -std::numeric_limits<float>::infinity();
        for (int d = 0; d < 2; ++d) f->packed[d] = pack_rows(f->S1, f->mat[d].rowptr, f->mat[d].col, f->mat[d].val, f->s2p, NINF);
    });
}

static int fsm_create_impl(int semiring, int64_t S1, int64_t nnz, int layout, int index_bytes, int index_base, int val_bytes,
                           const void *ptr, const void *idx, const void *val, int64_t n_init, const void *init_idx,
                           const void *init_val, const int32_t *state2pdf, int32_t P1, mm_fsm_t *out) {
    if (!out) return fail(MM_ERR_INVALID, "mm_fsm_create: out is NULL");
    *out = nullptr;
    if (semiring != MM_LOG && semiring != MM_TROPICAL && semiring != MM_PROB) return fail(MM_ERR_INVALID, "mm_fsm_create: unknown semiring");
    if (layout != MM_CSC && layout != MM_CSR) return fail(MM_ERR_INVALID, "mm_fsm_create: unknown layout");
    if ((index_bytes != 4 && index_bytes != 8) || (val_bytes != 4 && val_bytes != 8) ||
        (index_base != 0 && index_base != 1))
        return fail(MM_ERR_INVALID, "mm_fsm_create: index_bytes/val_bytes must be 4 or 8, index_base 0 or 1");
    if (S1 < 2 || P1 < 2 || nnz < 0 || n_init < 0) return fail(MM_ERR_DIM, "mm_fsm_create: need S1 >= 2, P1 >= 2");
    if (S1 > (int64_t(1) << 30)) return fail(MM_ERR_UNSUPPORTED, "mm_fsm_create: too many states");
    if (!ptr || !state2pdf || (nnz && (!idx || !val)) || (n_init && (!init_idx || !init_val)))
        return fail(MM_ERR_INVALID, "mm_fsm_create: NULL array");
    if (rd_index(ptr, index_bytes, 0) != index_base || rd_index(ptr, index_bytes, S1) - index_base != nnz)
        return fail(MM_ERR_DIM, "mm_fsm_create: ptr[0]/ptr[S1] do not match index_base/nnz");
    Csr given;
    given.rowptr.resize(S1 + 1);
    given.col.resize(nnz);
    given.val.resize(nnz);
    const float scale = semiring == MM_LOG ? MM_LOG2E : 1.0f;
    for (int64_t i = 0; i <= S1; ++i) {
        given.rowptr[i] = rd_index(ptr, index_bytes, i) - index_base;
        if (given.rowptr[i] < 0 || given.rowptr[i] > nnz || (i && given.rowptr[i] < given.rowptr[i - 1]))
            return fail(MM_ERR_DIM, "mm_fsm_create: ptr is not monotone within [0, nnz]");
    }
    for (int64_t k = 0; k < nnz; ++k) {
        int64_t c = rd_index(idx, index_bytes, k) - index_base;
        if (c < 0 || c >= S1) return fail(MM_ERR_DIM, "mm_fsm_create: state index out of range");
        given.col[k] = int32_t(c);
        given.val[k] = rd_val(val, val_bytes, k) * scale;
    }
    sort_rows(given, S1);
    // MM_CSC(T_hat) is CSR(T_hat') = the forward matrix; MM_CSR(T_hat) the backward one
    Csr other = transpose(given, S1);
    const Csr &fwd = layout == MM_CSC ? given : other;
    const Csr &bwd = layout == MM_CSC ? other : given;

    mm_fsm_s *f = new mm_fsm_s();
    f->semiring = semiring;
    f->S1 = S1;
    f->nnz = nnz;
    f->P1 = P1;
    f->S1p = int((S1 + 1 + 3) / 4 * 4);  // at least one slot beyond the last state (the row kernels' "no row" position)
    f->s2p.resize(S1);
    for (int64_t s = 0; s < S1; ++s) {
        int32_t pdf = state2pdf[s] - index_base;
        if (pdf < 0 || pdf >= P1) {
            delete f;
            return fail(MM_ERR_DIM, "mm_fsm_create: state2pdf out of range");
        }
        f->s2p[s] = pdf;
    }
    if (f->s2p[S1 - 1] != P1 - 1) {
        delete f;
        return fail(MM_ERR_DIM, "mm_fsm_create: the final state must map to the last (phony) pdf");
    }
    const float NINF = -std::numeric_limits<float>::infinity();
    f->init.assign(S1, NINF);
    for (int64_t k = 0; k < n_init; ++k) {
        int64_t s = rd_index(init_idx, index_bytes, k) - index_base;
        if (s < 0 || s >= S1) {
            delete f;
            return fail(MM_ERR_DIM, "mm_fsm_create: initial state out of range");
        }
        f->init[s] = rd_val(init_val, val_bytes, k) * scale;
    }
    {   // the generic path's copy (double, natural units) is built when mm_pdfposteriors_ex first asks for it: gen_build()
        f->gen_layout = layout;
        f->raw_ptr = given.rowptr;
        f->raw_col.resize(static_cast<size_t>(nnz));
        f->raw_val.resize(static_cast<size_t>(nnz));
        for (int64_t k = 0; k < nnz; ++k) {
            f->raw_col[size_t(k)] = int32_t(rd_index(idx, index_bytes, k) - index_base);
            f->raw_val[size_t(k)] = val_bytes == 4 ? double(static_cast<const float *>(val)[k]) : static_cast<const double *>(val)[k];
        }
        const double zero = semiring == MM_PROB ? 0.0 : -std::numeric_limits<double>::infinity();
        f->gen_init.assign(size_t(S1), zero);
        for (int64_t k = 0; k < n_init; ++k) {
            const int64_t s = rd_index(init_idx, index_bytes, k) - index_base;
            f->gen_init[size_t(s)] = val_bytes == 4 ? double(static_cast<const float *>(init_val)[k]) : static_cast<const double *>(init_val)[k];
        }
        f->gen.semiring = semiring;
        f->gen.S1 = S1;
        f->gen.P1 = P1;
        f->gen.init = f->gen_init.data();
        f->gen.s2p = f->s2p.data();
    }
    if (semiring == MM_PROB) {
        // the generic path (mm_pdfposteriors_ex) works on the FSM as given; Float32 FSMs with non-negative weights also get their
        // log-semiring twin for the fast kernels (mm_pdfposteriors_f32)
        bool twin = val_bytes == 4;
        std::vector<float> lv(static_cast<size_t>(nnz)), liv(static_cast<size_t>(n_init));
        for (int64_t k = 0; k < nnz && twin; ++k) {
            const float x = static_cast<const float *>(val)[k];
            twin = x >= 0.f;
            lv[size_t(k)] = std::log(x);  // (log 0 = -inf = zero(LogSemiring))
        }
        for (int64_t k = 0; k < n_init && twin; ++k) {
            const float x = static_cast<const float *>(init_val)[k];
            twin = x >= 0.f;
            liv[size_t(k)] = std::log(x);
        }
        if (twin) {
            mm_fsm_t tw = nullptr;
            // (a graph the log forms refuse stays what it was before round 6: a ProbSemiring FSM of the generic entry, without a twin)
            if (fsm_create_impl(MM_LOG, S1, nnz, layout, index_bytes, index_base, 4, ptr, idx, lv.data(), n_init, init_idx, liv.data(), state2pdf, P1, &tw) == MM_OK)
                f->log_twin = tw;
        }
        *out = f;
        return MM_OK;
    }
    f->mat[0] = fwd;
    f->mat[1] = bwd;
    if (semiring == MM_LOG) {
        // useful states: forward reachable (over out-arcs = rows of T_hat) and co-reachable (over in-arcs)
        std::vector<char> reach(S1, 0), coreach(S1, 0);
        std::vector<int32_t> stack;
        for (int64_t s = 0; s < S1; ++s)
            if (f->init[s] > NINF) {
                reach[s] = 1;
                stack.push_back(int32_t(s));
            }
        while (!stack.empty()) {
            const int32_t i = stack.back();
            stack.pop_back();
            for (int64_t k = bwd.rowptr[i]; k < bwd.rowptr[i + 1]; ++k)
                if (bwd.val[k] > NINF && !reach[bwd.col[k]]) {
                    reach[bwd.col[k]] = 1;
                    stack.push_back(bwd.col[k]);
                }
        }
        coreach[S1 - 1] = 1;
        stack.push_back(int32_t(S1 - 1));
        while (!stack.empty()) {
            const int32_t j = stack.back();
            stack.pop_back();
            for (int64_t k = fwd.rowptr[j]; k < fwd.rowptr[j + 1]; ++k)
                if (fwd.val[k] > NINF && !coreach[fwd.col[k]]) {
                    coreach[fwd.col[k]] = 1;
                    stack.push_back(fwd.col[k]);
                }
        }
        // Which states the kernel forms keep.  A state that cannot reach the final state has beta = zero(K), one that cannot be
        // reached alpha = zero(K): neither contributes to any posterior, and until round 6 both were dropped.  Now a FEW unreachable
        // states that do reach the final state stay (config 3's generator leaves 4 of 2000 units without a predecessor): their rows
        // cost nothing to speak of, their alpha is exactly 0 on every path of the kernels, and the backward forms then hold every
        // state with a non-zero beta -- the beta-recursion export (mm_betarecursion_f32) can run on them.  The alpha-recursion export
        // needs every reachable state to reach the final state (export_on_pairs).
        int64_t n_extra = 0;
        f->export_ok[0] = true;
        for (int64_t s = 0; s < S1; ++s) {
            if (reach[s] && !coreach[s]) f->export_ok[0] = false;
            n_extra += coreach[s] && !reach[s];
        }
        const bool keep_extra = n_extra * 64 <= S1;
        f->export_ok[1] = keep_extra || n_extra == 0;
        std::vector<char> useful(S1, 0);
        for (int64_t s = 0; s < S1; ++s) useful[s] = coreach[s] && (reach[s] || keep_extra);
        for (int d = 0; d < 2; ++d) {
            const Csr &m = d == 0 ? fwd : bwd;
            Csr &q = f->qmat[d];
            q.rowptr.assign(S1 + 1, 0);
            for (int64_t r = 0; r < S1; ++r) {
                if (useful[r])
                    for (int64_t k = m.rowptr[r]; k < m.rowptr[r + 1]; ++k)
                        if (m.val[k] > NINF && useful[m.col[k]]) {
                            q.col.push_back(m.col[k]);
                            q.val.push_back(m.val[k]);
                        }
                q.rowptr[r + 1] = int64_t(q.col.size());
            }
            f->nquads[d] = count_quads(S1, q.rowptr);
        }
        f->fast_ok = quad_range_ok(S1, f->qmat[0].rowptr, f->qmat[0].val, P1) &&
                     quad_range_ok(S1, f->qmat[1].rowptr, f->qmat[1].val, P1);
        // depth = the most arcs any state is away from the initial states.  A deep (left-to-right) graph keeps
        // states alive whose values differ by more than the float range within one frame, so most rows of the
        // quad kernels' linear-domain sums would take the exact fallback: such graphs run on the item kernel.
        std::vector<int32_t> seeds;
        for (int64_t s = 0; s < S1; ++s)
            if (f->init[s] > NINF) seeds.push_back(int32_t(s));
        for (uint16_t d : reach_distance(S1, f->qmat[1].rowptr, f->qmat[1].col, seeds))
            if (d != 0xffff) f->depth = std::max(f->depth, int(d));
    }
    *out = f;
    return MM_OK;
}

static int upload(const Blob &bl, void **out) {
    void *blob = nullptr;
    HIP_TRY(hipMalloc(&blob, bl.host.size() ? bl.host.size() : 256));
    hipError_t e = hipMemcpy(blob, bl.host.data(), bl.host.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return fail(MM_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
    }
    *out = blob;
    return MM_OK;
}

static int fsm_to_device(mm_fsm_t f) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (f->dev_blob && f->device == dev) return MM_OK;
    if (f->dev_blob) return fail(MM_ERR_INVALID, "FSM already resident on another device");
    ensure_packed(f);
    Blob bl;
    size_t o_items[2], o_rows[2], o_slots[2];
    for (int d = 0; d < 2; ++d) {
        o_items[d] = bl.add(f->packed[d].items);
        o_rows[d] = bl.add(f->packed[d].rowinfo);
        o_slots[d] = bl.add(f->packed[d].slots);
    }
    const size_t o_init = bl.add(f->init), o_s2p = bl.add(f->s2p);
    // pdf -> states (CSR): in deterministic mode the item kernel sums a pdf's state posteriors over this list, in this order
    std::vector<int32_t> pdf_ptr(size_t(f->P1) + 1, 0), pdf_rows(size_t(f->S1), 0);
    for (int64_t s = 0; s < f->S1; ++s) ++pdf_ptr[size_t(f->s2p[s]) + 1];
    for (int32_t q = 0; q < f->P1; ++q) pdf_ptr[q + 1] += pdf_ptr[q];
    {
        std::vector<int32_t> fill(pdf_ptr.begin(), pdf_ptr.end() - 1);
        for (int64_t s = 0; s < f->S1; ++s) pdf_rows[size_t(fill[f->s2p[s]]++)] = int32_t(s);
    }
    const size_t o_pptr = bl.add(pdf_ptr), o_prows = bl.add(pdf_rows);
    void *blob = nullptr;
    int rc = upload(bl, &blob);
    if (rc) return rc;
    char *base = static_cast<char *>(blob);
    for (int d = 0; d < 2; ++d) {
        f->gdev[d].items = reinterpret_cast<const ItemMeta *>(base + o_items[d]);
        f->gdev[d].rowinfo = reinterpret_cast<const RowInfo *>(base + o_rows[d]);
        f->gdev[d].slots = reinterpret_cast<const Slot *>(base + o_slots[d]);
        f->gdev[d].n_items = int(f->packed[d].items.size());
        f->gdev[d].n_short = 0;
        for (const auto &im : f->packed[d].items) f->gdev[d].n_short += im.R <= 4 ? 1 : 0;
    }
    f->d_init = reinterpret_cast<const float *>(base + o_init);
    f->d_s2p = reinterpret_cast<const int *>(base + o_s2p);
    f->d_pdf_ptr = reinterpret_cast<const int *>(base + o_pptr);
    f->d_pdf_rows = reinterpret_cast<const int *>(base + o_prows);
    f->dev_blob = blob;
    f->dev_bytes = bl.host.size();
    f->device = dev;
    return MM_OK;
}

// build (once per direction and KQ) and upload the quad form of an FSM
static int quad_variant(mm_fsm_t f, int dir, int KQ, bool verbose, QuadVariant **out) {
    auto it = f->variants.find(2 * KQ + dir);
    if (it != f->variants.end()) {
        *out = it->second;
        return MM_OK;
    }
    QuadVariant *v = new QuadVariant();
    v->KQ = KQ;
    v->dir = dir;
    v->g = make_quads(f->S1, f->qmat[dir].rowptr, f->qmat[dir].col, f->qmat[dir].val, f->s2p, f->P1, dir == 1, KQ);
    if (verbose)
        fprintf(stderr, "[mm] quad form dir %d: KQ %d, %zu quads, %lld arcs, LDS cycles/gather (bank model) %.2f -> %.2f\n",
                dir, KQ, v->g.quads.size(), (long long)f->qmat[dir].rowptr[f->S1], v->g.conflict_before,
                v->g.conflict_after);
    if (dir == 0) {
        v->init_f.resize(f->S1);
        for (int64_t i = 0; i < f->S1; ++i) v->init_f[i] = f->init[v->g.order[i]];
    } else {
        // the forward numbering does not depend on KQ (rows by decreasing quads): take it from any forward form
        std::vector<int32_t> order_f, pos_f;
        quad_order(f->S1, f->qmat[0].rowptr, f->s2p, f->P1, false, order_f, pos_f);
        v->map_bf.resize(f->S1);
        for (int64_t i = 0; i < f->S1; ++i) v->map_bf[i] = uint16_t(pos_f[v->g.order[i]]);
    }
    {   // distances in original numbering: forward = arcs from an initial state (successor lists = rows of
        // T_hat, qmat[1]); backward = arcs to the phony final state (predecessor lists = rows of T_hat', qmat[0])
        std::vector<int32_t> seeds;
        if (dir == 0) {
            for (int64_t s = 0; s < f->S1; ++s)
                if (f->init[s] > -std::numeric_limits<float>::infinity()) seeds.push_back(int32_t(s));
        } else {
            seeds.push_back(int32_t(f->S1 - 1));
        }
        const std::vector<uint16_t> d0 = reach_distance(f->S1, f->qmat[1 - dir].rowptr, f->qmat[1 - dir].col, seeds);
        v->dist.resize(f->S1);
        for (int64_t i = 0; i < f->S1; ++i) v->dist[i] = d0[v->g.order[i]];
    }
    Blob bl;
    const size_t o_q = bl.add(v->g.quads), o_rec = bl.add(v->g.recs), o_ptr = bl.add(v->g.rowptr);
    const size_t o_col = bl.add(v->g.col), o_w = bl.add(v->g.w), o_pse = bl.add(v->g.pdfstart);
    const size_t o_dist = bl.add(v->dist), o_initf = bl.add(v->init_f), o_map = bl.add(v->map_bf);
    int rc = upload(bl, &v->blob);
    if (rc) {
        delete v;
        return rc;
    }
    char *base = static_cast<char *>(v->blob);
    v->qdev.quads = reinterpret_cast<const Quad *>(base + o_q);
    v->qdev.recs = reinterpret_cast<const RowRec *>(base + o_rec);
    v->qdev.rowptr = reinterpret_cast<const int *>(base + o_ptr);
    v->qdev.col = reinterpret_cast<const int *>(base + o_col);
    v->qdev.w = reinterpret_cast<const float *>(base + o_w);
    v->qdev.pdfse = reinterpret_cast<const unsigned short *>(base + o_pse);
    v->qdev.dist = reinterpret_cast<const unsigned short *>(base + o_dist);
    v->qdev.nq = int(v->g.quads.size());
    v->qdev.fpos = v->g.pos[f->S1 - 1];
    v->qdev.ncopy = v->g.ncopy;
    v->qdev.pad = 0;
    v->d_init_f = reinterpret_cast<const float *>(base + o_initf);
    v->d_map_bf = reinterpret_cast<const unsigned short *>(base + o_map);
    f->variants[2 * KQ + dir] = v;
    *out = v;
    return MM_OK;
}

// the device image of a row-lane form: sections of one blob (offsets in RowBlob) ...
struct RowBlob {
    Blob bl;
    size_t o_w, o_a, o_s, o_sc, o_ptr, o_col, o_cw, o_pdf, o_pse, o_init, o_ord, o_pt;
};
static void row_variant_blob(RowVariant *v, bool pad, RowBlob &rb) {
    // (zero rows up to MM_ROW_KA_PAD arc slots: the kernels load their whole register window unconditionally)
    if (pad) {
        v->g.w.resize(size_t(MM_ROW_KA_PAD) * 64 * v->g.NWC, 0.f);
        v->g.addr.resize(size_t(MM_ROW_KA_PAD) * 64 * v->g.NWC, 0u);
    }
    Blob &bl = rb.bl;
    rb.o_w = bl.add(v->g.w), rb.o_a = bl.add(v->g.addr), rb.o_s = bl.add(v->g.slots), rb.o_sc = bl.add(v->g.sched);
    rb.o_ptr = bl.add(v->g.rowptr), rb.o_col = bl.add(v->g.col), rb.o_cw = bl.add(v->g.cw);
    rb.o_pdf = bl.add(v->g.rowpdf), rb.o_pse = bl.add(v->g.pdfse), rb.o_init = bl.add(v->init), rb.o_ord = bl.add(v->g.order);
    rb.o_pt = bl.add(v->ptab);
}
// ... and the form's device descriptor once the blob sits at `base`
static void row_variant_bind(mm_fsm_t f, RowVariant *v, const RowBlob &rb, char *base, float thr) {
    RowDev &d = v->rdev;
    d.w = reinterpret_cast<const float *>(base + rb.o_w);
    d.addr = reinterpret_cast<const unsigned *>(base + rb.o_a);
    d.slots = reinterpret_cast<const unsigned *>(base + rb.o_s);
    d.sched = reinterpret_cast<const RowSched *>(base + rb.o_sc);
    d.rowptr = reinterpret_cast<const int *>(base + rb.o_ptr);
    d.col = reinterpret_cast<const int *>(base + rb.o_col);
    d.cw = reinterpret_cast<const float *>(base + rb.o_cw);
    d.rowpdf = reinterpret_cast<const unsigned short *>(base + rb.o_pdf);
    d.pdfse = reinterpret_cast<const unsigned short *>(base + rb.o_pse);
    d.init = reinterpret_cast<const float *>(base + rb.o_init);
    d.order = reinterpret_cast<const int *>(base + rb.o_ord);
    d.ptab = v->ptab.empty() ? nullptr : reinterpret_cast<const unsigned *>(base + rb.o_pt);
    d.KA = v->g.KA;
    d.NWC = v->g.NWC;
    d.nslotrows = v->g.nslotrows;
    d.fpos = v->g.pos[f->S1 - 1];
    d.rows = int(f->S1);
    d.thr = thr;
}
static int upload_row_variant(mm_fsm_t f, RowVariant *v, int dir, float thr, bool pad = true) {
    RowBlob rb;
    row_variant_blob(v, pad, rb);
    int rc = upload(rb.bl, &v->blob);
    if (rc) return rc;
    row_variant_bind(f, v, rb, static_cast<char *>(v->blob), thr);
    (void)dir;
    return MM_OK;
}

// options of the split pair forms (mm_rows.h make_rows_split; the kernels: mm_kernel_pairs.hip with H > 1)
static void split_pack_opts(const DebugOpts &dbg, RowPackOpts &opt, RowPackOpts &optb, int H = 2) {
    opt = RowPackOpts();
    opt.rs = mm_split_rs(H);
    opt.ka_max = mm_split_ka(H);
    opt.nwc_max = MM_SPLIT_NWC;
    opt.pair = true;
    opt.pdf_halves = !dbg.no_pdf_halves;
    for (float &x : opt.group_speed) x = 1.f;
    if (dbg.finish_cost > 0) opt.finish_cost = dbg.finish_cost;
    opt.ka_choices[0] = mm_split_ka(H);
    optb = opt;
    if (dbg.finish_cost <= 0) optb.finish_cost = 24;
}

// the pair variants of the row-lane forms (same schedule rules; 8-byte positions, one copy of the vector, the other
// direction's numbering in the slot table)
static int pair_variants(mm_fsm_t f, const DebugOpts &dbg, bool *ok) {
    const bool verbose = dbg.verbose;
    *ok = f->prows[0] && f->prows[1];
    if (*ok || f->pairs_tried) return MM_OK;
    f->pairs_tried = true;
    // (the row forms' conditions, with the pair kernels' own pdf capacity: 251 .. 506 pdfs run their NJ = 8 instances, which the row
    // kernels do not have)
    if (f->semiring != MM_LOG || !f->fast_ok || f->P1 > MM_PAIR_P1MAX || (f->S1 + 1) * 4 > MM_ROW_RS) {
        if (verbose) fprintf(stderr, "[mm] pair form: not tried (semiring %d, fast_ok %d, P1 %d, S1 %lld)\n", f->semiring, int(f->fast_ok), int(f->P1), (long long)f->S1);
        return MM_OK;
    }
    RowPackOpts opt;
    opt.rs = MM_ROW_RS;
    opt.ka_max = MM_PAIR_KA;
    opt.pair = true;
    opt.pdf_halves = !dbg.no_pdf_halves;
    for (float &x : opt.group_speed) x = 1.f;  // (the waves of a SIMD progress together: mm_rows.h)
    if (dbg.finish_cost > 0) opt.finish_cost = dbg.finish_cost;
    if (dbg.group_speed[0] > 0)
        for (int i = 0; i < 4; ++i) opt.group_speed[i] = dbg.group_speed[i];
    opt.ka_choices[0] = MM_PAIR_KA;
    opt.bank_opt = dbg.bankopt ? 1 : 0;
    RowVariant *rv[2] = {new RowVariant(), new RowVariant()};
    const std::vector<int32_t> none;
    // (cost of a finish in arcs: measured with cycle stamps on config 3 -- the backward agent's finishes are the dearer
    // ones, its phase B is the longest kernel of a call: 8 / 24 against 8 / 8 shortens it by 4 %)
    RowPackOpts optb = opt;
    if (dbg.finish_cost <= 0) optb.finish_cost = 24;
    if (dbg.group_speed[0] <= 0) {  // (the backward agent's younger waves still lag a little: stamps, -1..2 % with these weights)
        const float sp[4] = {1.08f, 1.04f, 0.97f, 0.92f};
        for (int i = 0; i < 4; ++i) optb.group_speed[i] = sp[i];
    }
    bool fits = make_rows(f->S1, f->qmat[0].rowptr, f->qmat[0].col, f->qmat[0].val, f->s2p, f->P1, false, none, opt, rv[0]->g) &&
                make_rows(f->S1, f->qmat[1].rowptr, f->qmat[1].col, f->qmat[1].val, f->s2p, f->P1, true, rv[0]->g.pos, optb, rv[1]->g);
    if (verbose && !fits) fprintf(stderr, "[mm] pair form: the graph does not fit the register windows (KA %d)\n", MM_PAIR_KA);
    // (arc weights below 2^-60 leave too little of the float range to the values: such graphs run on the other kernels)
    if (fits && std::min(rv[0]->g.wmin_log2, rv[1]->g.wmin_log2) < -60.f) {
        if (verbose) fprintf(stderr, "[mm] pair form: arc weights down to 2^%.0f\n", std::min(rv[0]->g.wmin_log2, rv[1]->g.wmin_log2));
        fits = false;
    }
    if (fits) set_partner(rv[0]->g, rv[1]->g.pos);
    const float thr = fits ? 125.f + std::min(rv[0]->g.wmin_log2, rv[1]->g.wmin_log2) : 0.f;
    int rc = MM_OK;
    for (int dir = 0; dir < 2 && fits && !rc; ++dir) {
        if (verbose)
            fprintf(stderr, "[mm] pair form dir %d: KA %d, %d compute waves, %d segments, arcs/slots %.3f, cost %d..%d, "
                            "LDS cycles/gather (bank model) %.2f -> %.2f\n",
                    dir, rv[dir]->g.KA, rv[dir]->g.NWC, rv[dir]->g.nslotrows - 2, rv[dir]->g.pad_eff, rv[dir]->g.mincost,
                    rv[dir]->g.maxcost, rv[dir]->g.conflict_before, rv[dir]->g.conflict_after);
        if (dir == 0) {
            rv[0]->init.resize(size_t(f->S1));
            for (int64_t i = 0; i < f->S1; ++i) rv[0]->init[i] = f->init[rv[0]->g.order[i]];
        }
        rc = upload_row_variant(f, rv[dir], dir, thr);
    }
    if (!fits || rc) {
        for (RowVariant *x : rv) {
            if (x->blob) (void)hipFree(x->blob);
            delete x;
        }
        return rc;
    }
    f->prows[0] = rv[0];
    f->prows[1] = rv[1];
    *ok = true;
    return MM_OK;
}

// the Viterbi form of a tropical FSM (built once; *ok = false if it does not fit: a row of more than 255 arcs, more than 8
// segments per wave -- 5760 rows at most)
static int vit_variant(mm_fsm_t f, const DebugOpts &dbg, bool *ok) {
    *ok = f->vrow != nullptr;
    if (*ok || f->vit_tried) return MM_OK;
    f->vit_tried = true;
    if (f->semiring != MM_TROPICAL || f->P1 > 256) return MM_OK;
    RowPackOpts opt;
    opt.rs = 65536;
    opt.nwc_max = 15;
    opt.copies = 1;
    opt.acap_force = 4;
    opt.log_weights = true;
    opt.keep_order = true;
    opt.finish_cost = 4;
    for (float &x : opt.group_speed) x = 1.f;
    RowVariant *v = new RowVariant();
    const std::vector<int32_t> none;
    // per wave N4 positions of 4 arc slots and N2 of 2: the cheapest layout the graph fits (the kernel's instances)
    static const int shapes[3][2] = {{1, 5}, {2, 4}, {6, 0}};
    bool fits = false;
    for (const auto &sh : shapes) {
        opt.mix_n4 = sh[0];
        opt.mix_n2 = sh[1];
        opt.ka_max = 4 * sh[0] + 2 * sh[1];
        if (make_rows(f->S1, f->mat[0].rowptr, f->mat[0].col, f->mat[0].val, f->s2p, f->P1, false, none, opt, v->g)) {
            fits = true;
            f->vit_n4 = sh[0];
            f->vit_n2 = sh[1];
            break;
        }
    }
    if (!fits) {
        delete v;
        return MM_OK;
    }
    v->init.resize(size_t(f->S1));
    for (int64_t i = 0; i < f->S1; ++i) v->init[i] = f->init[v->g.order[i]];
    if (dbg.verbose)
        fprintf(stderr, "[mm] Viterbi form: %d waves, %d segments, %d x 4 + %d x 2 arc slots per lane, arcs/slots %.3f\n", v->g.NWC,
                v->g.nslotrows - 2, f->vit_n4, f->vit_n2, v->g.pad_eff);
    int rc = upload_row_variant(f, v, 0, 0.f, false);
    if (rc) {
        delete v;
        return rc;
    }
    f->vrow = v;
    *ok = true;
    return MM_OK;
}

// The per-pdf sums of the wave kernel (C_hat' * (A .* B), src/inference.jl:155) as packed segments of their own: pdf p
// with n_p states gets a group of L_p = pow2(ceil(n_p / 4)) adjacent lanes, each of which reads 4 of the states' values
// (byte addresses relative to the vector u in the numbering of `g`: 4 * position; unused slots read the trash position,
// which holds zero(K)); the lanes of a group combine by a butterfly.  Groups are placed largest first, so every group
// starts at a multiple of its size; a segment is 64 lanes.  Segment s runs on compute wave 3 - s % 4 as its pdf segment
// s / 4.  tab[((w * 2 + j) * 5 + k) * 64 + lane]: k < 4 the addresses, k = 4: 4 * pdf of the group's first lane (the lane
// that stores; others: the trash slot 4 * P1p) | log2(L) << 16.  Returns the segments per wave (1 or 2), 0 if it does not fit.
// the lane form of an FSM (mm_kernel_lane.hip): up to 64 real states, every one on a real pdf, up to 64 pdfs; built once
static int lane_variant(mm_fsm_t f, bool *ok) {
    *ok = f->lane_blob != nullptr;
    if (*ok || f->lane_tried) return MM_OK;
    f->lane_tried = true;
    const int64_t S = f->S1 - 1;
    const int P = f->P1 - 1;
    if (f->semiring != MM_LOG || S < 1 || S > 64 || P < 1 || P > 64) return MM_OK;
    for (int64_t s = 0; s < S; ++s)
        if (f->s2p[size_t(s)] >= P) return MM_OK;  // (a real state on the phony pdf)
    std::vector<double> w0(64 * 64, 0.0), w1(64 * 64, 0.0);
    std::vector<float> init(64, -std::numeric_limits<float>::infinity()), fin(64, -std::numeric_limits<float>::infinity());
    std::vector<int32_t> s2p(64, 0), pdf_ptr(size_t(P) + 1, 0), pdf_states(64, 0);
    const Csr &fwd = f->mat[0], &bwd = f->mat[1];
    for (int64_t j = 0; j < S; ++j)  // forward: row j of T_hat' = the arcs INTO j
        for (int64_t k = fwd.rowptr[size_t(j)]; k < fwd.rowptr[size_t(j) + 1]; ++k)
            if (fwd.col[size_t(k)] < S) w0[size_t(fwd.col[size_t(k)]) * 64 + size_t(j)] += std::exp2(double(fwd.val[size_t(k)]));
    for (int64_t i = 0; i < S; ++i)  // backward: row i of T_hat = the arcs OUT OF i
        for (int64_t k = bwd.rowptr[size_t(i)]; k < bwd.rowptr[size_t(i) + 1]; ++k) {
            const int64_t d = bwd.col[size_t(k)];
            if (d < S) w1[size_t(d) * 64 + size_t(i)] += std::exp2(double(bwd.val[size_t(k)]));
            else if (bwd.val[size_t(k)] > -std::numeric_limits<float>::infinity())  // the arc to the final state: omega_i
                fin[size_t(i)] = fin[size_t(i)] > -std::numeric_limits<float>::infinity()
                                     ? float(std::log2(std::exp2(double(fin[size_t(i)])) + std::exp2(double(bwd.val[size_t(k)]))))
                                     : bwd.val[size_t(k)];
        }
    bool ident = true;
    for (int64_t s = 0; s < S; ++s) {
        init[size_t(s)] = f->init[size_t(s)];
        s2p[size_t(s)] = f->s2p[size_t(s)];
        ident = ident && f->s2p[size_t(s)] == int32_t(s);
        ++pdf_ptr[size_t(f->s2p[size_t(s)]) + 1];
    }
    for (int q = 0; q < P; ++q) pdf_ptr[size_t(q) + 1] += pdf_ptr[size_t(q)];
    {
        std::vector<int32_t> fill(pdf_ptr.begin(), pdf_ptr.end() - 1);
        for (int64_t s = 0; s < S; ++s) pdf_states[size_t(fill[size_t(f->s2p[size_t(s)])]++)] = int32_t(s);
    }
    Blob bl;
    std::vector<char> head(align_up(mm_lane_dev_bytes(), 256), 0);
    const size_t o_head = bl.add(head), o_w0 = bl.add(w0), o_w1 = bl.add(w1), o_init = bl.add(init), o_fin = bl.add(fin), o_s2p = bl.add(s2p),
                 o_pp = bl.add(pdf_ptr), o_ps = bl.add(pdf_states);
    void *blob = nullptr;
    HIP_TRY(hipMalloc(&blob, bl.host.size()));
    char *base = static_cast<char *>(blob);
    mm_lane_dev_fill(bl.host.data() + o_head, reinterpret_cast<const double *>(base + o_w0), reinterpret_cast<const double *>(base + o_w1),
                     reinterpret_cast<const float *>(base + o_init), reinterpret_cast<const float *>(base + o_fin),
                     reinterpret_cast<const int *>(base + o_s2p), reinterpret_cast<const int *>(base + o_pp),
                     reinterpret_cast<const int *>(base + o_ps), int(S), P, ident ? 1 : 0);
    if (hipMemcpy(blob, bl.host.data(), bl.host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(blob);
        return fail(MM_ERR_HIP, "lane form: upload failed");
    }
    f->lane_blob = blob;
    *ok = true;
    return MM_OK;
}

// the stream form of an FSM for teams of H workgroups (mm_stream.hip): built once per H; *ok = false if the graph does not fit it
static int stream_hidx(int H) { return H == 4 ? 2 : (H == 2 ? 1 : 0); }
static int stream_variant(mm_fsm_t f, int H, bool *ok) {
    const int k = stream_hidx(H);
    *ok = f->stream_h[k] != nullptr;
    if (f->stream_h[k] || f->stream_tried[k]) return MM_OK;
    f->stream_tried[k] = true;
    if (f->semiring != MM_LOG) return MM_OK;
    const int64_t *rp[2] = {f->mat[0].rowptr.data(), f->mat[1].rowptr.data()};
    const int32_t *cl[2] = {f->mat[0].col.data(), f->mat[1].col.data()};
    const float *vl[2] = {f->mat[0].val.data(), f->mat[1].val.data()};
    int dev = -1;
    const bool have_dev = hipGetDevice(&dev) == hipSuccess;
    int rc = mm_stream_build(f->S1, f->P1, rp, cl, vl, f->init.data(), f->s2p.data(), have_dev, H, &f->stream_h[k]);
    if (rc) return rc;
    *ok = f->stream_h[k] != nullptr;
    return MM_OK;
}

static int wave_pdf_table(const RowGraph &g, const std::vector<int32_t> &s2p, int64_t S1, int32_t P1, std::vector<uint32_t> &tab) {
    std::vector<std::vector<uint32_t>> src(static_cast<size_t>(P1));
    std::vector<std::pair<int32_t, int32_t>> bypos;  // (position, pdf): a fixed order of the states of a pdf
    for (int64_t r = 0; r < S1; ++r) bypos.emplace_back(g.pos[size_t(r)], s2p[size_t(r)]);
    std::sort(bypos.begin(), bypos.end());
    for (const auto &pp : bypos) src[size_t(pp.second)].push_back(4u * uint32_t(pp.first));
    struct Item {
        int32_t pdf, lg;
    };
    std::vector<Item> items;
    for (int32_t p = 0; p < P1; ++p) {
        const size_t n = src[size_t(p)].size();
        if (n == 0) continue;  // (no state emits it: its sum stays zero(K), the buffers start that way)
        int lg = 0;
        while ((size_t(4) << lg) < n) ++lg;
        if (lg > 6) return 0;
        items.push_back({p, lg});
    }
    std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.lg > b.lg; });
    const uint32_t trash_src = 4u * uint32_t(S1), trash_pdf = 4u * uint32_t((P1 + 3) & ~3);
    tab.assign(size_t(4) * 2 * 5 * 64, 0u);
    for (int w = 0; w < 4; ++w)
        for (int j = 0; j < 2; ++j) {
            for (int k = 0; k < 4; ++k)
                for (int l = 0; l < 64; ++l) tab[size_t(((w * 2 + j) * 5 + k) * 64 + l)] = trash_src;
            for (int l = 0; l < 64; ++l) tab[size_t(((w * 2 + j) * 5 + 4) * 64 + l)] = trash_pdf;
        }
    int seg = 0, lane = 0;
    for (const Item &it : items) {
        const int L = 1 << it.lg;
        if (lane + L > 64) {
            ++seg;
            lane = 0;
        }
        if (seg >= 8) return 0;
        const int w = 3 - seg % 4, j = seg / 4;  // (the largest groups to the last wave: the first has the widest rows of the graph)
        const std::vector<uint32_t> &sv = src[size_t(it.pdf)];
        for (int i = 0; i < L; ++i) {
            for (int k = 0; k < 4; ++k) {
                const size_t q = size_t(4 * i + k);
                if (q < sv.size()) tab[size_t(((w * 2 + j) * 5 + k) * 64 + lane + i)] = sv[q];
            }
            tab[size_t(((w * 2 + j) * 5 + 4) * 64 + lane + i)] = (i == 0 ? 4u * uint32_t(it.pdf) : trash_pdf) | (uint32_t(it.lg) << 16);
        }
        lane += L;
    }
    const int nseg = items.empty() ? 1 : seg + 1;
    return nseg <= 4 ? 1 : 2;
}

// host part of the wave forms (no device call: mm_batch_create runs it for the FSMs of a batch on several host threads --
// an LF-MMI step brings a batch of numerator graphs that were never seen before, and packing them one after the other
// cost 150 ms per 128 graphs against 0.4 ms of kernel time).  Leaves the packed forms in f->wpend, or nothing if the FSM
// does not fit them.
static void wave_pack(mm_fsm_t f) {
    if (f->wrows[0] || f->wave_tried || f->wave_packed) return;
    f->wave_packed = true;
    if (f->semiring != MM_LOG || f->P1 > 256 || f->qmat[0].rowptr.empty()) return;
    RowPackOpts opt;
    opt.rs = MM_WAVE_RS;
    opt.nwc_max = MM_WAVE_WAVES;
    opt.ka_max = 16;  // (4 segments of 4 slots per wave)
    opt.copies = 1;
    opt.acap_force = 4;
    opt.seg_stride = 4;
    opt.log_weights = true;
    opt.want_partner = true;
    opt.spread_pdf = true;
    opt.finish_cost = 4;
    for (float &x : opt.group_speed) x = 1.f;
    // (greedy placement without the local search: the wave kernel is bound by its waves' latency chains, not by LDS cycles --
    // 0.434 -> 0.437 ms on the WSJ numerators x 128 -- and the search is half the host time of packing a small graph)
    opt.place = 1;
    opt.naive_stats = false;
    opt.q_positions = false;  // (the wave kernel sums the posteriors per pdf through its own tables: wave_pdf_table)
    if (process_debug_opts().wave_place >= 0) opt.place = process_debug_opts().wave_place;
    // (owned until they are handed to the FSM: the packer may throw -- an allocation that fails)
    std::unique_ptr<RowVariant> rv[2] = {std::make_unique<RowVariant>(), std::make_unique<RowVariant>()};
    const std::vector<int32_t> none;
    bool fits = make_rows(f->S1, f->qmat[0].rowptr, f->qmat[0].col, f->qmat[0].val, f->s2p, f->P1, false, none, opt, rv[0]->g) &&
                make_rows(f->S1, f->qmat[1].rowptr, f->qmat[1].col, f->qmat[1].val, f->s2p, f->P1, true, rv[0]->g.pos, opt, rv[1]->g);
    for (int d = 0; d < 2 && fits; ++d) {
        rv[d]->pdf_nps = wave_pdf_table(rv[d]->g, f->s2p, f->S1, f->P1, rv[d]->ptab);
        fits = rv[d]->pdf_nps > 0;
    }
    if (!fits) return;
    set_partner(rv[0]->g, rv[1]->g.pos);
    rv[0]->init.resize(size_t(f->S1));
    for (int64_t i = 0; i < f->S1; ++i) rv[0]->init[i] = f->init[rv[0]->g.order[i]];
    f->wpend[0] = rv[0].release();
    f->wpend[1] = rv[1].release();
}

// the wave forms of an FSM (built once; *ok = false if it does not fit them: more than 16 segments of 64 rows, ...)
static int wave_variants(mm_fsm_t f, const DebugOpts &dbg, bool *ok) {
    *ok = f->wrows[0] && f->wrows[1];
    if (*ok || f->wave_tried) return MM_OK;
    wave_pack(f);
    f->wave_tried = true;
    RowVariant *rv[2] = {f->wpend[0], f->wpend[1]};
    f->wpend[0] = f->wpend[1] = nullptr;
    if (!rv[0]) return MM_OK;
    int rc = MM_OK;
    for (int d = 0; d < 2 && !rc; ++d) {
        if (dbg.verbose)
            fprintf(stderr, "[mm] wave form dir %d: %d segments, arcs/slots %.3f, LDS cycles/gather (bank model) %.2f -> %.2f\n", d,
                    rv[d]->g.nslotrows - 2, rv[d]->g.pad_eff, rv[d]->g.conflict_before, rv[d]->g.conflict_after);
        rc = upload_row_variant(f, rv[d], d, 0.f, false);
    }
    if (rc) {
        for (RowVariant *x : rv) {
            if (x->blob) (void)hipFree(x->blob);
            delete x;
        }
        return rc;
    }
    f->wrows[0] = rv[0];
    f->wrows[1] = rv[1];
    *ok = true;
    return MM_OK;
}

// the split pair forms of an FSM for teams of H workgroups (built once; *ok = false if it does not fit them)
static int split_variants(mm_fsm_t f, const DebugOpts &dbg, int H, bool *ok) {
    *ok = f->split.H == H && f->srows[0][0] != nullptr;
    if (*ok || f->srows[0][0] != nullptr || (f->split_tried >> H) & 1) return MM_OK;  // (one team size per FSM: the first that fits)
    f->split_tried |= 1 << H;
    if (f->semiring != MM_LOG || !f->fast_ok || mm_pair_nj(f->P1, H) == 0 || H > MM_SPLIT_HMAX) return MM_OK;
    RowPackOpts opt, optb;
    split_pack_opts(dbg, opt, optb, H);
    std::vector<RowGraph> gs;
    SplitInfo info;
    if (!make_rows_split(H, f->S1, f->qmat[0].rowptr, f->qmat[0].col, f->qmat[0].val, f->qmat[1].rowptr, f->qmat[1].col,
                         f->qmat[1].val, f->s2p, f->P1, opt, optb, gs, info)) {
        if (dbg.verbose) fprintf(stderr, "[mm] split forms for teams of %d: the graph does not fit them\n", H);
        return MM_OK;
    }
    float wmin = 0.f;
    for (const RowGraph &g : gs) wmin = std::min(wmin, g.wmin_log2);
    if (wmin < -60.f) return MM_OK;  // (as for the row forms: too little of the float range would be left to the values)
    for (int h = 0; h < H; ++h)
        if (size_t(info.count[h] + 1) * 8 > size_t(mm_split_rsh(H))) {
            if (dbg.verbose) fprintf(stderr, "[mm] split forms for teams of %d: set %d has %d rows\n", H, h, info.count[h]);
            return MM_OK;
        }
    const float NINF = -std::numeric_limits<float>::infinity();
    int rc = MM_OK;
    for (int d = 0; d < 2 && !rc; ++d) {
        // rowpdf / init by position of the TEAM's vector (0xffff / -inf at the alignment padding between the regions)
        std::vector<uint16_t> rowpdf_g(size_t(info.total) + 1, uint16_t(0xffff));
        std::vector<float> init_g(size_t(info.total) + 1, NINF);
        for (int64_t r = 0; r < f->S1; ++r) {
            rowpdf_g[size_t(info.gpos[d][size_t(r)])] = uint16_t(f->s2p[size_t(r)]);
            if (d == 0) init_g[size_t(info.gpos[d][size_t(r)])] = f->init[size_t(r)];
        }
        for (int h = 0; h < H && !rc; ++h) {
            RowVariant *v = new RowVariant();
            v->g = std::move(gs[size_t(d * H + h)]);
            if (dbg.verbose)
                fprintf(stderr, "[mm] split form dir %d set %d/%d: %d rows at %d, KA %d, %d compute waves, %d segments, arcs/slots %.3f, "
                                "cost %d..%d, LDS cycles/gather (bank model) %.2f -> %.2f\n",
                        d, h, H, info.count[h], info.base[h], v->g.KA, v->g.NWC, v->g.nslotrows - 2, v->g.pad_eff, v->g.mincost,
                        v->g.maxcost, v->g.conflict_before, v->g.conflict_after);
            v->g.rowpdf = rowpdf_g;
            v->init = init_g;
            rc = upload_row_variant(f, v, d, 125.f + wmin);
            v->rdev.rows = info.total;
            v->rdev.fpos = info.gpos[d][size_t(f->S1 - 1)];
            f->srows[d][h] = v;
        }
    }
    if (rc) {
        for (int d = 0; d < 2; ++d)
            for (int h = 0; h < H; ++h)
                if (f->srows[d][h]) {
                    if (f->srows[d][h]->blob) (void)hipFree(f->srows[d][h]->blob);
                    delete f->srows[d][h];
                    f->srows[d][h] = nullptr;
                }
        return rc;
    }
    f->split = std::move(info);
    *ok = true;
    return MM_OK;
}

// build (once) and upload the row-lane forms of both directions of an FSM; *ok = false if it does not fit them
static int row_variants(mm_fsm_t f, bool verbose, bool *ok) {
    *ok = f->rows[0] && f->rows[1];
    if (f->rows_tried) return MM_OK;
    f->rows_tried = true;
    if (f->semiring != MM_LOG || !f->fast_ok || f->P1 > 250 || (f->S1 + 1) * 4 > MM_ROW_RS) return MM_OK;
    RowPackOpts opt;
    opt.rs = MM_ROW_RS;
    opt.ka_max = kRowKA[sizeof(kRowKA) / sizeof(kRowKA[0]) - 1];
    for (size_t i = 0; i < sizeof(kRowKA) / sizeof(kRowKA[0]); ++i) opt.ka_choices[i] = kRowKA[i];
    RowVariant *rv[2] = {new RowVariant(), new RowVariant()};
    const std::vector<int32_t> none;
    // (arc weights below 2^-60 leave too little of the float range to the values: such graphs run on the other kernels)
    bool fits = make_rows(f->S1, f->qmat[0].rowptr, f->qmat[0].col, f->qmat[0].val, f->s2p, f->P1, false, none, opt, rv[0]->g) &&
                make_rows(f->S1, f->qmat[1].rowptr, f->qmat[1].col, f->qmat[1].val, f->s2p, f->P1, true, rv[0]->g.pos, opt, rv[1]->g);
    fits = fits && std::min(rv[0]->g.wmin_log2, rv[1]->g.wmin_log2) >= -60.f;
    if (!fits) {
        delete rv[0];
        delete rv[1];
        return MM_OK;
    }
    for (int dir = 0; dir < 2; ++dir) {
        RowVariant *v = rv[dir];
        if (verbose)
            fprintf(stderr, "[mm] row form dir %d: KA %d, %d compute waves, %d segments, arcs/slots %.3f, cost %d..%d, "
                            "LDS cycles/gather (bank model) %.2f -> %.2f\n",
                    dir, v->g.KA, v->g.NWC, v->g.nslotrows - 2, v->g.pad_eff, v->g.mincost, v->g.maxcost, v->g.conflict_before,
                    v->g.conflict_after);
        if (verbose)
            for (int w = 0; w < v->g.NWC; ++w) {
                const RowSched &sc = v->g.sched[w];
                int last = 0;
                for (int k = 0; k < 64; ++k)
                    if ((sc.endmask >> k) & 1) last = k;
                fprintf(stderr, "[mm]   wave %2d: %u segments, %d arcs, lg %llx\n", w, sc.nslots & 0xffffu,
                        2 * (last + 1 - int(sc.nslots >> 16)), (unsigned long long)sc.lg);
            }
        if (dir == 0) {
            v->init.resize(f->S1);
            for (int64_t i = 0; i < f->S1; ++i) v->init[i] = f->init[v->g.order[i]];
        }
        int rc = upload_row_variant(f, v, dir, 125.f + std::min(rv[0]->g.wmin_log2, rv[1]->g.wmin_log2));
        if (rc) {
            for (RowVariant *x : rv) {
                if (x->blob) (void)hipFree(x->blob);
                delete x;
            }
            return rc;
        }
    }
    f->rows[0] = rv[0];
    f->rows[1] = rv[1];
    *ok = true;
    return MM_OK;
}

int mm_fsm_destroy(mm_fsm_t f) {
    if (!f) return MM_OK;
    if (f->log_twin) (void)mm_fsm_destroy(f->log_twin);
    for (void *&d : f->gen.dev)
        if (d) {
            mm_generic_free(d);
            d = nullptr;
        }
    if (f->dev_blob) (void)hipFree(f->dev_blob);
    if (f->lane_blob) (void)hipFree(f->lane_blob);
    for (StreamForm *sf : f->stream_h) mm_stream_free(sf);
    for (auto &kv : f->variants) {
        if (kv.second->blob) (void)hipFree(kv.second->blob);
        delete kv.second;
    }
    auto drop = [](RowVariant *rv) {
        if (!rv) return;
        if (rv->blob) (void)hipFree(rv->blob);
        delete rv;
    };
    for (RowVariant *rv : {f->rows[0], f->rows[1], f->prows[0], f->prows[1], f->wrows[0], f->wrows[1], f->wpend[0], f->wpend[1], f->vrow}) drop(rv);
    for (int d = 0; d < 2; ++d)
        for (int hh = 0; hh < MM_SPLIT_HMAX; ++hh) drop(f->srows[d][hh]);  // (teams of up to MM_SPLIT_HMAX sets)
    delete f;
    return MM_OK;
}

int mm_fsm_info(mm_fsm_t f, int64_t *S1, int64_t *nnz, int32_t *P1, int64_t packed_slots[2], int64_t packed_items[2]) {
    if (!f) return fail(MM_ERR_INVALID, "mm_fsm_info: NULL handle");
    if (S1) *S1 = f->S1;
    if (nnz) *nnz = f->nnz;
    if (P1) *P1 = f->P1;
    for (int d = 0; d < 2; ++d) {
        ensure_packed(f);
        if (packed_slots) packed_slots[d] = f->packed[d].n_slot_rows * 64;
        if (packed_items) packed_items[d] = int64_t(f->packed[d].items.size());
    }
    return MM_OK;
}

int mm_debug_packed_product(mm_fsm_t f, int direction, const float *in, float *out, int32_t *argmax) {
    if (!f || !in || !out || direction < 0 || direction > 1) return fail(MM_ERR_INVALID, "mm_debug_packed_product");
    std::vector<float> x(f->S1);
    const float s = f->semiring == MM_LOG ? MM_LOG2E : 1.0f;
    for (int64_t i = 0; i < f->S1; ++i) x[i] = in[i] * s;
    ensure_packed(f);
    eval_packed(f->packed[direction], f->semiring, x.data(), out, argmax, f->S1);
    if (f->semiring == MM_LOG)
        for (int64_t i = 0; i < f->S1; ++i) out[i] *= MM_LN2;
    return MM_OK;
}

int mm_debug_reach_distance(mm_fsm_t f, int direction, int32_t *out) {
    if (!f || !out || direction < 0 || direction > 1) return fail(MM_ERR_INVALID, "mm_debug_reach_distance: bad argument");
    if (f->qmat[0].rowptr.empty()) return fail(MM_ERR_INVALID, "mm_debug_reach_distance: log-semiring FSMs only");
    std::vector<int32_t> seeds;
    if (direction == 0) {
        for (int64_t s = 0; s < f->S1; ++s)
            if (f->init[s] > -std::numeric_limits<float>::infinity()) seeds.push_back(int32_t(s));
    } else {
        seeds.push_back(int32_t(f->S1 - 1));
    }
    const std::vector<uint16_t> d = reach_distance(f->S1, f->qmat[1 - direction].rowptr, f->qmat[1 - direction].col, seeds);
    for (int64_t s = 0; s < f->S1; ++s) out[s] = d[s] == 0xffff ? -1 : int32_t(d[s]);
    return MM_OK;
}

int mm_debug_quad_product(mm_fsm_t f, int direction, int KQ, const float *in, float *out, double stats[4]) {
    if (!f || !in || !out || direction < 0 || direction > 1 || KQ < 1 || KQ > 32)
        return fail(MM_ERR_INVALID, "mm_debug_quad_product: bad argument");
    if (f->semiring != MM_LOG) return fail(MM_ERR_INVALID, "mm_debug_quad_product: log-semiring FSMs only");
    const Csr &m = f->mat[direction];
    QuadGraph g = make_quads(f->S1, m.rowptr, m.col, m.val, f->s2p, f->P1, direction == 1, KQ);
    const int64_t S1 = f->S1, nq = int64_t(g.quads.size());
    // p = 2^(in * log2e - max) in the internal numbering
    float mx = -std::numeric_limits<float>::infinity();
    for (int64_t s = 0; s < S1; ++s) mx = std::max(mx, in[s] * MM_LOG2E);
    if (!(mx > -std::numeric_limits<float>::infinity())) mx = 0.f;
    std::vector<float> p(S1), qs(nq, 0.f);
    for (int64_t i = 0; i < S1; ++i) p[i] = std::exp2(in[g.order[i]] * MM_LOG2E - mx);
    const int64_t lanes = (nq + KQ - 1) / KQ;
    for (int64_t l = 0; l < lanes; ++l) {
        const uint32_t mask = g.quads[size_t(l * KQ)].mask;
        float run = 0.f;
        for (int j = 0; j < KQ && l * KQ + j < nq; ++j) {
            const Quad &q = g.quads[size_t(l * KQ + j)];
            float s = 0.f;
            const bool cont = std::signbit(q.wl[0]);
            if (cont != bool((mask >> j) & 1u)) return fail(MM_ERR_INVALID, "mm_debug_quad_product: sign flags and lane mask disagree");
            for (int k = 0; k < 4; ++k)
                s = std::fmaf(std::fabs(q.wl[k]), p[(q.off[k] / 4) % quad_pstride(f->S1p, g.ncopy)], s);
            run = cont ? run + s : s;
            qs[size_t(l * KQ + j)] = run;
        }
    }
    for (int64_t i = 0; i < S1; ++i) {
        const RowRec &r = g.recs[i];
        float acc = 0.f;
        if (r.qe) {  // the kernel's row_total(): qs2 = {0, 0, qs...}
            auto at = [&](int idx) { return idx < MM_QS_PAD ? 0.f : qs[size_t(idx - MM_QS_PAD)]; };
            if (r.i2 == MM_ROW_LONG) {
                acc = at(r.qe);
                for (int q = r.i1; q < r.qe; q += KQ) acc += at(q);
            } else {
                acc = at(r.qe) + (at(r.i1) + at(r.i2));
            }
        }
        out[g.order[i]] = (std::log2(acc) + mx) * MM_LN2;
    }
    if (stats) {
        stats[0] = double(nq);
        stats[1] = double(lanes);
        stats[2] = g.conflict_before;
        stats[3] = g.conflict_after;
    }
    return MM_OK;
}

int mm_debug_row_product(mm_fsm_t f, int direction, const float *in, float *out, double stats[8]) {
    return mm_debug_row_product_ex(f, direction, 0, in, out, stats);
}

int mm_debug_row_product_ex(mm_fsm_t f, int direction, int flags, const float *in, float *out, double stats[8]) {
    if (!f || !in || !out || direction < 0 || direction > 1 || flags < 0 || flags > 31)
        return fail(MM_ERR_INVALID, "mm_debug_row_product: bad argument");
    if (f->semiring != MM_LOG) return fail(MM_ERR_INVALID, "mm_debug_row_product: log-semiring FSMs only");
    RowPackOpts opt;
    opt.rs = MM_ROW_RS;
    opt.ka_max = kRowKA[sizeof(kRowKA) / sizeof(kRowKA[0]) - 1];
    if (flags & 1) {  // the pair form as pair_variants() builds it
        opt.pair = true;
        opt.pdf_halves = !process_debug_opts().no_pdf_halves;
        opt.ka_max = MM_PAIR_KA;
        opt.ka_choices[0] = MM_PAIR_KA;
        for (float &x : opt.group_speed) x = 1.f;
        if (direction == 1) opt.finish_cost = 24;
    }
    opt.copies = (flags >> 1) & 3;
    opt.copy_perm = (flags & 8) != 0;
    opt.bank_opt = (flags & 16) ? 1 : 0;
    if (opt.copies > 2) return fail(MM_ERR_INVALID, "mm_debug_row_product: at most two copies");
    RowGraph gf, g;
    const std::vector<int32_t> none;
    const Csr &mf = f->mat[0], &m = f->mat[direction];
    if ((f->S1 + 1) * 4 > MM_ROW_RS || !make_rows(f->S1, mf.rowptr, mf.col, mf.val, f->s2p, f->P1, false, none, opt, gf))
        return fail(MM_ERR_UNSUPPORTED, "mm_debug_row_product: the FSM does not fit the row-lane form");
    if (direction == 1) {
        if (!make_rows(f->S1, m.rowptr, m.col, m.val, f->s2p, f->P1, true, gf.pos, opt, g))
            return fail(MM_ERR_UNSUPPORTED, "mm_debug_row_product: the FSM does not fit the row-lane form");
    } else {
        g = gf;
    }
    const int64_t S1 = f->S1;
    float mx = -std::numeric_limits<float>::infinity();
    for (int64_t s = 0; s < S1; ++s) mx = std::max(mx, in[s] * MM_LOG2E);
    if (!(mx > -std::numeric_limits<float>::infinity())) mx = 0.f;
    std::vector<float> pl(S1 + 1, 0.f), ol(S1 + 1, 0.f);
    for (int64_t i = 0; i < S1; ++i) pl[i] = std::exp2(in[g.order[i]] * MM_LOG2E - mx);
    eval_rows(g, pl.data(), ol.data());
    for (int64_t i = 0; i < S1; ++i) out[g.order[i]] = (std::log2(ol[i]) + mx) * MM_LN2;
    if (stats) {
        stats[0] = g.KA;
        stats[1] = g.NWC;
        stats[2] = g.nslotrows - 2;
        stats[3] = g.pad_eff;
        stats[4] = g.maxcost;
        stats[5] = g.mincost;
        stats[6] = g.conflict_before;
        stats[7] = g.conflict_after;
    }
    return MM_OK;
}

int mm_debug_stream_product(mm_fsm_t f, int direction, const float *in, float *out, double stats[4]) {
    if (!f || !in || !out || (direction != 0 && direction != 1)) return fail(MM_ERR_INVALID, "mm_debug_stream_product: bad argument");
    return mm_debug_stream_team_product(f, 1, direction, in, out, stats);
}

int mm_debug_stream_team_product(mm_fsm_t f, int H, int direction, const float *in, float *out, double stats[4]) {
    if (!f || !in || !out || (direction != 0 && direction != 1) || (H != 1 && H != 2 && H != 4))
        return fail(MM_ERR_INVALID, "mm_debug_stream_team_product: bad argument");
    return no_throw("mm_debug_stream_team_product", [&]() {
        bool ok = false;
        int rc = stream_variant(f, H, &ok);
        if (rc) return rc;
        if (!ok) return fail(MM_ERR_UNSUPPORTED, "mm_debug_stream_team_product: the FSM does not fit the stream form");
        mm_stream_eval(f->stream_h[stream_hidx(H)], direction, in, out, stats);
        return int(MM_OK);
    });
}

int mm_debug_wave_product(mm_fsm_t f, int direction, const float *in, float *out, double stats[4]) {
    if (!f || !in || !out || direction < 0 || direction > 1) return fail(MM_ERR_INVALID, "mm_debug_wave_product: bad argument");
    if (f->semiring != MM_LOG) return fail(MM_ERR_INVALID, "mm_debug_wave_product: log-semiring FSMs only");
    RowPackOpts opt;
    opt.rs = MM_WAVE_RS;
    opt.nwc_max = MM_WAVE_WAVES;
    opt.ka_max = 16;
    opt.finish_cost = 4;
    opt.copies = 1;
    opt.acap_force = 4;
    opt.seg_stride = 4;
    opt.log_weights = true;
    opt.want_partner = true;
    opt.spread_pdf = true;
    for (float &x : opt.group_speed) x = 1.f;
    RowGraph gf, g;
    const std::vector<int32_t> none;
    if (!make_rows(f->S1, f->mat[0].rowptr, f->mat[0].col, f->mat[0].val, f->s2p, f->P1, false, none, opt, gf) ||
        (direction == 1 && !make_rows(f->S1, f->mat[1].rowptr, f->mat[1].col, f->mat[1].val, f->s2p, f->P1, true, gf.pos, opt, g)))
        return fail(MM_ERR_UNSUPPORTED, "mm_debug_wave_product: the FSM does not fit the wave form");
    if (direction == 0) g = gf;
    const float NINF = -std::numeric_limits<float>::infinity();
    std::vector<float> x(size_t(f->S1) + 1, NINF), y(size_t(f->S1) + 1, NINF);
    for (int64_t i = 0; i < f->S1; ++i) x[size_t(i)] = in[g.order[size_t(i)]] * MM_LOG2E;
    eval_rows_log(g, 4, x.data(), y.data());
    for (int64_t i = 0; i < f->S1; ++i) out[g.order[size_t(i)]] = y[size_t(i)] * MM_LN2;
    if (stats) {
        stats[0] = g.KA;
        stats[1] = g.nslotrows - 2;
        stats[2] = g.pad_eff;
        stats[3] = g.conflict_after;
    }
    return MM_OK;
}

int mm_debug_split_product(mm_fsm_t f, int H, int direction, const float *in, float *out, double stats[8]) {
    if (!f || !in || !out || direction < 0 || direction > 1 || H < 2 || H > 8)
        return fail(MM_ERR_INVALID, "mm_debug_split_product: bad argument");
    if (f->semiring != MM_LOG) return fail(MM_ERR_INVALID, "mm_debug_split_product: log-semiring FSMs only");
    RowPackOpts opt, optb;
    split_pack_opts(DebugOpts(), opt, optb, H);
    std::vector<RowGraph> gs;
    SplitInfo info;
    if (!make_rows_split(H, f->S1, f->mat[0].rowptr, f->mat[0].col, f->mat[0].val, f->mat[1].rowptr, f->mat[1].col,
                         f->mat[1].val, f->s2p, f->P1, opt, optb, gs, info))
        return fail(MM_ERR_UNSUPPORTED, "mm_debug_split_product: the FSM does not fit the split forms");
    const int64_t S1 = f->S1;
    float mx = -std::numeric_limits<float>::infinity();
    for (int64_t s = 0; s < S1; ++s) mx = std::max(mx, in[s] * MM_LOG2E);
    if (!(mx > -std::numeric_limits<float>::infinity())) mx = 0.f;
    std::vector<float> pl(size_t(info.total) + 1, 0.f), ol(size_t(info.total) + 1, 0.f);
    for (int64_t r = 0; r < S1; ++r) pl[size_t(info.gpos[direction][size_t(r)])] = std::exp2(in[r] * MM_LOG2E - mx);
    double ka = 0, segs = 0, eff = 0, maxc = 0, minc = 1e30, cb = 0, ca = 0;
    for (int h = 0; h < H; ++h) {
        const RowGraph &g = gs[size_t(direction * H + h)];
        eval_rows(g, pl.data(), ol.data());
        ka = std::max(ka, double(g.KA));
        segs += g.nslotrows - 2;
        eff += g.pad_eff / H;
        maxc = std::max(maxc, double(g.maxcost));
        minc = std::min(minc, double(g.mincost));
        cb += g.conflict_before / H;
        ca += g.conflict_after / H;
    }
    for (int64_t r = 0; r < S1; ++r) out[r] = (std::log2(ol[size_t(info.gpos[direction][size_t(r)])]) + mx) * MM_LN2;
    if (stats) {
        stats[0] = ka;
        stats[1] = double(info.total);
        stats[2] = segs;
        stats[3] = eff;
        stats[4] = maxc;
        stats[5] = minc;
        stats[6] = cb;
        stats[7] = ca;
    }
    return MM_OK;
}

int mm_fsm_create(int semiring, int64_t S1, int64_t nnz, int layout, int index_bytes, int index_base, int val_bytes,
                  const void *ptr, const void *idx, const void *val, int64_t n_init, const void *init_idx,
                  const void *init_val, const int32_t *state2pdf, int32_t P1, mm_fsm_t *out) {
    return no_throw("mm_fsm_create", [&]() {
        return fsm_create_impl(semiring, S1, nnz, layout, index_bytes, index_base, val_bytes, ptr, idx, val, n_init, init_idx, init_val,
                               state2pdf, P1, out);
    });
}

// Pinned staging buffer of mm_fsm_create_many (grow only; one upload at a time).
namespace {
std::mutex g_stage_lock;
char *g_stage = nullptr;
size_t g_stage_bytes = 0;
}  // namespace

extern "C" std::atomic<long long> mm_rows_prof_ns[8];  // (mm_rows.cpp: where the time of make_rows goes, MM_VERBOSE)
static int fsm_create_many_impl(int64_t n, int semiring, int layout, int index_bytes, int index_base, int val_bytes, const int64_t *S1,
                                const int64_t *nnz, const void *const *ptr, const void *const *idx, const void *const *val,
                                const int64_t *n_init, const void *const *init_idx, const void *const *init_val,
                                const int32_t *const *state2pdf, const int32_t *P1, int threads, mm_fsm_t *out) {
    if (!out || n < 1) return fail(MM_ERR_INVALID, "mm_fsm_create_many: bad argument");
    for (int64_t i = 0; i < n; ++i) out[i] = nullptr;
    if (!S1 || !nnz || !ptr || !idx || !val || !n_init || !init_idx || !init_val || !state2pdf || !P1)
        return fail(MM_ERR_INVALID, "mm_fsm_create_many: NULL array");
    const bool verbose = process_debug_opts().verbose;
    const auto tc0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (verbose) fprintf(stderr, "[mm] create_many: %s at %.2f ms\n", what, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count());
    };
    const size_t nthr = size_t(std::max(1, std::min<int>({threads > 0 ? threads : 16, int(std::max(1u, std::thread::hardware_concurrency())), int((n + 3) / 4), 64})));
    std::vector<int> rcs(static_cast<size_t>(n), MM_OK);
    std::vector<std::string> msgs{size_t(n), std::string()};
    std::vector<size_t> bytes(static_cast<size_t>(n), 0);  // device bytes of the FSM's wave forms (0: none)
    // ---- 1. every graph on a host thread: the FSM (mm_fsm_create), its wave forms, the size of their device image
    auto run_pool = [&](auto &&body) {
        std::atomic<int64_t> next{0};
        auto work = [&]() {
            for (int64_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) body(i);
        };
        if (nthr <= 1) {
            work();
            return;
        }
        std::vector<std::thread> pool;
        for (size_t t = 0; t < nthr; ++t) pool.emplace_back(work);
        for (std::thread &t : pool) t.join();
    };
    std::atomic<long long> ns_create{0}, ns_pack{0};
    run_pool([&](int64_t i) {
        try {
            const auto ta = std::chrono::steady_clock::now();
            rcs[size_t(i)] = fsm_create_impl(semiring, S1[i], nnz[i], layout, index_bytes, index_base, val_bytes, ptr[i], idx[i], val[i], n_init[i],
                                             init_idx[i], init_val[i], state2pdf[i], P1[i], &out[i]);
            if (rcs[size_t(i)]) {
                msgs[size_t(i)] = mm_last_error();
                return;
            }
            mm_fsm_t f = out[i];
            const auto tb = std::chrono::steady_clock::now();
            ns_create += std::chrono::duration_cast<std::chrono::nanoseconds>(tb - ta).count();
            if (f->semiring == MM_LOG && f->P1 <= 250 && f->S1 <= 1023 && f->qmat[0].rowptr[size_t(f->S1)] <= 16 * 64 * 4) {
                wave_pack(f);
                ns_pack += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tb).count();
                if (f->wpend[0] && f->wpend[1]) {
                    size_t tot = 0;
                    for (int d = 0; d < 2; ++d) {
                        RowBlob rb;
                        rb.bl.dry = true;
                        row_variant_blob(f->wpend[d], false, rb);
                        tot += align_up(rb.bl.size(), 256);
                    }
                    bytes[size_t(i)] = tot;
                }
            }
        } catch (const std::exception &e) {
            rcs[size_t(i)] = MM_ERR_NOMEM;
            msgs[size_t(i)] = std::string("mm_fsm_create_many: ") + e.what();
        }
    });
    lap("graphs compiled and packed");
    if (verbose) {
        fprintf(stderr, "[mm] make_rows sections (us per graph): plan %.0f, numbering %.0f, csr %.0f, slots+placement %.0f\n", 1e-3 * double(mm_rows_prof_ns[0].exchange(0)) / double(n),
                1e-3 * double(mm_rows_prof_ns[1].exchange(0)) / double(n), 1e-3 * double(mm_rows_prof_ns[2].exchange(0)) / double(n), 1e-3 * double(mm_rows_prof_ns[3].exchange(0)) / double(n));
    }
    if (verbose) fprintf(stderr, "[mm] create_many: per graph %.0f us mm_fsm_create + %.0f us wave forms (%zu threads)\n", 1e-3 * double(ns_create.load()) / double(n), 1e-3 * double(ns_pack.load()) / double(n), nthr);
    auto undo = [&](int rc, const std::string &msg) {
        for (int64_t i = 0; i < n; ++i)
            if (out[i]) {
                (void)mm_fsm_destroy(out[i]);
                out[i] = nullptr;
            }
        return fail(rc, msg);
    };
    for (int64_t i = 0; i < n; ++i)
        if (rcs[size_t(i)]) return undo(rcs[size_t(i)], "graph " + std::to_string(i) + ": " + msgs[size_t(i)]);
    // ---- 2. ONE device allocation and ONE copy for the wave forms of all graphs
    std::vector<size_t> off(size_t(n) + 1, 0);
    for (int64_t i = 0; i < n; ++i) off[size_t(i) + 1] = off[size_t(i)] + bytes[size_t(i)];
    const size_t total = off[size_t(n)];
    if (total == 0) return MM_OK;
    {   // (without a device the packed forms stay with their FSMs, like after mm_fsm_create: mm_batch_create uploads them)
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
            (void)hipGetLastError();
            return MM_OK;
        }
    }
    void *dev = nullptr;
    if (hipMalloc(&dev, total) != hipSuccess) return undo(MM_ERR_HIP, "mm_fsm_create_many: device allocation failed");
    std::shared_ptr<void> arena(dev, [](void *q) { (void)hipFree(q); });
    lap("arena allocated");
    std::lock_guard<std::mutex> guard(g_stage_lock);
    if (g_stage_bytes < total) {
        if (g_stage) (void)hipHostFree(g_stage);
        g_stage = nullptr;
        g_stage_bytes = 0;
        void *q = nullptr;
        const size_t want = std::max(total + total / 2, size_t(1) << 22);
        if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) return undo(MM_ERR_HIP, "mm_fsm_create_many: staging allocation failed");
        g_stage = static_cast<char *>(q);
        g_stage_bytes = want;
    }
    std::vector<RowBlob> rbs(size_t(n) * 2);
    run_pool([&](int64_t i) {
        if (!bytes[size_t(i)]) return;
        mm_fsm_t f = out[i];
        size_t o = off[size_t(i)];
        for (int d = 0; d < 2; ++d) {
            RowBlob &rb = rbs[size_t(i) * 2 + d];
            rb.bl.ext = g_stage + o;
            row_variant_blob(f->wpend[d], false, rb);
            o += align_up(rb.bl.size(), 256);
        }
    });
    lap("staged");
    if (hipMemcpy(dev, g_stage, total, hipMemcpyHostToDevice) != hipSuccess) return undo(MM_ERR_HIP, "mm_fsm_create_many: upload failed");
    lap("uploaded");
    for (int64_t i = 0; i < n; ++i) {
        if (!bytes[size_t(i)]) continue;
        mm_fsm_t f = out[i];
        size_t o = off[size_t(i)];
        for (int d = 0; d < 2; ++d) {
            RowVariant *rv = f->wpend[d];
            const RowBlob &rb = rbs[size_t(i) * 2 + d];
            rv->arena = arena;
            row_variant_bind(f, rv, rb, static_cast<char *>(dev) + o, 0.f);
            o += align_up(rb.bl.size(), 256);
            f->wrows[d] = rv;
            f->wpend[d] = nullptr;
        }
        f->wave_tried = true;
    }
    return MM_OK;
}

int mm_fsm_create_many(int64_t n, int semiring, int layout, int index_bytes, int index_base, int val_bytes, const int64_t *S1,
                       const int64_t *nnz, const void *const *ptr, const void *const *idx, const void *const *val, const int64_t *n_init,
                       const void *const *init_idx, const void *const *init_val, const int32_t *const *state2pdf, const int32_t *P1,
                       int threads, mm_fsm_t *out) {
    return no_throw("mm_fsm_create_many", [&]() {
        return fsm_create_many_impl(n, semiring, layout, index_bytes, index_base, val_bytes, S1, nnz, ptr, idx, val, n_init, init_idx, init_val,
                                    state2pdf, P1, threads, out);
    });
}

// a batch of ProbSemiring FSMs answers for its settings and counters through the batch of its log twins (what runs its fast calls)
static mm_batch_t twin_of(mm_batch_t h) { return h && h->log_twin ? h->log_twin : h; }

static int batch_create_impl(const mm_fsm_t *fsms, int64_t B, mm_batch_t *out) {
    if (!out) return fail(MM_ERR_INVALID, "mm_batch_create: out is NULL");
    *out = nullptr;
    if (!fsms || B < 1) return fail(MM_ERR_INVALID, "mm_batch_create: empty batch");
    const auto tb0 = std::chrono::steady_clock::now();
    for (int64_t b = 0; b < B; ++b) {
        if (!fsms[b]) return fail(MM_ERR_INVALID, "mm_batch_create: NULL FSM handle");
        if (fsms[b]->semiring != fsms[0]->semiring)
            return fail(MM_ERR_INVALID, "mm_batch_create: FSMs of one batch must share the semiring (FSM{K})");
    }
    // (owned here until it is handed out: every early return and every exception -- the packers allocate -- destroys it)
    struct Drop {
        void operator()(mm_batch_s *x) const { (void)mm_batch_destroy(x); }
    };
    std::unique_ptr<mm_batch_s, Drop> hold(new mm_batch_s());
    mm_batch_s *h = hold.get();
    h->dbg = read_debug_opts();
    h->B = B;
    h->semiring = fsms[0]->semiring;
    h->fsms.assign(fsms, fsms + B);
    if (hipGetDevice(&h->device) != hipSuccess) return fail(MM_ERR_HIP, "mm_batch_create: no device");
    if (h->semiring == MM_PROB) {  // the generic path (mm_pdfposteriors_ex): no kernel forms, no descriptors of its own
        bool twins = true;
        for (int64_t b = 0; b < B; ++b) {
            h->total_states += fsms[b]->S1;
            h->max_P1 = std::max(h->max_P1, int(fsms[b]->P1));
            twins = twins && fsms[b]->log_twin != nullptr;
        }
        if (twins) {  // ... and the batch of the log twins for the fast entry (a repeated handle repeats its twin: stored once)
            std::vector<mm_fsm_t> tw(static_cast<size_t>(B));
            for (int64_t b = 0; b < B; ++b) tw[size_t(b)] = fsms[b]->log_twin;
            const int rc = batch_create_impl(tw.data(), B, &h->log_twin);
            if (rc) return rc;
        }
        *out = hold.release();
        return MM_OK;
    }
    for (int64_t b = 0; b < B; ++b) h->max_P1 = std::max(h->max_P1, int(fsms[b]->P1));  // (the choice of kernels below reads it)
    std::vector<UttDesc> utts(B);
    // quad kernels: one geometry (quads per lane, waves) per direction for the whole batch
    int64_t nq_max[2] = {0, 0};
    for (int64_t b = 0; b < B; ++b) {
        h->fast_ok = h->fast_ok && fsms[b]->fast_ok;
        for (int d = 0; d < 2; ++d) nq_max[d] = std::max(nq_max[d], fsms[b]->nquads[d]);
        h->max_depth = std::max(h->max_depth, fsms[b]->depth);
    }
    h->fast_ok = h->fast_ok && h->semiring == MM_LOG;
    if (h->fast_ok) {
        for (int d = 0; d < 2; ++d) {
            QuadGeometry geo = pick_quad_geometry(nq_max[d]);
            if (h->dbg.kq >= 1 && h->dbg.kq <= 29) geo.KQ = h->dbg.kq;
            // enough waves for the quads, and for at most two rows per thread where the workgroup can be that large
            int64_t s1_max = 0;
            for (int64_t b = 0; b < B; ++b) s1_max = std::max(s1_max, fsms[b]->S1);
            geo.NW = int(std::min<int64_t>(geo.KQ > 13 ? 8 : MM_MAX_WAVES,
                                           std::max<int64_t>({1, (nq_max[d] + 64 * geo.KQ - 1) / (64 * geo.KQ),
                                                              (s1_max + 127) / 128})));
            if (h->dbg.nwaves >= 1 && h->dbg.nwaves <= (geo.KQ > 13 ? 8 : MM_MAX_WAVES)) geo.NW = h->dbg.nwaves;
            h->geo_kq[d] = geo.KQ;
            h->geo_nw[d] = geo.NW;
            h->max_quads[d] = int(nq_max[d]);
        }
        // small graphs keep the CSR the exact fallback walks in LDS (left-to-right graphs spread the
        // values of one frame over far more than the float range: most of their rows take that path)
        for (int64_t b = 0; b < B; ++b)
            h->max_xcsr = std::max<int64_t>(h->max_xcsr, fsms[b]->S1 + 1 + 2 * std::max(fsms[b]->qmat[0].rowptr[fsms[b]->S1],
                                                                                  fsms[b]->qmat[1].rowptr[fsms[b]->S1]));
    }
    // The wave forms of every FSM of the batch (packed on the host's cores for the FSMs that are new): h->wave_ok, h->wave_nseg.
    auto try_wave = [&]() -> int {
        h->wave_ok = true;
        {   // pack the forms of the FSMs that are new, on the host's cores
            std::vector<mm_fsm_t> todo;
            for (int64_t b = 0; b < B; ++b)
                if (!fsms[b]->wrows[0] && !fsms[b]->wave_tried && !fsms[b]->wave_packed &&
                    std::find(todo.begin(), todo.end(), fsms[b]) == todo.end())
                    todo.push_back(fsms[b]);
            const size_t nthr = std::min<size_t>({todo.size() / 4, size_t(std::max(1u, std::thread::hardware_concurrency())), size_t(16)});
            if (nthr > 1) {
                std::atomic<size_t> next{0};
                std::vector<std::thread> pool;
                for (size_t t = 0; t < nthr; ++t)
                    pool.emplace_back([&]() {
                        // (an exception must not leave a thread: an FSM whose packing failed is packed again, and fails
                        // again, on the calling thread -- wave_variants below)
                        for (size_t i = next.fetch_add(1); i < todo.size(); i = next.fetch_add(1)) {
                            try {
                                wave_pack(todo[i]);
                            } catch (...) {
                                todo[i]->wave_packed = false;
                            }
                        }
                    });
                for (std::thread &t : pool) t.join();
            }
        }
        for (int64_t b = 0; b < B && h->wave_ok; ++b) {
            bool ok = false;
            int rc = wave_variants(fsms[b], h->dbg, &ok);
            if (rc) return rc;
            h->wave_ok = ok;
            // (segments of a wave's registers: the state segments, and twice the pdf segments -- the kernel has NSEG / 2 of those)
            if (ok)
                h->wave_nseg = std::max({h->wave_nseg, std::max(fsms[b]->wrows[0]->g.KA, fsms[b]->wrows[1]->g.KA) / 4,
                                         2 * std::max(fsms[b]->wrows[0]->pdf_nps, fsms[b]->wrows[1]->pdf_nps)});
        }
        return MM_OK;
    };
    // The wave kernel FIRST wherever every graph of the batch fits it (up to 1023 states, 16 segments of 64 lanes x 4 arcs
    // per direction, 250 pdfs): one workgroup per utterance runs both directions at once in the log domain, without the
    // marks and the exact second pass of the linear-domain kernels -- measured against the row kernels on batches of
    // different graphs (1.18 -> 0.72 ms: 128 lexicon graphs of 150..400 states, T = 700) and against the pair kernels on one
    // shared small graph (0.80 -> 0.47 ms: 300 states, B = 256, T = 500; 3-state HMM, B = 1, T = 100: 0.18 -> 0.06 ms).  The
    // exception: one shared DENSE graph (more than 16 arcs per state, more than 2 segments per wave) on a batch of more than
    // two utterances per compute unit, where the pair kernels' two utterances per workgroup win (32-state ergodic HMM,
    // B = 1024: 1.90 against 2.33 ms).
    // The lane kernel FIRST for batches of tiny graphs (every FSM: up to 64 states and 64 pdfs): one wave per utterance and
    // direction with the graph in its registers, float64, exact -- BASELINE config 2 (dense 64-state HMM, B = 32, T = 500):
    // 0.60 ms on the pair kernels; config 1 (3-state HMM, one utterance).
    if (h->semiring == MM_LOG && (h->dbg.kernel == DebugOpts::K_AUTO || h->dbg.kernel == DebugOpts::K_LANE)) {
        h->lane_ok = true;
        for (int64_t b = 0; b < B && h->lane_ok; ++b) {
            bool ok = false;
            int rc = lane_variant(fsms[b], &ok);
            if (rc) return rc;
            h->lane_ok = ok;
            h->lane_S = std::max(h->lane_S, int(fsms[b]->S1 - 1));
        }
    }
    // What the lane kernel marks (mass beyond the double's range: the one path of a sharp left-to-right graph) is computed again in the
    // log domain: by the WAVE kernel when every graph has its wave forms (up to 4096 arc slots per direction; packed with the graph by
    // mm_fsm_create_many, nothing to upload at the first call, capturable from the first call on), else -- a dense 64-state HMM has
    // 4161 arcs -- by the item kernel, whose forms then go to the device here.
    if (h->lane_ok) {
        int rc = try_wave();
        if (rc) return rc;
        h->lane_redo_wave = h->wave_ok;
        h->wave_ok = false;
        if (!h->lane_redo_wave) h->wave_nseg = 0;
    }
    bool wave_first_tried = false;
    if (h->semiring == MM_LOG && h->dbg.kernel == DebugOpts::K_AUTO && !h->lane_ok) {
        bool small = h->max_P1 <= 250, same = true;
        for (int64_t b = 0; b < B && small; ++b)
            small = fsms[b]->S1 <= 1023 && fsms[b]->qmat[0].rowptr[fsms[b]->S1] <= 16 * 64 * 4;
        for (int64_t b = 1; b < B && same; ++b) same = fsms[b] == fsms[0];
        const bool dense_many = same && B > 2 * int64_t(h->n_cus) && fsms[0]->qmat[0].rowptr[fsms[0]->S1] > 16 * fsms[0]->S1;
        if (small) {
            wave_first_tried = true;
            int rc = try_wave();
            if (rc) return rc;
            // (the exception; the forms stay with the FSM.  Graphs of up to 2 segments per wave run the kernel instance of
            // which two workgroups fit a compute unit and win at every batch size: 16-state ergodic HMM, B = 1024: 1.18
            // against 1.88 ms)
            if (h->wave_ok && dense_many && h->wave_nseg > 2) h->wave_ok = false;
        }
    }
    // row kernels: every FSM of the batch needs its row-lane forms.  Small deep (left-to-right) graphs keep states
    // alive whose values differ by more than the float range within one frame, so most of their rows would take the
    // exact fallback of the linear-domain kernels: they run on the item kernel (unless a kernel is forced).
    const bool linear_first = !h->wave_ok && !h->lane_ok && h->fast_ok && h->dbg.kernel != DebugOpts::K_ITEM && h->dbg.kernel != DebugOpts::K_QUAD &&
                              h->dbg.kernel != DebugOpts::K_WAVE && h->dbg.kernel != DebugOpts::K_STREAM &&
                              !(h->max_depth >= 64 && nq_max[0] <= 3 * 1024 && nq_max[1] <= 3 * 1024 && h->dbg.kernel == DebugOpts::K_AUTO);
    h->rows_ok = linear_first;
    for (int64_t b = 0; b < B && h->rows_ok; ++b) {
        bool ok = false;
        int rc = row_variants(fsms[b], h->dbg.verbose, &ok);
        if (rc) return rc;
        h->rows_ok = ok;
    }
    // pair kernels: all utterances on ONE FSM (the graph registers are shared by the two utterances of a workgroup)
    // (a batch of ONE utterance too: its pair runs the utterance twice, the copy writes nothing outside the workspace -- both
    // directions at once instead of the row kernels' two passes: 4.1 -> 2.6 ms on config 3's graph)
    // (the row kernels take up to 250 pdfs, the pair kernels 506: a shared graph of more pdfs has no row forms)
    h->pairs_ok = linear_first && (h->rows_ok || h->max_P1 > 250) && h->dbg.kernel != DebugOpts::K_ROW && h->dbg.kernel != DebugOpts::K_SPLIT;
    for (int64_t b = 1; b < B && h->pairs_ok; ++b) h->pairs_ok = fsms[b] == fsms[0];
    if (h->pairs_ok) {
        bool ok = false;
        int rc = pair_variants(fsms[0], h->dbg, &ok);
        if (rc) return rc;
        h->pairs_ok = ok;
        if (ok) {
            h->pair_ka = std::max(fsms[0]->prows[0]->g.KA, fsms[0]->prows[1]->g.KA);
            h->pair_nwc = std::max(fsms[0]->prows[0]->g.NWC, fsms[0]->prows[1]->g.NWC);
            h->pair_slotrows = std::max(fsms[0]->prows[0]->g.nslotrows, fsms[0]->prows[1]->g.nslotrows);
            h->pairs_ok = h->pair_ka <= MM_PAIR_KA && mm_pair_lds_bytes(1, h->pair_slotrows, h->max_P1) <= 160 * 1024;
        }
    }
    // split pair kernels: one shared FSM that is too large for the pair kernels proper (more arcs than the registers of a
    // compute unit hold, more states than half its LDS) -- teams of 2 workgroups per utterance pair and direction
    if (!h->wave_ok && !h->lane_ok && !h->pairs_ok && h->fast_ok && h->dbg.kernel != DebugOpts::K_ITEM && h->dbg.kernel != DebugOpts::K_QUAD &&
        h->dbg.kernel != DebugOpts::K_ROW && h->dbg.kernel != DebugOpts::K_WAVE && h->dbg.kernel != DebugOpts::K_STREAM &&
        !(h->max_depth >= 64 && nq_max[0] <= 3 * 1024 && nq_max[1] <= 3 * 1024 && h->dbg.kernel == DebugOpts::K_AUTO)) {
        bool same = true;
        for (int64_t b = 1; b < B && same; ++b) same = fsms[b] == fsms[0];
        if (same) {
            bool ok = false;
            int rc = split_variants(fsms[0], h->dbg, 2, &ok);
            // (a graph beyond the teams of 2 -- more than 3070 states or 2 x 14 x 64 x 36 arcs: teams of 4; beyond those -- more
            // than 4094 states: teams of 8, up to 6014 states and 314 pdfs)
            if (!rc && !ok) rc = split_variants(fsms[0], h->dbg, 4, &ok);
            if (!rc && !ok) rc = split_variants(fsms[0], h->dbg, 8, &ok);
            if (rc) return rc;
            if (ok) {
                const mm_fsm_t f0 = fsms[0];
                h->pair_H = f0->split.H;
                h->pair_ka = 0;
                h->pair_nwc = 1;
                h->pair_slotrows = 0;
                for (int d = 0; d < 2; ++d)
                    for (int s = 0; s < h->pair_H; ++s) {
                        h->pair_ka = std::max(h->pair_ka, f0->srows[d][s]->g.KA);
                        h->pair_nwc = std::max(h->pair_nwc, f0->srows[d][s]->g.NWC);
                        h->pair_slotrows = std::max(h->pair_slotrows, f0->srows[d][s]->g.nslotrows);
                    }
                h->split_s1p = (f0->split.total + 2 + 3) & ~3;
                const size_t lds = mm_split_lds_bytes(h->pair_H, 1, h->pair_slotrows, h->max_P1);
                h->pairs_ok = h->pair_ka <= mm_split_ka(h->pair_H) && h->pair_nwc <= MM_SPLIT_NWC && lds > 0 && lds <= 160 * 1024;
                if (h->dbg.verbose) fprintf(stderr, "[mm] teams of %d: LDS %zu bytes in phase B\n", h->pair_H, lds);
                if (!h->pairs_ok) h->pair_H = 1;
            }
        }
    }
    if (h->semiring == MM_TROPICAL && h->dbg.kernel != DebugOpts::K_ITEM) {
        h->vit_ok = true;
        for (int64_t b = 0; b < B && h->vit_ok; ++b) {
            bool ok = false;
            int rc = vit_variant(fsms[b], h->dbg, &ok);
            if (rc) return rc;
            h->vit_ok = ok;
            if (ok) {  // (one layout per batch: the first FSM's; an FSM that needed another one keeps the batch on the item kernel)
                if (b == 0) {
                    h->vit_n4 = fsms[b]->vit_n4;
                    h->vit_n2 = fsms[b]->vit_n2;
                }
                h->vit_ok = fsms[b]->vit_n4 == h->vit_n4 && fsms[b]->vit_n2 == h->vit_n2;
                h->vit_arcs = std::max(h->vit_arcs, int(fsms[b]->vrow->g.col.size()));
                h->vit_ok = h->vit_ok && size_t(fsms[b]->S1p) * 4 <= 24576;  // (the kernel's state vectors: mm_vit_tu.hip)
            }
        }
    }
    // wave kernel, the other case: small graphs that none of the linear-domain kernels takes (deep left-to-right graphs:
    // numerators), or the kernel is asked for by name
    if (h->semiring == MM_LOG && !h->wave_ok && !h->lane_ok && !wave_first_tried && !h->rows_ok && !h->pairs_ok &&
        (h->dbg.kernel == DebugOpts::K_WAVE ||
         (h->dbg.kernel == DebugOpts::K_AUTO && (!h->fast_ok || (h->max_depth >= 64 && h->geo_kq[0] <= 3 && h->geo_kq[1] <= 3))))) {
        // (= where the item kernel would run: quad_kernel_usable() says no for these; see there)
        int rc = try_wave();
        if (rc) return rc;
    }
    // (a batch of the wave kernel never runs the quad kernels, and one whose marked utterances go to the float64 pair kernels
    // has the item kernel behind those: their quad forms are not built)
    // stream kernels (mm_stream.hip): graphs beyond every register-resident form and beyond the quad kernels' LDS (more than ~3000
    // states, more than 250 pdfs on different graphs / 506 on a shared one, weights outside the float range) -- what the item
    // kernel ran until round 5
    {
        int64_t s1_max = 0;
        for (int64_t b = 0; b < B; ++b) s1_max = std::max(s1_max, fsms[b]->S1);
        const bool beyond = s1_max > 3000 || h->max_P1 > 250 || !h->fast_ok;
        if (h->semiring == MM_LOG && !h->lane_ok && !h->wave_ok && !h->pairs_ok && !h->rows_ok &&
            (h->dbg.kernel == DebugOpts::K_STREAM || (h->dbg.kernel == DebugOpts::K_AUTO && beyond))) {
            h->stream_ok = true;
            // teams of 2 or 4 workgroups per utterance and direction when the batch leaves compute units idle (round 6: a direction's
            // record stream through ONE compute unit's memory path is what bounds a frame)
            {
                int dev = 0, cus = 0;
                if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) h->n_cus = cus;
            }
            h->stream_H = (h->dbg.stream_h == 1 || h->dbg.stream_h == 2 || h->dbg.stream_h == 4) ? h->dbg.stream_h : mm_stream_pick_h(B, h->n_cus);
            for (int64_t b = 0; b < B && h->stream_ok; ++b) {
                bool ok = false;
                int rc = stream_variant(fsms[b], h->stream_H, &ok);
                if (rc) return rc;
                h->stream_ok = ok && mm_stream_dev(fsms[b]->stream_h[stream_hidx(h->stream_H)]) != nullptr;
            }
            h->stream_S1 = int(s1_max);
        }
    }
    const bool want_dpair = h->pairs_ok && !h->dbg.no_dpair;
    h->quad_built = h->fast_ok && !h->wave_ok && !h->lane_ok && !h->stream_ok && !(want_dpair && h->dbg.kernel != DebugOpts::K_QUAD);
    // (the item forms -- the general fallback, the alpha / beta export, the total-sum family -- of a batch of the wave kernel go to
    // the device when an entry first needs them, ensure_item_forms(): a batch of new numerator graphs every training step
    // never does)
    h->items_resident = !h->wave_ok && !(h->lane_ok && h->lane_redo_wave);
    for (int64_t b = 0; b < B; ++b) {
        mm_fsm_t f = fsms[b];
        int rc = h->items_resident ? fsm_to_device(f) : MM_OK;
        QuadVariant *qv[2] = {nullptr, nullptr};
        for (int d = 0; d < 2 && !rc && h->quad_built; ++d) rc = quad_variant(f, d, h->geo_kq[d], h->dbg.verbose, &qv[d]);
        if (rc) return rc;
        UttDesc &u = utts[b];
        memset(&u, 0, sizeof(u));
        u.g[0] = f->gdev[0];
        u.g[1] = f->gdev[1];
        if (qv[0] && qv[1]) {
            u.q[0] = qv[0]->qdev;
            u.q[1] = qv[1]->qdev;
            u.init_f = qv[0]->d_init_f;
            u.map_bf = qv[1]->d_map_bf;
        }
        if (h->lane_ok) u.lane = static_cast<const LaneDev *>(f->lane_blob);
        if (h->stream_ok) u.stream = mm_stream_dev(f->stream_h[stream_hidx(h->stream_H)]);
        if (h->vit_ok) u.rv = f->vrow->rdev;
        if (h->wave_ok || h->lane_redo_wave)
            for (int d = 0; d < 2; ++d) u.rw[d] = f->wrows[d]->rdev;
        if (h->pairs_ok && h->pair_H == 1)
            for (int d = 0; d < 2; ++d) u.rp[d] = f->prows[d]->rdev;
        if (h->pairs_ok && h->pair_H > 1)
            for (int d = 0; d < 2; ++d)
                for (int s = 0; s < h->pair_H; ++s) u.rps[d][s] = f->srows[d][s]->rdev;
        if (h->rows_ok)
            for (int d = 0; d < 2; ++d) {
                u.r[d] = f->rows[d]->rdev;
                h->row_ka[d] = std::max(h->row_ka[d], f->rows[d]->g.KA);
                h->row_nwc[d] = std::max(h->row_nwc[d], f->rows[d]->g.NWC);
                h->row_slotrows[d] = std::max(h->row_slotrows[d], f->rows[d]->g.nslotrows);
            }
        u.init = f->d_init;
        u.s2p = f->d_s2p;
        u.pdf_ptr = f->d_pdf_ptr;
        u.pdf_rows = f->d_pdf_rows;
        u.S1 = int(f->S1);
        u.S1p = f->S1p;
        u.P1 = f->P1;
        u.pad = 0;
        u.state_off = h->total_states;
        u.s1p_prefix = h->total_s1p;
        h->total_states += f->S1;
        h->total_s1p += f->S1p;
        h->max_S1p = std::max(h->max_S1p, f->S1p);
        h->max_P1 = std::max(h->max_P1, int(f->P1));
        if (h->items_resident) h->max_items = std::max(h->max_items, int(std::max(f->packed[0].items.size(), f->packed[1].items.size())));
    }
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            h->n_cus = cus;
    }
    if (hipGetDevice(&h->device) != hipSuccess || hipMalloc(&h->d_utts, sizeof(UttDesc) * B) != hipSuccess ||
        hipMemcpy(h->d_utts, utts.data(), sizeof(UttDesc) * B, hipMemcpyHostToDevice) != hipSuccess) {
        return fail(MM_ERR_HIP, "mm_batch_create: device allocation failed");
    }
    h->utts_host = std::move(utts);
    {   // FSMs whose state vectors do not fit the LDS: the item / tropical kernels keep them in global memory
        const int P1p = (h->max_P1 + 3) & ~3;
        if (size_t(lds_plan(h->max_S1p, P1p, true).total) * 4 > 160 * 1024 || h->dbg.bigv) {
            if (hipMalloc(&h->ws_big, size_t(B) * 4 * size_t(h->max_S1p) * sizeof(float)) != hipSuccess) {
                h->ws_big = nullptr;
                return fail(MM_ERR_HIP, "mm_batch_create: device allocation failed");
            }
        }
    }
    if (want_dpair) {
        void *hp = nullptr;
        if (hipMalloc(&h->stat_dev, 4 * sizeof(int)) == hipSuccess && hipMemset(h->stat_dev, 0, 4 * sizeof(int)) == hipSuccess &&
            hipHostMalloc(&hp, 4 * sizeof(int), hipHostMallocMapped) == hipSuccess) {
            h->stat_host = static_cast<volatile int *>(hp);
            for (int k = 0; k < 4; ++k) h->stat_host[k] = 0;  // ([2], [3]: utterances the last alpha / beta export handed to the item kernel)
            h->dpair_ok = true;
            h->exact_first = h->dbg.exact_first;
            {
                PairLaunch pl;
                pl.B = h->B;
                pl.nwc = h->pair_nwc;
                pl.slotrows = h->pair_slotrows;
                pl.max_P1 = h->max_P1;
                pl.pair_ka = h->pair_ka;
                pl.H = h->pair_H;
                h->wpair_ok = !h->dbg.no_wpair && mm_wpair_fits(pl);
            }
        }
    }
    if (h->fast_ok && h->geo_kq[0] <= 3 && h->geo_kq[1] <= 3 && !h->dbg.no_xcsr) {
        h->xcsr = int((h->max_xcsr + 3) & ~int64_t(3));
        if (h->max_xcsr > 16 * 1024 || quad_lds_bytes(h, 0) > 128 * 1024 || quad_lds_bytes(h, 1) > 128 * 1024) h->xcsr = 0;
    }
    if (h->dbg.verbose)
        fprintf(stderr, "[mm] batch of %lld created in %.1f ms\n", (long long)B, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count());
    *out = hold.release();
    return MM_OK;
}

int mm_batch_create(const mm_fsm_t *fsms, int64_t B, mm_batch_t *out) {
    return no_throw("mm_batch_create", [&]() { return batch_create_impl(fsms, B, out); });
}

#ifdef MM_STAMPS
// diagnostic build: copy the per-wave phase cycle sums of the last pdfposteriors call to the host
int mm_debug_read_stamps(unsigned long long *out, int64_t n) {
    if (!g_dbg) return fail(MM_ERR_INVALID, "no stamps recorded");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, g_dbg, sizeof(unsigned long long) * size_t(std::min<int64_t>(n, int64_t(g_dbg_n))),
                      hipMemcpyDeviceToHost));
    return MM_OK;
}
#endif

int mm_batch_destroy(mm_batch_t h) {
    if (!h) return MM_OK;
    if (h->log_twin) (void)mm_batch_destroy(h->log_twin);
    if (h->prob_logv) (void)hipFree(h->prob_logv);
    if (h->d_utts) (void)hipFree(h->d_utts);
    if (h->ws_big) (void)hipFree(h->ws_big);
    if (h->ws) (void)hipFree(h->ws);
    if (h->stat_dev) (void)hipFree(h->stat_dev);
    if (h->stat_host) (void)hipHostFree(const_cast<int *>(h->stat_host));
    if (h->gen.ws) (void)hipFree(h->gen.ws);
    if (h->gen.d_utts) (void)hipFree(h->gen.d_utts);
    delete h;
    return MM_OK;
}

int64_t mm_batch_total_states(mm_batch_t h) { return h ? h->total_states : -1; }

}  // extern "C"
namespace mm {
FsmGenView *mm_fsm_gen_view(mm_fsm_t f) {
    if (!f) return nullptr;
    std::call_once(f->gen_once, [&]() { gen_build(f); });
    return &f->gen;
}
GenScratch *mm_batch_gen_scratch(mm_batch_t h) { return h ? &h->gen : nullptr; }
int mm_batch_gen_view(mm_batch_t h, int64_t *B, const mm_fsm_t **fsms, int *semiring, int *device) {
    if (!h) return MM_ERR_INVALID;
    *B = h->B;
    *fsms = h->fsms.data();
    *semiring = h->semiring;
    *device = h->device;
    return MM_OK;
}
}  // namespace mm
extern "C" {

int mm_batch_set_posterior_floor(mm_batch_t h, float floor) {
    h = twin_of(h);
    if (!h) return fail(MM_ERR_INVALID, "mm_batch_set_posterior_floor: NULL batch");
    if (!(floor >= 1e-30f && floor <= 1e-6f)) return fail(MM_ERR_INVALID, "mm_batch_set_posterior_floor: floor outside [1e-30, 1e-6]");
    // a term that dropped out of the linear path would have had a posterior below 2^(-120 - L_n) (mm_pair_finish_kernel):
    // L_n >= -120 - log2(floor) keeps every loss below the floor (1e-30 -> -20.3, the default -20; 1e-12 -> -80)
    h->lt_floor = std::min(-20.f, -120.f - std::log2(floor));
    if (h->stat_host) h->stat_host[0] = 0;  // (what was hard under the old floor need not be under the new one: float32 kernels first)
    return MM_OK;
}

int mm_batch_set_exact_policy(mm_batch_t h, int policy) {
    h = twin_of(h);
    if (!h) return fail(MM_ERR_INVALID, "mm_batch_set_exact_policy: NULL batch");
    if (policy != MM_EXACT_AUTO && policy != MM_EXACT_F32_FIRST && policy != MM_EXACT_F64_FIRST)
        return fail(MM_ERR_INVALID, "mm_batch_set_exact_policy: unknown policy");
    h->exact_first = policy == MM_EXACT_AUTO ? -1 : (policy == MM_EXACT_F64_FIRST ? 1 : 0);
    return MM_OK;
}

int mm_batch_set_gamma_mode(mm_batch_t h, int accumulate, float scale) {
    h = twin_of(h);
    if (!h) return fail(MM_ERR_INVALID, "mm_batch_set_gamma_mode: NULL batch");
    if (!(scale == scale) || std::isinf(scale)) return fail(MM_ERR_INVALID, "mm_batch_set_gamma_mode: scale is not finite");
    const bool plain = !accumulate && scale == 1.f;
    // (every other kernel family may compute an utterance TWICE -- the linear-domain kernels first, the exact ones for what they
    // mark -- and the second result must replace the first, not add to it)
    if (!plain && !h->wave_ok)
        return fail(MM_ERR_UNSUPPORTED, "mm_batch_set_gamma_mode: only batches of the wave kernel (every graph <= 1023 states, <= 4096 arc slots per direction: "
                                        "LF-MMI numerators) scale or accumulate their posteriors");
    h->g_acc = accumulate != 0;
    h->g_scale = scale;
    return MM_OK;
}

int mm_batch_set_mark_policy(mm_batch_t h, int policy) {
    h = twin_of(h);
    if (!h) return fail(MM_ERR_INVALID, "mm_batch_set_mark_policy: NULL batch");
    if (policy != MM_MARKS_DECIDE && policy != MM_MARKS_KEEP) return fail(MM_ERR_INVALID, "mm_batch_set_mark_policy: unknown policy");
    h->keep_marks = policy == MM_MARKS_KEEP;
    if (h->stat_host) h->stat_host[0] = 0;  // (what the last call counted was counted under the other policy: float32 kernels first)
    return MM_OK;
}

int mm_batch_set_deterministic(mm_batch_t h, int on) {
    h = twin_of(h);
    if (!h) return fail(MM_ERR_INVALID, "mm_batch_set_deterministic: NULL batch");
    h->deterministic = on != 0;
    return MM_OK;
}

int mm_batch_last_redo_count(mm_batch_t h, void *stream, int64_t *n) {
    h = twin_of(h);
    if (!h || !n) return fail(MM_ERR_INVALID, "mm_batch_last_redo_count: bad argument");
    *n = 0;
    if (!h->last_redo) return MM_OK;
    std::vector<int> marks(size_t(h->B), 0);
    HIP_TRY(hipMemcpyAsync(marks.data(), h->last_redo, size_t(h->B) * sizeof(int), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    for (int m : marks) *n += m != 0;
    if (h->dbg.verbose && h->last_z) {  // what mm_pair_finish_kernel decided on
        std::vector<double> z(size_t(h->B) * 6);
        HIP_TRY(hipMemcpy(z.data(), h->last_z, z.size() * 8, hipMemcpyDeviceToHost));
        double sp = 0, lm = 0;
        int worst_b = -1, below[5] = {0, 0, 0, 0, 0};
        const double thr[5] = {-8, -11, -14, -17, -20};
        for (int64_t b = 0; b < h->B; ++b) {
            const double zmin = std::min(z[6 * b], z[6 * b + 1]), zmax = std::max(z[6 * b + 2], z[6 * b + 3]);
            if (zmax - zmin > sp) sp = zmax - zmin, worst_b = int(b);
            const double l = std::min(z[6 * b + 4], z[6 * b + 5]);
            lm = std::min(lm, l);
            for (int k = 0; k < 5; ++k) below[k] += l < thr[k];
        }
        fprintf(stderr, "[mm] per-frame log2 normalisers: largest spread %.3g (utterance %d), smallest overlap term %.3g; utterances with an overlap term "
                        "below -8 / -11 / -14 / -17 / -20: %d / %d / %d / %d / %d of %lld\n", sp, worst_b, lm, below[0], below[1], below[2], below[3], below[4],
                (long long)h->B);
    }
    return MM_OK;
}

int mm_batch_last_fallback_count(mm_batch_t h, void *stream, int64_t *n) {
    h = twin_of(h);
    if (!h || !n) return fail(MM_ERR_INVALID, "mm_batch_last_fallback_count: bad argument");
    *n = 0;
    if (!h->last_redo2) return MM_OK;
    std::vector<int> marks(size_t(h->B), 0);
    HIP_TRY(hipMemcpyAsync(marks.data(), h->last_redo2, size_t(h->B) * sizeof(int), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    for (int m : marks) *n += m != 0;
    return MM_OK;
}

int mm_batch_last_exact_first(mm_batch_t h) {
    h = twin_of(h);
    return h && h->last_exact_first ? 1 : 0;
}

int mm_batch_team_xcd_stats(mm_batch_t h, int out[2]) {
    h = twin_of(h);
    if (!h || !out) return fail(MM_ERR_INVALID, "mm_batch_team_xcd_stats: bad argument");
    out[0] = out[1] = 0;
    if (!h->stat_dev) return MM_OK;
    h->xcd_counting = true;  // (the calls from here on count; the first call returns the zeros nothing has been added to)
    return no_throw("mm_batch_team_xcd_stats", [&]() {
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, h->stat_dev + 2, 2 * sizeof(int), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(h->stat_dev + 2, 0, 2 * sizeof(int)));
        return int(MM_OK);
    });
}

static bool export_on_pairs(mm_batch_t h, int dir);
int mm_batch_kernels(mm_batch_t h, int entry, char *buf, size_t n) {
    if (!h || !buf || n < 2) return fail(MM_ERR_INVALID, "mm_batch_kernels: bad argument");
    if (entry == 0) h = twin_of(h);  // (ProbSemiring: the fast entry runs the log twins' kernels)
    if (entry == 0 && h->semiring == MM_PROB)
        return fail(MM_ERR_UNSUPPORTED, "mm_batch_kernels: this ProbSemiring batch has no log twins (Float64 FSMs, or negative weights): mm_pdfposteriors_ex only");
    std::string s;
    if (entry == 0) {  // mm_pdfposteriors_f32
        const bool quad = quad_kernel_usable(h);
        const std::string exact = quad ? "mm_fbq_kernel<" + std::to_string(h->geo_kq[0]) + ",*,0> + mm_fbq_kernel<" +
                                             std::to_string(h->geo_kq[1]) + ",*,1>"
                                       : std::string("mm_log_kernel<MODE_FB> (forward, backward)");
        if (h->lane_ok) {
            s = "mm_lane_kernel<" + std::to_string(h->lane_S <= 8 ? 8 : h->lane_S <= 16 ? 16 : h->lane_S <= 32 ? 32 : 64) +
                "> (one wave per utterance and direction, the graph in its registers, float64)";
        } else if (h->wave_ok) {
            s = "mm_wave_kernel<" + std::to_string(h->wave_nseg <= 2 ? 2 : 4) + "," + std::to_string(h->max_P1 <= 128 ? 2 : 4) +
                (h->wave_nseg <= 2 && h->B > h->n_cus ? ",two per CU>" : ">");
        } else if (h->stream_ok) {
            s = "mm_stream_kernel (forward and backward recursions as workgroups of one grid" +
                (h->stream_H > 1 ? ", teams of " + std::to_string(h->stream_H) + " workgroups per utterance and direction" : std::string()) +
                "; arcs streamed from L2, the vector in LDS as wide-exponent "
                "32-bit values), mm_stream_combine_kernel, mm_stream_finish_kernel, then for marked utterances only " + exact;
        } else if (h->pairs_ok && h->pair_H > 1) {
            const std::string k = std::to_string(mm_pair_nj(h->max_P1, h->pair_H)), H = std::to_string(h->pair_H);
            s = "mm_fbs_kernel<" + k + ",A," + H + ">, then <" + k + ",B," + H + "> (forward and backward agents in one grid, teams of " + H +
                " workgroups), mm_pair_finish_kernel, then for marked utterances only " +
                (h->dpair_ok ? "mm_fbds_kernel<" + k + ",A," + H + ">, then <" + k + ",B," + H + "> (float64, one utterance per team" +
                                   (h->wpair_ok ? std::string("); FIRST and alone while the inputs are hard: mm_fbws_kernel<") + k + ",A," + H + ">, then <" + k +
                                                      ",B," + H + "> (wide-exponent pairs, two utterances per team)"
                                                : std::string("; FIRST and alone while the inputs are hard)")) +
                                   ", mm_dpair_finish_kernel, then for what those mark "
                             : std::string()) +
                exact;
        } else if (h->pairs_ok) {
            const std::string k = std::to_string(mm_pair_nj(h->max_P1));
            s = "mm_fbp_kernel<" + k + ",A>, then <" + k + ",B> (forward and backward agents in one grid), mm_pair_finish_kernel, then for marked "
                "utterances only " +
                (h->dpair_ok ? "mm_fbd_kernel<" + k + ",A>, then <" + k + ",B> (float64, one utterance per workgroup" +
                                   (h->wpair_ok ? std::string("); FIRST and alone while the inputs are hard: mm_fbw_kernel<") + k + ",A>, then <" + k +
                                                      ",B> (wide-exponent pairs, two utterances per workgroup)"
                                                : std::string("; FIRST and alone while the inputs are hard)")) +
                                   ", mm_dpair_finish_kernel, then for what those mark "
                             : std::string()) +
                exact;
        } else if (h->rows_ok) {
            auto ka = [&](int d) {
                for (int k : kRowKA)
                    if (h->row_ka[d] <= k) return k;
                return 0;
            };
            s = "mm_fbr_kernel<" + std::to_string(ka(0)) + ",8192,0> + mm_fbr_kernel<" + std::to_string(ka(1)) +
                ",8192,1>, then for marked utterances only " + exact;
        } else {
            s = exact;
        }
    } else if (entry == 1) {  // mm_viterbi_f32
        s = h->vit_ok ? "mm_vit_kernel + mm_vit_backtrace_kernel (mm_tropical_kernel + mm_backtrace_kernel when the int32 back-pointers are asked for)"
                      : "mm_tropical_kernel + mm_backtrace_kernel";
    } else if (entry == 3) {  // mm_alpharecursion_f32 / mm_betarecursion_f32
        const bool xa = export_on_pairs(h, 0), xb = export_on_pairs(h, 1);
        const std::string fast = (h->pair_H > 1 ? "mm_fbsx_kernel<2," + std::to_string(h->pair_H) + "> (phase A of one direction over all frames, teams of " + std::to_string(h->pair_H) + " workgroups) + "
                                                : "mm_fbx_kernel<" + std::to_string(mm_pair_nj(h->max_P1)) + "> (phase A of one direction over all frames, two utterances per workgroup) + ") +
                                 "mm_pair_export_kernel, then for marked utterances only the item kernel";
        s = h->semiring == MM_TROPICAL ? std::string("mm_tropical_kernel / mm_log_kernel<MODE_BETA, TROP>")
            : "alpha: " + (xa ? fast : std::string("mm_log_kernel<MODE_ALPHA>")) + "; beta: " + (xb ? fast : std::string("mm_log_kernel<MODE_BETA>"));
    } else if (entry == 2) {  // mm_pdfposteriors_ex: what its last call on this batch launched
        s = h->gen.last_kernels.empty() ? std::string("mm_generic_kernel (not called yet)") : h->gen.last_kernels;
    } else {
        return fail(MM_ERR_INVALID, "mm_batch_kernels: unknown entry");
    }
    snprintf(buf, n, "%s", s.c_str());
    return MM_OK;
}

// (the pair kernels keep N + 2 vectors per utterance and one workspace slot more than utterances)
static size_t ws_alpha_bytes(mm_batch_t h, int64_t N) {
    // (+ 16 KB: the service waves copy whole LDS regions of a stored row without clamping, dma_row_b128)
    if (h->pairs_ok) return align_up(size_t(h->B + 1) * size_t(h->pair_H > 1 ? h->split_s1p : h->max_S1p) * size_t(N + 2) * 4 + 16384, 256);
    return align_up(size_t(h->total_s1p) * size_t(N + 1) * 4, 256);
}
static size_t ws_c_bytes(mm_batch_t h, int64_t N) { return align_up(size_t(h->B + 1) * size_t(N + 2) * 8, 256); }
// (split kernels) what the workgroups of a team send each other: [2 phases][pairs][2 directions][H sets][2 slots][2 * split_s1p]
// floats of rows, then [pairs][2][H][4 slots][512] floats of per-pdf partial sums; zeroed before every call
static size_t ws_x_rows_bytes(mm_batch_t h) {
    return h->pair_H > 1 ? size_t(2) * size_t((h->B + 1) / 2) * 2 * size_t(h->pair_H) * 2 * (2 * size_t(h->split_s1p)) * 4 : 0;
}
static size_t ws_x_bytes(mm_batch_t h) {
    return h->pair_H > 1 ? align_up(ws_x_rows_bytes(h) + size_t((h->B + 1) / 2) * 2 * size_t(h->pair_H) * 4 * size_t(mm_pair_xps(h->max_P1, h->pair_H)) * 4, 256) : 0;
}
// ... and for the teams of the float64 kernels (one utterance per team: B "pairs")
static size_t ws_xd_rows_bytes(mm_batch_t h) {
    return h->pair_H > 1 && h->dpair_ok ? size_t(2) * size_t(h->B) * 2 * size_t(h->pair_H) * 2 * (2 * size_t(h->split_s1p)) * 4 : 0;
}
static size_t ws_xd_bytes(mm_batch_t h) {
    // (+ 64 KB: the wide pair teams lay their slots of per-pdf sums -- two doubles per pdf, 16 bytes more per slot for 128 pdfs -- over
    // the same area, (B + 1) / 2 teams instead of B)
    return h->pair_H > 1 && h->dpair_ok ? align_up(ws_xd_rows_bytes(h) + size_t(h->B) * 2 * size_t(h->pair_H) * 4 * size_t(mm_pair_xps(h->max_P1, h->pair_H)) * 4 + 65536, 256) : 0;
}
static size_t ws_tail_bytes(mm_batch_t h) {  // longest-first order, redo marks, pair hand-over, per-direction log Z minima, team buffers
    return 2 * align_up(size_t(h->B + 1) * 4, 256) + align_up(size_t(h->B + 1) * 2 * mm_pair_hand_bytes(), 256) +
           align_up(size_t(h->B) * 6 * 8, 256) + ws_x_bytes(h) + ws_xd_bytes(h) + align_up(size_t(h->B + 1) * 4, 256);  // (last: redo2)
}

// (quad kernels) the emissions shifted by their per-frame maxima [B][N][P], and the maxima [B][N]
static size_t ws_shift_bytes(mm_batch_t h, int64_t N) {
    if (h->stream_ok) {  // (stream kernels: the backward direction's vectors and offsets, the frames' log Z, the teams' exchange area)
        size_t off[4];
        return mm_stream_extra_bytes(h->B, h->total_s1p, N, off, h->stream_H, h->stream_S1);
    }
    if (!h->fast_ok || !h->quad_built) return 0;
    return align_up(size_t(h->B) * size_t(N) * size_t(h->max_P1 - 1) * 4, 256) + align_up(size_t(h->B) * size_t(N) * 4, 256);
}
size_t mm_batch_workspace_bytes(mm_batch_t h, int64_t N) {
    if (!h || N < 0) return 0;
    if (h->log_twin) return mm_batch_workspace_bytes(h->log_twin, N) + align_up(size_t(h->B) * size_t(N) * size_t(h->max_P1 - 1) * 4, 256);
    return ws_alpha_bytes(h, N) + ws_c_bytes(h, N) + ws_tail_bytes(h) + ws_shift_bytes(h, N);
}


// Viterbi on the row-lane form (mm_kernel_vit.hip): one-byte back-pointers [B][N + 1][row], rows padded to 256 bytes, + the
// KB the back-trace's last DMA may read past the end
static size_t vit_bp_row(mm_batch_t h) { return (size_t(h->max_S1p) + 255) & ~size_t(255); }
static size_t ws_vit_bytes(mm_batch_t h, int64_t N) {
    if (h->semiring != MM_TROPICAL) return 0;
    return h->vit_ok ? size_t(h->B) * size_t(N + 1) * vit_bp_row(h) + 1024 : align_up(size_t(h->total_states) * size_t(N + 1) * 4, 256);
}

static int ensure_ws(mm_batch_t h, size_t bytes, void *stream = nullptr) {
    if (h->ws_bytes >= bytes) return MM_OK;
    // growing frees the old workspace: never while the caller's stream is capturing (a graph captured earlier on this
    // batch has the old pointers baked in; mm_batch_reserve is the way to size it up front)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(MM_ERR_INVALID, "the workspace would have to grow during stream capture: call mm_batch_reserve first");
    if (h->ws) {
        HIP_TRY(hipFree(h->ws));  // synchronises: only on growth
        h->ws = nullptr;
        h->ws_bytes = 0;
        h->last_redo = nullptr;
        h->last_z = nullptr;
    }
    HIP_TRY(hipMalloc(&h->ws, bytes));
    h->ws_bytes = bytes;
    return MM_OK;
}

// (ProbSemiring batches) room for the logarithms of a call's likelihoods [B][N][P]; grown like the workspace: never during a capture
static int ensure_prob_logv(mm_batch_t h, int64_t N, void *stream) {
    const size_t bytes = align_up(size_t(h->B) * size_t(N) * size_t(h->max_P1 - 1) * 4, 256);
    if (h->prob_logv_bytes >= bytes) return MM_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(MM_ERR_INVALID, "the workspace would have to grow during stream capture: call mm_batch_reserve first");
    if (h->prob_logv) {
        HIP_TRY(hipFree(h->prob_logv));
        h->prob_logv = nullptr;
        h->prob_logv_bytes = 0;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->prob_logv), bytes));
    h->prob_logv_bytes = bytes;
    return MM_OK;
}

int mm_batch_reserve(mm_batch_t h, int64_t N) {
    if (!h || N < 1) return fail(MM_ERR_INVALID, "mm_batch_reserve: bad argument");
    if (h->log_twin) {
        const int rc = ensure_prob_logv(h, N, nullptr);
        return rc ? rc : mm_batch_reserve(h->log_twin, N);
    }
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != h->device) return fail(MM_ERR_INVALID, "mm_batch_reserve: batch lives on another device");
    // (the total-sum rows and the max-marginals share the workspace: [N + 1][total states] words; the Viterbi back-pointers:
    // one-byte rows for the row-lane kernels, int32 rows for the item kernel)
    return ensure_ws(h, std::max({mm_batch_workspace_bytes(h, N), align_up(size_t(h->total_states) * size_t(N + 1) * 4, 256), ws_vit_bytes(h, N)}));
}

static int check_run(mm_batch_t h, const char *who, const float *V, int64_t N, int want_semiring) {
    if (!h) return fail(MM_ERR_INVALID, std::string(who) + ": NULL batch");
    if (!V) return fail(MM_ERR_INVALID, std::string(who) + ": V is NULL");
    if (N < 1 || N > (int64_t(1) << 30)) return fail(MM_ERR_DIM, std::string(who) + ": need N >= 1");
    if (want_semiring >= 0 && h->semiring != want_semiring)
        return fail(MM_ERR_INVALID, std::string(who) + ": batch was built for another semiring");
    if (h->semiring == MM_PROB)
        return fail(MM_ERR_UNSUPPORTED, std::string(who) + ": ProbSemiring batches run through mm_pdfposteriors_f32 (Float32 FSMs) or mm_pdfposteriors_ex");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != h->device) return fail(MM_ERR_INVALID, std::string(who) + ": batch lives on another device");
    // (every run entry reuses the workspace the redo marks of the last pdfposteriors call live in: mm_batch_last_redo_count
    // must not read what another entry wrote there)
    h->last_redo = nullptr;
    h->last_redo2 = nullptr;
    h->last_z = nullptr;
    return MM_OK;
}

// ProbSemiring batches on the fast kernels: V holds LIKELIHOODS (values of the semiring, as the reference's Array{ProbSemiring} does);
// one pass takes their logarithms, the log twins' kernels do the work, ttl comes back as the probability exp(log Z) -- gamma is a
// probability in both semirings (src/inference.jl:158-160: the quotient, and exp() of it for the log-like semirings only)
static __global__ void mm_prob_log_kernel(const float *V, long long vsb, long long vsn, int N, int P, float *out) {
    const long long row = (long long)blockIdx.x;  // (b, n)
    const float *src = V + (row / N) * vsb + (row % N) * vsn;
    float *dst = out + row * P;
    for (int q = threadIdx.x; q < P; q += blockDim.x) dst[q] = logf(src[q]);  // (log 0 = -inf = zero(LogSemiring))
}
static __global__ void mm_prob_exp_kernel(float *ttl, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) ttl[b] = expf(ttl[b]);
}
static int prob_pdfposteriors(mm_batch_t h, const float *V, int64_t vsb, int64_t vsn, const int32_t *lens, int64_t N, float *gamma,
                              int64_t gsb, int64_t gsn, int64_t gsp, float *ttl, void *stream) {
    if (!V || !gamma || !ttl) return fail(MM_ERR_INVALID, "mm_pdfposteriors_f32: V / gamma / ttl is NULL");
    if (N < 1 || N > (int64_t(1) << 30)) return fail(MM_ERR_DIM, "mm_pdfposteriors_f32: need N >= 1");
    if (!h->log_twin)
        return fail(MM_ERR_UNSUPPORTED, "mm_pdfposteriors_f32: this ProbSemiring batch has no log twins (Float64 FSMs, or negative weights): mm_pdfposteriors_ex");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != h->device) return fail(MM_ERR_INVALID, "mm_pdfposteriors_f32: batch lives on another device");
    int rc = ensure_prob_logv(h, N, stream);
    if (rc) return rc;
    const int P = h->max_P1 - 1;
    hipLaunchKernelGGL(mm_prob_log_kernel, dim3(unsigned(h->B * N)), dim3(P <= 64 ? 64 : (P <= 128 ? 128 : 256)), 0, static_cast<hipStream_t>(stream), V,
                       (long long)vsb, (long long)vsn, int(N), P, h->prob_logv);
    HIP_TRY(hipGetLastError());
    rc = mm_pdfposteriors_f32(h->log_twin, h->prob_logv, int64_t(N) * P, P, lens, N, gamma, gsb, gsn, gsp, ttl, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(mm_prob_exp_kernel, dim3(unsigned((h->B + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), ttl, int(h->B));
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

int mm_pdfposteriors_f32(mm_batch_t h, const float *V, int64_t vsb, int64_t vsn, const int32_t *lens, int64_t N,
                         float *gamma, int64_t gsb, int64_t gsn, int64_t gsp, float *ttl, void *stream) {
    if (h && h->semiring == MM_PROB) return prob_pdfposteriors(h, V, vsb, vsn, lens, N, gamma, gsb, gsn, gsp, ttl, stream);
    int rc = check_run(h, "mm_pdfposteriors_f32", V, N, MM_LOG);
    if (rc) return rc;
    if (!gamma || !ttl) return fail(MM_ERR_INVALID, "mm_pdfposteriors_f32: gamma/ttl is NULL");
    h->last_redo = nullptr;
    h->last_z = nullptr;
    // more utterances than CUs and different lengths: hand the workgroups out longest first; the pair kernels also
    // pair the utterances in that order (the two of a pair run the same number of frames)
    const bool ordered = lens && (h->B > h->n_cus || h->pairs_ok) && h->B <= 8192;
    rc = ensure_ws(h, mm_batch_workspace_bytes(h, N), stream);
    if (rc) return rc;
    RunParams p{};
    p.utts = h->d_utts;
    p.V = V;
    p.vsb = vsb;
    p.vsn = vsn;
    p.lens = lens;
    p.N = int(N);
    p.B = int(h->B);
    p.x_sleep = h->dbg.x_sleep;
    // (the backward agent's phase-B steps are ~10 % dearer than the forward agent's on the pair kernels -- cycle stamps, config 3:
    // 4990 against 4520 --, the phase-A steps alike: the cut that levels both launches lies at 0.48 of the frames; the steps of
    // the team kernels are bound by the exchange, the same in both directions)
    p.split_q10 = h->dbg.split_q10 > 0 ? h->dbg.split_q10 : (h->pair_H == 1 ? MM_PAIR_SPLIT_Q10 : 512);
    p.x_timeout = std::min<unsigned long long>(10000000ull, std::max<unsigned long long>(200000ull, 1000ull * (unsigned long long)N));
    p.lt_floor = h->lt_floor;
    p.clear_marks = h->keep_marks ? 0 : 1;
    p.g_scale = h->g_scale;
    p.g_acc = h->g_acc ? 1 : 0;
    p.ws_alpha = static_cast<float *>(h->ws);
    p.ws_c = reinterpret_cast<double *>(static_cast<char *>(h->ws) + ws_alpha_bytes(h, N));
    p.gamma = gamma;
    p.gsb = gsb;
    p.gsn = gsn;
    p.gsp = gsp;
    p.ttl = ttl;
    p.xcsr = h->xcsr;
    char *const tail0 = static_cast<char *>(h->ws) + ws_alpha_bytes(h, N) + ws_c_bytes(h, N);
    int *const order = ordered ? reinterpret_cast<int *>(tail0) : nullptr;  // (first of the tail)
    p.order = order;
    const bool marks = !h->wave_ok && !h->lane_ok && (h->rows_ok || h->pairs_ok || h->stream_ok);
    // Which kernels first?  The float32 pair kernels, unless the inputs of the last finished call were beyond them for some of
    // its utterances (a sharp acoustic model marks every utterance) AND starting with the float64 kernels costs no more rounds
    // of workgroups than the float32 kernels followed by the float64 kernels for that many utterances would -- a launch lasts as
    // long as its rounds (one workgroup per compute unit), whatever the number of workgroups in the last of them: config 3
    // (B = 256: one round of pairs, two of single utterances) with 38 utterances marked is 6.0 ms float32-first, 5.5 ms
    // float64-first; B = 512 with the same 38 stays float32-first.  (Until round 4: "more than a quarter of the utterances".)
    // The float64 kernels keep reporting (how many utterances have an overlap term below the float32 kernels' floor), so the
    // choice follows the data back as well.  Read without synchronising: the count of whatever call finished last.
    bool exact_first = false;
    if (marks) {
        p.redo = reinterpret_cast<int *>(tail0 + align_up(size_t(h->B + 1) * 4, 256));
        if (h->dpair_ok) {
            const int64_t M = h->stat_host[0], H = h->pair_H, cus = std::max(1, h->n_cus);
            auto rounds = [&](int64_t wgs) { return (wgs + cus - 1) / cus; };
            // (the wide pair kernels take a whole batch in the float32 kernels' one grid per phase, a third slower: cheaper than the
            // float32 kernels followed by a round of the float64 ones for ANY number of marked utterances)
            const bool by_rounds = M > 0 && (h->wpair_ok || rounds(2 * h->B * H) <= rounds(2 * ((h->B + 1) / 2) * H) + rounds(2 * std::min<int64_t>(M, h->B) * H));
            // (a capture must not bake in the marks of whatever call finished last: the automatic choice is float32-first there)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
            exact_first = h->exact_first >= 0 ? h->exact_first != 0 : (by_rounds && !capturing);
            p.redo2 = reinterpret_cast<int *>(tail0 + ws_tail_bytes(h) - align_up(size_t(h->B + 1) * 4, 256));
            p.stat_dev = h->stat_dev;
            p.stat_xcd = h->xcd_counting ? 1 : 0;
            p.stat_host = h->stat_host;
            p.stat_seq = ++h->stat_seq;
            p.stat_mode = exact_first ? 1 : 0;
        }
    }
    if (ordered || marks) {
        // (the marks are set here, on the caller's stream, ahead of everything: the forward and the backward agents run
        // concurrently and either may mark an utterance first)
        hipLaunchKernelGGL(mm_prologue_kernel, dim3(unsigned((h->B + 1 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           lens, int(h->B), int(N), order, p.redo, p.redo2, exact_first ? 1 : 0);
        HIP_TRY(hipGetLastError());
    }
#ifdef MM_STAMPS
    if (!g_dbg) {
        g_dbg_n = size_t(16) * MM_MAX_WAVES * size_t(h->B);
        (void)hipMalloc(&g_dbg, sizeof(unsigned long long) * g_dbg_n);
    }
    p.dbg = g_dbg;
#endif
    if (h->lane_ok) {
        // the lane kernel decides per utterance whether its float64 range was enough (every workgroup writes its own mark: no
        // zeroing pass); what it marks -- the one path of a left-to-right graph through values 1000 log2 below their frames'
        // maxima -- goes to the item kernel, both passes in ONE launch whose workgroups leave at once otherwise
        p.redo = reinterpret_cast<int *>(tail0 + align_up(size_t(h->B + 1) * 4, 256));
        h->last_redo = p.redo;
        rc = mm_launch_lane(h->B, h->lane_S, p, static_cast<hipStream_t>(stream));
        if (rc || h->dbg.no_redo) return rc;
        if (h->lane_redo_wave) {  // (the wave kernel's workgroups of unmarked utterances leave at once; same workspace layout)
            p.pair_zmin = reinterpret_cast<double *>(tail0 + 2 * align_up(size_t(h->B + 1) * 4, 256) + align_up(size_t(h->B + 1) * 2 * mm_pair_hand_bytes(), 256));
            WaveLaunch wlc;
            wlc.B = h->B;
            wlc.nseg = h->wave_nseg;
            wlc.max_P1 = h->max_P1;
            wlc.n_cus = h->n_cus;
            return mm_launch_wave(wlc, p, static_cast<hipStream_t>(stream));
        }
        return launch(mm_log_kernel<MODE_FB, 0, 0, false, false>, mm_log_kernel<MODE_FB, 0, 0, false, true>, h, p, true, pick_geometry(h).NW, stream);
    }
    if (h->stream_ok) {
        // the stream kernels (forward launch, backward launch, finish), then -- for the utterances whose range marks stay: values
        // beyond the double's range that carry mass, normally none -- the item kernel, both passes in one launch
        p.pair_zmin = reinterpret_cast<double *>(tail0 + 2 * align_up(size_t(h->B + 1) * 4, 256) + align_up(size_t(h->B + 1) * 2 * mm_pair_hand_bytes(), 256));
        h->last_redo = p.redo;
        h->last_z = p.pair_zmin;
        {
            size_t off[4];
            (void)mm_stream_extra_bytes(h->B, h->total_s1p, N, off, h->stream_H, h->stream_S1);
            char *base = tail0 + ws_tail_bytes(h);
            p.xbuf = reinterpret_cast<float *>(base + off[0]);    // the backward direction's vectors
            p.xbuf_d = reinterpret_cast<float *>(base + off[1]);  // ... and offsets (doubles)
            p.xps = reinterpret_cast<float *>(base + off[2]);     // the frames' {log2 Z, overlap term} (doubles)
            if (h->stream_H > 1) {  // the teams' exchange area: zeroed before every call (a zero dword = not yet arrived)
                p.sx = reinterpret_cast<unsigned *>(base + off[3]);
                p.sx_slot = (long long)mm_stream_slot(h->stream_S1);
                const size_t zn = mm_stream_exchange_bytes(h->B, h->stream_H, h->stream_S1) / 16;
                hipLaunchKernelGGL(mm_zero_kernel, dim3(unsigned(std::min<size_t>(2048, (zn + 255) / 256))), dim3(256), 0, static_cast<hipStream_t>(stream),
                                   reinterpret_cast<char *>(p.sx), (unsigned long long)zn);
                HIP_TRY(hipGetLastError());
            }
        }
        rc = mm_launch_stream(h->B, h->n_cus, h->stream_S1, h->max_P1, h->stream_H, p, static_cast<hipStream_t>(stream));
        if (rc || h->dbg.no_redo) return rc;
        return launch(mm_log_kernel<MODE_FB, 0, 0, false, false>, mm_log_kernel<MODE_FB, 0, 0, false, true>, h, p, true, pick_geometry(h).NW, stream);
    }
    if (h->wave_ok) {
        char *tail = static_cast<char *>(h->ws) + ws_alpha_bytes(h, N) + ws_c_bytes(h, N);
        p.pair_zmin = reinterpret_cast<double *>(tail + 2 * align_up(size_t(h->B + 1) * 4, 256) + align_up(size_t(h->B + 1) * 2 * mm_pair_hand_bytes(), 256));
        WaveLaunch wlc;
        wlc.B = h->B;
        wlc.nseg = h->wave_nseg;
        wlc.max_P1 = h->max_P1;
        wlc.n_cus = h->n_cus;
        return mm_launch_wave(wlc, p, static_cast<hipStream_t>(stream));
    }
    if (h->rows_ok || h->pairs_ok) {
        // the pair or row kernels, then -- for the utterances they marked (linear sums outside the trusted range),
        // normally none: every workgroup then leaves at once -- the exact kernels
        char *tail = tail0;
        h->last_redo = p.redo;
        h->last_exact_first = exact_first;
        if (h->pairs_ok) {
            p.pair_s1p = h->pair_H > 1 ? h->split_s1p : h->max_S1p;
            p.pair_hand = tail + 2 * align_up(size_t(h->B + 1) * 4, 256);
            p.pair_zmin = reinterpret_cast<double *>(static_cast<char *>(p.pair_hand) + align_up(size_t(h->B + 1) * 2 * mm_pair_hand_bytes(), 256));
            if (h->pair_H > 1) {
                const SplitInfo &si = h->fsms[0]->split;
                for (int s = 0; s < h->pair_H; ++s) {
                    p.sp_base[s] = si.base[s];
                    p.sp_cnt[s] = si.count[s];
                }
                p.xbuf = reinterpret_cast<float *>(reinterpret_cast<char *>(p.pair_zmin) + align_up(size_t(h->B) * 6 * 8, 256));
                p.xps = reinterpret_cast<float *>(reinterpret_cast<char *>(p.xbuf) + ws_x_rows_bytes(h));
                p.x_slot = 2ll * h->split_s1p;
                p.x_psn = mm_pair_xps(h->max_P1, h->pair_H);
                p.x_phase = (long long)(ws_x_rows_bytes(h) / 8);
                p.x_sleep = h->dbg.x_sleep;
                p.xbuf_d = reinterpret_cast<float *>(reinterpret_cast<char *>(p.xbuf) + ws_x_bytes(h));
                p.xps_d = reinterpret_cast<float *>(reinterpret_cast<char *>(p.xbuf_d) + ws_xd_rows_bytes(h));
                p.x_phase_d = (long long)(ws_xd_rows_bytes(h) / 8);
                // (one memset node over what this call can touch: a call that goes to the float64 kernels with the whole batch leaves
                // the float32 kernels' area alone, one that starts with the float32 kernels zeroes both -- marks are not known here)
                p.x_H = h->dpair_ok ? h->pair_H : 0;
                // (one memset node: the float64 kernels' area when they take the whole batch, else the float32 kernels' -- the
                // finish kernel then zeroes the float64 area of the utterances it leaves marked)
                // (a kernel of the library's own, not hipMemsetAsync: inside a captured hipGraph the runtime's memset node has been seen
                // to overlap the team kernel behind it on replay -- the handshake granules of a launch wiped after they were
                // published, every team timing out and every utterance recomputed by the exact kernels: round 6,
                // tests/test_gpu_lfmmi.py::test_lfmmi_step_in_one_hip_graph; and the runtime's memset is two fill kernels, this is one)
                {
                    char *zp = reinterpret_cast<char *>(exact_first ? p.xbuf_d : p.xbuf);
                    const size_t zn = (exact_first ? ws_xd_bytes(h) : ws_x_bytes(h)) / 16;  // (both sizes are multiples of 256)
                    const unsigned zb = unsigned(std::min<size_t>(2048, (zn + 255) / 256));
                    hipLaunchKernelGGL(mm_zero_kernel, dim3(zb), dim3(256), 0, static_cast<hipStream_t>(stream), zp, (unsigned long long)zn);
                    HIP_TRY(hipGetLastError());
                }
            }
            h->last_z = p.pair_zmin;
            rc = exact_first ? MM_OK : launch_pairs(h, p, stream);
        } else {
            rc = launch_rows(h, p, stream);
        }
        if (rc) return rc;
        if (h->dbg.no_redo && !exact_first) return MM_OK;
        if (h->dpair_ok) {
            // the float64 pair kernels for the marked utterances (workgroups of the others leave at once); what THEY mark --
            // values beyond the double's range that carry mass -- is left in redo2 for the log-domain kernels below
            PairLaunch pl;
            pl.B = h->B;
            pl.nwc = h->pair_nwc;
            pl.slotrows = h->pair_slotrows;
            pl.max_P1 = h->max_P1;
            pl.pair_ka = h->pair_ka;
            pl.H = h->pair_H;
            // a whole batch (exact_first: every utterance is marked): two utterances per workgroup on the wide pair kernels
            rc = exact_first && h->wpair_ok ? mm_launch_wpairs(pl, p, static_cast<hipStream_t>(stream)) : mm_launch_dpairs(pl, p, static_cast<hipStream_t>(stream));
            if (rc) return rc;
            h->last_redo2 = p.redo2;
            p.redo = p.redo2;
            if (h->dbg.no_redo || h->dbg.no_fallback) return MM_OK;
            // what the float64 kernels hand on (values beyond a double's range that carry mass: normally nothing) goes to the
            // item kernel, both passes in ONE launch (streamed items: its speed does not matter, the empty launch's does)
            if (!quad_kernel_usable(h))
                return launch(mm_log_kernel<MODE_FB, 0, 0, false, false>, mm_log_kernel<MODE_FB, 0, 0, false, true>, h, p, true, pick_geometry(h).NW, stream);
        }
    }
    if (quad_kernel_usable(h)) {
        // the quad kernels run on emissions shifted by their per-frame maxima (see mm_shift_em_kernel)
        bool same_P = true;
        for (int64_t b = 1; b < h->B && same_P; ++b) same_P = h->fsms[b]->P1 == h->fsms[0]->P1;
        RunParams q = p;
        float *Vs = nullptr, *E = nullptr;
        if (same_P) {
            char *base = static_cast<char *>(h->ws) + ws_alpha_bytes(h, N) + ws_c_bytes(h, N) + ws_tail_bytes(h);
            Vs = reinterpret_cast<float *>(base);
            E = reinterpret_cast<float *>(base + align_up(size_t(h->B) * size_t(N) * size_t(h->max_P1 - 1) * 4, 256));
            hipLaunchKernelGGL(mm_shift_em_kernel, dim3(unsigned(h->B), 8), dim3(256), 0, static_cast<hipStream_t>(stream), p, Vs, E);
            HIP_TRY(hipGetLastError());
            q.V = Vs;
            q.vsb = N * int64_t(h->max_P1 - 1);
            q.vsn = h->max_P1 - 1;
        }
        rc = launch_quad(h, q, stream);
        if (rc != MM_ERR_UNSUPPORTED) {
            if (!rc && same_P) {
                hipLaunchKernelGGL(mm_shift_ttl_kernel, dim3(unsigned(h->B)), dim3(256), 0, static_cast<hipStream_t>(stream), p, E);
                HIP_TRY(hipGetLastError());
            }
            return rc;
        }
    }
    return launch_log<MODE_FB>(h, p, stream);
}

// alpha / beta export on the pair kernels (mm_pairs_tu.hip: mm_fbx_kernel + mm_pair_export_kernel): one shared graph in the pair
// form whose every state takes part (the packer drops states that cannot be reached or cannot reach the final state -- their
// posteriors are zero, their alpha / beta are not), log semiring, up to 250 pdfs
static PairLaunch pair_launch_of(mm_batch_t h) {
    PairLaunch pl;
    pl.B = h->B;
    pl.nwc = h->pair_nwc;
    pl.slotrows = h->pair_slotrows;
    pl.max_P1 = h->max_P1;
    pl.pair_ka = h->pair_ka;
    pl.H = h->pair_H;
    pl.small = h->pair_H == 1 && h->max_S1p <= 128;
    return pl;
}
static bool export_on_pairs(mm_batch_t h, int dir) {
    if (h->semiring != MM_LOG || !h->pairs_ok) return false;
    if (!(h->pair_H == 1 ? mm_pair_export_fits(pair_launch_of(h)) : mm_split_export_fits(pair_launch_of(h)))) return false;
    return h->fsms[0]->export_ok[dir];
}

static int run_export(mm_batch_t h, int mode, const float *V, int64_t vsb, int64_t vsn, const int32_t *lens, int64_t N,
                      float *out, int64_t out_stride_n, void *stream) {
    const char *who = mode == MODE_ALPHA ? "mm_alpharecursion_f32" : "mm_betarecursion_f32";
    int rc = check_run(h, who, V, N, -1);
    if (rc) return rc;
    if (!out) return fail(MM_ERR_INVALID, std::string(who) + ": out is NULL");
    if (out_stride_n < h->total_states) return fail(MM_ERR_DIM, std::string(who) + ": out_stride_n < total states");
    RunParams p{};
    p.utts = h->d_utts;
    p.V = V;
    p.vsb = vsb;
    p.vsn = vsn;
    p.lens = lens;
    p.N = int(N);
    p.B = int(h->B);
    p.out = out;
    p.out_stride_n = out_stride_n;
    // (inputs whose vectors leave float32's range in most utterances -- the reference's WSJ denominator in the forward direction: its
    // initial-context states decay 2 log2 per frame against the rest, and the reference's alpha holds them as finite logarithms -- go
    // straight to the item kernel while the last finished export of the direction marked more than half of its utterances (read from
    // pinned host memory without synchronising, like the exact policy of mm_pdfposteriors_f32); every 32nd call tries again)
    bool linear_first = export_on_pairs(h, mode == MODE_ALPHA ? 0 : 1);
    // (mm_batch_set_exact_policy pins this choice too: F32_FIRST = always the linear-domain kernels first, F64_FIRST = always the item
    // kernel alone -- with a fixed policy the launches of a call, and the last bits of its result, are a function of the call)
    if (linear_first && h->exact_first == 1) linear_first = false;
    else if (linear_first && h->stat_host && h->exact_first < 0) {
        const int dirx = mode == MODE_ALPHA ? 0 : 1;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        const bool probe = (++h->export_calls[dirx] & 31u) == 0u;
        if (!capturing && !probe && 2 * int64_t(h->stat_host[2 + dirx]) > h->B) linear_first = false;
    }
    if (linear_first) {
        // phase A of ONE direction over all N + 1 frames with two utterances per workgroup (linear domain, float32), the layout
        // pass, then -- for the utterances whose values left float32's range (marked; sharp emissions) -- the item kernel
        rc = ensure_ws(h, mm_batch_workspace_bytes(h, N), stream);
        if (rc) return rc;
        char *const tail0 = static_cast<char *>(h->ws) + ws_alpha_bytes(h, N) + ws_c_bytes(h, N);
        p.ws_alpha = static_cast<float *>(h->ws);
        p.ws_c = reinterpret_cast<double *>(static_cast<char *>(h->ws) + ws_alpha_bytes(h, N));
        p.redo = reinterpret_cast<int *>(tail0 + align_up(size_t(h->B + 1) * 4, 256));
        p.pair_s1p = h->pair_H > 1 ? h->split_s1p : h->max_S1p;
        p.pair_hand = tail0 + 2 * align_up(size_t(h->B + 1) * 4, 256);
        p.pair_zmin = reinterpret_cast<double *>(static_cast<char *>(p.pair_hand) + align_up(size_t(h->B + 1) * 2 * mm_pair_hand_bytes(), 256));
        p.split_q10 = 512;
        p.lt_floor = h->lt_floor;
        p.x_sleep = h->dbg.x_sleep;
        p.x_timeout = std::min<unsigned long long>(10000000ull, std::max<unsigned long long>(200000ull, 1000ull * (unsigned long long)N));
        hipLaunchKernelGGL(mm_prologue_kernel, dim3(unsigned((h->B + 1 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           (const int *)nullptr, int(h->B), int(N), (int *)nullptr, p.redo, (int *)nullptr, 0);
        HIP_TRY(hipGetLastError());
        if (h->pair_H > 1) {  // the teams' exchange areas (the float32 team kernels': phase A's rows), zeroed like before a pdfposteriors call
            const SplitInfo &si = h->fsms[0]->split;
            for (int s = 0; s < h->pair_H; ++s) {
                p.sp_base[s] = si.base[s];
                p.sp_cnt[s] = si.count[s];
            }
            p.xbuf = reinterpret_cast<float *>(reinterpret_cast<char *>(p.pair_zmin) + align_up(size_t(h->B) * 6 * 8, 256));
            p.xps = reinterpret_cast<float *>(reinterpret_cast<char *>(p.xbuf) + ws_x_rows_bytes(h));
            p.x_slot = 2ll * h->split_s1p;
            p.x_psn = mm_pair_xps(h->max_P1, h->pair_H);
            p.x_phase = (long long)(ws_x_rows_bytes(h) / 8);
            const size_t zn = ws_x_bytes(h) / 16;
            hipLaunchKernelGGL(mm_zero_kernel, dim3(unsigned(std::min<size_t>(2048, (zn + 255) / 256))), dim3(256), 0, static_cast<hipStream_t>(stream),
                               reinterpret_cast<char *>(p.xbuf), (unsigned long long)zn);
            HIP_TRY(hipGetLastError());
            rc = mm_launch_split_export(pair_launch_of(h), p, mode == MODE_ALPHA ? 0 : 1, static_cast<hipStream_t>(stream));
        } else {
            rc = mm_launch_pair_export(pair_launch_of(h), p, mode == MODE_ALPHA ? 0 : 1, static_cast<hipStream_t>(stream));
        }
        if (rc) return rc;
        if (h->stat_host) {  // how many utterances were marked: what the next export of this direction starts with
            hipLaunchKernelGGL(mm_count_marks_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), p.redo, int(h->B),
                               const_cast<int *>(h->stat_host) + 2 + (mode == MODE_ALPHA ? 0 : 1));
            HIP_TRY(hipGetLastError());
        }
        h->last_redo = p.redo;  // (mm_batch_last_redo_count: how many utterances the item kernel computed instead)
        if (h->dbg.no_redo) return MM_OK;
        p.ws_alpha = nullptr;   // (the item kernel's export modes keep nothing in the workspace)
        p.ws_c = nullptr;
        return mode == MODE_ALPHA ? launch_log<MODE_ALPHA>(h, p, stream) : launch_log<MODE_BETA>(h, p, stream);
    }
    if (h->semiring == MM_TROPICAL) {
        if (mode == MODE_ALPHA) return launch_tropical(h, p, stream);
        const Geometry g = pick_geometry(h);
        if (g.NI == 0) return launch(mm_log_kernel<MODE_BETA, 0, 0, true, false>, mm_log_kernel<MODE_BETA, 0, 0, true, true>, h, p, false, g.NW, stream);
        return launch(mm_log_kernel<MODE_BETA, 8, 0, true, false>, mm_log_kernel<MODE_BETA, 8, 0, true, true>, h, p, false, g.NW, stream);
    }
    if (mode == MODE_ALPHA) return launch_log<MODE_ALPHA>(h, p, stream);
    return launch_log<MODE_BETA>(h, p, stream);
}

int mm_alpharecursion_f32(mm_batch_t h, const float *V, int64_t vsb, int64_t vsn, const int32_t *lens, int64_t N,
                          float *out, int64_t out_stride_n, void *stream) {
    return run_export(h, MODE_ALPHA, V, vsb, vsn, lens, N, out, out_stride_n, stream);
}

int mm_betarecursion_f32(mm_batch_t h, const float *V, int64_t vsb, int64_t vsn, const int32_t *lens, int64_t N,
                         float *out, int64_t out_stride_n, void *stream) {
    return run_export(h, MODE_BETA, V, vsb, vsn, lens, N, out, out_stride_n, stream);
}

int mm_maxstateposteriors_f32(mm_batch_t h, const float *V, int64_t vsb, int64_t vsn, const int32_t *lens, int64_t N,
                              float *out, int64_t out_stride_n, void *stream) {
    int rc = check_run(h, "mm_maxstateposteriors_f32", V, N, MM_TROPICAL);
    if (rc) return rc;
    if (!out) return fail(MM_ERR_INVALID, "mm_maxstateposteriors_f32: out is NULL");
    if (out_stride_n < h->total_states) return fail(MM_ERR_DIM, "mm_maxstateposteriors_f32: out_stride_n < total states");
    // tropical alpha into `out`, tropical beta into the workspace, then mu = alpha (*) beta (/) best in place
    const size_t beta_bytes = align_up(size_t(h->total_states) * size_t(N + 1) * 4, 256);
    rc = ensure_ws(h, beta_bytes + align_up(size_t(h->B) * 4, 256), stream);
    if (rc) return rc;
    float *beta = static_cast<float *>(h->ws);
    float *best = reinterpret_cast<float *>(static_cast<char *>(h->ws) + beta_bytes);
    rc = run_export(h, MODE_ALPHA, V, vsb, vsn, lens, N, out, out_stride_n, stream);
    if (rc) return rc;
    rc = run_export(h, MODE_BETA, V, vsb, vsn, lens, N, beta, h->total_states, stream);
    if (rc) return rc;
    const int bt = 64;
    hipLaunchKernelGGL(mm_pick_final_kernel, dim3(unsigned((h->B + bt - 1) / bt)), dim3(bt), 0, static_cast<hipStream_t>(stream),
                       h->d_utts, int(h->B), out, (long long)out_stride_n, int(N), best);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(mm_maxmarginal_kernel, dim3(unsigned(h->B), unsigned(N + 1)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       h->d_utts, out, (long long)out_stride_n, beta, (long long)h->total_states, best);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

// ---- RCCL (resolved at run time: the communicator belongs to the RCCL of the calling process) ----
namespace {
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
enum { MM_NCCL_SUM = 0, MM_NCCL_FLOAT32 = 7, MM_NCCL_FLOAT64 = 8 };  // rccl.h: ncclSum, ncclFloat32, ncclFloat64
// The RCCL the caller's communicators belong to: handed over with mm_set_rccl (a dlopen handle), else whatever the
// process exposes globally.  The library never opens an RCCL of its own: an ncclComm_t made by one build of RCCL passed
// to another is undefined behaviour.
void *g_rccl_handle = nullptr;
std::mutex g_rccl_lock;
void *rccl_symbol(const char *name) {
    std::lock_guard<std::mutex> guard(g_rccl_lock);
    if (g_rccl_handle)
        if (void *f = dlsym(g_rccl_handle, name)) return f;
    return dlsym(RTLD_DEFAULT, name);
}
__global__ void mm_sum_f64_kernel(const float *x, long long n, double *out) {
    __shared__ double part[16];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) s += (double)x[i];
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (unsigned w = 0; w < blockDim.x / 64; ++w) t += part[w];  // (fixed order: the same sum on every run)
        *out = t;
    }
}
}  // namespace

int mm_set_rccl(void *dl_handle) {
    std::lock_guard<std::mutex> guard(g_rccl_lock);
    g_rccl_handle = dl_handle;
    return MM_OK;
}

int mm_allreduce_logz(void *comm, const float *ttl, int64_t B_local, double *sum, void *stream) {
    if (!comm || !sum || B_local < 0 || (B_local > 0 && !ttl)) return fail(MM_ERR_INVALID, "mm_allreduce_logz: bad argument");
    const nccl_allreduce_fn allreduce = reinterpret_cast<nccl_allreduce_fn>(rccl_symbol("ncclAllReduce"));
    if (!allreduce) return fail(MM_ERR_UNSUPPORTED, "mm_allreduce_logz: RCCL is not visible in this process: call mm_set_rccl with the handle of the RCCL `comm` came from");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mm_sum_f64_kernel, dim3(1), dim3(1024), 0, st, ttl, (long long)B_local, sum);
    HIP_TRY(hipGetLastError());
    if (allreduce(sum, sum, 1, MM_NCCL_FLOAT64, MM_NCCL_SUM, comm, st) != 0) return fail(MM_ERR_HIP, "mm_allreduce_logz: ncclAllReduce failed");
    return MM_OK;
}

int mm_allgather_ttl(void *comm, const float *ttl, int64_t B_max, float *all, void *stream) {
    if (!comm || !ttl || !all || B_max < 1) return fail(MM_ERR_INVALID, "mm_allgather_ttl: bad argument");
    const nccl_allgather_fn allgather = reinterpret_cast<nccl_allgather_fn>(rccl_symbol("ncclAllGather"));
    if (!allgather) return fail(MM_ERR_UNSUPPORTED, "mm_allgather_ttl: RCCL is not visible in this process: call mm_set_rccl with the handle of the RCCL `comm` came from");
    if (allgather(ttl, all, size_t(B_max), MM_NCCL_FLOAT32, comm, static_cast<hipStream_t>(stream)) != 0)
        return fail(MM_ERR_HIP, "mm_allgather_ttl: ncclAllGather failed");
    return MM_OK;
}

int mm_totalsum_f32(mm_batch_t h, int64_t n, int cumulative, float *out, void *stream) {
    static const float dummy = 0.f;
    int rc = check_run(h, "mm_totalsum_f32", &dummy, n, -1);
    if (rc) return rc;
    if (!out) return fail(MM_ERR_INVALID, "mm_totalsum_f32: out is NULL");
    // v_k and the running total live in the extended system (src/fsm.jl:19-28): the final state's self
    // loop of weight one makes it the accumulator of omega . v_k; n + 1 frames of the alpha recursion
    rc = ensure_ws(h, align_up(size_t(h->total_states) * size_t(n + 1) * 4, 256), stream);
    if (rc) return rc;
    RunParams p{};
    p.utts = h->d_utts;
    p.N = int(n);
    p.B = int(h->B);
    p.out = static_cast<float *>(h->ws);
    p.out_stride_n = h->total_states;
    p.free_run = cumulative ? 2 : 1;
    rc = h->semiring == MM_TROPICAL ? launch_tropical(h, p, stream) : launch_log<MODE_ALPHA>(h, p, stream);
    if (rc) return rc;
    const int bt = 64;
    hipLaunchKernelGGL(mm_pick_final_kernel, dim3(unsigned((h->B + bt - 1) / bt)), dim3(bt), 0,
                       static_cast<hipStream_t>(stream), h->d_utts, int(h->B), p.out, p.out_stride_n, int(n), out);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

int mm_viterbi_f32(mm_batch_t h, const float *V, int64_t vsb, int64_t vsn, const int32_t *lens, int64_t N,
                   int32_t *path, int64_t path_stride_b, float *score, int32_t *bp, int64_t bp_stride_n,
                   void *stream) {
    int rc = check_run(h, "mm_viterbi_f32", V, N, MM_TROPICAL);
    if (rc) return rc;
    if (!path || !score) return fail(MM_ERR_INVALID, "mm_viterbi_f32: path/score is NULL");
    if (path_stride_b < N) return fail(MM_ERR_DIM, "mm_viterbi_f32: path_stride_b < N");
    RunParams p{};
    if (!bp) {
        // (int32 rows for the item kernel, or one-byte rows padded to 256 bytes for the row-lane kernels: what this batch runs)
        rc = ensure_ws(h, ws_vit_bytes(h, N), stream);
        if (rc) return rc;
        bp = static_cast<int32_t *>(h->ws);
        bp_stride_n = h->total_states;
        p.stop_at_len = 1;  // nobody reads the back-pointers of the frames beyond len_b + 1
    } else if (bp_stride_n < h->total_states) {
        return fail(MM_ERR_DIM, "mm_viterbi_f32: bp_stride_n < total states");
    }
    p.utts = h->d_utts;
    p.V = V;
    p.vsb = vsb;
    p.vsn = vsn;
    p.lens = lens;
    p.N = int(N);
    p.B = int(h->B);
    p.bp = bp;
    p.bp_stride_n = bp_stride_n;
    p.path = path;
    p.path_stride_b = path_stride_b;
    p.score = score;
#ifdef MM_STAMPS
    if (!g_dbg) {
        g_dbg_n = size_t(16) * MM_MAX_WAVES * size_t(h->B);
        (void)hipMalloc(&g_dbg, sizeof(unsigned long long) * g_dbg_n);
    }
    p.dbg = g_dbg;
#endif
    if (h->vit_ok && p.stop_at_len) {  // (internal back-pointers: the compact form)
        VitLaunch vl;
        vl.B = h->B;
        vl.n4 = h->vit_n4;
        vl.n2 = h->vit_n2;
        vl.max_P1 = h->max_P1;
        vl.max_S1p = h->max_S1p;
        vl.max_arcs = h->vit_arcs;
        vl.bp_row = int(vit_bp_row(h));  // bytes of a row of one-byte back-pointers, padded to 256
        RunParams q = p;
        q.bp_stride_n = vl.bp_row;
        return mm_launch_viterbi(vl, q, static_cast<hipStream_t>(stream));
    }
    rc = launch_tropical(h, p, stream);
    if (rc) return rc;
    const int bt = 64;
    hipLaunchKernelGGL(mm_backtrace_kernel, dim3(unsigned((h->B + bt - 1) / bt)), dim3(bt), 0,
                       static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

}  // extern "C"
