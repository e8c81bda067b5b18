// mm_engine.hip -- C ABI (include/markovmodels_amd.h) over the gfx950 kernels.
// Handles, host-side compile (packing), workspace management, launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/markovmodels_amd.h"
#include "mm_kernels.hip"
#include "mm_kernel_quad.hip"
#include "mm_pack.h"

using namespace mm;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(MM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

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

struct mm_fsm_s {
    int semiring;
    int64_t S1, nnz;
    int32_t P1;
    int S1p;
    Packed packed[2];
    QuadGraph quad[2];
    bool fast_ok = false;
    QuadDev qdev[2];
    std::vector<float> init;  // dense alpha_hat, engine domain
    std::vector<int32_t> s2p;
    int device = -1;
    void *dev_blob = nullptr;
    size_t dev_bytes = 0;
    GraphDev gdev[2];
    const float *d_init = nullptr;
    const int *d_s2p = nullptr;
};

struct mm_batch_s {
    std::vector<mm_fsm_t> fsms;
    int semiring;
    int64_t B;
    int64_t total_states = 0;
    int64_t total_s1p = 0;
    int max_S1p = 0, max_P1 = 0, max_items = 0, max_quads = 0;
    bool fast_ok = true;
    int device = -1;
    UttDesc *d_utts = nullptr;
    void *ws = nullptr;
    size_t ws_bytes = 0;
};

// Launch geometry.  NW waves per workgroup, NI register-resident items per wave
// (mm_kernels.hip, "Register-resident graph"): NI <= 8 keeps 16 waves per CU inside the
// 128-VGPR budget, NI = 24 uses 8 waves with up to 256 VGPRs.
struct Geometry {
    int NW, NI;
};

static Geometry pick_geometry(mm_batch_t h) {
    Geometry g{16, 8};
    const int it = h->max_items;
    if (it <= 8 * MM_MAX_WAVES) {
        g.NI = 8;
        g.NW = std::max(1, (it + 7) / 8);
    } else {
        g.NI = 24;
        g.NW = 8;
    }
    if (const char *e = getenv("MM_NWAVES")) {
        int v = atoi(e);
        if (v >= 1 && v <= MM_MAX_WAVES) g.NW = v;
    }
    if (const char *e = getenv("MM_NITEMS")) {
        int v = atoi(e);
        if (v == 0 || v == 8 || v == 24) g.NI = v;
    }
    if (g.NI == 24 && g.NW > 8) g.NW = 8;
    return g;
}

template <typename K>
static int launch(K kernel, mm_batch_t h, const RunParams &p, bool with_stage, int NW, void *stream) {
    const int P1p = (h->max_P1 + 3) & ~3;
    const LdsPlan L = lds_plan(h->max_S1p, P1p, with_stage);
    const size_t lds = size_t(L.total) * 4;
    if (lds > 160 * 1024)
        return fail(MM_ERR_UNSUPPORTED, "FSM too large: " + std::to_string(lds) + " B of LDS needed, 163840 available");
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(lds)));
    hipLaunchKernelGGL(kernel, dim3(unsigned(h->B)), dim3(64 * NW), lds, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

template <int MODE>
static int launch_log(mm_batch_t h, const RunParams &p, void *stream) {
    const Geometry g = pick_geometry(h);
    const bool st = MODE == MODE_FB;
    switch (g.NI) {
        case 0: return launch(mm_log_kernel<MODE, 0>, h, p, st, g.NW, stream);
        case 8: return launch(mm_log_kernel<MODE, 8>, h, p, st, g.NW, stream);
        default: return launch(mm_log_kernel<MODE, 24>, h, p, st, g.NW, stream);
    }
}

// The quad kernel (mm_kernel_quad.hip): KQ register-resident quads per lane, NW waves.
template <int KQ>
static int launch_quad_kq(mm_batch_t h, const RunParams &p, int NW, void *stream) {
    const int P1p = (h->max_P1 + 3) & ~3;
    const size_t lds = size_t(lds_plan_q(h->max_S1p, P1p, std::max(h->max_quads, 64 * NW * KQ)).total) * 4;
    if (lds > 160 * 1024) return fail(MM_ERR_UNSUPPORTED, "quad kernel: LDS plan does not fit");
    auto kernel = mm_fbq_kernel<KQ>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(lds)));
    hipLaunchKernelGGL(kernel, dim3(unsigned(h->B)), dim3(64 * NW), lds, static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

static bool quad_kernel_usable(mm_batch_t h) {
    if (const char *e = getenv("MM_KERNEL"))
        if (!strcmp(e, "item")) return false;
    if (!h->fast_ok || h->semiring != MM_LOG) return false;
    const int P1p = (h->max_P1 + 3) & ~3;
    return size_t(lds_plan_q(h->max_S1p, P1p, std::max(h->max_quads, 64 * MM_MAX_WAVES)).total) * 4 <= 150 * 1024;
}

static int launch_quad(mm_batch_t h, const RunParams &p, void *stream) {
    static const int kqs[] = {1, 2, 3, 5, 7, 9, 11, 13};
    int KQ = 13;
    for (int k : kqs)
        if (64 * MM_MAX_WAVES * k >= h->max_quads) {
            KQ = k;
            break;
        }
    if (const char *e = getenv("MM_KQ")) {
        int v = atoi(e);
        for (int k : kqs)
            if (k == v) KQ = v;
    }
    int NW = std::min(MM_MAX_WAVES, std::max(1, (h->max_quads + 64 * KQ - 1) / (64 * KQ)));
    if (const char *e = getenv("MM_NWAVES")) {
        int v = atoi(e);
        if (v >= 1 && v <= MM_MAX_WAVES) NW = v;
    }
    switch (KQ) {
        case 1: return launch_quad_kq<1>(h, p, NW, stream);
        case 2: return launch_quad_kq<2>(h, p, NW, stream);
        case 3: return launch_quad_kq<3>(h, p, NW, stream);
        case 5: return launch_quad_kq<5>(h, p, NW, stream);
        case 7: return launch_quad_kq<7>(h, p, NW, stream);
        case 9: return launch_quad_kq<9>(h, p, NW, stream);
        case 11: return launch_quad_kq<11>(h, p, NW, stream);
        default: return launch_quad_kq<13>(h, p, NW, stream);
    }
}

static int tropical_waves(mm_batch_t h) {
    Geometry g = pick_geometry(h);
    return g.NI == 24 ? 16 : g.NW;
}

extern "C" {

int mm_abi_version(void) { return MM_ABI_VERSION; }
const char *mm_last_error(void) { return g_err.c_str(); }

int mm_fsm_create(int semiring, int64_t S1, int64_t nnz, int layout, int index_bytes, int index_base, int val_bytes,
                  const void *ptr, const void *idx, const void *val, int64_t n_init, const void *init_idx,
                  const void *init_val, const int32_t *state2pdf, int32_t P1, mm_fsm_t *out) {
    if (!out) return fail(MM_ERR_INVALID, "mm_fsm_create: out is NULL");
    *out = nullptr;
    if (semiring != MM_LOG && semiring != MM_TROPICAL) return fail(MM_ERR_INVALID, "mm_fsm_create: unknown semiring");
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
    f->S1p = int((S1 + 3) / 4 * 4);
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
    f->packed[0] = pack_rows(S1, fwd.rowptr, fwd.col, fwd.val, f->s2p, NINF);
    f->packed[1] = pack_rows(S1, bwd.rowptr, bwd.col, bwd.val, f->s2p, NINF);
    if (semiring == MM_LOG) {
        f->quad[0] = make_quads(S1, fwd.rowptr, fwd.col, fwd.val);
        f->quad[1] = make_quads(S1, bwd.rowptr, bwd.col, bwd.val);
        f->fast_ok = f->quad[0].fast_ok && f->quad[1].fast_ok && P1 <= 65535;
    }
    *out = f;
    return MM_OK;
}

static int fsm_to_device(mm_fsm_t f) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (f->dev_blob && f->device == dev) return MM_OK;
    if (f->dev_blob) return fail(MM_ERR_INVALID, "FSM already resident on another device");
    size_t off = 0, o_items[2], o_rows[2], o_slots[2], o_init, o_s2p, o_quads[2], o_qptr[2], o_qcol[2], o_qw[2], o_qst[2], o_qord[2];
    for (int d = 0; d < 2; ++d) {
        o_quads[d] = off;
        off = align_up(off + f->quad[d].quads.size() * sizeof(Quad), 256);
        o_qptr[d] = off;
        off = align_up(off + f->quad[d].rowptr.size() * sizeof(int32_t), 256);
        o_qcol[d] = off;
        off = align_up(off + f->quad[d].col.size() * sizeof(int32_t), 256);
        o_qw[d] = off;
        off = align_up(off + f->quad[d].w.size() * sizeof(float), 256);
        o_qst[d] = off;
        off = align_up(off + f->quad[d].qstart.size() * sizeof(uint16_t), 256);
        o_qord[d] = off;
        off = align_up(off + f->quad[d].rord.size() * sizeof(uint16_t), 256);
    }
    for (int d = 0; d < 2; ++d) {
        o_items[d] = off;
        off = align_up(off + f->packed[d].items.size() * sizeof(ItemMeta), 256);
        o_rows[d] = off;
        off = align_up(off + f->packed[d].rowinfo.size() * sizeof(RowInfo), 256);
        o_slots[d] = off;
        off = align_up(off + f->packed[d].slots.size() * sizeof(Slot), 256);
    }
    o_init = off;
    off = align_up(off + f->init.size() * sizeof(float), 256);
    o_s2p = off;
    off = align_up(off + f->s2p.size() * sizeof(int32_t), 256);
    std::vector<char> host(off, 0);
    for (int d = 0; d < 2; ++d) {
        memcpy(host.data() + o_items[d], f->packed[d].items.data(), f->packed[d].items.size() * sizeof(ItemMeta));
        memcpy(host.data() + o_rows[d], f->packed[d].rowinfo.data(), f->packed[d].rowinfo.size() * sizeof(RowInfo));
        memcpy(host.data() + o_slots[d], f->packed[d].slots.data(), f->packed[d].slots.size() * sizeof(Slot));
    }
    for (int d = 0; d < 2; ++d) {
        memcpy(host.data() + o_quads[d], f->quad[d].quads.data(), f->quad[d].quads.size() * sizeof(Quad));
        memcpy(host.data() + o_qptr[d], f->quad[d].rowptr.data(), f->quad[d].rowptr.size() * sizeof(int32_t));
        memcpy(host.data() + o_qcol[d], f->quad[d].col.data(), f->quad[d].col.size() * sizeof(int32_t));
        memcpy(host.data() + o_qw[d], f->quad[d].w.data(), f->quad[d].w.size() * sizeof(float));
        memcpy(host.data() + o_qst[d], f->quad[d].qstart.data(), f->quad[d].qstart.size() * sizeof(uint16_t));
        memcpy(host.data() + o_qord[d], f->quad[d].rord.data(), f->quad[d].rord.size() * sizeof(uint16_t));
    }
    memcpy(host.data() + o_init, f->init.data(), f->init.size() * sizeof(float));
    memcpy(host.data() + o_s2p, f->s2p.data(), f->s2p.size() * sizeof(int32_t));
    void *blob = nullptr;
    HIP_TRY(hipMalloc(&blob, off));
    hipError_t e = hipMemcpy(blob, host.data(), off, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        return fail(MM_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
    }
    char *base = static_cast<char *>(blob);
    for (int d = 0; d < 2; ++d) {
        f->gdev[d].items = reinterpret_cast<const ItemMeta *>(base + o_items[d]);
        f->gdev[d].rowinfo = reinterpret_cast<const RowInfo *>(base + o_rows[d]);
        f->gdev[d].slots = reinterpret_cast<const Slot *>(base + o_slots[d]);
        f->gdev[d].n_items = int(f->packed[d].items.size());
        f->gdev[d].pad = 0;
        f->qdev[d].quads = reinterpret_cast<const Quad *>(base + o_quads[d]);
        f->qdev[d].rowptr = reinterpret_cast<const int *>(base + o_qptr[d]);
        f->qdev[d].col = reinterpret_cast<const int *>(base + o_qcol[d]);
        f->qdev[d].w = reinterpret_cast<const float *>(base + o_qw[d]);
        f->qdev[d].qstart = reinterpret_cast<const unsigned short *>(base + o_qst[d]);
        f->qdev[d].rord = reinterpret_cast<const unsigned short *>(base + o_qord[d]);
        f->qdev[d].nq = int(f->quad[d].quads.size());
        f->qdev[d].pad = 0;
    }
    f->d_init = reinterpret_cast<const float *>(base + o_init);
    f->d_s2p = reinterpret_cast<const int *>(base + o_s2p);
    f->dev_blob = blob;
    f->dev_bytes = off;
    f->device = dev;
    return MM_OK;
}

int mm_fsm_destroy(mm_fsm_t f) {
    if (!f) return MM_OK;
    if (f->dev_blob) (void)hipFree(f->dev_blob);
    delete f;
    return MM_OK;
}

int mm_fsm_info(mm_fsm_t f, int64_t *S1, int64_t *nnz, int32_t *P1, int64_t packed_slots[2], int64_t packed_items[2]) {
    if (!f) return fail(MM_ERR_INVALID, "mm_fsm_info: NULL handle");
    if (S1) *S1 = f->S1;
    if (nnz) *nnz = f->nnz;
    if (P1) *P1 = f->P1;
    for (int d = 0; d < 2; ++d) {
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
    eval_packed(f->packed[direction], f->semiring, x.data(), out, argmax, f->S1);
    if (f->semiring == MM_LOG)
        for (int64_t i = 0; i < f->S1; ++i) out[i] *= MM_LN2;
    return MM_OK;
}

int mm_batch_create(const mm_fsm_t *fsms, int64_t B, mm_batch_t *out) {
    if (!out) return fail(MM_ERR_INVALID, "mm_batch_create: out is NULL");
    *out = nullptr;
    if (!fsms || B < 1) return fail(MM_ERR_INVALID, "mm_batch_create: empty batch");
    for (int64_t b = 0; b < B; ++b) {
        if (!fsms[b]) return fail(MM_ERR_INVALID, "mm_batch_create: NULL FSM handle");
        if (fsms[b]->semiring != fsms[0]->semiring)
            return fail(MM_ERR_INVALID, "mm_batch_create: FSMs of one batch must share the semiring (FSM{K})");
    }
    mm_batch_s *h = new mm_batch_s();
    h->B = B;
    h->semiring = fsms[0]->semiring;
    h->fsms.assign(fsms, fsms + B);
    std::vector<UttDesc> utts(B);
    for (int64_t b = 0; b < B; ++b) {
        mm_fsm_t f = fsms[b];
        int rc = fsm_to_device(f);
        if (rc) {
            delete h;
            return rc;
        }
        UttDesc &u = utts[b];
        u.g[0] = f->gdev[0];
        u.g[1] = f->gdev[1];
        u.q[0] = f->qdev[0];
        u.q[1] = f->qdev[1];
        h->fast_ok = h->fast_ok && f->fast_ok;
        h->max_quads = std::max(h->max_quads, std::max(f->qdev[0].nq, f->qdev[1].nq));
        u.init = f->d_init;
        u.s2p = f->d_s2p;
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
        h->max_items = std::max(h->max_items, std::max(f->gdev[0].n_items, f->gdev[1].n_items));
    }
    if (hipGetDevice(&h->device) != hipSuccess || hipMalloc(&h->d_utts, sizeof(UttDesc) * B) != hipSuccess ||
        hipMemcpy(h->d_utts, utts.data(), sizeof(UttDesc) * B, hipMemcpyHostToDevice) != hipSuccess) {
        if (h->d_utts) (void)hipFree(h->d_utts);
        delete h;
        return fail(MM_ERR_HIP, "mm_batch_create: device allocation failed");
    }
    *out = h;
    return MM_OK;
}

int mm_batch_destroy(mm_batch_t h) {
    if (!h) return MM_OK;
    if (h->d_utts) (void)hipFree(h->d_utts);
    if (h->ws) (void)hipFree(h->ws);
    delete h;
    return MM_OK;
}

int64_t mm_batch_total_states(mm_batch_t h) { return h ? h->total_states : -1; }

static size_t ws_alpha_bytes(mm_batch_t h, int64_t N) { return align_up(size_t(h->total_s1p) * size_t(N + 1) * 4, 256); }
static size_t ws_c_bytes(mm_batch_t h, int64_t N) { return align_up(size_t(h->B) * size_t(N + 2) * 8, 256); }

size_t mm_batch_workspace_bytes(mm_batch_t h, int64_t N) {
    if (!h || N < 0) return 0;
    return ws_alpha_bytes(h, N) + ws_c_bytes(h, N);
}

static int ensure_ws(mm_batch_t h, size_t bytes) {
    if (h->ws_bytes >= bytes) return MM_OK;
    if (h->ws) {
        HIP_TRY(hipFree(h->ws));  // synchronises: only on growth
        h->ws = nullptr;
        h->ws_bytes = 0;
    }
    HIP_TRY(hipMalloc(&h->ws, bytes));
    h->ws_bytes = bytes;
    return MM_OK;
}

static int check_run(mm_batch_t h, const char *who, const float *V, int64_t N, int want_semiring) {
    if (!h) return fail(MM_ERR_INVALID, std::string(who) + ": NULL batch");
    if (!V) return fail(MM_ERR_INVALID, std::string(who) + ": V is NULL");
    if (N < 1 || N > (int64_t(1) << 30)) return fail(MM_ERR_DIM, std::string(who) + ": need N >= 1");
    if (want_semiring >= 0 && h->semiring != want_semiring)
        return fail(MM_ERR_INVALID, std::string(who) + ": batch was built for another semiring");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != h->device) return fail(MM_ERR_INVALID, std::string(who) + ": batch lives on another device");
    return MM_OK;
}

int mm_pdfposteriors_f32(mm_batch_t h, const float *V, int64_t vsb, int64_t vsn, const int32_t *lens, int64_t N,
                         float *gamma, int64_t gsb, int64_t gsn, int64_t gsp, float *ttl, void *stream) {
    int rc = check_run(h, "mm_pdfposteriors_f32", V, N, MM_LOG);
    if (rc) return rc;
    if (!gamma || !ttl) return fail(MM_ERR_INVALID, "mm_pdfposteriors_f32: gamma/ttl is NULL");
    rc = ensure_ws(h, ws_alpha_bytes(h, N) + ws_c_bytes(h, N));
    if (rc) return rc;
    RunParams p{};
    p.utts = h->d_utts;
    p.V = V;
    p.vsb = vsb;
    p.vsn = vsn;
    p.lens = lens;
    p.N = int(N);
    p.B = int(h->B);
    p.ws_alpha = static_cast<float *>(h->ws);
    p.ws_c = reinterpret_cast<double *>(static_cast<char *>(h->ws) + ws_alpha_bytes(h, N));
    p.gamma = gamma;
    p.gsb = gsb;
    p.gsn = gsn;
    p.gsp = gsp;
    p.ttl = ttl;
    if (quad_kernel_usable(h)) {
        rc = launch_quad(h, p, stream);
        if (rc != MM_ERR_UNSUPPORTED) return rc;
    }
    return launch_log<MODE_FB>(h, p, stream);
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
    if (h->semiring == MM_TROPICAL) {
        if (mode != MODE_ALPHA) return fail(MM_ERR_UNSUPPORTED, "tropical beta-recursion export is not implemented");
        return launch(mm_tropical_kernel, h, p, false, tropical_waves(h), stream);
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

int mm_viterbi_f32(mm_batch_t h, const float *V, int64_t vsb, int64_t vsn, const int32_t *lens, int64_t N,
                   int32_t *path, int64_t path_stride_b, float *score, int32_t *bp, int64_t bp_stride_n,
                   void *stream) {
    int rc = check_run(h, "mm_viterbi_f32", V, N, MM_TROPICAL);
    if (rc) return rc;
    if (!path || !score) return fail(MM_ERR_INVALID, "mm_viterbi_f32: path/score is NULL");
    if (path_stride_b < N) return fail(MM_ERR_DIM, "mm_viterbi_f32: path_stride_b < N");
    RunParams p{};
    if (!bp) {
        rc = ensure_ws(h, align_up(size_t(h->total_states) * size_t(N + 1) * 4, 256));
        if (rc) return rc;
        bp = static_cast<int32_t *>(h->ws);
        bp_stride_n = h->total_states;
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
    rc = launch(mm_tropical_kernel, h, p, false, tropical_waves(h), stream);
    if (rc) return rc;
    const int bt = 64;
    hipLaunchKernelGGL(mm_backtrace_kernel, dim3(unsigned((h->B + bt - 1) / bt)), dim3(bt), 0,
                       static_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

}  // extern "C"
