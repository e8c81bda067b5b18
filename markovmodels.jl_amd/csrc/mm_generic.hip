// mm_generic.hip -- the GENERIC pdfposteriors path: any semiring (Log, Tropical, Prob), float32 or float64, any sparse
// state map C_hat, any (P+1) x (N+1) matrices V_hat -- the reference's pdfposteriors(fsm, V_hats, C_hats) as it is
// declared (src/inference.jl:145-161: generic in K; FSM{LogSemiring{Float64}} is what its tests build, test/test_fsms.jl:3-7;
// its linear algebra is tested for 3 semirings x 2 float types, test/test_linalg.jl:88-108).
//
// Correctness first: one workgroup per utterance, a thread per CSR row and frame, alpha and beta materialised in the
// workspace like the reference does (:152-153), every reduction in a fixed order.  The fast kernels (float32, Log /
// Tropical, one-hot C_hat, V_hat of expand()'s form) are elsewhere; this path is what makes the entry a drop-in for the
// rest of the reference's argument space.  Z = 0 gives gamma = 0 and ttl = zero(K) (the reference: 0/0 = NaN, :158).
#define MM_SECONDARY_TU
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "mm_internal.h"

namespace mm {

struct mm_statemap_view {  // device CSR of C_hat (S1 x P1) and of its transpose, in T
    int64_t S1;
    int32_t P1;
    const int *cptr, *ccol;    // C_hat rows: state -> (pdf, weight)
    const void *cval;
    const int *tptr, *tcol;    // C_hat' rows: pdf -> (state, weight)
    const void *tval;
    // ProbSemiring, float32, a general map: C_hat as a DENSE matrix [S1 rounded up to 32][dense_ld] (zero padded, dense_ld a multiple of 16)
    // -- the B operand of the emission GEMM on the matrix cores (mm_prob_emission_mfma_kernel); NULL otherwise
    const float *dense;
    int dense_ld, pad_;
};

template <typename T>
struct GenFsmDev {  // one FSM on the device in T
    int S1, P1;
    const int *ptr[2], *col[2];
    const T *val[2];
    const T *init;
    mm_statemap_view own;  // the FSM's one-hot state map
};

template <typename T>
struct GenUtt {
    GenFsmDev<T> f;
    mm_statemap_view c;
    long long ws_off;  // offset of the utterance's alpha / beta arrays in the workspace (elements)
    // (ProbSemiring, float32, a general map) the state-level emissions C_hat * V_hat of every frame, computed ahead by the
    // emission GEMM: [N1][E_ld] (E_ld = S1 rounded up to 32); NULL: the recursion gathers them itself
    const T *E;
    long long E_ld;
};

template <typename T, int SR>
struct Sem {
    static __device__ __forceinline__ T zero() { return SR == MM_PROB ? T(0) : -std::numeric_limits<T>::infinity(); }
    static __device__ __forceinline__ T one() { return SR == MM_PROB ? T(1) : T(0); }
    static __device__ __forceinline__ T mul(T a, T b) {
        if (SR == MM_PROB) return a * b;
        if (a == zero() || b == zero()) return zero();  // (zero annihilates: never -inf + inf)
        return a + b;
    }
    static __device__ __forceinline__ T add(T a, T b) {
        if (SR == MM_PROB) return a + b;
        if (SR == MM_TROPICAL) return a > b ? a : b;
        const T m = a > b ? a : b;
        if (m == zero()) return zero();
        const T d = a > b ? b - a : a - b;  // -|a - b|
        return m + T(log1p(exp((double)d)));
    }
    static __device__ __forceinline__ T div(T a, T b) { return SR == MM_PROB ? a / b : a - b; }
    static __device__ __forceinline__ T out(T a) { return SR == MM_PROB ? a : T(exp((double)a)); }  // (:160 exp for the log-like semirings)
};
// (float32 Log: log1p / exp in double keep the path within rounding of the float64 oracle; speed is not this path's business)

template <typename T, int SR>
__global__ void __launch_bounds__(256) mm_generic_kernel(const GenUtt<T> *utts, const T *V, long long vsb, long long vsn, int N1, T *ws,
                                                         T *gamma, long long gsb, long long gsn, long long gsp, T *ttl) {
    using K = Sem<T, SR>;
    const GenUtt<T> &u = utts[blockIdx.x];
    const int S1 = u.f.S1, P1 = u.c.P1, P = P1 - 1, N = N1 - 1, tid = threadIdx.x, NT = blockDim.x;
    const T *Vb = V + (long long)blockIdx.x * vsb;
    T *A = ws + u.ws_off, *Bv = A + (long long)N1 * S1;
    extern __shared__ char smem[];
    T *zp = reinterpret_cast<T *>(smem);  // [P1] per-pdf sums of a frame, then [1] the frame's sum
    // state-level emission (C_hat * V_hat)[s, n]   (src/inference.jl:150)
    auto em = [&](int s, int n) {
        if (u.E) return u.E[(long long)n * u.E_ld + s];
        const T *cv = static_cast<const T *>(u.c.cval);
        T acc = K::zero();
        for (int k = u.c.cptr[s]; k < u.c.cptr[s + 1]; ++k) acc = K::add(acc, K::mul(cv[k], Vb[(long long)n * vsn + u.c.ccol[k]]));
        return acc;
    };
    // ---- alpha-recursion (src/inference.jl:62-74)
    for (int s = tid; s < S1; s += NT) A[s] = K::mul(u.f.init[s], em(s, 0));
    __syncthreads();
    for (int n = 1; n < N1; ++n) {
        const T *prev = A + (long long)(n - 1) * S1;
        T *cur = A + (long long)n * S1;
        for (int j = tid; j < S1; j += NT) {
            T acc = K::zero();
            for (int k = u.f.ptr[0][j]; k < u.f.ptr[0][j + 1]; ++k) acc = K::add(acc, K::mul(prev[u.f.col[0][k]], u.f.val[0][k]));
            cur[j] = K::mul(acc, em(j, n));
        }
        __syncthreads();
    }
    // ---- beta-recursion (:99-110): B[:, N+1] = one, B[:, n] = T_hat (B[:, n+1] (*) lhs[:, n+1])
    for (int s = tid; s < S1; s += NT) Bv[(long long)(N1 - 1) * S1 + s] = K::one();
    __syncthreads();
    for (int n = N1 - 2; n >= 0; --n) {
        const T *nxt = Bv + (long long)(n + 1) * S1;
        T *cur = Bv + (long long)n * S1;
        for (int i = tid; i < S1; i += NT) {
            T acc = K::zero();
            for (int k = u.f.ptr[1][i]; k < u.f.ptr[1][i + 1]; ++k) {
                const int j = u.f.col[1][k];
                acc = K::add(acc, K::mul(u.f.val[1][k], K::mul(nxt[j], em(j, n + 1))));
            }
            cur[i] = acc;
        }
        __syncthreads();
    }
    // ---- A .* B, C_hat' * AB, per-frame sum, divide, minimum, exp   (:154-160)
    T tmin = std::numeric_limits<T>::infinity();
    const T *tv = static_cast<const T *>(u.c.tval);
    for (int n = 0; n < N1; ++n) {
        for (int q = tid; q < P1; q += NT) {
            T acc = K::zero();
            for (int k = u.c.tptr[q]; k < u.c.tptr[q + 1]; ++k) {
                const int s = u.c.tcol[k];
                acc = K::add(acc, K::mul(tv[k], K::mul(A[(long long)n * S1 + s], Bv[(long long)n * S1 + s])));
            }
            zp[q] = acc;
        }
        __syncthreads();
        if (tid == 0) {
            T s = K::zero();
            for (int q = 0; q < P1; ++q) s = K::add(s, zp[q]);
            zp[P1] = s;
        }
        __syncthreads();
        const T sum = zp[P1];
        tmin = sum < tmin ? sum : tmin;
        if (n < N)
            for (int q = tid; q < P; q += NT)
                gamma[(long long)blockIdx.x * gsb + (long long)n * gsn + (long long)q * gsp] = sum == K::zero() ? T(0) : K::out(K::div(zp[q], sum));
        __syncthreads();
    }
    if (tid == 0) ttl[blockIdx.x] = tmin;
}

// The one place of the path where the matrix cores apply (BASELINE north star: "MFMA only for the dense emission GEMM in the
// probability semiring"): lhs = C_hat * V_hat (src/inference.jl:150) is a plain GEMM when K is the ProbSemiring -- multiply and
// add of real numbers -- and C_hat is a general (mixture) map; in the Log / Tropical semirings the same product is a
// log-sum-exp / a maximum per entry and stays in the recursion kernel.  Per utterance E[N1 x S1] = V_hat'[N1 x P1] * C_hat'[P1 x S1]
// in float32 on v_mfma_f32_32x32x2f32: a wave owns a 32 x 32 tile of E (32 frames x 32 states), K runs over the pdfs two at a
// time; operands straight from global memory (both matrices are small and cache resident: this is the correctness-first
// path's GEMM, not a tuned one).  Lane l supplies A[row l % 32][a pdf of half l / 32] and B[the same pdf][column l % 32];
// accumulator v of lane l is D[8 (v / 4) + 4 (l / 32) + v % 4][l % 32].
typedef float mm_acc16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mm_prob_emission_mfma_kernel(const GenUtt<float> *utts, const float *V, long long vsb, long long vsn,
                                                                    int N1) {
    const GenUtt<float> &u = utts[blockIdx.z];
    if (!u.E || !u.c.dense) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, kk = lane >> 5;
    const int n0 = (int)blockIdx.x * 32, s0 = ((int)blockIdx.y * 4 + wave) * 32;
    if (s0 >= (int)u.E_ld) return;
    const int P1 = u.c.P1, ld = u.c.dense_ld;
    const int n = n0 + r < N1 ? n0 + r : N1 - 1;
    const float *arow = V + (long long)blockIdx.z * vsb + (long long)n * vsn;  // V_hat[:, n]
    const float *brow = u.c.dense + (long long)(s0 + r) * ld;                  // C_hat[s, :]
    mm_acc16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    // 16 pdfs per trip.  Which pdf a (lane half, instruction) pair takes is free as long as A and B agree: half kk takes the 8
    // CONSECUTIVE pdfs k0 + 8 kk .. + 7, so a lane reads its 8 values of either operand with two 16-byte loads from its own row
    // (one load, one wait, one instruction at a time, a pdf per load, the kernel ran at 15 TFLOP/s; with the 16 scalar loads of
    // a trip issued ahead of its 8 instructions at 17: every load instruction touched 64 cache lines for 4 bytes of each).
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // (a row of V_hat starts at any multiple of 4 bytes)
    for (int k0 = 0; k0 < ld; k0 += 16) {
        const int q0 = k0 + 8 * kk;
        float a[8], b[8];
        const f4u b0 = *reinterpret_cast<const f4u *>(brow + q0), b1 = *reinterpret_cast<const f4u *>(brow + q0 + 4);
        if (q0 + 8 <= P1) {
            const f4u a0 = *reinterpret_cast<const f4u *>(arow + q0), a1 = *reinterpret_cast<const f4u *>(arow + q0 + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = a0[i], a[4 + i] = a1[i];
        } else {  // (the last pdfs of a row: nothing is read beyond them)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = q0 + i < P1 ? arow[q0 + i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = b0[i], b[4 + i] = b1[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc, 0, 0, 0);
    }
    float *E = const_cast<float *>(u.E);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int nr = n0 + 8 * (v / 4) + 4 * kk + (v % 4);
        if (nr < N1) E[(long long)nr * u.E_ld + s0 + r] = acc[v];
    }
}

// ---- host side
namespace {

struct DevFsm {  // owns the device copy of one FSM in one precision
    void *blob = nullptr;
    int S1 = 0, P1 = 0;
    size_t off_ptr[2], off_col[2], off_val[2], off_init, off_cptr, off_ccol, off_cval, off_tptr, off_tcol, off_tval;
    size_t off_dense = 0;
    int dense_ld = 0;  // > 0: the blob holds C_hat as a dense float matrix (ProbSemiring maps in float32)
};

template <typename T>
void push(std::vector<char> &h, size_t &off, const std::vector<T> &v) {
    off = (h.size() + 255) / 256 * 256;
    h.resize(off + v.size() * sizeof(T));
    if (!v.empty()) memcpy(h.data() + off, v.data(), v.size() * sizeof(T));
}

// C_hat as CSR + its transpose, values in T
template <typename T>
void statemap_arrays(int64_t S1, int32_t P1, const std::vector<int64_t> &ptr, const std::vector<int32_t> &col, const std::vector<double> &val,
                     std::vector<int> &cptr, std::vector<int> &ccol, std::vector<T> &cval, std::vector<int> &tptr, std::vector<int> &tcol,
                     std::vector<T> &tval) {
    cptr.assign(ptr.begin(), ptr.end());
    ccol.assign(col.begin(), col.end());
    cval.resize(val.size());
    for (size_t k = 0; k < val.size(); ++k) cval[k] = T(val[k]);
    tptr.assign(size_t(P1) + 1, 0);
    for (int32_t c : col) tptr[size_t(c) + 1]++;
    for (int32_t q = 0; q < P1; ++q) tptr[size_t(q) + 1] += tptr[size_t(q)];
    std::vector<int> cur(tptr.begin(), tptr.end() - 1);
    tcol.resize(col.size());
    tval.resize(col.size());
    for (int64_t s = 0; s < S1; ++s)
        for (int64_t k = ptr[size_t(s)]; k < ptr[size_t(s) + 1]; ++k) {
            const int d = cur[size_t(col[size_t(k)])]++;
            tcol[size_t(d)] = int(s);
            tval[size_t(d)] = T(val[size_t(k)]);
        }
}

template <typename T>
int fsm_to_device(FsmGenView *g, DevFsm **out) {
    const int slot = sizeof(T) == 4 ? 0 : 1;
    if (g->dev[slot]) {
        *out = static_cast<DevFsm *>(g->dev[slot]);
        return MM_OK;
    }
    DevFsm *d = new DevFsm();
    d->S1 = int(g->S1);
    d->P1 = int(g->P1);
    std::vector<char> h;
    for (int dir = 0; dir < 2; ++dir) {
        const int64_t nnz = g->ptr[dir][g->S1];
        std::vector<int> ptr(g->ptr[dir], g->ptr[dir] + g->S1 + 1), col(g->col[dir], g->col[dir] + nnz);
        std::vector<T> val(static_cast<size_t>(nnz));
        for (int64_t k = 0; k < nnz; ++k) val[size_t(k)] = T(g->val[dir][k]);
        push(h, d->off_ptr[dir], ptr);
        push(h, d->off_col[dir], col);
        push(h, d->off_val[dir], val);
    }
    std::vector<T> init(static_cast<size_t>(g->S1));
    for (int64_t s = 0; s < g->S1; ++s) init[size_t(s)] = T(g->init[s]);
    push(h, d->off_init, init);
    // the FSM's own one-hot state map: row s = {(pdf(s), one)}
    std::vector<int64_t> optr(size_t(g->S1) + 1);
    std::vector<int32_t> ocol(g->s2p, g->s2p + g->S1);
    std::vector<double> oval(size_t(g->S1), g->semiring == MM_PROB ? 1.0 : 0.0);
    for (int64_t s = 0; s <= g->S1; ++s) optr[size_t(s)] = s;
    std::vector<int> cptr, ccol, tptr, tcol;
    std::vector<T> cval, tval;
    statemap_arrays<T>(g->S1, g->P1, optr, ocol, oval, cptr, ccol, cval, tptr, tcol, tval);
    push(h, d->off_cptr, cptr);
    push(h, d->off_ccol, ccol);
    push(h, d->off_cval, cval);
    push(h, d->off_tptr, tptr);
    push(h, d->off_tcol, tcol);
    push(h, d->off_tval, tval);
    if (hipMalloc(&d->blob, h.size()) != hipSuccess || hipMemcpy(d->blob, h.data(), h.size(), hipMemcpyHostToDevice) != hipSuccess) {
        if (d->blob) (void)hipFree(d->blob);
        delete d;
        return mm_fail(MM_ERR_HIP, "mm_pdfposteriors_ex: device allocation failed");
    }
    g->dev[slot] = d;
    *out = d;
    return MM_OK;
}

mm_statemap_view view_of(const DevFsm *d, int val_bytes) {
    const char *b = static_cast<const char *>(d->blob);
    (void)val_bytes;
    mm_statemap_view v;
    v.S1 = d->S1;
    v.P1 = d->P1;
    v.cptr = reinterpret_cast<const int *>(b + d->off_cptr);
    v.ccol = reinterpret_cast<const int *>(b + d->off_ccol);
    v.cval = b + d->off_cval;
    v.tptr = reinterpret_cast<const int *>(b + d->off_tptr);
    v.tcol = reinterpret_cast<const int *>(b + d->off_tcol);
    v.tval = b + d->off_tval;
    v.dense = d->dense_ld > 0 ? reinterpret_cast<const float *>(b + d->off_dense) : nullptr;
    v.dense_ld = d->dense_ld;
    v.pad_ = 0;
    return v;
}

}  // namespace

void mm_generic_free(void *dev) {
    DevFsm *d = static_cast<DevFsm *>(dev);
    if (!d) return;
    if (d->blob) (void)hipFree(d->blob);
    delete d;
}

}  // namespace mm

using namespace mm;

// a general sparse state map C_hat (S1 x P1), handed over as CSR
struct mm_statemap_s {
    int semiring;
    int64_t S1;
    int32_t P1;
    std::vector<int64_t> ptr;
    std::vector<int32_t> col;
    std::vector<double> val;
    DevFsm *dev[2] = {nullptr, nullptr};  // device copies (float32, float64): only the C_hat parts of DevFsm are used
};

template <typename T>
static int statemap_to_device(mm_statemap_s *m, DevFsm **out) {
    const int slot = sizeof(T) == 4 ? 0 : 1;
    if (m->dev[slot]) {
        *out = m->dev[slot];
        return MM_OK;
    }
    DevFsm *d = new DevFsm();
    d->S1 = int(m->S1);
    d->P1 = int(m->P1);
    std::vector<char> h;
    std::vector<int> cptr, ccol, tptr, tcol;
    std::vector<T> cval, tval;
    statemap_arrays<T>(m->S1, m->P1, m->ptr, m->col, m->val, cptr, ccol, cval, tptr, tcol, tval);
    push(h, d->off_cptr, cptr);
    push(h, d->off_ccol, ccol);
    push(h, d->off_cval, cval);
    push(h, d->off_tptr, tptr);
    push(h, d->off_tcol, tcol);
    push(h, d->off_tval, tval);
    if (m->semiring == MM_PROB && sizeof(T) == 4) {  // the B operand of the emission GEMM (at most 64 MB: else the recursion gathers)
        const size_t S1r = (size_t(m->S1) + 31) / 32 * 32, ld = (size_t(m->P1) + 15) / 16 * 16;
        if (S1r * ld * 4 <= (size_t(64) << 20)) {
            std::vector<float> dense(S1r * ld, 0.f);
            for (int64_t s = 0; s < m->S1; ++s)
                for (int64_t k = m->ptr[size_t(s)]; k < m->ptr[size_t(s) + 1]; ++k) dense[size_t(s) * ld + size_t(m->col[size_t(k)])] += float(m->val[size_t(k)]);
            push(h, d->off_dense, dense);
            d->dense_ld = int(ld);
        }
    }
    if (hipMalloc(&d->blob, h.size() ? h.size() : 256) != hipSuccess ||
        hipMemcpy(d->blob, h.data(), h.size(), hipMemcpyHostToDevice) != hipSuccess) {
        if (d->blob) (void)hipFree(d->blob);
        delete d;
        return mm_fail(MM_ERR_HIP, "mm_pdfposteriors_ex: device allocation failed");
    }
    m->dev[slot] = d;
    *out = d;
    return MM_OK;
}

// workspace elements (of T) of one call: alpha and beta of every utterance, [N1][S1] each
// (ProbSemiring: + the state-level emissions of every frame, [N1][S1 rounded up to 32], which the emission GEMM writes)
static size_t generic_ws_elems(int64_t B, const mm_fsm_t *fsms, int64_t N1, int semiring) {
    size_t n = 0;
    for (int64_t b = 0; b < B; ++b) {
        const size_t S1 = size_t(mm_fsm_gen_view(fsms[b])->S1);
        n += size_t(2) * size_t(N1) * S1 + (semiring == MM_PROB ? size_t(N1) * ((S1 + 31) / 32 * 32) : 0);
    }
    return n;
}
// (MM_GENERIC_NO_MFMA, read once: the recursion kernel gathers the emissions itself -- what the emission GEMM is measured against)
static bool generic_no_mfma() {
    static const bool off = getenv("MM_GENERIC_NO_MFMA") != nullptr;
    return off;
}
// grow a device buffer of the batch (never while the stream is capturing: the old pointer may be baked into a graph)
static int grow(void **buf, size_t *have, size_t want, hipStream_t stream) {
    if (*have >= want) return MM_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return mm_fail(MM_ERR_INVALID, "mm_pdfposteriors_ex: the workspace would have to grow during stream capture: call mm_batch_reserve_ex first");
    if (*buf) {
        HIP_TRY(hipFree(*buf));  // synchronises: only on growth
        *buf = nullptr;
        *have = 0;
    }
    HIP_TRY(hipMalloc(buf, want));
    *have = want;
    return MM_OK;
}

template <typename T, int SR>
static int run_generic(mm_batch_t batch, int64_t B, const mm_fsm_t *fsms, const mm_statemap_t *maps, int32_t P1, const T *V, int64_t vsb,
                       int64_t vsn, int64_t N1, T *gamma, int64_t gsb, int64_t gsn, int64_t gsp, T *ttl, hipStream_t stream) {
    std::vector<GenUtt<T>> utts(static_cast<size_t>(B));
    std::vector<long long> e_off(static_cast<size_t>(B), -1);
    long long ws_elems = 0, max_S1r = 0;
    int maxP1 = 0, n_mfma = 0;
    for (int64_t b = 0; b < B; ++b) {
        FsmGenView *g = mm_fsm_gen_view(fsms[b]);
        DevFsm *d = nullptr;
        int rc = fsm_to_device<T>(g, &d);
        if (rc) return rc;
        const char *base = static_cast<const char *>(d->blob);
        GenUtt<T> &u = utts[size_t(b)];
        memset(&u, 0, sizeof(u));  // (the descriptors are compared bytewise with the last call's: no stray padding)
        u.f.S1 = d->S1;
        u.f.P1 = d->P1;
        for (int dir = 0; dir < 2; ++dir) {
            u.f.ptr[dir] = reinterpret_cast<const int *>(base + d->off_ptr[dir]);
            u.f.col[dir] = reinterpret_cast<const int *>(base + d->off_col[dir]);
            u.f.val[dir] = reinterpret_cast<const T *>(base + d->off_val[dir]);
        }
        u.f.init = reinterpret_cast<const T *>(base + d->off_init);
        u.f.own = view_of(d, sizeof(T));
        u.c = u.f.own;
        if (maps && maps[b]) {
            mm_statemap_s *m = maps[b];
            if (m->S1 != g->S1) return mm_fail(MM_ERR_DIM, "mm_pdfposteriors_ex: a state map's rows do not match its FSM's states");
            DevFsm *md = nullptr;
            rc = statemap_to_device<T>(m, &md);
            if (rc) return rc;
            u.c = view_of(md, sizeof(T));
        }
        // the kernel reads P1 rows of V_hat and writes P1 - 1 rows of gamma per utterance, P1 = the columns of the map in force
        // (src/inference.jl:146-150: vcat(V_hats...) must conform to blockdiag(C_hats...)' -- a DimensionMismatch there)
        if (u.c.P1 != P1)
            return mm_fail(MM_ERR_DIM, "mm_pdfposteriors_ex: V_hat has " + std::to_string(P1) + " rows but the state map of utterance " +
                                           std::to_string(b) + " has " + std::to_string(u.c.P1) + " pdfs (P + 1: was expand() applied?)");
        maxP1 = std::max(maxP1, int(u.c.P1));
        u.ws_off = ws_elems;
        ws_elems += 2ll * N1 * d->S1;
        if (SR == MM_PROB && sizeof(T) == 4 && u.c.dense && !generic_no_mfma()) {  // (its address is filled in below: the workspace may move)
            u.E_ld = (long long)(d->S1 + 31) / 32 * 32;
            e_off[size_t(b)] = ws_elems;
            ws_elems += N1 * u.E_ld;
            n_mfma++;
            max_S1r = std::max<long long>(max_S1r, u.E_ld);
        }
    }
    // workspace and descriptors live with the batch: no allocation, no synchronisation in the steady state
    GenScratch *sc = mm_batch_gen_scratch(batch);
    int rc = grow(&sc->ws, &sc->ws_bytes, sizeof(T) * size_t(ws_elems), stream);
    if (rc) return rc;
    for (int64_t b = 0; b < B; ++b)
        if (e_off[size_t(b)] >= 0) utts[size_t(b)].E = static_cast<const T *>(sc->ws) + e_off[size_t(b)];
    const size_t ub = sizeof(GenUtt<T>) * size_t(B);
    if (sc->utts_bytes < ub) sc->host.clear();
    rc = grow(&sc->d_utts, &sc->utts_bytes, ub, stream);
    if (rc) return rc;
    if (sc->host.size() != ub || memcmp(sc->host.data(), utts.data(), ub) != 0) {
        // (stream-ordered behind the last call's kernel; the source is the batch's own image, alive as long as the batch)
        sc->host.assign(reinterpret_cast<const char *>(utts.data()), reinterpret_cast<const char *>(utts.data()) + ub);
        HIP_TRY(hipMemcpyAsync(sc->d_utts, sc->host.data(), ub, hipMemcpyHostToDevice, stream));
    }
    sc->last_kernels = std::string("mm_generic_kernel<") + (sizeof(T) == 4 ? "float" : "double") + "," + (SR == MM_LOG ? "log" : SR == MM_TROPICAL ? "tropical" : "prob") + ">";
    if constexpr (SR == MM_PROB && sizeof(T) == 4) {
        if (n_mfma > 0) {  // C_hat * V_hat of every frame on the matrix cores, for the utterances whose map is a general one
            hipLaunchKernelGGL(mm_prob_emission_mfma_kernel, dim3(unsigned((N1 + 31) / 32), unsigned((max_S1r / 32 + 3) / 4), unsigned(B)), dim3(256), 0,
                               stream, static_cast<const GenUtt<float> *>(sc->d_utts), reinterpret_cast<const float *>(V), (long long)vsb, (long long)vsn,
                               int(N1));
            HIP_TRY(hipGetLastError());
            sc->last_kernels = "mm_prob_emission_mfma_kernel (v_mfma_f32_32x32x2f32: C_hat * V_hat of " + std::to_string(n_mfma) + " utterances), then " +
                               sc->last_kernels;
        }
    }
    hipLaunchKernelGGL((mm_generic_kernel<T, SR>), dim3(unsigned(B)), dim3(256), size_t(maxP1 + 1) * sizeof(T), stream,
                       static_cast<const GenUtt<T> *>(sc->d_utts), V, (long long)vsb, (long long)vsn, int(N1), static_cast<T *>(sc->ws), gamma,
                       (long long)gsb, (long long)gsn, (long long)gsp, ttl);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

extern "C" {

int mm_statemap_create(int semiring, int64_t S1, int32_t P1, int64_t nnz, int index_bytes, int index_base, int val_bytes, const void *rowptr,
                       const void *colidx, const void *val, mm_statemap_t *out) {
    if (!out) return mm_fail(MM_ERR_INVALID, "mm_statemap_create: out is NULL");
    *out = nullptr;
    if (semiring < MM_LOG || semiring > MM_PROB || (index_bytes != 4 && index_bytes != 8) || (val_bytes != 4 && val_bytes != 8) ||
        (index_base != 0 && index_base != 1) || S1 < 1 || P1 < 1 || nnz < 0 || !rowptr || (nnz && (!colidx || !val)))
        return mm_fail(MM_ERR_INVALID, "mm_statemap_create: bad argument");
    auto rd = [&](const void *p, int64_t i) {
        return index_bytes == 4 ? int64_t(static_cast<const int32_t *>(p)[i]) : static_cast<const int64_t *>(p)[i];
    };
    mm_statemap_s *m = new mm_statemap_s();
    m->semiring = semiring;
    m->S1 = S1;
    m->P1 = P1;
    m->ptr.resize(size_t(S1) + 1);
    for (int64_t i = 0; i <= S1; ++i) {
        m->ptr[size_t(i)] = rd(rowptr, i) - index_base;
        if (m->ptr[size_t(i)] < 0 || m->ptr[size_t(i)] > nnz || (i && m->ptr[size_t(i)] < m->ptr[size_t(i) - 1])) {
            delete m;
            return mm_fail(MM_ERR_DIM, "mm_statemap_create: rowptr is not monotone within [0, nnz]");
        }
    }
    m->col.resize(size_t(nnz));
    m->val.resize(size_t(nnz));
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t c = rd(colidx, k) - index_base;
        if (c < 0 || c >= P1) {
            delete m;
            return mm_fail(MM_ERR_DIM, "mm_statemap_create: pdf index out of range");
        }
        m->col[size_t(k)] = int32_t(c);
        m->val[size_t(k)] = val_bytes == 4 ? double(static_cast<const float *>(val)[k]) : static_cast<const double *>(val)[k];
    }
    *out = m;
    return MM_OK;
}

int mm_statemap_destroy(mm_statemap_t m) {
    if (!m) return MM_OK;
    for (DevFsm *d : m->dev) mm_generic_free(d);
    delete m;
    return MM_OK;
}

int mm_batch_reserve_ex(mm_batch_t batch, int val_bytes, int64_t N1) {
    int64_t B = 0;
    const mm_fsm_t *fsms = nullptr;
    int semiring = 0, device = -1, dev = -1;
    if (mm_batch_gen_view(batch, &B, &fsms, &semiring, &device)) return mm_fail(MM_ERR_INVALID, "mm_batch_reserve_ex: NULL batch");
    if ((val_bytes != 4 && val_bytes != 8) || N1 < 2) return mm_fail(MM_ERR_INVALID, "mm_batch_reserve_ex: bad argument");
    HIP_TRY(hipGetDevice(&dev));
    if (dev != device) return mm_fail(MM_ERR_INVALID, "mm_batch_reserve_ex: batch lives on another device");
    GenScratch *sc = mm_batch_gen_scratch(batch);
    int rc = grow(&sc->ws, &sc->ws_bytes, size_t(val_bytes) * generic_ws_elems(B, fsms, N1, semiring), nullptr);
    if (rc) return rc;
    if (sc->utts_bytes < sizeof(GenUtt<double>) * size_t(B)) sc->host.clear();
    return grow(&sc->d_utts, &sc->utts_bytes, sizeof(GenUtt<double>) * size_t(B), nullptr);
}

int mm_pdfposteriors_ex(mm_batch_t batch, const mm_statemap_t *maps, int val_bytes, int32_t P1, const void *Vhat, int64_t v_stride_b,
                        int64_t v_stride_n, int64_t N1, void *gamma, int64_t g_stride_b, int64_t g_stride_n, int64_t g_stride_p, void *ttl,
                        void *stream) {
    int64_t B = 0;
    const mm_fsm_t *fsms = nullptr;
    int semiring = 0, device = -1, dev = -1;
    if (mm_batch_gen_view(batch, &B, &fsms, &semiring, &device)) return mm_fail(MM_ERR_INVALID, "mm_pdfposteriors_ex: NULL batch");
    if (!Vhat || !gamma || !ttl || (val_bytes != 4 && val_bytes != 8)) return mm_fail(MM_ERR_INVALID, "mm_pdfposteriors_ex: bad argument");
    if (N1 < 2) return mm_fail(MM_ERR_DIM, "mm_pdfposteriors_ex: V_hat needs at least two columns (N + 1)");
    if (P1 < 2) return mm_fail(MM_ERR_DIM, "mm_pdfposteriors_ex: V_hat needs at least two rows (P + 1)");
    HIP_TRY(hipGetDevice(&dev));
    if (dev != device) return mm_fail(MM_ERR_INVALID, "mm_pdfposteriors_ex: batch lives on another device");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define MM_GEN_CASE(T, SR)                                                                                                              \
    return run_generic<T, SR>(batch, B, fsms, maps, P1, static_cast<const T *>(Vhat), v_stride_b, v_stride_n, N1, static_cast<T *>(gamma), \
                              g_stride_b, g_stride_n, g_stride_p, static_cast<T *>(ttl), st)
    if (val_bytes == 4) {
        if (semiring == MM_LOG) MM_GEN_CASE(float, MM_LOG);
        if (semiring == MM_TROPICAL) MM_GEN_CASE(float, MM_TROPICAL);
        MM_GEN_CASE(float, MM_PROB);
    }
    if (semiring == MM_LOG) MM_GEN_CASE(double, MM_LOG);
    if (semiring == MM_TROPICAL) MM_GEN_CASE(double, MM_TROPICAL);
    MM_GEN_CASE(double, MM_PROB);
#undef MM_GEN_CASE
}

}  // extern "C"
