/* CPU oracle for the MarkovModels.jl inference hot path -- plain C restatement.
 *
 * TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into
 * oracle/libmm_oracle.so and loaded only by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.  Never linked into the product library.
 *
 * Parity status: see the header of oracle/mm_oracle.py (pinned against the
 * reference's known-answer vectors only; the reference is Julia and cannot be
 * run or built in the image; oracle/_ref is therefore absent).
 *
 * The operation ORDER follows the reference's CPU path so that the timing is
 * a fair "port" baseline and float32 rounding matches it:
 *   - sparse * dense = Julia stdlib generic mul!(C, A::SparseMatrixCSC, B):
 *     fill C with zero(K), then for each column `col` of A, for each stored
 *     entry j:  C[rv[j]] (+)= nzv[j] (*) B[col]     (a scatter, one logaddexp
 *     per arc)                       -- called from src/inference.jl:70,107
 *   - alpha-recursion  src/inference.jl:62-74   (A materialised, S1 x N1)
 *   - beta-recursion   src/inference.jl:99-110  (B materialised, S1 x N1)
 *   - pdfposteriors    src/inference.jl:145-161 (C*V gather, A.*B, C'*AB
 *     reduce, per-frame sums, divide, min, exp)
 *   - expand           src/inference.jl:54-60
 * Indices are 0-based.  REAL is float or double (the file is compiled twice).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#error "compile with -DREAL=float -DSUF=f32 or -DREAL=double -DSUF=f64"
#endif
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

#define SR_LOG 0
#define SR_TROPICAL 1

static inline REAL r_exp(REAL x) { return sizeof(REAL) == 4 ? (REAL)expf((float)x) : (REAL)exp((double)x); }
static inline REAL r_log1p(REAL x) { return sizeof(REAL) == 4 ? (REAL)log1pf((float)x) : (REAL)log1p((double)x); }
static inline REAL r_abs(REAL x) { return x < 0 ? -x : x; }

/* LogSemiring (+): logaddexp, (-inf)(+)(-inf) = -inf  (test/test_semirings.jl:4-6) */
static inline REAL sr_add(int sr, REAL x, REAL y) {
    REAL m = x > y ? x : y;
    if (sr == SR_TROPICAL) return m;
    if (m == (REAL)-INFINITY) return m;
    return m + r_log1p(r_exp(-r_abs(x - y)));
}
/* (*) = + ; zero annihilates */
static inline REAL sr_mul(REAL x, REAL y) { return x + y; }

/* C = A * b for CSC A (ncols columns): the Julia generic mul! scatter loop. */
static void spmv_csc(int sr, int64_t nrows, int64_t ncols, const int64_t *colptr, const int64_t *rowval,
                     const REAL *nzval, const REAL *b, REAL *c) {
    for (int64_t i = 0; i < nrows; ++i) c[i] = (REAL)-INFINITY;
    for (int64_t col = 0; col < ncols; ++col) {
        REAL x = b[col];
        for (int64_t j = colptr[col]; j < colptr[col + 1]; ++j) {
            int64_t r = rowval[j];
            c[r] = sr_add(sr, c[r], sr_mul(nzval[j], x));
        }
    }
}

/* One utterance.  T_* : CSC of T_hat (column j = in-arcs of j);  Tt_* : CSC of
 * copy(T_hat') (column i = out-arcs of i).  Vhat: expanded emissions, column
 * major (P1 x N1, pdf fastest).  state2pdf: len S1, last = P1-1.
 * gamma: (P1-1) x (N1-1) column major (pdf fastest) probabilities.
 * A_out/B_out: optional S1 x N1 column-major copies of alpha / beta.       */
/* Per-thread scratch (CV, A, B, buf, AB, V_hat), kept across calls and grown on demand: a malloc/free pair of
 * three 12 MB arrays per utterance is an mmap/munmap + first-touch page faults per utterance, which serialises
 * the threads of the all-cores baseline timing in the kernel (measured: 8-9x per-thread collapse at 256 threads).
 * The reference allocates its arrays per call as well (src/inference.jl:64,101), but once per BATCH. */
static _Thread_local REAL *tl_ws = NULL;
static _Thread_local size_t tl_ws_n = 0;
static REAL *thread_ws(size_t n) {
    if (n > tl_ws_n) {
        free(tl_ws);
        tl_ws = (REAL *)malloc(sizeof(REAL) * n);
        tl_ws_n = tl_ws ? n : 0;
        if (tl_ws) memset(tl_ws, 0, sizeof(REAL) * n); /* first touch here, not inside a timed region */
    }
    return tl_ws;
}
static size_t ws_need(int64_t S1, int64_t P1, int64_t N1) { return (size_t)(3 * S1 * N1 + S1 + P1 + P1 * N1); }

int FN(mmo_pdfposteriors)(int sr, int64_t S1, int64_t P1, int64_t N1, const int64_t *T_colptr,
                          const int64_t *T_rowval, const REAL *T_nzval, const int64_t *Tt_colptr,
                          const int64_t *Tt_rowval, const REAL *Tt_nzval, const REAL *alpha_hat,
                          const int32_t *state2pdf, const REAL *Vhat, REAL *gamma, REAL *ttl, REAL *A_out,
                          REAL *B_out) {
    REAL *ws = thread_ws(ws_need(S1, P1, N1));
    if (!ws) return -1;
    REAL *CV = ws;                /* C_hat * V_hat   :150 */
    REAL *A = CV + S1 * N1;       /* state_A         :152 */
    REAL *Bm = A + S1 * N1;       /* state_B         :153 */
    REAL *buf = Bm + S1 * N1;
    REAL *AB = buf + S1;
    for (int64_t n = 0; n < N1; ++n)
        for (int64_t s = 0; s < S1; ++s) CV[n * S1 + s] = Vhat[n * P1 + state2pdf[s]];
    /* alpha-recursion :62-74 */
    for (int64_t s = 0; s < S1; ++s) A[s] = sr_mul(alpha_hat[s], CV[s]);
    for (int64_t n = 1; n < N1; ++n) {
        spmv_csc(sr, S1, S1, Tt_colptr, Tt_rowval, Tt_nzval, A + (n - 1) * S1, buf);
        for (int64_t s = 0; s < S1; ++s) A[n * S1 + s] = sr_mul(buf[s], CV[n * S1 + s]);
    }
    /* beta-recursion :99-110 */
    for (int64_t s = 0; s < S1; ++s) Bm[(N1 - 1) * S1 + s] = 0;
    for (int64_t n = N1 - 2; n >= 0; --n) {
        for (int64_t s = 0; s < S1; ++s) buf[s] = sr_mul(Bm[(n + 1) * S1 + s], CV[(n + 1) * S1 + s]);
        spmv_csc(sr, S1, S1, T_colptr, T_rowval, T_nzval, buf, Bm + n * S1);
    }
    /* combine :154-160 */
    REAL tmin = (REAL)INFINITY;
    for (int64_t n = 0; n < N1; ++n) {
        for (int64_t p = 0; p < P1; ++p) AB[p] = (REAL)-INFINITY;
        for (int64_t s = 0; s < S1; ++s) {
            REAL ab = sr_mul(A[n * S1 + s], Bm[n * S1 + s]);
            AB[state2pdf[s]] = sr_add(sr, AB[state2pdf[s]], ab);
        }
        REAL sum = (REAL)-INFINITY;
        for (int64_t p = 0; p < P1; ++p) sum = sr_add(sr, sum, AB[p]);
        if (sum < tmin) tmin = sum;
        if (n < N1 - 1)
            for (int64_t p = 0; p < P1 - 1; ++p) gamma[n * (P1 - 1) + p] = r_exp(AB[p] - sum);
    }
    *ttl = tmin;
    if (A_out) memcpy(A_out, A, sizeof(REAL) * S1 * N1);
    if (B_out) memcpy(B_out, Bm, sizeof(REAL) * S1 * N1);
    return 0;
}

/* expand (src/inference.jl:54-60) for one utterance: lhs is P x N column major
 * (pdf fastest, frame stride ld); out is (P+1) x (N+1). */
static void expand_one(const REAL *lhs, int64_t ld, int64_t P, int64_t N, int64_t len, REAL *out) {
    int64_t P1 = P + 1;
    for (int64_t n = 0; n <= N; ++n) {
        for (int64_t p = 0; p < P; ++p) out[n * P1 + p] = (n < len) ? lhs[n * ld + p] : (REAL)-INFINITY;
        out[n * P1 + P] = (n < len) ? (REAL)-INFINITY : 0;
    }
}

/* B utterances sharing one graph (the denominator case,
 * examples/test_cuda.jl:112-113): rawunion is block diagonal, so the batched
 * reference computes exactly these B independent problems.
 * lhs: [B][N][P] (pdf fastest).  gamma: [B][N][P].  nthreads <= 1: serial
 * (the reference is single threaded); > 1: OpenMP over utterances.        */
int FN(mmo_batch_shared)(int sr, int64_t B, int64_t S1, int64_t P, int64_t N, const int64_t *T_colptr,
                         const int64_t *T_rowval, const REAL *T_nzval, const int64_t *Tt_colptr,
                         const int64_t *Tt_rowval, const REAL *Tt_nzval, const REAL *alpha_hat,
                         const int32_t *state2pdf, const REAL *lhs, const int32_t *lens, REAL *gamma, REAL *ttl,
                         int nthreads) {
    int rc = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int64_t b = 0; b < B; ++b) {
        REAL *ws = thread_ws(ws_need(S1, P + 1, N + 1));
        if (!ws) {
            rc = -1;
            continue;
        }
        REAL *Vh = ws + 3 * S1 * (N + 1) + S1 + (P + 1); /* behind mmo_pdfposteriors' own arrays */
        expand_one(lhs + b * N * P, P, P, N, lens ? lens[b] : N, Vh);
        int r = FN(mmo_pdfposteriors)(sr, S1, P + 1, N + 1, T_colptr, T_rowval, T_nzval, Tt_colptr, Tt_rowval,
                                      Tt_nzval, alpha_hat, state2pdf, Vh, gamma + b * N * P, ttl + b, NULL, NULL);
        if (r) rc = r;
    }
    return rc;
}

/* Allocate and first-touch the scratch of `nthreads` OpenMP threads for problems of this size, outside any
 * timed region (bench.py cpu_baseline). */
int FN(mmo_warm)(int64_t S1, int64_t P, int64_t N, int nthreads) {
    int rc = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        if (!thread_ws(ws_need(S1, P + 1, N + 1))) rc = -1;
    }
    return rc;
}

/* Tropical forward recursion + back-pointers + back-trace for one utterance.
 * Forward = alpha-recursion with K = TropicalSemiring (src/inference.jl:62-74);
 * back-pointers are this project's specification (the reference has no
 * bestpath at this commit): bp[n][j] = lowest source i maximising
 * T_hat[i,j] + A[i,n-1] over stored arcs, -1 if that maximum is -inf.
 * T_* is CSC of T_hat (in-arcs of j, ascending source).  lhs: [N][P].
 * bp: [N+1][S1] (row 0 = -1); path: [N] (-1 beyond len).                   */
int FN(mmo_viterbi)(int64_t S1, int64_t P, int64_t N, int64_t len, const int64_t *T_colptr, const int64_t *T_rowval,
                    const REAL *T_nzval, const REAL *alpha_hat, const int32_t *state2pdf, const REAL *lhs,
                    int32_t *bp, int32_t *path, REAL *score) {
    int64_t P1 = P + 1, N1 = N + 1;
    REAL *Vh = (REAL *)malloc(sizeof(REAL) * P1 * N1);
    REAL *a = (REAL *)malloc(sizeof(REAL) * S1), *a2 = (REAL *)malloc(sizeof(REAL) * S1);
    expand_one(lhs, P, P, N, len, Vh);
    for (int64_t s = 0; s < S1; ++s) {
        a[s] = alpha_hat[s] + Vh[state2pdf[s]];
        bp[s] = -1;
    }
    for (int64_t n = 1; n < N1; ++n) {
        for (int64_t j = 0; j < S1; ++j) {
            REAL best = (REAL)-INFINITY;
            int32_t arg = -1;
            for (int64_t k = T_colptr[j]; k < T_colptr[j + 1]; ++k) {
                REAL v = T_nzval[k] + a[T_rowval[k]];
                if (v > best) { best = v; arg = (int32_t)T_rowval[k]; }
            }
            a2[j] = best + Vh[n * P1 + state2pdf[j]];
            bp[n * S1 + j] = arg;
        }
        REAL *t = a; a = a2; a2 = t;
        if (n == len) *score = a[S1 - 1];
    }
    for (int64_t n = 0; n < N; ++n) path[n] = -1;
    if (*score > (REAL)-INFINITY) {
        int32_t s = (int32_t)(S1 - 1);
        for (int64_t n = len; n >= 1; --n) {
            s = bp[n * S1 + s];
            path[n - 1] = s;
        }
    }
    free(Vh); free(a); free(a2);
    return 0;
}
