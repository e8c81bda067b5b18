/* markovmodels_amd.h -- C ABI of the MI355X-native forward-backward / Viterbi
 * engine that sits behind MarkovModels.jl's inference API.
 *
 * Every entry point names the reference interface (file:line under the
 * MarkovModels.jl tree) it replaces.  The reference seam is Julia multiple
 * dispatch on CuArray storage (src/fsm.jl:42-48, src/inference.jl:14-26,
 * src/linalg.jl:163,240,335); this library moves the seam one level up: one
 * call per `pdfposteriors` / `alpha-recursion` / `beta-recursion` instead of
 * four kernel launches per frame.
 *
 * Conventions
 *   - plain C, no C++/torch types; all functions return an int status
 *     (MM_OK = 0, negative = error) and record a message for mm_last_error().
 *   - "device pointer" = memory of the HIP device that was current when the
 *     handle was created; "host pointer" = ordinary memory.  Run calls are
 *     asynchronous on the given hipStream_t (passed as void*, NULL = default
 *     stream) and never synchronise the device.
 *   - the caller owns every in/out buffer; the library owns its handles and
 *     an internal workspace (the alpha store) that grows lazily and is freed
 *     with the batch.  ONE workspace per batch: run calls on the same batch
 *     must be issued on one stream (or otherwise ordered); calls on different
 *     batches are independent.  Growing the workspace frees the old one
 *     (hipFree synchronises): size it once with mm_batch_reserve() before
 *     capturing run calls in a hipGraph -- a run call that would have to grow
 *     the workspace while its stream is capturing fails with MM_ERR_INVALID
 *     instead of invalidating the pointers earlier captures baked in.
 *   - weights/likelihoods are natural-log values of the Log/Tropical
 *     semirings (zero(K) = -inf, one(K) = 0), bit-compatible with the
 *     reference's Array{K} storage.  The fast entries (mm_*_f32) take and return float32 like the reference's
 *     Float32 FSMs; inside, the linear-domain kernels carry float32 or -- for inputs beyond float32's exponent range --
 *     float64 values (mm_batch_set_exact_policy); the generic entry (mm_pdfposteriors_ex) and the linear algebra
 *     (mm_spmv / mm_spmm / mm_svdv) compute in the caller's float type.
 *   - the FSM is the reference's *extended* system (src/fsm.jl:19-28): S1 =
 *     S + 1 states, the last one being the phony final state (self loop of
 *     weight one); P1 = P + 1 pdfs, the last one being the phony pdf that
 *     only the final state emits (examples/prepare-lfmmi-graphs.jl:15-23).
 *   - state / pdf indices RETURNED by the library are 0-based.
 */
#ifndef MARKOVMODELS_AMD_H
#define MARKOVMODELS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_ABI_VERSION 4 /* 2: mm_pdfposteriors_ex takes the rows of V_hat (P1); mm_batch_reserve_ex.  3: mm_spmv / mm_spmm / mm_svdv, mm_batch_set_exact_policy.  4: mm_batch_set_mark_policy, mm_batch_set_gamma_mode */

enum mm_status {
    MM_OK = 0,
    MM_ERR_INVALID = -1,     /* bad handle / argument */
    MM_ERR_DIM = -2,         /* the reference's DimensionMismatch (src/linalg.jl:166-167,242-244) */
    MM_ERR_HIP = -3,         /* a HIP runtime call failed */
    MM_ERR_UNSUPPORTED = -4, /* valid input the engine cannot run (e.g. graph too large for LDS) */
    MM_ERR_NOMEM = -5
};

enum mm_semiring { MM_LOG = 0, MM_TROPICAL = 1, MM_PROB = 2 }; /* Semirings.jl LogSemiring / TropicalSemiring / ProbSemiring */
enum mm_layout { MM_CSC = 0, MM_CSR = 1 };        /* how T_hat is handed over */

typedef struct mm_fsm_s *mm_fsm_t;     /* one compiled FSM   ~ CompiledFSM   (src/inference.jl:3-12)  */
typedef struct mm_batch_s *mm_batch_t; /* a batch of them    ~ batch()/rawunion (src/inference.jl:28-36, src/fsmops.jl:28-36) */

int mm_abi_version(void);
/* Message of the last error on the calling thread ("" if none). */
const char *mm_last_error(void);

/* compile(fsm, C_hat) (src/inference.jl:11-12) + adapt to the device
 * (src/inference.jl:14-26): takes the reference FSM fields as they are stored
 * on the host and builds the device-resident packed forms of T_hat' (forward)
 * and T_hat (backward).
 *   S1, nnz      size of T_hat (S1 x S1) and its stored entries
 *   layout       MM_CSC: ptr = colptr (S1+1), idx = rowval  -- SparseMatrixCSC as in src/fsm.jl:14
 *                MM_CSR: ptr = rowptr (S1+1), idx = colval  -- CuSparseMatrixCSR as in src/fsm.jl:45
 *   index_bytes  4 (Cint, src/linalg.jl:80) or 8 (Int64);  index_base 0 or 1 (Julia)
 *   val_bytes    4 (float) or 8 (double) for val / init_val
 *   init_idx/val the n_init stored entries of alpha_hat (src/fsm.jl:10)
 *   state2pdf    S1 entries (index_base based), the column of the single
 *                stored entry of each row of C_hat; the last must be P1-1
 * All pointers are host pointers and are not retained. */
int mm_fsm_create(int semiring, int64_t S1, int64_t nnz, int layout, int index_bytes, int index_base,
                  int val_bytes, const void *ptr, const void *idx, const void *val, int64_t n_init,
                  const void *init_idx, const void *init_val, const int32_t *state2pdf, int32_t P1,
                  mm_fsm_t *out);
/* `compile.(fsms, C_hats)` for a mini-batch of NEW graphs in one call (examples/test_cuda.jl:74-78 builds a fresh batch of
 * numerator graphs every training step): the same arguments as mm_fsm_create, one entry per graph (arrays of n pointers /
 * sizes; semiring, layout and the index / value types are shared).  The graphs are compiled on `threads` host threads
 * (<= 0: up to 16); small log-semiring graphs (the wave kernel's: <= 1023 states, <= 4096 arcs) get their kernel forms at
 * once, and the forms of ALL graphs go to the device as ONE allocation and ONE copy (a handle keeps its share alive:
 * destroy the handles in any order).  out: n handles, each as mm_fsm_create would have made it -- the same results bit for
 * bit; on error none is created.  (Without a HIP device the forms stay on the host until mm_batch_create, as after
 * mm_fsm_create.) */
int mm_fsm_create_many(int64_t n, int semiring, int layout, int index_bytes, int index_base, int val_bytes, const int64_t *S1,
                       const int64_t *nnz, const void *const *ptr, const void *const *idx, const void *const *val,
                       const int64_t *n_init, const void *const *init_idx, const void *const *init_val,
                       const int32_t *const *state2pdf, const int32_t *P1, int threads, mm_fsm_t *out);
int mm_fsm_destroy(mm_fsm_t fsm);
/* nstates + sizes (src/fsm.jl:84); any out pointer may be NULL.
 * packed_slots[d] = arc slots of the packed form, d = 0 forward, 1 backward. */
int mm_fsm_info(mm_fsm_t fsm, int64_t *S1, int64_t *nnz, int32_t *P1, int64_t packed_slots[2],
                int64_t packed_items[2]);

/* batch(cfsm...) (src/inference.jl:28-36) / rawunion(fsms...) (src/fsmops.jl:28-36):
 * B independent FSMs in one block-diagonal system.  Handles may repeat; when
 * all B are the same handle the graph is stored once (denominator case,
 * examples/test_cuda.jl:112).  The FSM handles must outlive the batch. */
int mm_batch_create(const mm_fsm_t *fsms, int64_t B, mm_batch_t *out);
int mm_batch_destroy(mm_batch_t batch);
/* Sum over the batch of S1 (rows of the block-diagonal system). */
int64_t mm_batch_total_states(mm_batch_t batch);
/* Names of the kernels a run entry launches for this batch (the engine picks them from the graphs' sizes and
 * shapes): entry 0 = mm_pdfposteriors_f32, 1 = mm_viterbi_f32, 2 = what the last mm_pdfposteriors_ex call on the batch launched
 * (the recursion kernel; for ProbSemiring FSMs in float32 with general state maps, the emission GEMM C_hat * V_hat on the matrix
 * cores before it), 3 = mm_alpharecursion_f32 / mm_betarecursion_f32.  Informational (bench.py quotes it). */
int mm_batch_kernels(mm_batch_t batch, int entry, char *buf, size_t n);
/* Allocate the internal workspace for runs of up to N frames now (synchronises if it has to grow). */
int mm_batch_reserve(mm_batch_t batch, int64_t N);
/* Bytes of internal workspace a run with N frames needs (informational). */
size_t mm_batch_workspace_bytes(mm_batch_t batch, int64_t N);

/* pdfposteriors(fsm, V_hats, C_hats) (src/inference.jl:145-161), with
 * expand() (src/inference.jl:54-60) done inside:
 *   V      device, log-likelihoods of the P = P1-1 real pdfs; element (b, n, p)
 *          at V[b*v_stride_b + n*v_stride_n + p], n = 0..N-1
 *   lens   device int32[B] sequence lengths (<= N), or NULL for all N
 *   gamma  device, out: posterior PROBABILITIES exp(log gamma) like
 *          src/inference.jl:160; element (b, n, p) at
 *          gamma[b*g_stride_b + n*g_stride_n + p*g_stride_p]; frames n >= len_b
 *          are written as exact zeros.  (The reference returns a B x P x N
 *          column-major array: g_stride_b = 1, g_stride_p = B, g_stride_n = B*P.)
 *   ttl    device float[B], out: min over frames of the per-frame log
 *          normaliser (src/inference.jl:159) = log Z_b
 * An utterance with no accepting path (Z = 0) yields gamma = 0, ttl = -inf
 * (the reference yields NaN: src/inference.jl:158; guarded only in the dead
 * code at :198-200).
 * MM_PROB batches of Float32 FSMs (ProbSemiring{Float32}, non-negative weights): V holds LIKELIHOODS -- the semiring's values, like
 * the reference's Array{ProbSemiring} --, the library keeps a log-semiring twin of every such FSM (weights = their logarithms),
 * takes log V in one pass, runs the same fast kernels and returns ttl as the probability Z_b = exp(log Z_b) (0 if no path); gamma is
 * the same quotient in both semirings (:158-160).  Float64 ProbSemiring FSMs, general state maps, V_hat that expand() did not make:
 * mm_pdfposteriors_ex. */
/* Streams: the call is a chain of kernel launches on `stream` and nothing else -- no library-owned streams, no events, no
 * host synchronisation (the forward and the backward agents of an utterance are workgroups of ONE grid per phase); it can be
 * captured in a hipGraph once the workspace is sized (mm_batch_reserve).  Two batches driven from two caller streams do
 * not wait for each other. */
int mm_pdfposteriors_f32(mm_batch_t batch, const float *V, int64_t v_stride_b, int64_t v_stride_n,
                         const int32_t *lens, int64_t N, float *gamma, int64_t g_stride_b, int64_t g_stride_n,
                         int64_t g_stride_p, float *ttl, void *stream);

/* alpha-recursion(alpha_hat, T_hat', C_hat*V_hat) (src/inference.jl:62-74) as
 * called from pdfposteriors (:150-152): out is the reference's state_A, a
 * (sum S1) x (N+1) column-major matrix: element (b, n, s) at
 * out[n*out_stride_n + state_offset_b + s] with state_offset_b the running sum
 * of S1 over the batch.  Values are natural-log (un-normalised).
 * One shared graph in the pair form (the batches of mm_fbp_kernel, up to 250 pdfs) runs phase A of the pair kernels over all
 * N + 1 frames -- two utterances per workgroup, linear domain -- and one layout pass; utterances whose values leave float32's
 * range (sharp emissions) and every other batch run the item kernel (log domain, one workgroup per utterance).  While the last
 * finished export of the direction handed more than half of its utterances to the item kernel, the next one starts there
 * (every 32nd call tries the linear-domain kernels again; the count is read from pinned memory without synchronising; never
 * during stream capture): the two paths agree within the parity bar, not in the last bits -- mm_batch_set_exact_policy pins the
 * choice (MM_EXACT_F32_FIRST: the linear-domain kernels first, always; MM_EXACT_F64_FIRST: the item kernel alone). */
int mm_alpharecursion_f32(mm_batch_t batch, const float *V, int64_t v_stride_b, int64_t v_stride_n,
                          const int32_t *lens, int64_t N, float *out, int64_t out_stride_n, void *stream);
/* beta-recursion(T_hat, C_hat*V_hat) (src/inference.jl:99-110); same layout (state_B).  Log and Tropical
 * batches (the latter is what maxstateposteriors -- docs/src/inference.md:5 -- combines with the tropical alpha). */
int mm_betarecursion_f32(mm_batch_t batch, const float *V, int64_t v_stride_b, int64_t v_stride_n,
                         const int32_t *lens, int64_t N, float *out, int64_t out_stride_n, void *stream);

/* maxstateposteriors (documented docs/src/inference.md:5, absent from src/ at this commit: src/MarkovModels.jl:56-57;
 * historical use test/test_algorithms.jl:279-281): the max-marginals of the tropical semiring,
 * mu = alpha (*) beta (/) best -- for every state and frame the weight of the best complete path through it relative
 * to the best path overall (0 on a best path, -inf where no complete path passes, -inf everywhere if the utterance
 * has no path).  MM_TROPICAL batches.  out: device, the layout of mm_alpharecursion_f32 (element (b, n, s) at
 * out[n*out_stride_n + state_offset_b + s], n = 0..N).  Computed on the device: tropical alpha and beta recursions
 * and the combination. */
int mm_maxstateposteriors_f32(mm_batch_t batch, const float *V, int64_t v_stride_b, int64_t v_stride_n,
                              const int32_t *lens, int64_t N, float *out, int64_t out_stride_n, void *stream);

/* bestpath (documented docs/src/inference.md:5-6, absent from src/ at this
 * commit: src/MarkovModels.jl:56-57; historical use examples/demo.ipynb cell 23).
 * Tropical alpha-recursion (src/inference.jl:62-74 with K = TropicalSemiring)
 * plus back-pointers and an on-device back-trace.  The batch must have been
 * built from MM_TROPICAL FSMs.
 *   path   device int32, element (b, n) at path[b*path_stride_b + n]: 0-based
 *          state at frame n < len_b, -1 for n >= len_b or when no path exists
 *   score  device float[B]: weight of the best path (-inf if none)
 *   bp     optional device int32 (NULL to use internal storage): back-pointers,
 *          element (b, n, s) at bp[n*bp_stride_n + state_offset_b + s], n = 0..N;
 *          bp = lowest source state among the maximisers, -1 if none. */
int mm_viterbi_f32(mm_batch_t batch, const float *V, int64_t v_stride_b, int64_t v_stride_n, const int32_t *lens,
                   int64_t N, int32_t *path, int64_t path_stride_b, float *score, int32_t *bp,
                   int64_t bp_stride_n, void *stream);

/* totalsum(alpha, T, omega, n) / totalcumsum(alpha, T, omega, n) (src/algorithms.jl:8-29;
 * totalweightsum(fsm, n) = totalcumsum, :36), one value per FSM of the batch, in the batch's semiring
 * (Log or Tropical):  v_1 = alpha, v_k = T' v_{k-1};
 *   cumulative = 0:  out[b] = omega . v_n            cumulative != 0:  out[b] = (+)_{k=1..n} omega . v_k
 * Runs the emission-free alpha-recursion on the extended system (the phony final state's self loop of
 * weight one is the accumulator) for n + 1 frames.  n >= 1; out: device float[B], natural log. */
int mm_totalsum_f32(mm_batch_t batch, int64_t n, int cumulative, float *out, void *stream);

/* ---- the generic entry: pdfposteriors(fsm, V_hats, C_hats) as the reference declares it (src/inference.jl:145-161) ----
 * Any semiring the FSMs were created with (MM_LOG, MM_TROPICAL, MM_PROB: the function is generic in K), float32 or
 * float64 (FSM{LogSemiring{Float64}} is what the reference's tests build: test/test_fsms.jl:3-7), any sparse state map
 * C_hat (not only the one-hot map of examples/prepare-lfmmi-graphs.jl:15-23), any (P+1) x (N+1) matrices V_hat (not
 * only those expand() makes).  Correctness first: a plain kernel that materialises alpha and beta like the reference.
 * Asynchronous on `stream` like every run call: its workspace (alpha and beta, 2 x N1 x sum S1 elements) and the
 * utterance descriptors live with the batch, grown lazily (growing synchronises; refused while the stream is capturing:
 * size them with mm_batch_reserve_ex first), so a steady-state call allocates nothing, waits for nothing and can be
 * captured in a hipGraph.  The fast kernels are behind mm_pdfposteriors_f32.
 *   maps     NULL (every FSM's own one-hot state map), or B handles (NULL entries: the FSM's own)
 *   val_bytes 4 / 8: the type of Vhat, gamma, ttl
 *   P1       rows of every V_hat_b (P + 1).  Must equal the number of pdfs of the state map in force for every utterance
 *            (src/inference.jl:146-150: vcat(V_hats...) against blockdiag(C_hats...)'); MM_ERR_DIM otherwise -- e.g. a
 *            P x N matrix that expand() (:54-60) was not applied to
 *   Vhat     device; element (b, n, p) of V_hat_b at Vhat[b*v_stride_b + n*v_stride_n + p], n = 0..N1-1 (N1 = N + 1 columns),
 *            p = 0..P1-1 (P1 = P + 1 rows: the last is the phony pdf); values in the semiring's own domain
 *   gamma    device, out: element (b, n, p), n < N1 - 1, p < P1 - 1 (the reference drops the last row and column, :160);
 *            exp() of the quotient for MM_LOG / MM_TROPICAL, the quotient itself for MM_PROB
 *   ttl      device [B], out: the minimum over ALL N1 columns of the per-column sum (:159), in the semiring's domain
 * Z = 0 yields gamma = 0 and ttl = zero(K) (the reference: 0/0). */
typedef struct mm_statemap_s *mm_statemap_t;
/* C_hat (S1 x P1) as CSR: rowptr[S1+1], colidx / val [nnz]; host pointers, not retained. */
int mm_statemap_create(int semiring, int64_t S1, int32_t P1, int64_t nnz, int index_bytes, int index_base, int val_bytes,
                       const void *rowptr, const void *colidx, const void *val, mm_statemap_t *out);
int mm_statemap_destroy(mm_statemap_t map);
int mm_pdfposteriors_ex(mm_batch_t batch, const mm_statemap_t *maps, int val_bytes, int32_t P1, const void *Vhat,
                        int64_t v_stride_b, int64_t v_stride_n, int64_t N1, void *gamma, int64_t g_stride_b, int64_t g_stride_n,
                        int64_t g_stride_p, void *ttl, void *stream);
/* Allocate the generic entry's workspace for calls with val_bytes-sized values and up to N1 = N + 1 columns now
 * (synchronises if it has to grow): call before capturing mm_pdfposteriors_ex in a hipGraph. */
int mm_batch_reserve_ex(mm_batch_t batch, int val_bytes, int64_t N1);

/* Deterministic mode (default off).  Every kernel but one reduces in a fixed order; the general ("item") kernel -- the
 * path of small deep graphs such as LF-MMI numerators -- adds a pdf's state posteriors with LDS float atomics, so the
 * last bits of its gamma can differ between runs.  on != 0 makes it sum over per-pdf state lists in a fixed order
 * instead (one more workgroup barrier per frame: ~20 % slower on such graphs).  The reference has no such switch: its
 * CPU path is deterministic, its CUDA path (src/linalg.jl:213-233) reduces in warp-shuffle order, also fixed. */
int mm_batch_set_deterministic(mm_batch_t batch, int on);

/* Posterior floor of the fast (linear-domain) kernels, default 1e-30.  Those kernels keep float32 products, so terms more
 * than ~126 log2 below their frame's scale drop out; an utterance is accepted from them only if every posterior that can
 * have been lost that way is below the floor (otherwise the exact kernels compute it again, mm_batch_last_redo_count).
 * With sharp emissions (a trained acoustic model: the forward and the backward mass of a frame sit on different states)
 * the default sends every utterance to the exact kernels -- 3 to 6 times the time of a call.  A caller to whom posteriors
 * below, say, 1e-12 are zero (LF-MMI gradients) says so here: results then differ from the reference's by less than the
 * floor in any posterior (those below it may come out as 0), log Z is unaffected beyond 1e-4 relative as before, and
 * inputs up to ~6 sigma of logit spread stay on the fast kernels.  floor in [1e-30, 1e-6]; the log-domain kernels (wave,
 * item, generic) are exact and ignore it.  The reference has no such switch (one algorithm, log domain throughout).
 * Caveat of the DEFAULT floor: the acceptance test bounds every dropped TERM, i.e. an absolute error
 * of K * 7.9e-31 for a pdf of K states -- the 1e-4 relative bar on log gamma holds by construction for posteriors above
 * ~1e-26; between 1e-30 and 1e-26 a posterior of an ACCEPTED utterance can be low by about a per cent (seen once in ~1000
 * fuzzer comparisons: 1.436e-30 computed as 1.423e-30).  mm_batch_set_exact_policy(batch, MM_EXACT_F64_FIRST) removes the
 * float32 kernels from the path altogether. */
int mm_batch_set_posterior_floor(mm_batch_t batch, float floor);

/* How many utterances of the LAST mm_pdfposteriors_f32 call on this batch the fast (linear-domain) kernels handed to
 * the exact kernels ("flag and redo": a value left the range in which float32 products are exact enough; results are
 * the log semiring's either way, only the time differs -- a redone utterance is computed twice).  Reads the marks the
 * call left in the batch's workspace: synchronises `stream` (the stream of that call), nothing on the hot path.
 * *n = 0 for batches that run on the exact kernels only, or before the first call.  The reference has no such
 * path (src/inference.jl:145-161 runs one algorithm for every input); this is an observability hook. */
int mm_batch_last_redo_count(mm_batch_t batch, void *stream, int64_t *n);
/* ... and how many of THOSE the float64 exact kernels (mm_kernel_dpair.hip: the first stop of a marked utterance of a
 * shared-graph batch) handed on to the log-domain kernels: values beyond the double's range that carry mass -- normally 0. */
int mm_batch_last_fallback_count(mm_batch_t batch, void *stream, int64_t *n);
/* 1 if the last mm_pdfposteriors_f32 call on this batch skipped the float32 kernels and ran the float64 exact kernels on
 * the whole batch (the engine does that while more than a quarter of the utterances of the last finished call were
 * beyond the float32 kernels -- a sharp acoustic model; mm_batch_last_redo_count then reports the whole batch), else 0.
 * No synchronisation.  Observability only: results are the same either way. */
int mm_batch_last_exact_first(mm_batch_t batch);
/* Team kernels (graphs beyond one compute unit): out[0] = workgroups of the float32 team kernels' phase-A launches, since the last
 * call of this function, that found their whole team on ONE XCD (they exchange rows by plain stores through that XCD's L2; the
 * others by write-through stores), out[1] = all such workgroups.  The kernels count from the FIRST call of this function on (which
 * returns zeros): a batch nobody asks carries no measurement atomics.  Synchronises the device; a measurement aid (bench.py). */
int mm_batch_team_xcd_stats(mm_batch_t batch, int out[2]);

/* Which linear-domain kernels a shared-graph batch starts with (the batches of the pair / split pair kernels; others ignore it).
 *   MM_EXACT_AUTO (default)  float32 first; float64 first while the last FINISHED call left utterances marked and that costs
 *                            no more rounds of workgroups.  The engine reads that call's count from pinned host memory without
 *                            synchronising, so WHICH kernels a pipelined caller's next call launches depends on host / device
 *                            timing (results are within the parity bar either way; the last bits of gamma and the time are not
 *                            reproducible from run to run), and a hipGraph capture freezes the choice current at capture time.
 *   MM_EXACT_F32_FIRST       always float32 first, marked utterances redone by the float64 kernels: the launches of a call
 *                            are a function of the call alone
 *   MM_EXACT_F64_FIRST       always the float64 kernels on the whole batch (a trained acoustic model's sharp outputs)
 * With a fixed policy two identical call sequences give bit-identical results.  During stream capture MM_EXACT_AUTO behaves
 * as MM_EXACT_F32_FIRST (a captured graph must not bake in the state of an unrelated earlier call).  The reference has no
 * such switch (src/inference.jl:145-161: one algorithm for every input). */
enum mm_exact_policy { MM_EXACT_AUTO = 0, MM_EXACT_F32_FIRST = 1, MM_EXACT_F64_FIRST = 2 };
int mm_batch_set_exact_policy(mm_batch_t batch, int policy);

/* What a RANGE MARK of the float32 linear-domain kernels means (the batches of the pair / split pair kernels; others ignore it).
 * Those kernels mark an utterance when a value of a state vector left the range in which float32 products keep every term.
 *   MM_MARKS_DECIDE (default)  the finish kernel clears the mark when two criteria say that nothing that matters was lost (the
 *                              frames' log Z agree within 2e-4 log2; every term that can have been dropped is below the posterior
 *                              floor): the reference's own benchmark graph marks every utterance after ~55 frames -- its
 *                              initial-context states decay out of the float range -- and none of those marks carries mass.
 *                              Contract: log gamma within 1e-4 relative for posteriors above ~1e-24, an ABSOLUTE error below
 *                              ~1e-27 for smaller ones (a flushed value of a recursion takes its descendants along: a posterior
 *                              of 8.5e-29 has been seen computed as 0 on a 5-frame utterance; DESIGN.md section 3,
 *                              tests/test_gpu_exact.py::test_fuzzer_findings_pin_the_parity_contract).
 *   MM_MARKS_KEEP              a range mark always stays: the exact kernels (float64 / wide-exponent values, 1022 log2 of range)
 *                              compute every utterance whose float32 vectors left the range, whatever the criteria say.  The
 *                              1e-4 relative bar on log gamma then holds down to 1e-30 like SURVEY section 8(d) states it; costs a
 *                              second pass on inputs that raise marks (the reference's WSJ denominator: 2.03 instead of 1.76 ms;
 *                              config 3 on the benchmark's inputs raises none: unchanged).
 * Per batch, takes effect at the next call; the reference has no such switch (src/inference.jl:145-161: one algorithm, log
 * domain, every input). */
enum mm_mark_policy { MM_MARKS_DECIDE = 0, MM_MARKS_KEEP = 1 };
int mm_batch_set_mark_policy(mm_batch_t batch, int policy);

/* What mm_pdfposteriors_f32 does with the posteriors, for the caller's step right behind the path (examples/test_cuda.jl:140-152:
 * the LF-MMI gradient gamma_den - gamma_num, formed by the reference in a separate broadcast over two B x P x N arrays):
 *   accumulate = 0, scale = 1 (default)   gamma_out  = gamma, frames n >= len_b zeroed
 *   accumulate = 0, scale = s             gamma_out  = s * gamma
 *   accumulate = 1, scale = s             gamma_out += s * gamma (frames n >= len_b left as they are): the numerator call with
 *                                         s = -1 on the buffer the denominator call has just written leaves the gradient there,
 *                                         no third pass.  Every element receives exactly one float add per call (a no-return
 *                                         atomic): the result does not depend on any order.
 * Supported by batches of the wave kernel (every graph <= 1023 states and <= 4096 arc slots per direction, any batch: LF-MMI
 * numerators; mm_batch_kernels says which kernel a batch runs); MM_ERR_UNSUPPORTED for the others (their float32 kernels may hand
 * an utterance to the exact kernels, whose result must REPLACE the first: the caller subtracts in a pass of its own).  Per batch,
 * takes effect at the next call. */
int mm_batch_set_gamma_mode(mm_batch_t batch, int accumulate, float scale);

/* ---- the reference's semiring linear algebra on caller-owned device arrays (src/linalg.jl) ------------------------------
 * Generic in K like the reference: semiring in {MM_LOG, MM_TROPICAL, MM_PROB}, val_bytes 4 (Float32) / 8 (Float64) for
 * every value array of a call.  A is a CuSparseMatrixCSR{K} (src/linalg.jl:80: Cint indices): rowptr[rows + 1],
 * colval[nnz], nzval[nnz], index_base 1 as Julia stores them (0 accepted).  All pointers are DEVICE pointers; the calls
 * are asynchronous on `stream`; sizes are checked like the reference's @boundscheck (MM_ERR_DIM = DimensionMismatch).
 * Values are in the semiring's own domain (natural log for Log / Tropical, zero(K) = -inf). */

/* LinearAlgebra.mul!(c, A, b) (src/linalg.jl:163-184; kernel _cukernel_mul_smdv! :213-233, warp_reduce :204-211):
 * c[r] = (+)_k nzval[k] (*) b[colval[k]] over row r.  b_len / c_len: the lengths of b and c (checked against cols / rows).
 * An A without stored entries leaves c untouched (`if length(A.nzVal) > 0`, :169). */
int mm_spmv(int semiring, int val_bytes, int64_t rows, int64_t cols, int64_t nnz, const int32_t *rowptr, const int32_t *colval,
            int index_base, const void *nzval, const void *b, int64_t b_len, void *c, int64_t c_len, void *stream);
/* LinearAlgebra.mul!(C, A, B, alpha, beta) (src/linalg.jl:240-262; kernel _cukernel_mul_smdm! :268-280):
 * C = (beta (*) C) (+) A (*) B with B (b_rows x b_cols) and C (c_rows x c_cols) column-major with leading dimensions ldb /
 * ldc.  beta = 0: C is overwritten (fill!(C, zero(K)), :247 -- what the 3-argument mul! passes); beta = 1: accumulated into;
 * any other beta: rmul!(C, beta) first (:247), beta a value of K's domain.  alpha is ignored like in the reference. */
int mm_spmm(int semiring, int val_bytes, int64_t rows, int64_t cols, int64_t nnz, const int32_t *rowptr, const int32_t *colval,
            int index_base, const void *nzval, const void *B, int64_t b_rows, int64_t b_cols, int64_t ldb, void *C, int64_t c_rows,
            int64_t c_cols, int64_t ldc, double beta, void *stream);
/* Sparse vector (.) dense vector broadcast, _copyto!(f, dest, x::CuSparseVector, y) (src/linalg.jl:294-315; kernel :320-328):
 * dest = zero(K) everywhere, then dest[nzind[i]] = f(nzval[i], y[nzind[i]]), f = (*) for op 0 (elmul!, :290), (/) for op 1
 * (eldiv!, :292).  n = length of x, y (y_len) and dest (dest_len). */
int mm_svdv(int semiring, int val_bytes, int op, int64_t n, int64_t nnz, const int32_t *nzind, int index_base, const void *nzval,
            const void *y, int64_t y_len, void *dest, int64_t dest_len, void *stream);

/* ---- multi-GPU boundary (one process per GPU, RCCL over xGMI) -------------------------------------------------
 * The batch is block diagonal (src/fsmops.jl:28-36, src/inference.jl:28-36): utterances shard over the ranks with no
 * collective on the data path.  The only exchange is the total log-likelihood the LF-MMI loss consumes
 * (examples/test_cuda.jl:140-152 sums ttl_num / ttl_den over the utterances), offered here over RCCL so that a host
 * binding without torch.distributed (julia/MarkovModelsAMD.jl) has it too.
 * comm: an ncclComm_t of the calling process.  The library does not link RCCL and never opens one of its own (an
 * ncclComm_t belongs to the RCCL build that made it): it resolves ncclAllReduce / ncclAllGather in the library handed
 * over with mm_set_rccl, else among the process's global symbols.  Both calls are asynchronous on `stream`;
 * MM_ERR_UNSUPPORTED if RCCL is not visible, MM_ERR_HIP if RCCL fails. */

/* dl_handle: what dlopen() returned for the RCCL `comm` was made with (PyTorch loads its bundled librccl.so with local
 * visibility; Libdl.dlopen in Julia likewise), or NULL to go back to the process's global symbols. */
int mm_set_rccl(void *dl_handle);

/* sum (device double[1]) = sum over ALL ranks of sum_b ttl[b], b < B_local (device float[B_local], the ttl output of
 * mm_pdfposteriors_f32), accumulated in float64: a one-block reduction and a one-element all-reduce. */
int mm_allreduce_logz(void *comm, const float *ttl, int64_t B_local, double *sum, void *stream);

/* all (device float[world * B_max]) = the ttl vectors of all ranks, rank r at all + r * B_max.  Every rank passes
 * the same B_max >= its B_local and a ttl buffer of B_max floats (pad with -inf: ranks may hold shards of different
 * sizes). */
int mm_allgather_ttl(void *comm, const float *ttl, int64_t B_max, float *all, void *stream);

/* Test aid (host only, no GPU): evaluate one semiring product out = M (x) in
 * THROUGH THE PACKED FORM the kernels consume, direction 0: M = T_hat'
 * (forward), 1: M = T_hat (backward).  in/out: host float[S1], natural log.
 * argmax (may be NULL): for MM_TROPICAL the back-pointer per row. */
int mm_debug_packed_product(mm_fsm_t fsm, int direction, const float *in, float *out, int32_t *argmax);

/* Test aid (host only, no GPU): the same product evaluated THROUGH THE QUAD FORM of the fast
 * pdfposteriors kernel (internal renumbering, quads of 4 arcs laid out for KQ quads per lane, per-lane
 * running sums, row totals from the lane partials), in the linear domain relative to max(in) like the
 * kernel does.  MM_LOG FSMs only.  stats (may be NULL) receives {quads, lanes, modelled LDS cycles per
 * gather instruction with arcs in CSR order, the same after the bank-aware placement}. */
int mm_debug_quad_product(mm_fsm_t fsm, int direction, int KQ, const float *in, float *out, double stats[4]);

/* Test aid (host only, no GPU): the same product evaluated THROUGH THE ROW-LANE FORM of the row kernels (mm_rows.h:
 * rows sorted by size and dealt to the compute waves as segments, arcs in per-lane register slots, group sums,
 * internal numbering = finishing order), in the linear domain relative to max(in) like the kernels do.  MM_LOG FSMs
 * only.  stats (may be NULL) receives {arc slots per lane (KA), compute waves, segments, real arcs / arc slots,
 * cost of the most loaded wave, of the least loaded one, modelled LDS cycles per gather instruction with arcs in
 * CSR order, the same after the bank-aware placement}.  Returns MM_ERR_UNSUPPORTED if the FSM does not fit the form. */
int mm_debug_row_product(mm_fsm_t fsm, int direction, const float *in, float *out, double stats[8]);
/* The same with the packer's options: flags bit 0 = the pair form of the pair kernels (8-byte positions, its cost model),
 * bits 1-2 = copies of the linear vector the arcs may read (0: the form's default, 1, 2), bit 3 = the second copy
 * scrambles the low five position bits with the next five instead of rotating by half the banks. */
int mm_debug_row_product_ex(mm_fsm_t fsm, int direction, int flags, const float *in, float *out, double stats[8]);

/* Test aid (host only, no GPU): the product evaluated THROUGH THE SPLIT FORMS of the team kernels (mm_rows.h
 * make_rows_split: the rows cut into H sets, one row-lane form per set whose arcs read the whole team's vector).
 * stats (may be NULL) receives {arc slots per lane, positions of the team's vector, segments, real arcs / arc slots,
 * cost of the most / least loaded wave, modelled LDS cycles per gather before / after the bank-aware placement}. */
int mm_debug_split_product(mm_fsm_t fsm, int H, int direction, const float *in, float *out, double stats[8]);

/* Test aid (host only, no GPU): the product evaluated THROUGH THE WAVE FORM of the wave kernel (one wave per direction:
 * segments of 64 / g rows, at most 4 arcs per lane and segment, log2 weights, lane-group log-sum-exp).  stats (may be
 * NULL) receives {arc slots per lane, segments, real arcs / arc slots, modelled LDS cycles per gather}.
 * MM_ERR_UNSUPPORTED if the FSM does not fit the form (more than 16 segments). */
int mm_debug_wave_product(mm_fsm_t fsm, int direction, const float *in, float *out, double stats[4]);

/* Test aid (host only, no GPU): the product evaluated THROUGH THE STREAM FORM of the stream kernels (mm_stream.hip: rows sorted
 * by length and cut into segments of 64, one lane per row, rows of more than 128 arcs on a whole wave; 8-byte arc records
 * {LDS address of the source, high dword of the weight's double}; float64 accumulation) exactly as a workgroup walks it.
 * MM_LOG FSMs of up to 16 370 states and 1024 pdfs (MM_ERR_UNSUPPORTED otherwise).  stats (may be NULL) receives {arc slots per
 * lane summed over the 15 waves, segments, real arcs / arc slots, arc slots of the most loaded wave}. */
int mm_debug_stream_product(mm_fsm_t fsm, int direction, const float *in, float *out, double stats[4]);
/* ... through the forms of a TEAM of H = 1, 2 or 4 workgroups (round 6: the rows of a direction dealt to H sets, one record stream per
 * set and wave, the sets' regions of the vector padded to multiples of 4 positions): the same product, set by set. */
int mm_debug_stream_team_product(mm_fsm_t fsm, int H, int direction, const float *in, float *out, double stats[4]);

/* Test aid (host only, no GPU): the static bound the fast kernels use to recognise dead rows without a walk --
 * the fewest arcs from an initial state to every state (direction 0) or from every state to the phony final
 * state (direction 1), on the pruned graph; -1 = unreachable (such states are dropped).  out: host int32[S1].
 * alpha_n[s] (resp. beta_n[s]) is zero(K) whenever n - 1 (resp. len + 1 - n) is smaller.  MM_LOG FSMs only. */
int mm_debug_reach_distance(mm_fsm_t fsm, int direction, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* MARKOVMODELS_AMD_H */
