"""CPU oracle for the MarkovModels.jl inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference algorithm.  It is imported
only by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- never by the product package (``markovmodels.jl_amd``), which
fails loudly when its HIP library is missing.

Parity status: the reference is Julia (no ``julia`` binary in the build image,
Semirings.jl 0.5 is an un-vendored dependency), so nothing here could be run
against the real package.  The oracle is pinned against the known-answer
vectors the reference's own (partly disabled) tests and demo notebook hold --
``tests/golden/known_answers.json``, see ``tests/golden/make_known_answers.py``
for the file:line of each.  For ``pdfposteriors`` the reference's *active* test
suite pins nothing (test/runtests.jl:25-27 comments the inference tests out), so
beyond those vectors: PARITY UNPINNED.  Viterbi back-pointers have no reference
implementation at this commit (src/MarkovModels.jl:56-57): the tie rule
(lowest source index) is ours.

Every function cites the reference file:line (relative to /root/reference) it
restates.  Indices are 0-based here; the reference is 1-based.

Semiring arithmetic (Semirings.jl 0.5, compat entry Project.toml:18; the
package's published definitions):
    LogSemiring      x (+) y = logaddexp(x, y)   x (*) y = x + y   x (/) y = x - y
                     zero = -inf, one = 0
    TropicalSemiring x (+) y = max(x, y)         rest as Log
    ProbSemiring     ordinary + * /              zero = 0, one = 1
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

# ----------------------------------------------------------------------------
# Semirings
# ----------------------------------------------------------------------------


def logaddexp_scalar(x: float, y: float) -> float:
    """logaddexp as in LogExpFunctions/Semirings.jl (and the check in
    test/test_semirings.jl:4-6): max + log1p(exp(-|x-y|)), with
    (-inf) (+) (-inf) = -inf (no NaN)."""
    m = max(x, y)
    if m == -math.inf:
        return -math.inf
    if x == math.inf or y == math.inf:
        return math.inf
    return m + math.log1p(math.exp(-abs(x - y)))


@dataclass(frozen=True)
class Semiring:
    name: str
    zero: float
    one: float

    def add(self, x, y):  # (+)
        if self.name == "log":
            with np.errstate(invalid="ignore", divide="ignore"):
                return np.logaddexp(x, y)
        if self.name == "tropical":
            return np.maximum(x, y)
        return x + y

    def mul(self, x, y):  # (*)
        if self.name == "prob":
            return x * y
        with np.errstate(invalid="ignore"):
            r = x + y
        # zero annihilates: (-inf) + (+inf) must stay zero, never NaN
        return r

    def div(self, x, y):  # (/)
        if self.name == "prob":
            return x / y
        with np.errstate(invalid="ignore"):
            return x - y

    def add_at(self, out, idx, vals):
        """out[idx[k]] (+)= vals[k], strictly in the order k = 0, 1, ...  (the
        order matters for float32 faithfulness of the CSC scatter loop)."""
        if self.name == "log":
            with np.errstate(invalid="ignore", divide="ignore"):
                np.logaddexp.at(out, idx, vals)
        elif self.name == "tropical":
            np.maximum.at(out, idx, vals)
        else:
            np.add.at(out, idx, vals)

    def reduce(self, x, axis):
        if self.name == "log":
            # sequential (+) along the axis, like a Julia reduction over a small dim
            x = np.moveaxis(x, axis, 0)
            acc = np.full(x.shape[1:], self.zero, dtype=x.dtype)
            for k in range(x.shape[0]):
                with np.errstate(invalid="ignore", divide="ignore"):
                    acc = np.logaddexp(acc, x[k])
            return acc
        if self.name == "tropical":
            return np.max(x, axis=axis)
        return np.sum(x, axis=axis)


LOG = Semiring("log", -math.inf, 0.0)
TROPICAL = Semiring("tropical", -math.inf, 0.0)
PROB = Semiring("prob", 0.0, 1.0)
SEMIRINGS = {"log": LOG, "tropical": TROPICAL, "prob": PROB}


# ----------------------------------------------------------------------------
# Sparse containers (CSC like Julia's SparseMatrixCSC, 0-based)
# ----------------------------------------------------------------------------


@dataclass
class CSC:
    colptr: np.ndarray  # int64, len ncols+1
    rowval: np.ndarray  # int64, len nnz, ascending inside a column
    nzval: np.ndarray  # float, len nnz
    shape: Tuple[int, int]

    @property
    def nnz(self) -> int:
        return int(self.rowval.shape[0])

    def colidx(self) -> np.ndarray:
        """column index of every stored entry (CSC traversal order)."""
        return np.repeat(np.arange(self.shape[1], dtype=np.int64), np.diff(self.colptr))

    def transpose(self) -> "CSC":
        """copy(A') -- materialised transpose, as src/inference.jl:149,151."""
        cols = self.colidx()
        return csc_from_coo(self.rowval, cols, self.nzval, (self.shape[1], self.shape[0]), K=None, transpose_in=True)

    def todense(self, K: Semiring) -> np.ndarray:
        d = np.full(self.shape, K.zero, dtype=self.nzval.dtype)
        d[self.rowval, self.colidx()] = self.nzval
        return d


def csc_from_coo(I, J, V, shape, K: Optional[Semiring], transpose_in=False) -> CSC:
    """sparse(I, J, V, m, n): duplicates are combined with (+) (Julia's
    ``sparse`` default combine is ``+``), entries sorted column-major."""
    I = np.asarray(I, dtype=np.int64)
    J = np.asarray(J, dtype=np.int64)
    V = np.asarray(V)
    if transpose_in:
        I, J = J, I
    order = np.lexsort((I, J))  # primary J (column), secondary I (row)
    I, J, V = I[order], J[order], V[order]
    if I.size and K is not None:
        keep = np.ones(I.size, dtype=bool)
        keep[1:] = (I[1:] != I[:-1]) | (J[1:] != J[:-1])
        if not keep.all():
            grp = np.cumsum(keep) - 1
            Vc = np.full(int(keep.sum()), K.zero, dtype=V.dtype)
            K.add_at(Vc, grp, V)
            I, J, V = I[keep], J[keep], Vc
    colptr = np.zeros(shape[1] + 1, dtype=np.int64)
    np.add.at(colptr, J + 1, 1)
    colptr = np.cumsum(colptr)
    return CSC(colptr, I.copy(), V.copy(), (int(shape[0]), int(shape[1])))


def blockdiag(mats: Sequence[CSC]) -> CSC:
    """SparseArrays.blockdiag -- used by rawunion (src/fsmops.jl:28-36) and by
    pdfposteriors for the state maps (src/inference.jl:148)."""
    colptr = [np.zeros(1, dtype=np.int64)]
    rowval, nzval = [], []
    r0 = 0
    n0 = 0
    for m in mats:
        colptr.append(m.colptr[1:] + n0)
        rowval.append(m.rowval + r0)
        nzval.append(m.nzval)
        r0 += m.shape[0]
        n0 += m.nnz
    c0 = sum(m.shape[1] for m in mats)
    return CSC(np.concatenate(colptr), np.concatenate(rowval), np.concatenate(nzval), (r0, c0))


def spmm_csc(A: CSC, Bd: np.ndarray, K: Semiring) -> np.ndarray:
    """C = A * B following Julia's generic ``mul!(C, A::SparseMatrixCSC, B, true, false)``
    loop (stdlib SparseArrays/linalg.jl): C is filled with zero(K); then for
    every column k of B, for every column ``col`` of A in order, for every
    stored entry j of that column in order: C[rv[j], k] (+)= nzv[j] (*) B[col, k].
    This is the loop behind src/inference.jl:70,107,150,155 on the CPU."""
    vec = Bd.ndim == 1
    B2 = Bd[:, None] if vec else Bd
    C = np.full((A.shape[0], B2.shape[1]), K.zero, dtype=B2.dtype)
    cols = A.colidx()
    for k in range(B2.shape[1]):
        contrib = K.mul(A.nzval.astype(B2.dtype, copy=False), B2[cols, k])
        K.add_at(C[:, k], A.rowval, contrib)
    return C[:, 0] if vec else C


# ----------------------------------------------------------------------------
# FSM  (src/fsm.jl)
# ----------------------------------------------------------------------------


@dataclass
class FSM:
    """FSM{K,L} (src/fsm.jl:7-17): alpha_hat is the initial weight vector
    extended with a zero for the phony final state; T_hat is the transition
    matrix extended with the final state ([T omega; 0 one], src/fsm.jl:19-28)."""

    K: Semiring
    alpha_hat: np.ndarray  # dense, len S+1
    T_hat: CSC  # (S+1) x (S+1), T_hat[i, j] = weight of arc i -> j
    labels: list

    @property
    def nstates(self) -> int:  # src/fsm.jl:84
        return self.alpha_hat.shape[0] - 1


def make_fsm(K: Semiring, initws, arcs, finalws, labels, dtype=np.float64) -> FSM:
    """FSM(initws, arcs, finalws, lambda) (src/fsm.jl:50-71) followed by the
    inner constructor (src/fsm.jl:19-28).  ``initws``/``finalws``: [(state, w)];
    ``arcs``: [((src, dst), w)]; all 0-based."""
    S = len(labels)
    alpha = np.full(S + 1, K.zero, dtype=dtype)
    for s, w in initws:
        alpha[s] = K.add(alpha[s], dtype(w))  # sparsevec combines duplicates with +
    I = [a[0][0] for a in arcs] + [f[0] for f in finalws] + [S]
    J = [a[0][1] for a in arcs] + [S] * len(finalws) + [S]
    V = [a[1] for a in arcs] + [f[1] for f in finalws] + [K.one]
    T_hat = csc_from_coo(I, J, np.asarray(V, dtype=dtype), (S + 1, S + 1), K)
    return FSM(K, alpha, T_hat, list(labels))


def fsm_from_json(s: str, dtype=np.float64) -> FSM:
    """FSM(::AbstractString) (src/fsm.jl:73-82); JSON states are 1-based."""
    data = json.loads(s)
    name = data["semiring"]
    K = LOG if "Log" in name else TROPICAL if "Tropical" in name else PROB
    return make_fsm(
        K,
        [(a - 1, b) for a, b in data["initstates"]],
        [((a - 1, b - 1), c) for a, b, c in data["arcs"]],
        [(a - 1, b) for a, b in data["finalstates"]],
        data["labels"],
        dtype,
    )


def fsm_parts(fsm: FSM):
    """.alpha / .T / .omega accessors (src/fsm.jl:30-40) as dense arrays."""
    S = fsm.nstates
    Td = fsm.T_hat.todense(fsm.K)
    return fsm.alpha_hat[:S].copy(), Td[:S, :S].copy(), Td[:S, S].copy()


def renorm(fsm: FSM) -> FSM:
    """renorm (src/fsmops.jl:71-79): Z = one ./ (sum(T, dims=2) .+ omega);
    alpha ./ sum(alpha); T .* Z; omega .* Z."""
    K = fsm.K
    S = fsm.nstates
    a, T, o = fsm_parts(fsm)
    rows = K.reduce(T, axis=1)
    Z = K.div(np.full(S, K.one, dtype=a.dtype), K.add(rows, o))
    a2 = K.div(a, K.reduce(a, axis=0))
    T2 = K.mul(T, Z[:, None])
    o2 = K.mul(o, Z)
    initws = [(i, a2[i]) for i in range(S) if a[i] != K.zero]
    arcs = [((i, j), T2[i, j]) for i in range(S) for j in range(S) if T[i, j] != K.zero]
    finalws = [(i, o2[i]) for i in range(S) if o[i] != K.zero]
    return make_fsm(K, initws, arcs, finalws, fsm.labels, dtype=a.dtype.type)


def rawunion(fsms: Sequence[FSM]) -> FSM:
    """rawunion (src/fsmops.jl:28-36): vcat the alpha_hat's, blockdiag the T_hat's."""
    return FSM(
        fsms[0].K,
        np.concatenate([f.alpha_hat for f in fsms]),
        blockdiag([f.T_hat for f in fsms]),
        sum((f.labels for f in fsms), []),
    )


def statemap(state2pdf: Sequence[int], numpdf: int, K: Semiring, dtype=np.float64) -> CSC:
    """statemap (examples/prepare-lfmmi-graphs.jl:15-23): (S+1) x (P+1) sparse
    with exactly one ``one(K)`` per row; state -> pdf id, final state -> P+1."""
    S = len(state2pdf)
    I = list(range(S + 1))
    J = list(state2pdf) + [numpdf]
    return csc_from_coo(I, J, np.full(S + 1, K.one, dtype=dtype), (S + 1, numpdf + 1), K)


# ----------------------------------------------------------------------------
# Inference  (src/inference.jl)
# ----------------------------------------------------------------------------


def expand(lhs: np.ndarray, seqlength: Optional[int], K: Semiring) -> np.ndarray:
    """expand (src/inference.jl:54-60): (P x N) -> (P+1) x (N+1)."""
    P, N = lhs.shape
    if seqlength is None:
        seqlength = N
    out = np.full((P + 1, N + 1), K.zero, dtype=lhs.dtype)
    out[:P, :N] = lhs
    out[:P, seqlength:] = K.zero
    out[P, seqlength:] = K.one
    return out


def alpharecursion(alpha_hat: np.ndarray, T_hat_T: CSC, lhs: np.ndarray, K: Semiring) -> np.ndarray:
    """alpha-recursion (src/inference.jl:62-74)."""
    S, N = alpha_hat.shape[0], lhs.shape[1]
    A = np.empty((S, N), dtype=lhs.dtype)
    A[:, 0] = K.mul(alpha_hat.astype(lhs.dtype), lhs[:, 0])
    for n in range(1, N):
        buf = spmm_csc(T_hat_T, A[:, n - 1], K)
        A[:, n] = K.mul(buf, lhs[:, n])
    return A


def betarecursion(T_hat: CSC, lhs: np.ndarray, K: Semiring) -> np.ndarray:
    """beta-recursion (src/inference.jl:99-110)."""
    S, N = T_hat.shape[0], lhs.shape[1]
    B = np.empty((S, N), dtype=lhs.dtype)
    B[:, N - 1] = K.one
    for n in range(N - 2, -1, -1):
        buf = K.mul(B[:, n + 1], lhs[:, n + 1])
        B[:, n] = spmm_csc(T_hat, buf, K)
    return B


# ----------------------------------------------------------------------------
# total-sum family  (src/algorithms.jl)
# ----------------------------------------------------------------------------


def _dense_Tt_v(T: np.ndarray, v: np.ndarray, K: Semiring) -> np.ndarray:
    """T' * v in the semiring K (dense; T[i, j] = weight of arc i -> j)."""
    return K.reduce(K.mul(T, v[:, None]), axis=0)


def _dot(v: np.ndarray, w: np.ndarray, K: Semiring):
    return K.reduce(K.mul(v, w), axis=0)


def totalcumsum(alpha: np.ndarray, T: np.ndarray, omega: np.ndarray, n: int, K: Semiring):
    """totalcumsum(alpha, T, omega, n) (src/algorithms.jl:8-16)."""
    v = alpha
    total = _dot(v, omega, K)
    for _ in range(2, n + 1):
        v = _dense_Tt_v(T, v, K)
        total = K.add(total, _dot(v, omega, K))
    return total


def totalsum(alpha: np.ndarray, T: np.ndarray, omega: np.ndarray, n: int, K: Semiring):
    """totalsum(alpha, T, omega, n) (src/algorithms.jl:23-29)."""
    v = alpha
    for _ in range(2, n + 1):
        v = _dense_Tt_v(T, v, K)
    return _dot(v, omega, K)


def totalweightsum(fsm: FSM, n: Optional[int] = None):
    """totalweightsum(fsm, n = nstates(fsm)) (src/algorithms.jl:36)."""
    a, T, w = fsm_parts(fsm)
    return totalcumsum(a, T, w, fsm.nstates if n is None else n, fsm.K)


def pdfposteriors(fsm: FSM, Vhats: Sequence[np.ndarray], Chats: Sequence[CSC]):
    """pdfposteriors(fsm, V_hats, C_hats) (src/inference.jl:145-161).

    ``fsm`` is the rawunion of the batch, ``Vhats`` the B expanded (P+1)x(N+1)
    log-likelihood matrices, ``Chats`` the B state maps.  Returns
    (gamma[B, P, N] probabilities, ttl[B])."""
    K = fsm.K
    V = np.concatenate(Vhats, axis=0)  # :146
    Vk = V.copy()  # :147
    C = blockdiag(Chats)  # :148
    Ct = C.transpose()  # :149
    CV = spmm_csc(C, Vk, K)  # :150
    Tt = fsm.T_hat.transpose()  # :151
    state_A = alpharecursion(fsm.alpha_hat, Tt, CV, K)  # :152
    state_B = betarecursion(fsm.T_hat, CV, K)  # :153
    state_AB = K.mul(state_A, state_B)  # :154
    AB = spmm_csc(Ct, state_AB, K)  # :155
    Bsz = len(Vhats)
    Z = AB.reshape(Bsz, -1, V.shape[1])  # :156  (B, P+1, N+1) after permutedims
    sums = K.reduce(Z, axis=1)[:, None, :]  # :157
    with np.errstate(invalid="ignore"):
        Zn = K.div(Z, sums)  # :158
    ttl = sums.min(axis=(1, 2))  # :159
    if K.name == "prob":
        return Zn[:, :-1, :-1], ttl
    return np.exp(Zn[:, :-1, :-1]), ttl  # :160


def pdfposteriors_batch(fsm1: FSM, state2pdf, numpdf, lhs_list: Sequence[np.ndarray], lens: Sequence[int]):
    """Convenience: B copies of one FSM (the denominator case,
    examples/test_cuda.jl:112-113) with per-utterance lengths, going through
    expand -> rawunion -> pdfposteriors exactly like examples/test_cuda.jl:124-128."""
    K = fsm1.K
    dtype = lhs_list[0].dtype
    Vh = [expand(l, n, K) for l, n in zip(lhs_list, lens)]
    Cs = [statemap(state2pdf, numpdf, K, dtype=dtype.type)] * len(lhs_list)
    return pdfposteriors(rawunion([fsm1] * len(lhs_list)), Vh, Cs)


# ----------------------------------------------------------------------------
# Independent dense cross-check (test/test_algorithms.jl:28-63)
# ----------------------------------------------------------------------------


def _logsumexp(x, axis):
    m = np.max(x, axis=axis, keepdims=True)
    m0 = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide="ignore"):
        return np.squeeze(m0 + np.log(np.sum(np.exp(x - m0), axis=axis, keepdims=True)), axis=axis)


def dense_forward_backward(A_hat: np.ndarray, init_hat: np.ndarray, lhs: np.ndarray):
    """forward / backward / forward_backward of test/test_algorithms.jl:28-63:
    dense log-domain recursions on the graph *without* the final state.
    A_hat is the dense (S+1)x(S+1) log transition matrix, init_hat len S+1."""
    final = A_hat[:-1, -1]
    A = A_hat[:-1, :-1]
    init = init_hat[:-1]
    S, N = lhs.shape
    la = np.empty_like(lhs)
    la[:, 0] = lhs[:, 0] + init
    for n in range(1, N):
        la[:, n] = lhs[:, n] + _logsumexp(A + la[:, n - 1][:, None], axis=0)
    lb = np.empty_like(lhs)
    lb[:, N - 1] = final
    for n in range(N - 2, -1, -1):
        lb[:, n] = _logsumexp(A.T + (lb[:, n + 1] + lhs[:, n + 1])[:, None], axis=0)
    lg = la + lb
    sums = _logsumexp(lg, axis=0)
    return np.exp(lg - sums), sums.min()


# ----------------------------------------------------------------------------
# Viterbi (tropical alpha-recursion is reference-defined; back-pointers are ours)
# ----------------------------------------------------------------------------


def viterbi(fsm1: FSM, state2pdf, numpdf, lhs: np.ndarray, seqlength: Optional[int] = None):
    """Best path through one FSM.

    The forward max-plus recursion is alpha-recursion (src/inference.jl:62-74)
    with K = TropicalSemiring over the expanded emissions (src/inference.jl:54-60).
    The reference has no ``bestpath`` at this commit (src/MarkovModels.jl:56-57;
    historical signature examples/demo.ipynb cell 23).  Our specification:
      bp[n, j] = the lowest source index i maximising T_hat[i, j] + A[i, n-1]
                 over the stored arcs i -> j; -1 if that maximum is -inf;
      the path is read back from the phony final state at frame len+1.
    Returns (path[len] 0-based states, score, A, bp)."""
    K = TROPICAL
    P, N = lhs.shape
    L = N if seqlength is None else seqlength
    Vh = expand(lhs, L, K)
    s2p = np.asarray(list(state2pdf) + [numpdf])
    em = Vh[s2p, :]  # C_hat * V_hat (src/inference.jl:150)
    S1 = fsm1.alpha_hat.shape[0]
    T = fsm1.T_hat
    cols = T.colidx()
    A = np.full((S1, N + 1), -np.inf, dtype=lhs.dtype)
    bp = np.full((N + 1, S1), -1, dtype=np.int32)
    A[:, 0] = fsm1.alpha_hat.astype(lhs.dtype) + em[:, 0]
    for n in range(1, N + 1):
        best = np.full(S1, -np.inf, dtype=lhs.dtype)
        arg = np.full(S1, -1, dtype=np.int32)
        # CSC(T_hat): column j holds its in-arcs with ascending source index, so a
        # strict '>' keeps the lowest source among ties
        for k in range(T.nnz):
            i, j = T.rowval[k], cols[k]
            v = T.nzval[k].astype(lhs.dtype) + A[i, n - 1]
            if v > best[j]:
                best[j] = v
                arg[j] = i
        with np.errstate(invalid="ignore"):
            A[:, n] = best + em[:, n]
        bp[n] = arg
    f = S1 - 1
    score = A[f, L]
    path = np.full(L, -1, dtype=np.int32)
    if np.isfinite(score):
        s = f
        for n in range(L, 0, -1):
            s = bp[n, s]
            path[n - 1] = s
    return path, score, A, bp


# ----------------------------------------------------------------------------
# OpenFst-text graphs (format written by misc/benchmark/generatefsm.jl:42-57)
# ----------------------------------------------------------------------------


def parse_openfst_text(text: str):
    """Lines ``0 i pdf pdf -log(pi_i)`` (initial), ``i j pdf pdf -log(T_ij)``
    (arc, pdf id of the destination j), ``i -log(omega_i)`` (final); states
    1-based, state 0 is the super-initial state.  Returns 0-based
    (S, init[(s,w)], arcs[((i,j),w)], final[(s,w)], state2pdf[S] 0-based, P)."""
    init, arcs, final = [], [], []
    pdf = {}
    S = 0
    for line in text.splitlines():
        t = line.split()
        if not t:
            continue
        if len(t) == 2:
            i = int(t[0])
            final.append((i - 1, -float(t[1])))
            S = max(S, i)
        else:
            i, j, p, w = int(t[0]), int(t[1]), int(t[2]), -float(t[4])
            pdf[j - 1] = p - 1
            S = max(S, i, j)
            if i == 0:
                init.append((j - 1, w))
            else:
                arcs.append(((i - 1, j - 1), w))
    s2p = np.zeros(S, dtype=np.int32)
    for s, p in pdf.items():
        s2p[s] = p
    return S, init, arcs, final, s2p, int(max(pdf.values())) + 1
