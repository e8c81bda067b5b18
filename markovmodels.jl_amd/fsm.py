"""Host-side FSM data model: the part of src/fsm.jl / src/fsmops.jl the inference
hot path consumes (the FSM type, its constructors, rawunion, the state map).

States, pdfs and labels are 0-based here (Python convention); the JSON and
OpenFst-text readers convert from the reference's 1-based files.
"""
from __future__ import annotations

import json
import re
from typing import List, Optional, Sequence, Tuple

import numpy as np

_ZERO = {"log": -np.inf, "tropical": -np.inf, "prob": 0.0}
_ONE = {"log": 0.0, "tropical": 0.0, "prob": 1.0}


def _oplus_at(semiring: str, out: np.ndarray, idx: np.ndarray, vals: np.ndarray) -> None:
    """out[idx] (+)= vals, the combine Julia's sparse()/sparsevec() apply to duplicates."""
    if semiring == "log":
        with np.errstate(invalid="ignore", divide="ignore"):
            np.logaddexp.at(out, idx, vals)
    elif semiring == "prob":
        np.add.at(out, idx, vals)
    else:
        np.maximum.at(out, idx, vals)


def semiring_name(s: str) -> str:
    """Accepts "log"/"tropical" or a Semirings.jl type name such as
    "LogSemiring{Float32}" (the JSON ``semiring`` field, src/fsm.jl:75)."""
    t = s.lower()
    if "tropical" in t:
        return "tropical"
    if "log" in t:
        return "log"
    if "prob" in t:
        return "prob"
    raise ValueError(f"unsupported semiring {s!r}: the engine implements LogSemiring, TropicalSemiring and ProbSemiring")


class FSM:
    """FSM{K,L} (src/fsm.jl:7-17): ``alpha_hat`` (sparse vector of S+1 initial
    weights, the last -- the phony final state -- being zero), ``T_hat`` (the
    (S+1)x(S+1) transition matrix [T omega; 0 one], CSC like
    SparseMatrixCSC: column j holds the arcs INTO j) and the labels.

    ``FSM(initws, arcs, finalws, labels)`` is the arc-list constructor
    (src/fsm.jl:50-71) followed by the inner one (:19-28):
    ``initws``/``finalws`` = [(state, weight)], ``arcs`` = [((src, dst), weight)].
    """

    def __init__(self, initws, arcs, finalws, labels, semiring: str = "log", dtype=np.float32):
        self.semiring = semiring_name(semiring)
        self.labels = list(labels)
        S = len(self.labels)
        self.dtype = np.dtype(dtype)
        zero = _ZERO[self.semiring]
        a = np.full(S + 1, zero, dtype=self.dtype)
        if len(initws):
            _oplus_at(self.semiring, a, np.asarray([s for s, _ in initws], dtype=np.int64),
                      np.asarray([w for _, w in initws], dtype=self.dtype))
        self.alpha_idx = np.flatnonzero(a != zero).astype(np.int64)
        self.alpha_val = a[self.alpha_idx]
        I = [s for (s, _), _ in arcs] + [s for s, _ in finalws] + [S]
        J = [d for (_, d), _ in arcs] + [S] * len(finalws) + [S]
        V = [w for _, w in arcs] + [w for _, w in finalws] + [_ONE[self.semiring]]
        self._set_coo(np.asarray(I, dtype=np.int64), np.asarray(J, dtype=np.int64), np.asarray(V, dtype=self.dtype), S + 1)
        self._parts: Optional[List["FSM"]] = None

    # -- construction helpers -------------------------------------------------
    def _set_coo(self, I, J, V, S1):
        if I.size and (I.min() < 0 or I.max() >= S1 or J.min() < 0 or J.max() >= S1):
            raise IndexError("state index out of range")
        order = np.lexsort((I, J))
        I, J, V = I[order], J[order], V[order]
        if I.size:
            first = np.ones(I.size, dtype=bool)
            first[1:] = (I[1:] != I[:-1]) | (J[1:] != J[:-1])
            if not first.all():
                grp = np.cumsum(first) - 1
                Vc = np.full(int(first.sum()), _ZERO[self.semiring], dtype=V.dtype)
                _oplus_at(self.semiring, Vc, grp, V)
                I, J, V = I[first], J[first], Vc
        colptr = np.zeros(S1 + 1, dtype=np.int64)
        np.add.at(colptr, J + 1, 1)
        self.colptr = np.cumsum(colptr)
        self.rowval = np.ascontiguousarray(I)
        self.nzval = np.ascontiguousarray(V)

    @classmethod
    def from_fields(cls, alpha_idx, alpha_val, colptr, rowval, nzval, labels, semiring="log") -> "FSM":
        """Directly from the struct fields (alpha_hat as (index, value) pairs, T_hat as CSC)."""
        self = cls.__new__(cls)
        self.semiring = semiring_name(semiring)
        self.labels = list(labels)
        self.nzval = np.ascontiguousarray(nzval)
        self.dtype = self.nzval.dtype
        self.alpha_idx = np.ascontiguousarray(alpha_idx, dtype=np.int64)
        self.alpha_val = np.ascontiguousarray(alpha_val, dtype=self.dtype)
        self.colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        self.rowval = np.ascontiguousarray(rowval, dtype=np.int64)
        self._parts = None
        return self

    @classmethod
    def from_json(cls, s: str, dtype=np.float32) -> "FSM":
        """FSM(::AbstractString) (src/fsm.jl:73-82): {"semiring", "initstates",
        "arcs", "finalstates", "labels"}, states 1-based in the file."""
        d = json.loads(s)
        return cls(
            [(a - 1, b) for a, b in d["initstates"]],
            [((a - 1, b - 1), c) for a, b, c in d["arcs"]],
            [(a - 1, b) for a, b in d["finalstates"]],
            d["labels"],
            semiring=d["semiring"],
            dtype=dtype,
        )

    @classmethod
    def from_openfst_text(cls, text: str, semiring="log", dtype=np.float32) -> Tuple["FSM", np.ndarray, int]:
        """Graphs in the text form misc/benchmark/generatefsm.jl:42-57 writes
        (den_fsm_wsj.txt / num_fsm_wsj.txt): ``0 i pdf pdf -log(pi)``,
        ``i j pdf pdf -log(T_ij)`` (pdf of the destination), ``i -log(omega)``.
        Returns (fsm, state2pdf[S] 0-based, number of pdfs)."""
        init, arcs, final, pdf, S = [], [], [], {}, 0

        def num(tok: str) -> float:  # (Julia prints a Float32 with an exponent as 1.0f-5, infinities as Inf / -Inf)
            return float(re.sub(r"(?<=\d)f(?=[-+]?\d)", "e", tok))

        for line in text.splitlines():
            t = line.split()
            if not t:
                continue
            if len(t) <= 2:
                i = int(t[0])
                final.append((i - 1, -num(t[1]) if len(t) == 2 else 0.0))
                S = max(S, i)
                continue
            i, j, p, w = int(t[0]), int(t[1]), int(t[2]), -num(t[4]) if len(t) > 4 else 0.0
            pdf[j - 1] = p - 1
            S = max(S, i, j)
            if i == 0:
                init.append((j - 1, w))
            else:
                arcs.append(((i - 1, j - 1), w))
        s2p = np.zeros(S, dtype=np.int32)
        for s, p in pdf.items():
            s2p[s] = p
        fsm = cls(init, arcs, final, list(range(S)), semiring=semiring, dtype=dtype)
        return fsm, s2p, int(max(pdf.values())) + 1

    def arc_lists(self):
        """(init_idx, init_w, src, dst, w, final_idx, final_w) of the real states, 0-based, in the order findnz gives
        them in the reference (initial states ascending; arcs by destination, then source; final states ascending):
        alpha_hat[1:end-1], T_hat[1:end-1, 1:end-1] and omega = T_hat[1:end-1, end] (src/fsm.jl:19-28)."""
        S = self.S1 - 1
        keep = self.alpha_idx < S
        dst_all = np.repeat(np.arange(self.S1, dtype=np.int64), np.diff(self.colptr))
        real = (dst_all < S) & (self.rowval < S)
        fin = (dst_all == S) & (self.rowval < S)
        return (self.alpha_idx[keep], self.alpha_val[keep], self.rowval[real], dst_all[real], self.nzval[real],
                self.rowval[fin], self.nzval[fin])

    def to_openfst_text(self, state2pdf) -> str:
        """The text form misc/benchmark/generatefsm.jl:42-57 writes (and from_openfst_text reads): one line per
        initial state ``0 i pdf pdf -w``, per arc ``i j pdf pdf -w`` (pdf of the destination) and per final state
        ``i -w``; states and pdfs 1-based, weights as negated natural logs in their shortest float32 form."""
        s2p = np.asarray(state2pdf)
        ii, iw, src, dst, w, fi, fw = self.arc_lists()

        def num(x) -> str:
            # numpy prints the shortest digits that give the value back, like Julia; the spellings that differ are mapped to
            # Julia's: a Float32 with an exponent is 1.0f-5 (numpy: 1e-05), infinities are Inf / -Inf
            v = self.dtype.type(-x)
            if not np.isfinite(v):
                return "NaN" if np.isnan(v) else ("Inf" if v > 0 else "-Inf")
            s = str(v)
            if "e" not in s and abs(v) >= 1e6:  # (Julia switches to the exponent form at 1e6, numpy at 1e16)
                s = np.format_float_scientific(v, unique=True, trim="0")
            if "e" in s:
                mant, ex = s.split("e")
                if "." not in mant:
                    mant += ".0"
                s = mant + ("f" if self.dtype == np.float32 else "e") + str(int(ex))
            return s

        out = []
        for i, v in zip(ii, iw):
            p = int(s2p[i]) + 1
            out.append(f"0 {i + 1} {p} {p} {num(v)}")
        for i, j, v in zip(src, dst, w):
            p = int(s2p[j]) + 1
            out.append(f"{i + 1} {j + 1} {p} {p} {num(v)}")
        for i, v in zip(fi, fw):
            out.append(f"{i + 1} {num(v)}")
        return "\n".join(out) + "\n"

    # -- accessors ---------------------------------------------------------------
    @property
    def S1(self) -> int:
        return int(self.colptr.shape[0] - 1)

    @property
    def nnz(self) -> int:
        return int(self.rowval.shape[0])

    def alpha_hat_dense(self) -> np.ndarray:
        a = np.full(self.S1, _ZERO[self.semiring], dtype=self.dtype)
        a[self.alpha_idx] = self.alpha_val
        return a

    def T_hat_dense(self) -> np.ndarray:
        d = np.full((self.S1, self.S1), _ZERO[self.semiring], dtype=self.dtype)
        cols = np.repeat(np.arange(self.S1), np.diff(self.colptr))
        d[self.rowval, cols] = self.nzval
        return d

    def with_semiring(self, semiring: str) -> "FSM":
        """Same storage under another semiring (convert(MatrixFSM{TropicalSemiring}, .) in
        test/test_algorithms.jl:280)."""
        f = FSM.from_fields(self.alpha_idx, self.alpha_val, self.colptr, self.rowval, self.nzval, self.labels, semiring)
        return f


def nstates(fsm: FSM) -> int:
    """nstates(fsm) (src/fsm.jl:84) = S, without the phony final state."""
    return fsm.S1 - 1


def rawunion(*fsms: FSM) -> FSM:
    """rawunion(fsms...) (src/fsmops.jl:28-36): vcat of the alpha_hat's and
    blockdiag of the T_hat's -- several independent FSMs (each keeping its own
    phony final state) packed in one structure."""
    if not fsms:
        raise ValueError("rawunion of nothing")
    if any(f.semiring != fsms[0].semiring for f in fsms):
        raise TypeError("rawunion: FSMs must share the semiring")
    parts: List[FSM] = []
    for f in fsms:
        parts.extend(f._parts if f._parts is not None else [f])
    off_s = np.cumsum([0] + [f.S1 for f in parts])
    off_z = np.cumsum([0] + [f.nnz for f in parts])
    u = FSM.from_fields(
        np.concatenate([f.alpha_idx + o for f, o in zip(parts, off_s)]),
        np.concatenate([f.alpha_val for f in parts]),
        np.concatenate([np.zeros(1, dtype=np.int64)] + [f.colptr[1:] + o for f, o in zip(parts, off_z)]),
        np.concatenate([f.rowval + o for f, o in zip(parts, off_s)]),
        np.concatenate([f.nzval for f in parts]),
        sum((f.labels for f in parts), []),
        parts[0].semiring,
    )
    u._parts = parts
    return u


def split_blocks(fsm: FSM, sizes: Sequence[int]) -> List[FSM]:
    """Inverse of rawunion for an FSM built elsewhere: cut the block-diagonal
    system into blocks of the given numbers of (extended) states."""
    if fsm._parts is not None and [p.S1 for p in fsm._parts] == list(sizes):
        return fsm._parts
    if sum(sizes) != fsm.S1:
        from ._lib import DimensionMismatch

        raise DimensionMismatch(-2, f"state maps cover {sum(sizes)} states, the FSM has {fsm.S1}")
    out, s0 = [], 0
    a_dense_idx = fsm.alpha_idx
    for n in sizes:
        cp = fsm.colptr[s0 : s0 + n + 1]
        rv = fsm.rowval[cp[0] : cp[-1]] - s0
        if rv.size and (rv.min() < 0 or rv.max() >= n):
            raise ValueError("the FSM is not block diagonal with the given block sizes")
        m = (a_dense_idx >= s0) & (a_dense_idx < s0 + n)
        out.append(FSM.from_fields(a_dense_idx[m] - s0, fsm.alpha_val[m], cp - cp[0], rv, fsm.nzval[cp[0] : cp[-1]],
                                   fsm.labels[s0 : s0 + n - 1] if len(fsm.labels) >= s0 + n - 1 else [], fsm.semiring))
        s0 += n
    return out


class StateMap:
    """C_hat (examples/prepare-lfmmi-graphs.jl:15-23): the (S+1)x(P+1) sparse
    matrix with exactly one ``one(K)`` per row, kept as the column index of that
    entry.  ``state2pdf``: S entries (0-based pdf of each real state); the phony
    final state maps to the phony pdf P."""

    def __init__(self, state2pdf, numpdf: int):
        s2p = np.asarray(state2pdf, dtype=np.int64).ravel()
        if s2p.size and (s2p.min() < 0 or s2p.max() >= numpdf):
            raise IndexError("pdf index out of range")
        self.numpdf = int(numpdf)
        self.state2pdf = np.concatenate([s2p, [numpdf]]).astype(np.int32)

    @property
    def shape(self):
        return (self.state2pdf.shape[0], self.numpdf + 1)

    @classmethod
    def from_matrix(cls, C) -> "StateMap":
        """From an explicit (S+1)x(P+1) matrix (dense ndarray whose non-zero(K)
        entries mark the map, or a scipy.sparse matrix)."""
        if hasattr(C, "tocsr"):
            m = C.tocsr()
            if (np.diff(m.indptr) != 1).any():
                raise ValueError("C_hat must have exactly one stored entry per row")
            cols = m.indices
        else:
            M = np.asarray(C)
            nz = np.isfinite(M) & (M == 0) if np.isneginf(M).any() else (M != 0)
            if (nz.sum(axis=1) != 1).any():
                raise ValueError("C_hat must have exactly one entry per row")
            cols = nz.argmax(axis=1)
        P1 = C.shape[1]
        if cols[-1] != P1 - 1:
            raise ValueError("the final state must map to the last (phony) pdf")
        return cls(cols[:-1], P1 - 1)


def statemap(state2pdf, numpdf: int) -> StateMap:
    return StateMap(state2pdf, numpdf)


class GeneralStateMap:
    """Any sparse C_hat ((S+1) x (P+1), semiring values; src/inference.jl:145-150 takes whatever the caller built): kept
    as CSR.  Only the generic entry (``pdfposteriors`` -> mm_pdfposteriors_ex) accepts it; the fast kernels need the
    one-hot ``StateMap``."""

    def __init__(self, matrix, semiring: str = "log"):
        self.semiring = semiring_name(semiring)
        zero = _ZERO[self.semiring]
        if hasattr(matrix, "tocsr"):
            m = matrix.tocsr()
            self.shape = tuple(m.shape)
            self.indptr, self.indices, self.data = m.indptr.astype(np.int64), m.indices.astype(np.int64), np.asarray(m.data, dtype=np.float64)
        else:
            M = np.asarray(matrix, dtype=np.float64)
            self.shape = M.shape
            nz = M != zero
            self.indptr = np.concatenate([[0], np.cumsum(nz.sum(axis=1))]).astype(np.int64)
            self.indices = np.nonzero(nz)[1].astype(np.int64)
            self.data = M[nz]
        self.numpdf = self.shape[1] - 1

    def one_hot(self):
        """The equivalent StateMap if every row holds exactly one entry of weight one(K) (and the last row maps to the
        last pdf), else None."""
        if (np.diff(self.indptr) != 1).any() or (self.data != _ONE[self.semiring]).any() or self.indices[-1] != self.numpdf:
            return None
        if (self.indices[:-1] >= self.numpdf).any():  # (a real state on the phony pdf: not what StateMap describes)
            return None
        return StateMap(self.indices[:-1], self.numpdf)
