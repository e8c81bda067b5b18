"""ctypes wrapper around oracle/libmm_oracle.so (the plain-C oracle).

TEST INFRASTRUCTURE ONLY -- see the header of mm_oracle.c / mm_oracle.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libmm_oracle.so")
    src = os.path.join(_HERE, "mm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmm_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _graph_args(fsm, dtype):
    """CSC of T_hat and of its transpose, from an oracle FSM (mm_oracle.FSM)."""
    T = fsm.T_hat
    Tt = T.transpose()
    ct = C.c_float if dtype == np.float32 else C.c_double
    arrs = dict(
        Tc=np.ascontiguousarray(T.colptr, dtype=np.int64),
        Tr=np.ascontiguousarray(T.rowval, dtype=np.int64),
        Tv=np.ascontiguousarray(T.nzval, dtype=dtype),
        Ttc=np.ascontiguousarray(Tt.colptr, dtype=np.int64),
        Ttr=np.ascontiguousarray(Tt.rowval, dtype=np.int64),
        Ttv=np.ascontiguousarray(Tt.nzval, dtype=dtype),
        a=np.ascontiguousarray(fsm.alpha_hat, dtype=dtype),
    )
    return arrs, ct


def batch_shared(fsm, state2pdf, P, lhs, lens=None, dtype=np.float32, nthreads=1, sr=0):
    """pdfposteriors for B utterances sharing ``fsm``.  lhs: [B, N, P].
    Returns (gamma [B, N, P], ttl [B])."""
    lhs = np.ascontiguousarray(lhs, dtype=dtype)
    B, N, P_ = lhs.shape
    assert P_ == P
    g, ct = _graph_args(fsm, dtype)
    S1 = fsm.alpha_hat.shape[0]
    s2p = np.ascontiguousarray(list(state2pdf) + [P], dtype=np.int32)
    gamma = np.zeros((B, N, P), dtype=dtype)
    ttl = np.zeros(B, dtype=dtype)
    lens_a = None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    fn = getattr(lib(), "mmo_batch_shared_f32" if dtype == np.float32 else "mmo_batch_shared_f64")
    fn.restype = C.c_int
    rc = fn(
        C.c_int(sr), C.c_int64(B), C.c_int64(S1), C.c_int64(P), C.c_int64(N),
        _p(g["Tc"], C.c_int64), _p(g["Tr"], C.c_int64), _p(g["Tv"], ct),
        _p(g["Ttc"], C.c_int64), _p(g["Ttr"], C.c_int64), _p(g["Ttv"], ct),
        _p(g["a"], ct), _p(s2p, C.c_int32), _p(lhs, ct), _p(lens_a, C.c_int32),
        _p(gamma, ct), _p(ttl, ct), C.c_int(nthreads),
    )
    if rc:
        raise RuntimeError("oracle failed")
    return gamma, ttl


def warm(S1, P, N, nthreads, dtype=np.float32):
    """Allocate + first-touch the per-thread scratch of the C oracle (outside a timed region)."""
    fn = getattr(lib(), "mmo_warm_f32" if dtype == np.float32 else "mmo_warm_f64")
    fn.restype = C.c_int
    if fn(C.c_int64(S1), C.c_int64(P), C.c_int64(N), C.c_int(nthreads)):
        raise RuntimeError("oracle: out of memory")


def single(fsm, state2pdf, P, Vhat, dtype=np.float64, sr=0, want_ab=False):
    """One utterance with already expanded emissions Vhat [(P+1), (N+1)] as in
    the NumPy oracle.  Returns gamma [P, N], ttl (and alpha, beta [S1, N1])."""
    P1, N1 = Vhat.shape
    assert P1 == P + 1
    Vf = np.ascontiguousarray(Vhat.T, dtype=dtype)  # [N1][P1]
    g, ct = _graph_args(fsm, dtype)
    S1 = fsm.alpha_hat.shape[0]
    s2p = np.ascontiguousarray(list(state2pdf) + [P], dtype=np.int32)
    gamma = np.zeros((N1 - 1, P), dtype=dtype)
    ttl = np.zeros(1, dtype=dtype)
    A = np.zeros((N1, S1), dtype=dtype) if want_ab else None
    Bm = np.zeros((N1, S1), dtype=dtype) if want_ab else None
    fn = getattr(lib(), "mmo_pdfposteriors_f32" if dtype == np.float32 else "mmo_pdfposteriors_f64")
    fn.restype = C.c_int
    rc = fn(
        C.c_int(sr), C.c_int64(S1), C.c_int64(P1), C.c_int64(N1),
        _p(g["Tc"], C.c_int64), _p(g["Tr"], C.c_int64), _p(g["Tv"], ct),
        _p(g["Ttc"], C.c_int64), _p(g["Ttr"], C.c_int64), _p(g["Ttv"], ct),
        _p(g["a"], ct), _p(s2p, C.c_int32), _p(Vf, ct), _p(gamma, ct), _p(ttl, ct),
        _p(A, ct), _p(Bm, ct),
    )
    if rc:
        raise RuntimeError("oracle failed")
    if want_ab:
        return gamma.T.copy(), ttl[0], A.T.copy(), Bm.T.copy()
    return gamma.T.copy(), ttl[0]


def viterbi(fsm, state2pdf, P, lhs, length=None, dtype=np.float32):
    """lhs [N, P].  Returns (path [N] 0-based, -1 beyond len; score; bp [N+1, S1])."""
    lhs = np.ascontiguousarray(lhs, dtype=dtype)
    N = lhs.shape[0]
    L = N if length is None else int(length)
    g, ct = _graph_args(fsm, dtype)
    S1 = fsm.alpha_hat.shape[0]
    s2p = np.ascontiguousarray(list(state2pdf) + [P], dtype=np.int32)
    bp = np.zeros((N + 1, S1), dtype=np.int32)
    path = np.zeros(N, dtype=np.int32)
    score = np.full(1, -np.inf, dtype=dtype)
    fn = getattr(lib(), "mmo_viterbi_f32" if dtype == np.float32 else "mmo_viterbi_f64")
    fn.restype = C.c_int
    fn(
        C.c_int64(S1), C.c_int64(P), C.c_int64(N), C.c_int64(L),
        _p(g["Tc"], C.c_int64), _p(g["Tr"], C.c_int64), _p(g["Tv"], ct),
        _p(g["a"], ct), _p(s2p, C.c_int32), _p(lhs, ct), _p(bp, C.c_int32), _p(path, C.c_int32), _p(score, ct),
    )
    return path, score[0], bp
