"""Host-side mirror of src/inference.jl: CompiledFSM / compile / batch / expand /
alpha-recursion / beta-recursion / pdfposteriors, plus bestpath
(docs/src/inference.md:4-6).  Every compute entry point is ONE call into the
HIP engine through the C ABI (include/markovmodels_amd.h); there is no CPU
path here.

Two ways in:
  * the reference call shapes -- ``pdfposteriors(fsm, Vhats, Chats)`` with
    ``Vhats`` made by ``expand`` (src/inference.jl:145-161), or the
    prepare-once variant ``pdfposteriors(batch(cfsm...), Vhats)``
    (``pdfposteriors2``, :164-180);
  * the native one the bench uses -- ``BatchedFSM.pdfposteriors(V, lens)`` on
    device-resident ``V[B, N, P]`` log-likelihoods and lengths, no host round
    trips.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict  # noqa: F401
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib
from ._lib import SEMIRING_ID, check, lib
from .fsm import FSM, GeneralStateMap, StateMap, split_blocks, statemap

_ZERO = -np.inf
_SEM_ZERO = {"log": -np.inf, "tropical": -np.inf, "prob": 0.0}
_SEM_ONE = {"log": 0.0, "tropical": 0.0, "prob": 1.0}


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("markovmodels_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch


class CompiledFSM:
    """CompiledFSM{K} (src/inference.jl:3-12): alpha_hat, T_hat, T_hat', C_hat, C_hat'
    prepared once -- here as the packed device forms the kernels stream."""

    def __init__(self, fsm: FSM, C_hat: Union[StateMap, np.ndarray]):
        if not isinstance(C_hat, StateMap):
            C_hat = StateMap.from_matrix(C_hat)
        if C_hat.shape[0] != fsm.S1:
            raise _lib.DimensionMismatch(-2, f"C_hat has {C_hat.shape[0]} rows, the FSM {fsm.S1} states")
        self.fsm = fsm
        self.C_hat = C_hat
        self.semiring = fsm.semiring
        self.S1 = fsm.S1
        self.P1 = C_hat.numpdf + 1
        # K of FSM{K}: the precision of the computation follows the FSM's, like the reference (src/inference.jl:147 converts
        # V_hat to K)
        self.dtype = np.dtype(np.float64) if np.asarray(fsm.nzval).dtype == np.float64 else np.dtype(np.float32)
        colptr = np.ascontiguousarray(fsm.colptr, dtype=np.int64)
        rowval = np.ascontiguousarray(fsm.rowval, dtype=np.int64)
        nzval = np.ascontiguousarray(fsm.nzval, dtype=np.float32 if fsm.nzval.dtype != np.float64 else np.float64)
        aidx = np.ascontiguousarray(fsm.alpha_idx, dtype=np.int64)
        aval = np.ascontiguousarray(fsm.alpha_val, dtype=nzval.dtype)
        s2p = np.ascontiguousarray(C_hat.state2pdf, dtype=np.int32)
        h = C.c_void_p()
        check(lib.mm_fsm_create(SEMIRING_ID[self.semiring], self.S1, fsm.nnz, _lib.MM_CSC, 8, 0, nzval.dtype.itemsize,
                                colptr.ctypes.data, rowval.ctypes.data, nzval.ctypes.data, aidx.shape[0],
                                aidx.ctypes.data, aval.ctypes.data, s2p.ctypes.data, self.P1, C.byref(h)))
        self._h = h

    @classmethod
    def _from_handle(cls, fsm: FSM, C_hat: StateMap, h) -> "CompiledFSM":
        """Around a handle the library has made already (compile_many)."""
        self = cls.__new__(cls)
        self.fsm = fsm
        self.C_hat = C_hat
        self.semiring = fsm.semiring
        self.S1 = fsm.S1
        self.P1 = C_hat.numpdf + 1
        self.dtype = np.dtype(np.float64) if np.asarray(fsm.nzval).dtype == np.float64 else np.dtype(np.float32)
        self._h = h
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:  # lib is None while the interpreter shuts down
            lib.mm_fsm_destroy(h)
            self._h = None

    def info(self) -> dict:
        S1, nnz, P1 = C.c_int64(), C.c_int64(), C.c_int32()
        slots, items = (C.c_int64 * 2)(), (C.c_int64 * 2)()
        check(lib.mm_fsm_info(self._h, C.byref(S1), C.byref(nnz), C.byref(P1), slots, items))
        return dict(S1=S1.value, nnz=nnz.value, P1=P1.value, packed_slots=list(slots), packed_items=list(items))

    def packed_product(self, x: np.ndarray, direction: int = 0):
        """Host evaluation of one semiring product through the packed form (test aid)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(self.S1, dtype=np.float32)
        arg = np.empty(self.S1, dtype=np.int32)
        check(lib.mm_debug_packed_product(self._h, direction, x.ctypes.data, out.ctypes.data, arg.ctypes.data))
        return out, arg


    def reach_distance(self, direction: int = 0) -> np.ndarray:
        """Fewest arcs from an initial state (direction 0) / to the phony final state (direction 1) for every
        state of the extended FSM, -1 = unreachable: the static dead-row bound of the fast kernels (host only)."""
        out = np.empty(self.S1, dtype=np.int32)
        check(lib.mm_debug_reach_distance(self._h, direction, out.ctypes.data))
        return out

    def stream_product(self, x: np.ndarray, direction: int = 0, H: int = 1):
        """Host evaluation of the product through the stream form of the stream kernels, for teams of H = 1, 2 or 4 workgroups (test
        aid).  Returns (out, stats = [arc slots per lane over all waves and sets, segments, real arcs / arc slots, slots of the most
        loaded wave])."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(self.S1, dtype=np.float32)
        stats = np.zeros(4, dtype=np.float64)
        check(lib.mm_debug_stream_team_product(self._h, int(H), direction, x.ctypes.data, out.ctypes.data, stats.ctypes.data))
        return out, stats

    def quad_product(self, x: np.ndarray, direction: int = 0, KQ: int = 5):
        """Host evaluation of the same product through the quad form of the fast kernel (test aid).
        Returns (out, stats = [quads, lanes, LDS cycles/gather naive, after placement])."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(self.S1, dtype=np.float32)
        stats = np.zeros(4, dtype=np.float64)
        check(lib.mm_debug_quad_product(self._h, direction, KQ, x.ctypes.data, out.ctypes.data, stats.ctypes.data))
        return out, stats


    def row_product(self, x: np.ndarray, direction: int = 0, pair: bool = False, copies: int = 0, scrambled: bool = False, bank_opt: bool = False):
        """Host evaluation of the same product through the row-lane form of the row kernels, or its pair variant
        (test aid); copies / scrambled: the copies of the linear vector the placement may use.
        Returns (out, stats = [KA, compute waves, segments, arcs / arc slots, max wave cost, min wave cost,
        LDS cycles/gather naive, after placement])."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(self.S1, dtype=np.float32)
        stats = np.zeros(8, dtype=np.float64)
        flags = (1 if pair else 0) | (int(copies) << 1) | (8 if scrambled else 0) | (16 if bank_opt else 0)
        check(lib.mm_debug_row_product_ex(self._h, direction, flags, x.ctypes.data, out.ctypes.data, stats.ctypes.data))
        return out, stats

    def split_product(self, x: np.ndarray, direction: int = 0, H: int = 2):
        """Host evaluation of the product through the split pair forms (teams of H workgroups; test aid).  Returns
        (out, stats = [KA, positions of the team's vector, segments, arcs / arc slots, max / min wave cost, LDS
        cycles/gather naive, after placement])."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(self.S1, dtype=np.float32)
        stats = np.zeros(8, dtype=np.float64)
        check(lib.mm_debug_split_product(self._h, int(H), direction, x.ctypes.data, out.ctypes.data, stats.ctypes.data))
        return out, stats

    def wave_product(self, x: np.ndarray, direction: int = 0):
        """Host evaluation of the product through the wave form (one wave per direction, log domain; test aid).
        Returns (out, stats = [arc slots per lane, segments, arcs / arc slots, LDS cycles/gather])."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(self.S1, dtype=np.float32)
        stats = np.zeros(4, dtype=np.float64)
        check(lib.mm_debug_wave_product(self._h, direction, x.ctypes.data, out.ctypes.data, stats.ctypes.data))
        return out, stats


def compile(fsm: FSM, C_hat) -> CompiledFSM:  # noqa: A001 - the reference's name
    """compile(fsm, C_hat) (src/inference.jl:11-12)."""
    return CompiledFSM(fsm, C_hat)


def compile_many(fsms, C_hats, threads: int = 0):
    """`compile.(fsms, C_hats)` (the reference broadcasts compile over a mini-batch of numerator graphs,
    examples/test_cuda.jl:76-78) in ONE call of the library (mm_fsm_create_many): the graphs are compiled on host threads,
    small graphs get the forms of their kernel at once, and everything goes to the device as one allocation and one
    copy.  C_hats: one state map for all, or one per FSM.  The handles are what ``compile`` would have made."""
    fsms = list(fsms)
    n = len(fsms)
    maps = list(C_hats) if isinstance(C_hats, (list, tuple)) else [C_hats] * n
    if len(maps) != n:
        raise ValueError("compile_many: one C_hat per FSM (or one for all)")
    if n == 0:
        return []
    maps = [m if isinstance(m, StateMap) else StateMap.from_matrix(m) for m in maps]
    sem = fsms[0].semiring
    f64 = np.asarray(fsms[0].nzval).dtype == np.float64
    vdt = np.float64 if f64 else np.float32
    if any(f.semiring != sem or (np.asarray(f.nzval).dtype == np.float64) != f64 for f in fsms):
        raise TypeError("compile_many: the FSMs of one call share the semiring and the float type (FSM{K})")
    keep = []  # the arrays the pointers below refer to

    def arr(a, dt):
        a = np.asarray(a)
        if a.dtype != dt or not a.flags.c_contiguous:
            a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.__array_interface__["data"][0]

    S1 = np.empty(n, np.int64)
    nnz = np.empty(n, np.int64)
    ninit = np.empty(n, np.int64)
    P1 = np.empty(n, np.int32)
    ptrs = np.empty((7, n), np.uint64)  # colptr, rowval, nzval, alpha_idx, alpha_val, state2pdf per graph
    for i, (f, m) in enumerate(zip(fsms, maps)):
        if m.shape[0] != f.S1:
            raise _lib.DimensionMismatch(-2, f"C_hat {i} has {m.shape[0]} rows, the FSM {f.S1} states")
        S1[i], nnz[i], P1[i] = f.S1, f.nnz, m.numpdf + 1
        ptrs[0, i] = arr(f.colptr, np.int64)
        ptrs[1, i] = arr(f.rowval, np.int64)
        ptrs[2, i] = arr(f.nzval, vdt)
        ai = np.asarray(f.alpha_idx)
        ninit[i] = ai.shape[0]
        ptrs[3, i] = arr(ai, np.int64)
        ptrs[4, i] = arr(f.alpha_val, vdt)
        ptrs[5, i] = arr(m.state2pdf, np.int32)
    out = (C.c_void_p * n)()
    check(lib.mm_fsm_create_many(n, SEMIRING_ID[sem], _lib.MM_CSC, 8, 0, 8 if f64 else 4, S1.ctypes.data, nnz.ctypes.data,
                                 ptrs[0].ctypes.data, ptrs[1].ctypes.data, ptrs[2].ctypes.data, ninit.ctypes.data, ptrs[3].ctypes.data,
                                 ptrs[4].ctypes.data, ptrs[5].ctypes.data, P1.ctypes.data, int(threads), out))
    return [CompiledFSM._from_handle(f, m, C.c_void_p(out[i])) for i, (f, m) in enumerate(zip(fsms, maps))]


class BatchedFSM:
    """batch(cfsm...) (src/inference.jl:28-36): B independent compiled FSMs as
    one block-diagonal system.  Repeating one CompiledFSM B times shares its
    device storage (the denominator case)."""

    def __init__(self, cfsms: Sequence[CompiledFSM]):
        self.cfsms = list(cfsms)
        self.B = len(self.cfsms)
        arr = (C.c_void_p * self.B)(*[c._h for c in self.cfsms])
        h = C.c_void_p()
        check(lib.mm_batch_create(arr, self.B, C.byref(h)))
        self._h = h
        self.semiring = self.cfsms[0].semiring
        self.dtype = np.dtype(np.float64) if any(c.dtype == np.float64 for c in self.cfsms) else np.dtype(np.float32)
        self.P = self.cfsms[0].P1 - 1
        if any(c.P1 != self.P + 1 for c in self.cfsms):
            raise _lib.DimensionMismatch(-2, "all FSMs of a batch must share the number of pdfs")
        self.total_states = int(lib.mm_batch_total_states(h))
        self.state_offsets = np.cumsum([0] + [c.S1 for c in self.cfsms])

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:
            lib.mm_batch_destroy(h)
            self._h = None

    # -- helpers -----------------------------------------------------------------
    def _prep(self, V, lens):
        torch = _torch()
        as_numpy = not isinstance(V, torch.Tensor)
        Vt = torch.as_tensor(np.ascontiguousarray(V, dtype=np.float32)).cuda() if as_numpy else V
        if Vt.dtype != torch.float32 or not Vt.is_cuda:
            raise TypeError("V must be a float32 tensor on the HIP device")
        if Vt.dim() != 3 or Vt.shape[0] != self.B or Vt.shape[2] != self.P:
            raise _lib.DimensionMismatch(-2, f"V must be [B={self.B}, N, P={self.P}], got {tuple(Vt.shape)}")
        if Vt.stride(2) != 1:
            Vt = Vt.contiguous()
        lt = None
        if lens is not None:
            lt = torch.as_tensor(np.asarray(lens, dtype=np.int32)).cuda() if not isinstance(lens, torch.Tensor) else lens
            lt = lt.to(device=Vt.device, dtype=torch.int32).contiguous()
            if lt.numel() != self.B:
                raise _lib.DimensionMismatch(-2, "lens must have B entries")
        return torch, Vt, lt, as_numpy

    @staticmethod
    def _stream(torch):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # -- compute -------------------------------------------------------------------
    def pdfposteriors(self, V, lens=None, out=None):
        """One call of the engine: gamma[B, N, P] (probabilities), ttl[B]."""
        torch, Vt, lt, as_numpy = self._prep(V, lens)
        B, N, P = Vt.shape
        if out is not None:
            # the kernels write through raw pointers and strides: a wrong buffer is memory corruption, not an exception
            if not isinstance(out, torch.Tensor) or out.dtype != torch.float32 or out.device != Vt.device:
                raise TypeError("out must be a float32 tensor on V's device")
            if out.dim() != 3 or tuple(out.shape) != (B, N, P):
                raise _lib.DimensionMismatch(-2, f"out must be [B={B}, N={N}, P={P}], got {tuple(out.shape)}")
        gamma = out if out is not None else torch.empty((B, N, P), dtype=torch.float32, device=Vt.device)
        ttl = torch.empty(B, dtype=torch.float32, device=Vt.device)
        check(lib.mm_pdfposteriors_f32(self._h, Vt.data_ptr(), Vt.stride(0), Vt.stride(1),
                                       lt.data_ptr() if lt is not None else None, N, gamma.data_ptr(), gamma.stride(0),
                                       gamma.stride(1), gamma.stride(2), ttl.data_ptr(), self._stream(torch)))
        if as_numpy:
            return gamma.cpu().numpy(), ttl.cpu().numpy()
        return gamma, ttl

    def has_fast_entry(self) -> bool:
        """True if ``pdfposteriors(V, lens)`` -- mm_pdfposteriors_f32, the fast kernels -- serves this batch: Log batches always,
        ProbSemiring batches of Float32 FSMs (the library keeps their log-semiring twins: ``kernels()`` names the twins' kernels),
        Tropical batches never (their fast entry is ``viterbi``)."""
        if self.semiring == "log":
            return True
        if self.semiring != "prob" or self.dtype == np.float64:
            return False
        try:
            return bool(self.kernels("log"))
        except _lib.MarkovModelsAMDError:
            return False

    def reserve_ex(self, dtype, N1: int):
        """Size the generic entry's workspace for V_hat of N1 = N + 1 columns (mm_batch_reserve_ex): before capturing
        ``pdfposteriors_ex`` in a hipGraph."""
        check(lib.mm_batch_reserve_ex(self._h, np.dtype(dtype).itemsize, int(N1)))

    def pdfposteriors_ex(self, Vhat, maps=None, gamma=None, ttl=None):
        """The generic entry, device resident and asynchronous like the fast one (mm_pdfposteriors_ex): ``Vhat`` a
        float32 / float64 tensor [B, N1, P1] on the device (element (b, n, p) = V_hat_b[p, n]), ``maps`` None or a ctypes
        array of B mm_statemap_t handles the caller keeps alive until the stream has run the call.  Returns the tensors
        (gamma[B, N, P], ttl[B])."""
        torch = _torch()
        if not isinstance(Vhat, torch.Tensor) or not Vhat.is_cuda or Vhat.dtype not in (torch.float32, torch.float64):
            raise TypeError("Vhat must be a float32 / float64 tensor on the HIP device")
        if Vhat.dim() != 3 or Vhat.shape[0] != self.B or Vhat.stride(2) != 1:
            raise _lib.DimensionMismatch(-2, f"Vhat must be [B={self.B}, N + 1, P + 1] with the pdfs contiguous, got {tuple(Vhat.shape)}")
        _, N1, P1 = Vhat.shape
        if gamma is None:
            gamma = torch.zeros((self.B, N1 - 1, P1 - 1), dtype=Vhat.dtype, device=Vhat.device)
        if ttl is None:
            ttl = torch.zeros(self.B, dtype=Vhat.dtype, device=Vhat.device)
        if gamma.dtype != Vhat.dtype or ttl.dtype != Vhat.dtype or tuple(gamma.shape) != (self.B, N1 - 1, P1 - 1) or ttl.numel() != self.B:
            raise _lib.DimensionMismatch(-2, "gamma must be [B, N, P] and ttl [B] of V_hat's dtype")
        check(lib.mm_pdfposteriors_ex(self._h, maps, Vhat.element_size(), P1, Vhat.data_ptr(), Vhat.stride(0), Vhat.stride(1), N1,
                                      gamma.data_ptr(), gamma.stride(0), gamma.stride(1), gamma.stride(2), ttl.data_ptr(),
                                      self._stream(torch)))
        return gamma, ttl

    def pdfposteriors_generic(self, Vhats, Chats=None, dtype=None):
        """The generic entry (mm_pdfposteriors_ex): any semiring of the batch (log / tropical / prob), float32 or
        float64, any sparse state maps (``GeneralStateMap``; None: every FSM's own), any (P+1) x (N+1) matrices V_hat.
        Returns NumPy (gamma[B, P, N], ttl[B]) like the reference (src/inference.jl:145-161)."""
        import ctypes

        torch = _torch()
        Vh = [np.asarray(v.cpu() if hasattr(v, "cpu") else v) for v in Vhats]
        if len(Vh) != self.B or any(v.shape != Vh[0].shape for v in Vh):
            raise _lib.DimensionMismatch(-2, "need B matrices V_hat of one (P+1) x (N+1) shape")
        P1, N1 = Vh[0].shape
        # precision: the FSM's K unless asked for (src/inference.jl:147: copyto!(similar(V_hat, K), V_hat))
        dt = np.dtype(dtype) if dtype is not None else self.dtype
        if Chats is not None:  # raw matrices (dense / scipy.sparse) become GeneralStateMap; one-hot StateMap = the FSM's own
            Chats = [c if (c is None or isinstance(c, (StateMap, GeneralStateMap))) else GeneralStateMap(c, self.semiring) for c in Chats]
            if len(Chats) != self.B:
                raise _lib.DimensionMismatch(-2, "need one C_hat per utterance")
        for b in range(self.B):  # rows of V_hat against the pdfs of the map in force (src/inference.jl:146-150)
            want = self.cfsms[b].P1 if (Chats is None or Chats[b] is None or isinstance(Chats[b], StateMap)) else Chats[b].shape[1]
            if P1 != want:
                raise _lib.DimensionMismatch(-2, f"V_hat has {P1} rows, the state map of utterance {b} has {want} pdfs "
                                                 f"(P + 1: was expand() applied?)")
        V = torch.from_numpy(np.ascontiguousarray(np.stack([v.T for v in Vh]), dtype=dt)).cuda()  # [B][N1][P1]
        handles, keep = None, []
        try:
            if Chats is not None:
                arr = (ctypes.c_void_p * self.B)()
                cache = {}
                for b, c in enumerate(Chats):
                    if c is None or isinstance(c, StateMap):
                        arr[b] = None
                        continue
                    if c.shape != (self.cfsms[b].S1, P1):
                        raise _lib.DimensionMismatch(-2, f"C_hat {b} is {c.shape}, expected {(self.cfsms[b].S1, P1)}")
                    if id(c) not in cache:
                        hm = ctypes.c_void_p()
                        ip, ix, dv = (np.ascontiguousarray(c.indptr, dtype=np.int64), np.ascontiguousarray(c.indices, dtype=np.int64),
                                      np.ascontiguousarray(c.data, dtype=np.float64))
                        check(lib.mm_statemap_create(SEMIRING_ID[self.semiring], c.shape[0], c.shape[1], ix.shape[0], 8, 0, 8,
                                                     ip.ctypes.data, ix.ctypes.data, dv.ctypes.data, ctypes.byref(hm)))
                        cache[id(c)] = hm
                        keep.append(hm)
                    arr[b] = cache[id(c)]
                handles = arr
            gamma, ttl = self.pdfposteriors_ex(V, handles)
            return np.ascontiguousarray(gamma.cpu().numpy().transpose(0, 2, 1)), ttl.cpu().numpy()  # (waits for the kernel)
        finally:
            torch.cuda.current_stream().synchronize()  # the call is asynchronous: the maps must outlive it
            for hm in keep:
                lib.mm_statemap_destroy(hm)

    def _export(self, fn, V, lens):
        torch, Vt, lt, as_numpy = self._prep(V, lens)
        B, N, P = Vt.shape
        out = torch.empty((N + 1, self.total_states), dtype=torch.float32, device=Vt.device)
        check(fn(self._h, Vt.data_ptr(), Vt.stride(0), Vt.stride(1), lt.data_ptr() if lt is not None else None, N,
                 out.data_ptr(), out.stride(0), self._stream(torch)))
        out = out.t()  # (sum S1) x (N+1) like the reference's state_A / state_B
        return out.cpu().numpy() if as_numpy else out

    def maxstateposteriors(self, V, lens=None):
        """Max-marginals of the tropical semiring, (sum S1) x (N+1), computed on the device."""
        return self._export(lib.mm_maxstateposteriors_f32, V, lens)

    def alpharecursion(self, V, lens=None):
        return self._export(lib.mm_alpharecursion_f32, V, lens)

    def betarecursion(self, V, lens=None):
        return self._export(lib.mm_betarecursion_f32, V, lens)

    def reserve(self, N: int):
        """Size the internal workspace for runs of up to N frames now (so that later calls -- e.g. ones captured in
        a hipGraph -- never reallocate)."""
        check(lib.mm_batch_reserve(self._h, int(N)))

    def set_posterior_floor(self, floor: float = 1e-30):
        """mm_batch_set_posterior_floor: posteriors below `floor` may come out as 0 from the fast kernels (default 1e-30);
        a relaxed floor (1e-12) keeps sharp emissions on the fast path.  Returns self."""
        check(lib.mm_batch_set_posterior_floor(self._h, float(floor)))
        return self

    def set_deterministic(self, on: bool = True):
        """No float atomics in the general kernel: bit-identical gamma on every run (slower on small deep graphs)."""
        check(lib.mm_batch_set_deterministic(self._h, 1 if on else 0))
        return self

    def last_redo_count(self) -> int:
        """Utterances of the last pdfposteriors call that the fast kernels handed to the exact ones (0 = the whole
        batch ran on the fast path).  Synchronises the current stream."""
        import ctypes

        n = ctypes.c_int64(0)
        check(lib.mm_batch_last_redo_count(self._h, self._stream(_torch()), ctypes.byref(n)))
        return int(n.value)

    def last_fallback_count(self) -> int:
        """... and how many of those the float64 exact kernels handed on to the log-domain kernels (normally 0).
        Synchronises the current stream."""
        import ctypes

        n = ctypes.c_int64(0)
        check(lib.mm_batch_last_fallback_count(self._h, self._stream(_torch()), ctypes.byref(n)))
        return int(n.value)

    def set_exact_policy(self, policy: str = "auto"):
        """mm_batch_set_exact_policy: which linear-domain kernels a shared-graph batch starts with -- "auto" (float32 first,
        float64 first while the last finished call left utterances marked: depends on host / device timing for pipelined
        callers), "f32_first" or "f64_first" (both: the launches of a call are a function of the call alone, identical call
        sequences give identical bits).  Pins the path of ``alpharecursion`` / ``betarecursion`` as well ("auto": the item kernel
        first while the last export handed it more than half of the utterances; "f64_first": the item kernel alone)."""
        pol = {"auto": _lib.MM_EXACT_AUTO, "f32_first": _lib.MM_EXACT_F32_FIRST, "f64_first": _lib.MM_EXACT_F64_FIRST}[policy]
        check(lib.mm_batch_set_exact_policy(self._h, pol))
        return self

    def set_mark_policy(self, policy: str = "decide"):
        """mm_batch_set_mark_policy: "decide" (default: the finish kernel clears a range mark of the float32 kernels when its two
        criteria hold -- log gamma within 1e-4 relative above ~1e-24, an absolute error below ~1e-27 for smaller posteriors) or "keep"
        (a range mark always stays: the exact kernels compute the utterance -- the relative bar down to 1e-30)."""
        check(lib.mm_batch_set_mark_policy(self._h, {"decide": _lib.MM_MARKS_DECIDE, "keep": _lib.MM_MARKS_KEEP}[policy]))
        return self

    def set_gamma_mode(self, accumulate: bool = False, scale: float = 1.0):
        """mm_batch_set_gamma_mode: what pdfposteriors writes -- gamma_out = scale * gamma, or (accumulate) gamma_out += scale * gamma
        into the ``out`` tensor of the call (frames beyond the lengths left alone).  Batches of the wave kernel (numerator graphs)
        only: MarkovModelsAMDError(-4) otherwise.  The LF-MMI gradient gamma_den - gamma_num without a third pass (lfmmi.py)."""
        check(lib.mm_batch_set_gamma_mode(self._h, 1 if accumulate else 0, float(scale)))
        return self

    def last_exact_first(self) -> bool:
        """True if the last pdfposteriors call skipped the float32 kernels (the inputs of the call before were hard: the
        float64 exact kernels then run the whole batch at once)."""
        return bool(lib.mm_batch_last_exact_first(self._h))

    def team_xcd_stats(self):
        """(workgroups of the team kernels' launches since the last call of this method whose whole team sat on ONE XCD, all
        such workgroups): how the hardware placed the teams (a measurement aid; synchronises the device)."""
        out = np.zeros(2, dtype=np.int32)
        check(lib.mm_batch_team_xcd_stats(self._h, out.ctypes.data))
        return int(out[0]), int(out[1])

    def kernels(self, semiring: str = "log") -> str:
        """The kernels the engine launches for this batch (informational): "log" = pdfposteriors, "tropical" = bestpath, "export" =
        alpharecursion / betarecursion."""
        import ctypes

        buf = ctypes.create_string_buffer(512)
        check(lib.mm_batch_kernels(self._h, {"log": 0, "tropical": 1, "export": 3}[semiring], buf, 512))
        return buf.value.decode()

    def kernels_generic(self) -> str:
        """What the last call of the generic entry (mm_pdfposteriors_ex) launched for this batch (informational)."""
        import ctypes

        buf = ctypes.create_string_buffer(512)
        check(lib.mm_batch_kernels(self._h, 2, buf, 512))
        return buf.value.decode()

    def viterbi(self, V, lens=None, return_backpointers=False):
        """Best paths: (path[B, N] 0-based states, -1 beyond len; score[B][, bp[N+1, sum S1]])."""
        torch, Vt, lt, as_numpy = self._prep(V, lens)
        B, N, P = Vt.shape
        path = torch.empty((B, N), dtype=torch.int32, device=Vt.device)
        score = torch.empty(B, dtype=torch.float32, device=Vt.device)
        bp = torch.empty((N + 1, self.total_states), dtype=torch.int32, device=Vt.device) if return_backpointers else None
        check(lib.mm_viterbi_f32(self._h, Vt.data_ptr(), Vt.stride(0), Vt.stride(1),
                                 lt.data_ptr() if lt is not None else None, N, path.data_ptr(), path.stride(0),
                                 score.data_ptr(), bp.data_ptr() if bp is not None else None,
                                 bp.stride(0) if bp is not None else 0, self._stream(torch)))
        res = (path, score) + ((bp,) if bp is not None else ())
        if as_numpy:
            res = tuple(r.cpu().numpy() for r in res)
        return res


    def totalsum(self, n: int, cumulative: bool = False):
        """Per FSM of the batch: omega . v_n (totalsum) or the semiring sum over k <= n of omega . v_k
        (totalcumsum), v_1 = alpha, v_k = T' v_{k-1} (src/algorithms.jl:8-29).  float32 tensor [B]."""
        torch = _torch()
        out = torch.empty(self.B, dtype=torch.float32, device="cuda")
        check(lib.mm_totalsum_f32(self._h, int(n), int(bool(cumulative)), out.data_ptr(), self._stream(torch)))
        return out


def batch(*cfsms: CompiledFSM) -> BatchedFSM:
    """batch(fsm1, fsms...) (src/inference.jl:28-36)."""
    return BatchedFSM(cfsms)


def expand(lhs, seqlength: Optional[int] = None, semiring: str = "log"):
    """expand(V, seqlength) (src/inference.jl:54-60): the P x N likelihoods
    become (P+1) x (N+1): a phony pdf row (zero(K) up to seqlength, one(K) after) and
    an extra frame; real pdfs are zero(K) beyond seqlength.  zero = -inf, one = 0 for the
    Log/Tropical semirings, 0 and 1 for ProbSemiring."""
    a = np.asarray(lhs)
    P, N = a.shape
    L = N if seqlength is None else int(seqlength)
    out = np.full((P + 1, N + 1), _SEM_ZERO[semiring], dtype=a.dtype if a.dtype.kind == "f" else np.float32)
    out[:P, :L] = a[:, :L]
    out[P, L:] = _SEM_ONE[semiring]
    return out


def _unexpand(Vhats: Sequence[np.ndarray], semiring: str = "log"):
    """Recover (V[B, N, P], lens) from matrices made by ``expand`` (what the fast kernels take: they implement expand's
    semantics themselves); None if a matrix is not of that form -- the caller then takes the generic path.  The form
    (src/inference.jl:54-60): the phony row zero(K) up to the length and one(K) after, the real rows zero(K) beyond it --
    zero(K) / one(K) = -inf / 0 for the Log and Tropical semirings, 0 / 1 for ProbSemiring."""
    Vh = [np.asarray(v.cpu() if hasattr(v, "cpu") else v) for v in Vhats]
    shp = Vh[0].shape
    if any(v.shape != shp for v in Vh):
        raise _lib.DimensionMismatch(-2, "all V_hat must share one (P+1) x (N+1) shape")
    P1, N1 = shp
    zero, one = _SEM_ZERO[semiring], _SEM_ONE[semiring]
    lens = []
    for v in Vh:
        ph = v[P1 - 1]
        L = int(np.argmax(ph == one)) if (ph == one).any() else N1
        ok = np.all(ph[:L] == zero) and np.all(ph[L:] == one) and np.all(v[: P1 - 1, L:] == zero) and L <= N1 - 1
        if not ok:
            return None
        lens.append(L)
    V = np.stack([v[: P1 - 1, : N1 - 1].T for v in Vh]).astype(np.float32)
    if semiring != "prob":
        V[~np.isfinite(V) & (V < 0)] = -np.inf
    return np.ascontiguousarray(V), np.asarray(lens, dtype=np.int32)


def _need_expanded(Vhats, semiring: str = "log"):
    un = _unexpand(Vhats, semiring)
    if un is None:
        raise ValueError("V_hat is not of the form expand(V, seqlength) produces (only pdfposteriors takes arbitrary V_hat)")
    return un


# ---- compiled graphs of the reference-shaped entries: a process-wide LRU keyed by CONTENT.  `pdfposteriors(fsm::FSM, V_hats,
# C_hats)` (src/inference.jl:145-161) takes plain FSMs and transposes / re-batches them on every call (:148-151); here a graph
# is compiled (packed into the kernels' forms, uploaded) the first time its content is seen and found again by a 128-bit hash of
# its fields afterwards -- the denominator graph of every call, the numerator graphs of an utterance across epochs.  Misses of
# one call are compiled together (compile_many: host threads, one allocation, one copy).
_COMPILED_LRU: "OrderedDict[bytes, CompiledFSM]" = None  # type: ignore[assignment]
_COMPILED_LRU_MAX = 8192            # entries ...
_COMPILED_LRU_MAX_BYTES = 16 << 30  # ... and an estimate of the device bytes their kernel forms hold (large graphs pin a lot of HBM each)
_COMPILED_LRU_BYTES = 0
_CACHE_STATS = {"hits": 0, "misses": 0, "memo_hits": 0}


def _device_bytes_estimate(cf: "CompiledFSM") -> int:
    """What a compiled graph holds on the device, roughly: both directions' arcs in two or three kernel forms (8 to 16 bytes per arc
    slot and form, padded) and per-state tables."""
    return int(96 * cf.fsm.nnz + 64 * cf.S1 + 4096)


def compiled_cache_stats(reset: bool = False) -> dict:
    """{"hits", "misses", "memo_hits", "entries"} of the compiled-graph cache behind the reference-shaped entries."""
    out = dict(_CACHE_STATS, entries=0 if _COMPILED_LRU is None else len(_COMPILED_LRU), bytes_estimate=_COMPILED_LRU_BYTES)
    if reset:
        for k in _CACHE_STATS:
            _CACHE_STATS[k] = 0
    return out


def compiled_cache_clear():
    global _COMPILED_LRU, _COMPILED_LRU_BYTES
    _COMPILED_LRU = None
    _COMPILED_LRU_BYTES = 0


def _content_key(part: FSM, c: StateMap) -> bytes:
    try:
        import xxhash

        h = xxhash.xxh3_128()
    except ImportError:  # pragma: no cover
        import hashlib

        h = hashlib.blake2b(digest_size=16)
    nz = np.ascontiguousarray(part.nzval)
    h.update(f"{part.semiring}|{part.S1}|{part.nnz}|{nz.dtype.str}|{c.numpdf}|".encode())
    for a in (part.colptr, part.rowval, nz, part.alpha_idx, part.alpha_val, c.state2pdf):
        a = np.ascontiguousarray(a)
        h.update(memoryview(a).cast("B"))
        h.update(b"|")
    return h.digest()


def _compiled_for(parts, Cs) -> List[CompiledFSM]:
    global _COMPILED_LRU, _COMPILED_LRU_BYTES
    from collections import OrderedDict

    if _COMPILED_LRU is None:
        _COMPILED_LRU = OrderedDict()
    lru = _COMPILED_LRU
    local, keys = {}, []
    for part, c in zip(parts, Cs):  # (the same objects repeated -- one denominator graph B times -- are hashed once)
        ik = (id(part), id(c))
        if ik not in local:
            local[ik] = _content_key(part, c)
        keys.append(local[ik])
    miss = {}
    for k, part, c in zip(keys, parts, Cs):
        if k in lru:
            lru.move_to_end(k)
        elif k not in miss:
            miss[k] = (part, c)
    _CACHE_STATS["hits"] += len(keys) - len(miss)
    _CACHE_STATS["misses"] += len(miss)
    if miss:
        # one call per (semiring, float type) group: compile_many's graphs share both
        groups = {}
        for k, (part, c) in miss.items():
            groups.setdefault((part.semiring, np.asarray(part.nzval).dtype == np.float64), []).append((k, part, c))
        for grp in groups.values():
            made = compile_many([g[1] for g in grp], [g[2] for g in grp]) if len(grp) > 1 else [CompiledFSM(grp[0][1], grp[0][2])]
            for (k, _, _), cf in zip(grp, made):
                lru[k] = cf
                _COMPILED_LRU_BYTES += _device_bytes_estimate(cf)
    out = [lru[k] for k in keys]  # (before anything is evicted: a batch larger than the bounds still gets its graphs)
    while len(lru) > 1 and (len(lru) > _COMPILED_LRU_MAX or _COMPILED_LRU_BYTES > _COMPILED_LRU_MAX_BYTES):
        _, old = lru.popitem(last=False)
        _COMPILED_LRU_BYTES -= _device_bytes_estimate(old)
    return out


def _as_batch(fsm, Chats) -> BatchedFSM:
    if isinstance(fsm, BatchedFSM):
        return fsm
    if isinstance(fsm, CompiledFSM):
        return BatchedFSM([fsm])
    if Chats is None:
        raise TypeError("pdfposteriors(fsm::FSM, V_hats, C_hats) needs the state maps")
    # the same FSM object with the same map objects as last time (a training loop's denominator): the batch as it was -- if the
    # object still holds what it held (a fingerprint of its arrays: their buffers, sizes and a strided sample of their values --
    # an FSM mutated in place between calls misses here and is found, or compiled, by its content below; the reference re-reads the
    # FSM on every call)
    memo = fsm.__dict__.get("_mm_batch_memo")
    if memo is not None and len(memo[0]) == len(Chats) and all(a is b for a, b in zip(memo[0], Chats)) and memo[2] == _fingerprint(fsm, Chats):
        _CACHE_STATS["memo_hits"] += 1
        return memo[1]
    # (a general sparse C_hat rides along as an argument of the generic entry; its FSM handle gets a placeholder map)
    Cs = [c if isinstance(c, StateMap) else
          (StateMap(np.zeros(c.shape[0] - 1, dtype=np.int64), c.numpdf) if isinstance(c, GeneralStateMap) else StateMap.from_matrix(c))
          for c in Chats]
    parts = split_blocks(fsm, [c.shape[0] for c in Cs])
    bf = BatchedFSM(_compiled_for(parts, Cs))
    fsm.__dict__["_mm_batch_memo"] = (list(Chats), bf, _fingerprint(fsm, Chats))
    return bf


def _fingerprint(fsm: FSM, Chats) -> tuple:
    """Cheap (O(1) in the size of the graph) and sensitive to what in-place edits do: for every array of the FSM and of the first and
    last state map its buffer address, size and 64 evenly spaced values."""
    def one(a):
        a = np.asarray(a)
        if a.size == 0:
            return (0, 0, b"")
        flat = a.reshape(-1)
        return (a.__array_interface__["data"][0], a.size, flat[:: max(1, a.size // 64)][:65].tobytes())

    arrs = [fsm.colptr, fsm.rowval, fsm.nzval, fsm.alpha_idx, fsm.alpha_val]
    for c in (Chats[0], Chats[-1]):
        s2p = getattr(c, "state2pdf", None)
        if s2p is not None:
            arrs.append(s2p)
    return tuple(one(a) for a in arrs)


def _device_vhats(Vhats):
    """The V_hats as ONE device tensor [B, P+1, N+1] if they are float32 tensors on the HIP device (a list of (P+1) x (N+1)
    matrices, or the stacked tensor itself), else None."""
    try:
        import torch
    except ImportError:  # pragma: no cover
        return None
    if isinstance(Vhats, torch.Tensor):
        return Vhats if (Vhats.is_cuda and Vhats.dim() == 3 and Vhats.dtype == torch.float32) else None
    Vh = list(Vhats)
    if not Vh or not all(isinstance(v, torch.Tensor) and v.is_cuda and v.dim() == 2 and v.dtype == torch.float32 for v in Vh):
        return None
    if any(v.shape != Vh[0].shape for v in Vh):
        raise _lib.DimensionMismatch(-2, "all V_hat must share one (P+1) x (N+1) shape")
    return torch.stack(Vh)


def _pdfposteriors_device(bf: BatchedFSM, Vd, seqlengths):
    """The reference call shape on device-resident V_hats, nothing crossing to the host unless asked to: the (P+1) x (N+1)
    matrices of expand() (src/inference.jl:54-60) become the engine's [B, N, P] log-likelihoods + lengths by one transposing
    copy; the lengths come from ``seqlengths`` (trusted, asynchronous) or -- one small reduction and a host read -- from the phony
    row, whose form is then checked like _unexpand checks it.  Returns device tensors (gamma[B, P, N] as a view, ttl[B])."""
    torch = _torch()
    B, P1, N1 = Vd.shape
    if B != bf.B:
        raise _lib.DimensionMismatch(-2, f"{B} matrices V_hat for a batch of {bf.B} FSMs")
    if seqlengths is None:
        zero, one = _SEM_ZERO[bf.semiring], _SEM_ONE[bf.semiring]
        ph = Vd[:, P1 - 1, :]
        lens = (ph == zero).sum(dim=1).to(torch.int32)
        step = torch.arange(N1, device=Vd.device)[None, :] < lens[:, None]
        ok = bool((torch.where(step, ph == zero, ph == one).all() & (lens <= N1 - 1).all()).item())
        if ok:  # real pdfs beyond the length are zero(K)
            ok = bool((Vd[:, : P1 - 1, :] == zero).masked_fill(step[:, None, :], True).all().item())
        if not ok:
            return None
    else:
        lens = torch.as_tensor(seqlengths, dtype=torch.int32, device=Vd.device)
    V = Vd[:, : P1 - 1, : N1 - 1].transpose(1, 2).contiguous()
    g, ttl = bf.pdfposteriors(V, lens)
    return g.transpose(1, 2), ttl


def pdfposteriors(fsm, Vhats, Chats=None, seqlengths=None):
    """pdfposteriors(fsm, V_hats, C_hats) (src/inference.jl:145-161) -- ``fsm`` the
    rawunion of the batch -- or pdfposteriors2(cfsm, V_hats) (:164-180) when
    given a BatchedFSM/CompiledFSM.  Returns (gamma[B, P, N] probabilities,
    ttl[B]): NumPy arrays for host inputs, like the reference returns fresh arrays; DEVICE tensors for float32 V_hats
    that live on the HIP device (a list of (P+1) x (N+1) tensors or one [B, P+1, N+1] tensor), with nothing but the
    compiled-graph cache between the call and the kernels (``seqlengths``: the lengths expand() was given, so that they
    need not be read back from the phony row)."""
    if not hasattr(Vhats, "dim"):  # (a generator is read once)
        Vhats = list(Vhats)
    Vd = _device_vhats(Vhats)
    if Chats is not None:  # general sparse maps that are one-hot after all take the fast kernels
        Chats = [(c.one_hot() or c) if isinstance(c, GeneralStateMap) else c for c in Chats]
    bf = _as_batch(fsm, Chats)
    general_c = Chats is not None and any(isinstance(c, GeneralStateMap) for c in Chats)
    # (ProbSemiring{Float32}: the fast entry too -- the library runs the log twins of the FSMs on log V_hat, mm_pdfposteriors_f32)
    fast = not (general_c or bf.dtype == np.float64) and (bf.semiring != "prob" or bf.has_fast_entry())
    if Vd is not None and fast:
        out = _pdfposteriors_device(bf, Vd, seqlengths)
        if out is not None:
            return out
    Vh = [np.asarray(v.cpu() if hasattr(v, "cpu") else v) for v in Vhats]
    # the precision follows the FSM's K like the reference (src/inference.jl:147 converts V_hat to K): a Float32 FSM computes
    # in float32 whatever the dtype of V_hat (NumPy's default float64 included), a Float64 FSM in float64
    un = _unexpand(Vh, bf.semiring) if fast else None
    if un is None:
        # a Float64 FSM, ProbSemiring, a general C_hat, or V_hat that expand() did not make: the generic entry
        return bf.pdfposteriors_generic(Vh, Chats if general_c else None)
    V, lens = un
    g, ttl = bf.pdfposteriors(V, lens)
    return np.ascontiguousarray(g.transpose(0, 2, 1)), ttl


def alpharecursion(fsm, Vhats, Chats=None):
    """alpha-recursion (src/inference.jl:62-74) on C_hat * V_hat as pdfposteriors calls
    it (:150-152): the (sum S1) x (N+1) matrix state_A."""
    bf = _as_batch(fsm, Chats)
    V, lens = _need_expanded(Vhats)
    return bf.alpharecursion(V, lens)


def betarecursion(fsm, Vhats, Chats=None):
    """beta-recursion (src/inference.jl:99-110): state_B."""
    bf = _as_batch(fsm, Chats)
    V, lens = _need_expanded(Vhats)
    return bf.betarecursion(V, lens)


def bestpath(fsm, Vhats, Chats=None):
    """bestpath (docs/src/inference.md:6; historical examples/demo.ipynb cell 23):
    per utterance the 0-based state sequence of the best path and its weight."""
    bf = _as_batch(fsm, Chats)
    V, lens = _need_expanded(Vhats)
    path, score = bf.viterbi(V, lens)
    return [path[b, : lens[b]].copy() for b in range(bf.B)], score


def maxstateposteriors(fsm, Vhats, Chats=None):
    """maxstateposteriors (docs/src/inference.md:5; absent from src/ at this commit, historical use
    test/test_algorithms.jl:280): the max-marginals of the tropical semiring, mu = alpha (*) beta (/) best --
    for every state and frame the weight of the best complete path through it, relative to the best path
    overall (0 on a best path, -inf where no complete path passes).  Tropical FSMs; returns the
    (sum S1) x (N+1) matrix in the layout of alpha-recursion / beta-recursion."""
    bf = _as_batch(fsm, Chats)
    if bf.semiring != "tropical":
        raise TypeError("maxstateposteriors needs TropicalSemiring FSMs")
    V, lens = _need_expanded(Vhats)
    return bf.maxstateposteriors(V, lens)


def _total(fsm, n, cumulative):
    if isinstance(fsm, (BatchedFSM, CompiledFSM)):
        bf = _as_batch(fsm, None)
    else:  # no emissions are involved: any state map will do
        bf = BatchedFSM([CompiledFSM(fsm, StateMap(np.zeros(fsm.S1 - 1, dtype=np.int64), 1))])
    out = bf.totalsum(n, cumulative).cpu().numpy()
    return out if isinstance(fsm, BatchedFSM) else float(out[0])


def totalsum(fsm, n: int):
    """totalsum(alpha, T, omega, n) (src/algorithms.jl:23-29) of an FSM: the weight of all paths of exactly n
    states, in the FSM's semiring, as a natural-log float (an array for a BatchedFSM)."""
    return _total(fsm, n, False)


def totalcumsum(fsm, n: int):
    """totalcumsum(alpha, T, omega, n) (src/algorithms.jl:8-16): paths of up to n states."""
    return _total(fsm, n, True)


def totalweightsum(fsm, n: Optional[int] = None):
    """totalweightsum(fsm, n = nstates(fsm)) (src/algorithms.jl:36)."""
    if n is None:
        if isinstance(fsm, BatchedFSM):
            raise TypeError("totalweightsum of a batch needs n")
        n = fsm.S1 - 1
    return _total(fsm, n, True)


# the reference's own (unicode) export names, src/MarkovModels.jl:40-45
αrecursion = alpharecursion
βrecursion = betarecursion
