"""Host-side mirror of src/linalg.jl: the semiring ``mul!`` methods and the sparse-vector broadcast the reference defines
on CuSparse containers, each ONE call into the HIP library (``mm_spmv`` / ``mm_spmm`` / ``mm_svdv``,
include/markovmodels_amd.h).  There is no CPU path here.

    A = SparseCSR.from_coo(I, J, V, (m, n), "log")        # ~ CuSparseMatrixCSR(adapt(CuArray, sparse(I, J, K.(V), m, n)))
    c = mul_(torch.empty(m, ...), A, b)                    # ~ mul!(similar(b, m), A, b)            src/linalg.jl:163-184
    C = mul_(torch.empty(m, k, ...), A, B)                 # ~ mul!(similar(B, m, k), A, B)          :240-262
    d = elmul_(torch.empty(n, ...), x, y)                  # ~ elmul!(d, y, x::CuSparseVector)       :290, 294-315

Arrays are torch tensors on the HIP device; a dense matrix is COLUMN-major like Julia's: pass ``M.t()`` of a contiguous
``(cols, rows)`` tensor, or any 2-D tensor with ``stride(0) == 1`` (``colmajor`` makes one).  Values are in the semiring's
domain (natural log for "log" / "tropical"; zero(K) = -inf), indices are stored 1-based ``Cint`` as CUDA.jl / AMDGPU.jl do.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import SEMIRING_ID, check, lib


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("markovmodels_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
    return torch


def _stream():
    return _torch().cuda.current_stream().cuda_stream


def colmajor(M):
    """A column-major copy of a 2-D tensor (``stride(0) == 1``), the layout of a Julia ``Matrix``."""
    return M.t().contiguous().t()


class SparseCSR:
    """CuSparseMatrixCSR{K} (src/linalg.jl:80-131 builds them with ``Cint`` indices, 1-based): rowPtr, colVal, nzVal on the device."""

    def __init__(self, rowptr, colval, nzval, shape: Tuple[int, int], semiring: str = "log", index_base: int = 1):
        torch = _torch()
        if semiring not in SEMIRING_ID:
            raise ValueError(f"unknown semiring {semiring!r}")
        self.semiring = semiring
        self.shape = (int(shape[0]), int(shape[1]))
        self.index_base = int(index_base)
        self.rowptr = torch.as_tensor(rowptr, dtype=torch.int32).cuda().contiguous()
        self.colval = torch.as_tensor(colval, dtype=torch.int32).cuda().contiguous()
        nz = torch.as_tensor(nzval).cuda().contiguous()
        if nz.dtype not in (torch.float32, torch.float64):
            nz = nz.to(torch.float32)
        self.nzval = nz
        if self.rowptr.numel() != self.shape[0] + 1 or self.colval.numel() != self.nzval.numel():
            raise _lib.DimensionMismatch(-2, "SparseCSR: rowptr / colval / nzval sizes do not match the shape")

    @property
    def nnz(self) -> int:
        return int(self.nzval.numel())

    @property
    def dtype(self):
        return self.nzval.dtype

    @classmethod
    def from_coo(cls, I: Sequence[int], J: Sequence[int], V: Sequence[float], shape, semiring: str = "log", dtype=np.float32,
                 index_base: int = 1) -> "SparseCSR":
        """``sparse(I, J, K.(V), m, n)`` (1-based ``I``, ``J`` when ``index_base`` is 1) as CSR; duplicates are not combined."""
        I = np.asarray(I, dtype=np.int64) - index_base
        J = np.asarray(J, dtype=np.int64) - index_base
        V = np.asarray(V, dtype=dtype)
        order = np.lexsort((J, I))
        rowptr = np.zeros(shape[0] + 1, dtype=np.int64)
        np.add.at(rowptr, I + 1, 1)
        rowptr = np.cumsum(rowptr) + index_base
        return cls(rowptr.astype(np.int32), (J[order] + index_base).astype(np.int32), V[order], shape, semiring, index_base)


class SparseVector:
    """CuSparseVector{K}: stored indices (1-based ``Cint``) and values of a length-n vector."""

    def __init__(self, nzind, nzval, n: int, semiring: str = "log", index_base: int = 1):
        torch = _torch()
        self.semiring = semiring
        self.n = int(n)
        self.index_base = int(index_base)
        self.nzind = torch.as_tensor(nzind, dtype=torch.int32).cuda().contiguous()
        nz = torch.as_tensor(nzval).cuda().contiguous()
        if nz.dtype not in (torch.float32, torch.float64):
            nz = nz.to(torch.float32)
        self.nzval = nz
        if self.nzind.numel() != self.nzval.numel():
            raise _lib.DimensionMismatch(-2, "SparseVector: nzind / nzval sizes differ")


def _check_dense(t, dtype, what):
    torch = _torch()
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError(f"{what} must be a tensor on the HIP device")
    if t.dtype != dtype:
        raise TypeError(f"{what} has dtype {t.dtype}, the sparse operand {dtype} (one K per call, like the reference's methods)")


def mul_(c, A: SparseCSR, b, alpha=True, beta=False):
    """``LinearAlgebra.mul!(c, A, b)`` / ``mul!(C, A, B, alpha, beta)`` on a CSR matrix over a semiring (src/linalg.jl:163-184,
    240-262).  Vectors: 1-D tensors.  Matrices: column-major 2-D tensors.  ``alpha`` is ignored like in the reference;
    ``beta`` False / 0 overwrites C, True / 1 accumulates into it (the vector method has no beta).  Returns ``c``."""
    vb = A.nzval.element_size()
    sr = SEMIRING_ID[A.semiring]
    _check_dense(b, A.dtype, "b")
    _check_dense(c, A.dtype, "c")
    if b.dim() == 1 and c.dim() == 1:
        if not (b.is_contiguous() and c.is_contiguous()):
            raise ValueError("mul_: vectors must be contiguous")
        check(lib.mm_spmv(sr, vb, A.shape[0], A.shape[1], A.nnz, A.rowptr.data_ptr(), A.colval.data_ptr(), A.index_base,
                          A.nzval.data_ptr(), b.data_ptr(), b.numel(), c.data_ptr(), c.numel(), _stream()))
        return c
    if b.dim() != 2 or c.dim() != 2:
        raise _lib.DimensionMismatch(-2, "mul_: b and c must both be vectors or both be matrices")
    for t, what in ((b, "B"), (c, "C")):
        if t.shape[0] > 1 and t.stride(0) != 1:
            raise ValueError(f"mul_: {what} must be column-major (stride(0) == 1): see linalg.colmajor")
    ldb = b.stride(1) if b.shape[1] > 1 else max(1, b.shape[0])
    ldc = c.stride(1) if c.shape[1] > 1 else max(1, c.shape[0])
    check(lib.mm_spmm(sr, vb, A.shape[0], A.shape[1], A.nnz, A.rowptr.data_ptr(), A.colval.data_ptr(), A.index_base, A.nzval.data_ptr(),
                      b.data_ptr(), b.shape[0], b.shape[1], ldb, c.data_ptr(), c.shape[0], c.shape[1], ldc, float(beta), _stream()))
    return c


def _svdv(op: int, out, x: SparseVector, y):
    _check_dense(y, x.nzval.dtype, "y")
    _check_dense(out, x.nzval.dtype, "out")
    if not (y.is_contiguous() and out.is_contiguous()):
        raise ValueError("vectors must be contiguous")
    check(lib.mm_svdv(SEMIRING_ID[x.semiring], x.nzval.element_size(), op, x.n, x.nzval.numel(), x.nzind.data_ptr(), x.index_base,
                      x.nzval.data_ptr(), y.data_ptr(), y.numel(), out.data_ptr(), out.numel(), _stream()))
    return out


def elmul_(out, x: SparseVector, y):
    """``elmul!(out, y, x::CuSparseVector)`` (src/linalg.jl:290): out = zero(K), out[i] = x[i] (*) y[i] at x's stored entries."""
    return _svdv(0, out, x, y)


def eldiv_(out, x: SparseVector, y):
    """``eldiv!(out, x::CuSparseVector, y)`` (src/linalg.jl:292): out = zero(K), out[i] = x[i] (/) y[i] at x's stored entries."""
    return _svdv(1, out, x, y)
