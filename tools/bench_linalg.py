#!/usr/bin/env python3
"""The reference's linear algebra at the boundary (mm_spmv / mm_spmm, src/linalg.jl:163-184, 240-262) at the sizes the reference
itself runs them at -- what ONE frame of its alpha-recursion and ONE call's emission gather cost on the HIP kernels:

  spmv_wsj_den   c = T_hat' (x) b on blockdiag(T_hat of the WSJ denominator x 128): 388 k rows, 6.7 M stored entries -- the product
                 the reference issues once per frame and direction (src/inference.jl:70, 107; 2 x 701 of them per call)
  spmm_config3   lhs = C_hat (x) V_hat at config 3's shape (src/inference.jl:150): blockdiag of 256 one-hot maps (512 256 x 21 760)
                 times a 21 760 x T block of frames (T = 64 columns here: the whole 1501 would be 3 GB of output)

GB/s = the bytes the product has to move (CSR arrays once, b and c once; SpMM: the gathered rows of B once per use) / time,
against the 8 TB/s HBM peak.  The engine's forward-backward kernels never call these (their products are fused): this is the
seam's own speed.      python tools/bench_linalg.py [out.json]          (GPU box)
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from srchash import source_hash  # noqa: E402

HBM = 8000.0


def timed(fn, K=20, W=5):
    for _ in range(W):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def blockdiag_transposed(g, B):
    """CSR of blockdiag(T_hat, B times)' (rows = destination states) of the extended FSM, 1-based Cint like CuSparseMatrixCSR"""
    S1 = g.S + 1
    I = np.concatenate([g.src, g.final_idx, [g.S]])
    J = np.concatenate([g.dst, np.full(g.final_idx.size, g.S), [g.S]])
    W = np.concatenate([g.w, g.final_w, [0.0]]).astype(np.float32)
    rows = np.concatenate([J + b * S1 for b in range(B)]) + 1  # transposed: row = destination
    cols = np.concatenate([I + b * S1 for b in range(B)]) + 1
    return mm.SparseCSR.from_coo(rows, cols, np.tile(W, B), (B * S1, B * S1), "log", np.float32), S1


rows = []
# ---- SpMV: one frame of the reference's forward recursion on its own benchmark graph, batch of 128
g = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz"))
A, S1 = blockdiag_transposed(g, 128)
b = torch.randn(A.shape[1], device="cuda")
c = torch.empty(A.shape[0], device="cuda")
ms = timed(lambda: mm.mul_(c, A, b))
bytes_ = A.nnz * 8 + (A.shape[0] + 1) * 4 + A.shape[1] * 4 + A.shape[0] * 4
rows.append(dict(op="mm_spmv", what="T_hat' (x) alpha on blockdiag(WSJ denominator x 128), log semiring, float32 (one frame of src/inference.jl:70)",
                 rows=A.shape[0], cols=A.shape[1], nnz=A.nnz, ms=ms, bytes=bytes_, GBps=bytes_ / ms / 1e6, frac_of_hbm_peak=bytes_ / ms / 1e6 / HBM,
                 note=f"the engine's fused recursion does 2 x 701 such frames (+ emissions, combine, posteriors) in 1.78 ms; 2 x 701 x this = {2 * 701 * ms:.0f} ms"))
print(rows[-1], flush=True)

# ---- SpMM: the emission gather C_hat (x) V_hat at config 3's shape, a block of 64 frames
g3 = wl.lfmmi_denominator(2000, 84, seed=0)
B3, T = 256, 64
S1, P1 = g3.S + 1, g3.P + 1
s2p = np.concatenate([g3.state2pdf, [g3.P]])
I = np.arange(B3 * S1) + 1
J = np.concatenate([s2p + b * P1 for b in range(B3)]) + 1
C = mm.SparseCSR.from_coo(I, J, np.zeros(I.size, np.float32), (B3 * S1, B3 * P1), "log", np.float32)
V = mm.linalg.colmajor(torch.randn(B3 * P1, T, device="cuda"))
out = mm.linalg.colmajor(torch.empty(B3 * S1, T, device="cuda"))
ms = timed(lambda: mm.mul_(out, C, V))
bytes_ = C.nnz * 8 + (C.shape[0] + 1) * 4 + C.shape[0] * T * 4 + C.shape[1] * T * 4
rows.append(dict(op="mm_spmm", what=f"C_hat (x) V_hat (src/inference.jl:150) at config 3's shape: blockdiag of 256 one-hot maps, {T} frames, log semiring, float32",
                 rows=C.shape[0], cols=C.shape[1], nnz=C.nnz, frames=T, ms=ms, bytes=bytes_, GBps=bytes_ / ms / 1e6, frac_of_hbm_peak=bytes_ / ms / 1e6 / HBM,
                 note="bytes = CSR once + C written once + V_hat read once (every gathered row of V_hat is re-read from cache by the ~24 states of its pdf)"))
print(rows[-1], flush=True)

if len(sys.argv) > 1:
    json.dump(dict(source_hash=source_hash(), hbm_peak_GBps=HBM, rows=rows), open(sys.argv[1], "w"), indent=1)
