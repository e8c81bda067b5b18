#!/usr/bin/env python3
"""Diagnostic: where a step of the lane kernel goes (cycles per step and section; -DMM_STAMPS build: make -C markovmodels.jl_amd/csrc stamps)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_AMD_LIB", os.path.join(ROOT, "gpurun_stamps", "libmarkovmodels_amd_stamps.so"))
import __graft_entry__ as ge
import torch
mm = ge.load_package(); wl = importlib.import_module(mm.__name__ + ".workloads"); L = importlib.import_module(mm.__name__ + "._lib")
S = int(os.environ.get("S", 64))
g, B, N = (wl.dense_ergodic(S, seed=0) if S > 3 else wl.l2r_hmm(3)), int(os.environ.get("B", 32)), int(os.environ.get("N", 500))
bf = mm.batch(*([mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * B))
V = torch.randn(B, N, g.P, device="cuda")
print(bf.kernels())
bf.pdfposteriors(V); bf.pdfposteriors(V); torch.cuda.synchronize()
n = B * 16 * 16
out = np.zeros(n, dtype=np.uint64)
L.lib.mm_debug_read_stamps.argtypes = [C.c_void_p, C.c_int64]
assert L.lib.mm_debug_read_stamps(out.ctypes.data, n) == 0
s = out[: B * 4 * 8].reshape(B, 4, 8).astype(np.float64) / N
an = ["reads, exponent, emissions, product", "scale, wait for the slot", "publish, request"]
en = ["DMA wait", "wait for the agent (slot)", "stage + request"]
fn = ["DMA wait", "wait for the agent", "store / combine + requests"]
for w, wn in enumerate(("forward agent", "backward agent", "forward service", "backward service")):
    nm = an if w < 2 else fn
    print(wn, "cycles per step %.0f:" % s[:, w].sum(-1).mean(), ", ".join(f"{nm[k]} {s[:, w, k].mean():.0f}" for k in range(len(nm))))
