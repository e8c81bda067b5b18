#!/usr/bin/env python3
"""Diagnostic: where a step of the wave kernel goes, per wave (needs the -DMM_STAMPS build: make -C markovmodels.jl_amd/csrc stamps)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_AMD_LIB", os.path.join(ROOT, "gpurun_stamps", "libmarkovmodels_amd_stamps.so"))
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
L = importlib.import_module(mm.__name__ + "._lib")
g, B = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz")), 128
N = int(os.environ.get("N", 400))
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
V = torch.randn(B, N, g.P, device="cuda")
print(bf.kernels())
bf.pdfposteriors(V)
bf.pdfposteriors(V)
torch.cuda.synchronize()
n = B * 16 * 16
out = np.zeros(n, dtype=np.uint64)
L.lib.mm_debug_read_stamps.argtypes = [C.c_void_p, C.c_int64]
assert L.lib.mm_debug_read_stamps(out.ctypes.data, n) == 0
sel = int(os.environ.get("MM_SPLIT_SLEEP", "0"))  # 0x1000: phase B only, 0x2000: phase A only (read by the kernel under MM_DEBUG)
s = out[: B * 16 * 8].reshape(B, 16, 8).astype(np.float64) / (N / 2 if sel & 0x3000 else N)
names = ["-", "in barrier (1)", "gather+lse (2)", "finish (3)", "-", "-"]
enames = ["-", "in barrier (1)", "wait DMA (2)", "stage+fetch (3)", "offset (4)", "-"]
pnames = ["-", "in barrier (1)", "scan max (2)", "-", "-", "-"]
fnames = ["-", "in barrier (1)", "-", "poff (3)", "frame_out (4)", "-"]
for wv in range(14):
    nm = enames if wv % 7 == 4 else pnames if wv % 7 == 5 else fnames if wv % 7 == 6 else names
    print("wave", wv, " ".join(f"{nm[k]} {s[:, wv, k].mean():6.0f}" for k in range(6)), " total %.0f" % s[:, wv, :6].sum(-1).mean())
