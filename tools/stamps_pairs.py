#!/usr/bin/env python3
"""Diagnostic: where a step of the pair kernels goes, per wave: cycles of work and cycles at the barrier
(needs the -DMM_STAMPS build: make -C markovmodels.jl_amd/csrc stamps).  Read SHARES, not totals."""
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
g, B = wl.lfmmi_denominator(2000, int(os.environ.get("P", 84)), seed=0), 256
if os.environ.get("WL") == "wsj_den":
    g, B = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz")), 128
if os.environ.get("WL") == "lfmmi_den4000":
    g, B = wl.lfmmi_denominator(4000, 84, seed=1), 32
if os.environ.get("WL") == "big6000":  # teams of 8
    g, B = wl.lfmmi_denominator(6000, int(os.environ.get("P", 300)), seed=1), 32
if os.environ.get("WL") == "ergodic64":
    g, B = wl.dense_ergodic(64, seed=0), 32
N = int(os.environ.get("N", 300))
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
V = torch.randn(B, N, g.P, device="cuda")
if os.environ.get("SHARP"):  # log-softmax(10 N(0,1)) and the exact kernels first: the stamps of the wide pair kernels
    V = torch.log_softmax(10.0 * V, dim=-1)
    bf.set_exact_policy("f64_first")
print(bf.kernels())
bf.pdfposteriors(V)
bf.pdfposteriors(V)
torch.cuda.synchronize()
print("redo", bf.last_redo_count())
n = B * 16 * 16
out = np.zeros(n, dtype=np.uint64)
L.lib.mm_debug_read_stamps.argtypes = [C.c_void_p, C.c_int64]
assert L.lib.mm_debug_read_stamps(out.ctypes.data, n) == 0
s = out[: (B // 2) * 16 * 8].reshape(B // 2, 16, 2, 2, 2).astype(np.float64) / (N / 2)
for ph, pn in enumerate(("phase A", "phase B")):
    for d, dn in enumerate(("forward", "backward")):
        work, wait = s[:, :, ph, d, 0].mean(0), s[:, :, ph, d, 1].mean(0)
        print(pn, dn, "cycles per step: total %.0f" % (work + wait).mean())
        print("  work per wave   :", " ".join(f"{v:5.0f}" for v in work))
        print("  barrier per wave:", " ".join(f"{v:5.0f}" for v in wait))
sv = out[(B // 2) * 16 * 8 : (B // 2) * 16 * 8 + (B // 2) * 32].reshape(B // 2, 2, 2, 8).astype(np.float64) / (N / 2)
names = ["rest", "barrier", "wait DMA", "scan max", "norm+stage", "DMA issue", "gamma frames", "wait partner"]
for ph, pn in enumerate(("phase A", "phase B")):
    for d, dn in enumerate(("forward", "backward")):
        print(pn, dn, "service wave:", ", ".join(f"{names[k]} {sv[:, ph, d, k].mean():.0f}" for k in range(8)))
