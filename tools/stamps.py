#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the quad kernel (needs the -DMM_STAMPS build:
make -C markovmodels.jl_amd/csrc stamps).  Read SHARES, not totals (the stamps fence overlaps)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_AMD_LIB", os.path.join(ROOT, "markovmodels.jl_amd", "libmarkovmodels_amd_stamps.so"))
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
L = importlib.import_module(mm.__name__ + "._lib")
which = sys.argv[1] if len(sys.argv) > 1 else "lfmmi_den"
if which == "wsj_num":
    g, B = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz")), 128
elif which == "wsj_den":
    g, B = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz")), 128
else:
    g, B = wl.lfmmi_denominator(2000, 84, seed=0), 256
N = int(os.environ.get("N", 300))
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
V = torch.randn(B, N, g.P, device="cuda")
bf.pdfposteriors(V)
bf.pdfposteriors(V)
torch.cuda.synchronize()
n = B * 16 * 16
out = np.zeros(n, dtype=np.uint64)
L.lib.mm_debug_read_stamps.argtypes = [C.c_void_p, C.c_int64]
assert L.lib.mm_debug_read_stamps(out.ctypes.data, n) == 0
s = out.reshape(B, 16, 2, 8).astype(np.float64) / N
names = ["top(part_max,copy,em)", "quad_phase", "barrier1", "rows(phaseB)", "barrier2"]
for d, dn in enumerate(("forward", "backward")):
    print(dn, "cycles per step: mean over all waves | wave 0 | max wave")
    tot = s[:, :, d, :5].sum(-1).mean()
    for k, nm in enumerate(names):
        x = s[:, :, d, k]
        print(f"  {nm:24s} {x.mean():8.0f} | {x[:, 0].mean():8.0f} | {x.mean(0).max():8.0f}   ({100 * x.mean() / tot:4.1f} %)")
    print(f"  total {tot:8.0f}")
    print("  per-wave exact rows/step:", " ".join(f"{v:.2f}" for v in s[:, :, d, 7].mean(0)))
    print("  per-wave B:totals     :", " ".join(f"{v:.0f}" for v in s[:, :, d, 5].mean(0)))
    print("  per-wave B:finish     :", " ".join(f"{v:.0f}" for v in s[:, :, d, 6].mean(0)))
    print("  per-wave rows(phaseB):", " ".join(f"{v:.0f}" for v in s[:, :, d, 3].mean(0)))
    print("  per-wave quad_phase  :", " ".join(f"{v:.0f}" for v in s[:, :, d, 1].mean(0)))
    print("  per-wave top         :", " ".join(f"{v:.0f}" for v in s[:, :, d, 0].mean(0)))
