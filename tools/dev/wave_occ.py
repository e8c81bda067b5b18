#!/usr/bin/env python3
"""one shared small graph on the wave kernel, B from the environment: for rocprofv3 --pmc SQ_WAVE_CYCLES (are two
workgroups of the wave kernel resident per compute unit?)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
B, N = int(os.environ.get("B", 256)), 500
g = wl.lexicon_fsm(200, 30, seed=1, hubs=2)
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
V = torch.randn(B, N, g.P, device="cuda")
out = torch.empty(B, N, g.P, device="cuda")
for _ in range(int(os.environ.get("REPS", 3))):
    bf.pdfposteriors(V, None, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
bf.pdfposteriors(V, None, out=out)
torch.cuda.synchronize()
print("B", B, bf.kernels()[:20], "%.3f ms" % ((time.perf_counter() - t0) * 1e3))
