#!/usr/bin/env python3
"""compile_many of 128 numerator graphs against the number of host threads"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
g = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz"))
sm = mm.statemap(g.state2pdf, g.P)
torch.zeros(1, device="cuda")
for thr in (1, 2, 4, 8, 16, 32):
    best = 1e9
    for rep in range(5):
        fs = [wl.to_fsm(mm, g) for _ in range(128)]
        t0 = time.perf_counter()
        cfs = mm.compile_many(fs, sm, threads=thr)
        best = min(best, time.perf_counter() - t0)
        del cfs
    print(f"threads {thr}: {best * 1e3:.2f} ms", flush=True)
