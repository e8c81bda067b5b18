#!/usr/bin/env python3
"""where compile_many's 4 ms go: cProfile of the Python side, and the C call alone (MM_VERBOSE prints its own sections)"""
import cProfile, importlib, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
g = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz"))
B = 128
sm = mm.statemap(g.state2pdf, g.P)
torch.zeros(1, device="cuda")
for rep in range(3):
    fs = [wl.to_fsm(mm, g) for _ in range(B)]
    t0 = time.perf_counter()
    cfs = mm.compile_many(fs, sm)
    t1 = time.perf_counter()
    print("compile_many %.2f ms" % (1e3 * (t1 - t0)))
    del cfs
fs = [wl.to_fsm(mm, g) for _ in range(B)]
pr = cProfile.Profile()
pr.enable()
cfs = mm.compile_many(fs, sm)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
