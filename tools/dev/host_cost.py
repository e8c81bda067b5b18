#!/usr/bin/env python3
"""host cost of building a numerator batch: FSM -> compile -> batch (128 graphs of the WSJ numerator's size), against the call"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
g = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz"))
B, N = 128, 700
V = torch.randn(B, N, g.P, device="cuda")
sm = mm.statemap(g.state2pdf, g.P)
for rep in range(3):
    t0 = time.perf_counter()
    fs = [wl.to_fsm(mm, g) for _ in range(B)]
    t1 = time.perf_counter()
    cfs = mm.compile_many(fs, sm) if os.environ.get("MANY") else [mm.compile(f, sm) for f in fs]
    t2 = time.perf_counter()
    bf = mm.batch(*cfs)
    t3 = time.perf_counter()
    gam, ttl = bf.pdfposteriors(V)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    gam, ttl = bf.pdfposteriors(V)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    t6 = time.perf_counter()
    bf2 = mm.batch(*cfs)
    t7 = time.perf_counter()
    print("batch of the same (known) FSMs again %.2f ms;" % (1e3 * (t7 - t6)), end=" ")
    print("FSM objects %.1f ms, compile (upload) %.1f ms, batch (pack + upload) %.1f ms, first call %.1f ms, second call %.2f ms" %
          (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4)), bf.kernels()[:30], flush=True)
