#!/usr/bin/env python3
"""config 3's graph family with 200 pdfs (the NJ = 4 instances): ms per call"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
for S, P, B, N in ((2000, 200, 256, 1500), (2000, 84, 256, 1500)):
    g = wl.lfmmi_denominator(S, P, seed=0)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    V = torch.randn(B, N, g.P, device="cuda")
    gam = torch.empty(B, N, g.P, device="cuda")
    for _ in range(3):
        bf.pdfposteriors(V, None, out=gam)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        bf.pdfposteriors(V, None, out=gam)
    torch.cuda.synchronize()
    print(f"S={S} P={P} B={B} T={N}: {(time.perf_counter() - t0) * 100:.3f} ms  redo {bf.last_redo_count()}  {bf.kernels()[:40]}", flush=True)
