#!/usr/bin/env python3
"""config 3's graph family with other pdf counts: ms per pdfposteriors call (B = 256, T = 1500)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
for P in (84, 128, 130, 200, 248):
    g = wl.lfmmi_denominator(2000, P, seed=0)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    B, N = 256, 1500
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
    print(P, "pdfs: %.2f ms" % ((time.perf_counter() - t0) * 100), "redo", bf.last_redo_count(), bf.kernels()[:40])
