#!/usr/bin/env python3
"""graphs beyond the fast paths (quad / item kernels): ms per pdfposteriors call"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
CASES = ((6000, 300, 128, 700), (4000, 84, 128, 700), (2000, 400, 256, 1500), (10000, 1000, 64, 700))
if len(sys.argv) > 1:  # S P B N
    CASES = (tuple(int(x) for x in sys.argv[1:5]),)
for S, P, B, N in CASES:
    g = wl.lfmmi_denominator(S, P, seed=1)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    V = torch.randn(B, N, g.P, device="cuda")
    gam = torch.empty(B, N, g.P, device="cuda")
    for _ in range(2):
        bf.pdfposteriors(V, None, out=gam)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        bf.pdfposteriors(V, None, out=gam)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 200
    print(f"S={S} arcs={g.n_arcs} P={P} B={B} T={N}: {ms:.2f} ms  {B * N / ms * 1e3:.3g} frames/s  redo {bf.last_redo_count()}  {bf.kernels()[:70]}", flush=True)
