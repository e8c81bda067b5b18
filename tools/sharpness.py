#!/usr/bin/env python3
"""How sharp may the emissions be before the linear-domain kernels hand utterances to the exact ones?
log-softmax of sigma * N(0,1) on config 3 (B = 256, T = 1500) and on the reference's WSJ denominator (B = 128, T = 700)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
cases = [(wl.lfmmi_denominator(2000, 84, seed=0), 256, 1500), (wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz")), 128, 700)]
for g, B, N in cases:
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    if os.environ.get("FLOOR"):
        bf.set_posterior_floor(float(os.environ["FLOOR"]))
    gam = torch.empty(B, N, g.P, device="cuda")
    for sigma in (1, 2, 3, 4, 5, 6, 8, 10):
        V = torch.log_softmax(sigma * torch.randn(B, N, g.P, device="cuda"), dim=-1)
        for _ in range(2):
            bf.pdfposteriors(V, None, out=gam)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            bf.pdfposteriors(V, None, out=gam)
        torch.cuda.synchronize()
        print(g.name, "sigma", sigma, "%.2f ms" % ((time.perf_counter() - t0) * 200), "redone", bf.last_redo_count(), "of", B, flush=True)
