#!/usr/bin/env python3
"""ONE utterance (B = 1) on graphs of the fast paths: latency per call"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
o, oc = ge.load_oracle()
sys.path.insert(0, os.path.join(ROOT, "tests"))
import graphs
for name, g, N in (("lfmmi 2000", wl.lfmmi_denominator(2000, 84, seed=0), 1500), ("wsj den", wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz")), 700),
                   ("lfmmi 600", wl.lfmmi_denominator(600, 40, seed=5), 500)):
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(cf)
    V = torch.randn(1, N, g.P, device="cuda")
    out = torch.empty(1, N, g.P, device="cuda")
    for _ in range(3): _, ttl = bf.pdfposteriors(V, None, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): bf.pdfposteriors(V, None, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    Vn = V.cpu().numpy()
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, Vn[:, :200], None, dtype=np.float64) if N <= 200 else (None, None)
    lens = torch.tensor([N - 7], dtype=torch.int32, device="cuda")
    g2, t2 = bf.pdfposteriors(V, lens)
    print(name, "B=1: %.3f ms" % ms, bf.kernels()[:40], "redo", bf.last_redo_count(), "ttl", float(ttl[0]), float(t2[0]), "gamma sums", float(out[0].sum(-1).min()), float(out[0].sum(-1).max()))
