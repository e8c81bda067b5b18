#!/usr/bin/env python3
"""Diagnostic: time of the forward half of the quad kernel alone.  The last frame's emissions are zero(K),
so Z = 0 and the kernel returns right after the forward pass (gamma = 0, ttl = -inf)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
which = sys.argv[1] if len(sys.argv) > 1 else "lfmmi_den"
if which == "wsj_num":
    g, B, N = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz")), 128, 700
elif which == "wsj_den":
    g, B, N = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz")), 128, 700
else:
    g, B, N = wl.lfmmi_denominator(2000, 84, seed=0), 256, 1500
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
V = torch.randn(B, N, g.P, device="cuda")
gamma = torch.empty(B, N, g.P, device="cuda")
for name, Vx in (("full", V), ("forward only", V.clone())):
    if name != "full":
        Vx[:, -1, :] = -float("inf")
    for _ in range(2):
        bf.pdfposteriors(Vx, out=gamma)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        _, ttl = bf.pdfposteriors(Vx, out=gamma)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"{which} {name:14s} {ev[0].elapsed_time(ev[1]) / 5:.3f} ms   ttl[0] = {float(ttl[0]):.3f}")
