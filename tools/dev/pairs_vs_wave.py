#!/usr/bin/env python3
"""ONE shared small graph for the whole batch: the pair kernels (default) against the wave kernel (forced)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")

def run(g, B, N, force):
    os.environ["MM_DEBUG"] = "1"
    if force: os.environ["MM_KERNEL"] = force
    else: os.environ.pop("MM_KERNEL", None)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    V = torch.randn(B, N, g.P, device="cuda")
    out = torch.empty(B, N, g.P, device="cuda")
    for _ in range(3): bf.pdfposteriors(V, None, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): bf.pdfposteriors(V, None, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 100, bf.kernels()[:22], bf.last_redo_count()

fam = {"l2r 3": wl.l2r_hmm(3), "l2r 30": wl.l2r_hmm(30), "lexicon 300": wl.lexicon_fsm(300, 40, seed=1, hubs=2),
       "lexicon 900": wl.lexicon_fsm(900, 84, seed=2, hubs=2), "random 300 deg 4": wl.random_fsm(300, 40, 4.0, seed=3),
       "random 120 deg 6": wl.random_fsm(120, 30, 6.0, seed=4), "ergodic 16": wl.dense_ergodic(16, seed=1), "ergodic 32": wl.dense_ergodic(32, seed=1),
       "lfmmi 250 (16 arcs per state)": wl.lfmmi_denominator(250, 40, seed=2), "random 400 deg 9": wl.random_fsm(400, 40, 9.0, seed=5)}
only = os.environ.get("ONLY")
for name, g in fam.items():
    if only and only not in name:
        continue
    for B in (32, 256, 512, 1024):
        res = []
        for force in ("pair", None, "wave"):
            try:
                res.append("%s %.3f ms (%s, redo %d)" % (((force or "auto"),) + run(g, B, 500, force)))
            except Exception as e:
                res.append(f"{force}: {type(e).__name__}")
        print(name, "B", B, "|", " | ".join(res), flush=True)
