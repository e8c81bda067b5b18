#!/usr/bin/env python3
"""per-utterance graphs (no pair kernels): the row kernels (default for shallow graphs) against the wave kernel (forced)"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")

def run(gs, B, N, force):
    os.environ["MM_DEBUG"] = "1"
    if force: os.environ["MM_KERNEL"] = force
    else: os.environ.pop("MM_KERNEL", None)
    P = max(g.P for g in gs)
    cfs = [mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, P)) for g in gs]
    bf = mm.batch(*[cfs[b % len(cfs)] for b in range(B)])
    V = torch.randn(B, N, P, device="cuda")
    out = torch.empty(B, N, P, device="cuda")
    for _ in range(3): bf.pdfposteriors(V, None, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): bf.pdfposteriors(V, None, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 50, bf.kernels()[:28], bf.last_redo_count()

fam = {
    "l2r3 B=1 T=100": ([wl.l2r_hmm(3)], 1, 100),
    "l2r3 B=64 T=100": ([wl.l2r_hmm(3)] , 64, 100),
    "lexicon 150-400 states, 128 graphs T=700": ([wl.lexicon_fsm(150 + 2 * b, 40, seed=b, hubs=1 + b % 2) for b in range(128)], 128, 700),
    "random 300 states deg 4, 64 graphs B=128 T=700": ([wl.random_fsm(300, 40, 4.0, seed=b) for b in range(64)], 128, 700),
    "random 300 states deg 8, 64 graphs B=128 T=700": ([wl.random_fsm(300, 40, 8.0, seed=b) for b in range(64)], 128, 700),
    "random 900 states deg 3, 64 graphs B=256 T=700": ([wl.random_fsm(900, 80, 3.0, seed=b) for b in range(64)], 256, 700),
    "ergodic 32 states, B=1 T=500": ([wl.dense_ergodic(32, seed=1)], 1, 500),
}
for name, (gs, B, N) in fam.items():
    res = []
    for force in (None, "row", "wave"):
        try:
            res.append("%s %.3f ms (%s, redo %d)" % (((force or "auto"),) + run(gs, B, N, force)))
        except Exception as e:
            res.append(f"{force}: {type(e).__name__}")
    print(name, "|", " | ".join(res), flush=True)
