#!/usr/bin/env python3
"""Graphs of config 3's family beyond config 3's size -- more pdfs (251 .. 506: the NJ = 8 instances of the pair kernels), more
states (teams of 4 and of 8 workgroups), and beyond every register-resident form (the stream kernels; the item kernel until round 5): ms per
pdfposteriors call.

    python tools/bench_big.py [S P B N] [--json out.json]      # on the GPU box (tools/measure_all.sh -> profiles/<tag>_bench_big.json)
"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
import json
CASES = ((2000, 400, 256, 1500), (2000, 200, 256, 1500), (4000, 84, 128, 700), (5000, 100, 128, 700), (6000, 300, 128, 700), (7000, 300, 64, 700),
         (10000, 1000, 64, 700), (10000, 1000, 256, 700), (14000, 1000, 64, 300))
args = [a for a in sys.argv[1:] if a != "--json" and not a.endswith(".json")]
out_json = next((a for a in sys.argv[1:] if a.endswith(".json")), None)
if len(args) >= 4:  # S P B N
    CASES = (tuple(int(x) for x in args[:4]),)
rows = []
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
    rows.append(dict(states=S, arcs=int(g.n_arcs), pdfs=P, B=B, T=N, ms_per_call=round(ms, 3), frames_per_s=B * N / ms * 1e3, redo_utterances=int(bf.last_redo_count()),
                     kernels=bf.kernels().split(" (")[0]))
    del bf, cf, V, gam
if out_json:
    json.dump(dict(what="lfmmi_denominator(S, P) graphs (workloads.py), randn emissions, full lengths; 5 timed calls after 2", rows=rows), open(out_json, "w"), indent=1)
