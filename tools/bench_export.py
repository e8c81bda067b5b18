#!/usr/bin/env python3
"""alpha-recursion / beta-recursion export (mm_alpharecursion_f32 / mm_betarecursion_f32, src/inference.jl:62-74, 99-110) at config 3's
size next to the pdfposteriors call of the same batch: ms per call, which kernels ran, GB/s of the exported matrix.
    python tools/bench_export.py [out.json]      (GPU box)"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge
import torch
from srchash import source_hash
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
L = importlib.import_module(mm.__name__ + "._lib")


def timed(fn, K=10, W=3):
    for _ in range(W):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


rows = []
for name, g, B, N in (("config 3 (lfmmi_den)", wl.lfmmi_denominator(2000, 84, seed=0), 256, 1500), ("WSJ denominator", wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz")), 128, 700),
                      ("lfmmi_den4000 (teams of 4)", wl.lfmmi_denominator(4000, 84, seed=1), 128, 700)):
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    V = torch.randn(B, N, g.P, device="cuda")
    gam = torch.empty(B, N, g.P, device="cuda")
    out = torch.empty((N + 1, bf.total_states), dtype=torch.float32, device="cuda")
    st = lambda: torch.cuda.current_stream().cuda_stream

    def export(fn):
        L.check(fn(bf._h, V.data_ptr(), V.stride(0), V.stride(1), None, N, out.data_ptr(), out.stride(0), st()))

    t_post = timed(lambda: bf.pdfposteriors(V, None, out=gam))
    t_a = timed(lambda: export(L.lib.mm_alpharecursion_f32))
    redo_a = bf.last_redo_count()
    t_b = timed(lambda: export(L.lib.mm_betarecursion_f32))
    redo_b = bf.last_redo_count()
    bytes_out = out.numel() * 4
    rows.append(dict(workload=name, states=g.S, pdfs=g.P, B=B, T=N, pdfposteriors_ms=t_post, alpharecursion_ms=t_a, betarecursion_ms=t_b,
                     alpha_over_pdfposteriors=t_a / t_post, beta_over_pdfposteriors=t_b / t_post, exported_GB=bytes_out / 1e9,
                     alpha_export_GBps=bytes_out / t_a / 1e6, item_kernel_utterances=[redo_a, redo_b], kernels=bf.kernels("export")))
    print(rows[-1], flush=True)
    del bf, V, gam, out
if len(sys.argv) > 1:
    json.dump(dict(source_hash=source_hash(), rows=rows), open(sys.argv[1], "w"), indent=1)
