#!/usr/bin/env python3
"""Diagnostic: the quad kernels against the item kernel (independent implementation) on random graphs of
many shapes -- every geometry (quads per lane, waves, rows per thread) gets exercised.  GPU only."""
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import __graft_entry__ as ge
    import torch

    mm = ge.load_package()
    wl = importlib.import_module(mm.__name__ + ".workloads")
    out = {}
    rng = np.random.default_rng(0)
    cases = [(13, 3, 2.0), (70, 5, 3.0), (300, 9, 2.3), (700, 20, 6.0), (1500, 40, 12.0), (2500, 84, 17.0),
             (3000, 84, 4.0), (4500, 60, 9.0), (6000, 84, 3.0), (900, 84, 40.0)]
    for ci, (S, P, deg) in enumerate(cases):
        g = wl.random_fsm(S, P, deg, seed=ci)
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        B, N = 3, 40
        V = torch.from_numpy((2.0 * rng.standard_normal((B, N, g.P))).astype(np.float32)).cuda()
        lens = torch.tensor([N, N // 2, 1], dtype=torch.int32).cuda()
        gam, ttl = mm.batch(*([cf] * B)).pdfposteriors(V, lens)
        out[f"g{ci}"] = gam.cpu().numpy()
        out[f"t{ci}"] = ttl.cpu().numpy()
    np.savez(sys.argv[2], **out)
    sys.exit(0)

res = {}
for kern in ("quad", "item"):
    env = dict(os.environ)
    env["MM_DEBUG"] = "1"
    env["MM_KERNEL"] = kern  # ("quad" switches the depth heuristic off)
    f = f"/tmp/fuzz_{kern}.npz"
    subprocess.check_call([sys.executable, __file__, "child", f], env=env)
    res[kern] = np.load(f)
bad = 0
for k in res["quad"].files:
    a, b = res["quad"][k].astype(np.float64), res["item"][k].astype(np.float64)
    if k.startswith("t"):
        same_inf = np.isinf(a) & np.isinf(b) & (a == b)
        fin = ~same_inf
        err = (np.abs(a[fin] - b[fin]).max() / max(1.0, np.abs(b[fin]).max())) if fin.any() else 0.0
        err = err if np.isfinite(err) else 1.0
    else:
        m = b > 1e-30
        err = max(np.abs(a - b).max(), (np.abs(np.log(np.maximum(a[m], 1e-300)) - np.log(b[m])) / np.maximum(np.abs(np.log(b[m])), 1)).max())
    flag = "" if err < 1e-4 else "   <-- MISMATCH"
    bad += err >= 1e-4
    print(f"{k}: max err {err:.2e}{flag}")
print("FAILED" if bad else "all geometries agree")
sys.exit(1 if bad else 0)
