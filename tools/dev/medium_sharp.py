#!/usr/bin/env python3
"""Medium-sharp emissions (log-softmax of sigma * N(0,1), sigma = 2 .. 6) on config 3's graph and the WSJ denominator: the engine's own
choice against the exact kernels first (MM_EXACT_F64_FIRST) -- posteriors above 1e-30 within the parity bar, redo counts."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
graphs = [("lfmmi_den2000", wl.lfmmi_denominator(2000, 84, seed=0)),
          ("den_fsm_wsj", wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz")))]
B, N = 8, 300
bad = 0
for name, g in graphs:
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    rng = np.random.default_rng(5)
    lens = np.array([N, N, 257, N, 100, N, 33, N], dtype=np.int32)
    for sigma in (1, 2, 3, 4, 5, 6, 8):
        V = sigma * rng.standard_normal((B, N, g.P))
        V = V - np.log(np.exp(V - V.max(-1, keepdims=True)).sum(-1, keepdims=True)) - V.max(-1, keepdims=True)
        V = V.astype(np.float32)
        res = {}
        for pol in ("auto", "f64_first"):
            bf = mm.batch(*([cf] * B))
            bf.set_exact_policy(pol)
            gam, ttl = bf.pdfposteriors(V, lens)
            res[pol] = (np.asarray(gam, dtype=np.float64), np.asarray(ttl, dtype=np.float64), bf.last_redo_count(), bf.last_fallback_count() if hasattr(bf, "last_fallback_count") else -1)
        ga, ta, ra, fa = res["auto"]
        ge_, te, re_, fe = res["f64_first"]
        m = ge_ > 1e-30
        with np.errstate(divide="ignore", invalid="ignore"):
            d = np.abs(np.log(ga[m]) - np.log(ge_[m])) / np.maximum(np.abs(np.log(ge_[m])), 1.0)
        worst = float(d.max()) if d.size else 0.0
        lost = int((ga[m] == 0).sum())
        dt = float(np.abs(ta - te).max() / max(1.0, np.abs(te).max()))
        flag = "" if (worst <= 1e-4 and lost == 0 and dt <= 1e-5) else "   <-- BAD"
        bad += bool(flag)
        print(f"{name} sigma {sigma}: auto redo {ra}, exact redo {re_}; worst rel dlog gamma {worst:.2e}, zeros where exact > 1e-30: {lost}, ttl rel {dt:.1e}{flag}", flush=True)
sys.exit(1 if bad else 0)
