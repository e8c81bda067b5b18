import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import importlib, graphs
mm = ge.load_package(); wl = importlib.import_module(mm.__name__ + ".workloads"); o, oc = ge.load_oracle()
sigma = float(os.environ.get("SIGMA", "10"))
g = wl.lfmmi_denominator(2000, 84, seed=0)
rng = np.random.default_rng(int(sigma))
B, N = 7, 130
lens = np.array([130, 130, 87, 1, 45, 0, 129], dtype=np.int32)
x = sigma * rng.standard_normal((B, N, g.P))
V = (x - np.log(np.exp(x - x.max(-1, keepdims=True)).sum(-1, keepdims=True)) - x.max(-1, keepdims=True)).astype(np.float32)
bf = mm.batch(*([mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * B))
g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64, nthreads=4)
for it in range(2):
    gam, ttl = bf.pdfposteriors(V, lens)
    print("call", it, "exact_first", bf.last_exact_first(), "redo", bf.last_redo_count(), "fallback", bf.last_fallback_count())
    for b in range(B):
        L = lens[b]
        m = g_ref[b] > 1e-30
        z = m & (gam[b] == 0)
        rel = np.abs(np.log(np.maximum(gam[b][m & ~z], 1e-300)) - np.log(g_ref[b][m & ~z])) / np.maximum(np.abs(np.log(g_ref[b][m & ~z])), 1)
        print(b, "len", L, "ttl", ttl[b], t_ref[b], "zeros where ref>1e-30:", int(z.sum()), "largest ref there", g_ref[b][z].max() if z.any() else 0,
              "max rel log err", rel.max() if rel.size else 0, "frames with zeros", np.unique(np.nonzero(z)[0])[:10])
