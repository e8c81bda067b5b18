#!/usr/bin/env python3
"""Debug aid: the wide pair kernels alone (MM_EXACT_FIRST=1, MM_NO_FALLBACK=1) against the float64 oracle, per utterance."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.update(MM_DEBUG="1", MM_EXACT_FIRST="1", MM_NO_FALLBACK="1")
import __graft_entry__ as ge
import graphs
mm = ge.load_package(); o, oc = ge.load_oracle()
wl = importlib.import_module(mm.__name__ + ".workloads")
def peaky(rng, shape, sigma):
    x = sigma * rng.standard_normal(shape)
    return (x - np.log(np.exp(x - x.max(-1, keepdims=True)).sum(-1, keepdims=True)) - x.max(-1, keepdims=True)).astype(np.float32)
g = wl.lfmmi_denominator(2000, 84, seed=0)
for name, lens, sigma in (("even", [130] * 4, 10.0), ("mixed", [130, 130, 87, 1, 45, 0, 129], 10.0), ("randn", [130, 100, 64, 130], 0.0), ("s25", [130] * 4, 25.0)):
    rng = np.random.default_rng(1)
    lens = np.array(lens, dtype=np.int32); B, N = len(lens), 130
    V = peaky(rng, (B, N, g.P), sigma) if sigma else rng.standard_normal((B, N, g.P)).astype(np.float32)
    bf = mm.batch(*([mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * B))
    gam, ttl = bf.pdfposteriors(V, lens)
    print(name, "exact_first", bf.last_exact_first(), "redo", bf.last_redo_count(), "fallback", bf.last_fallback_count())
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64, nthreads=4)
    for b in range(B):
        L = lens[b]
        if L == 0: print("  utt", b, "len 0 ttl", ttl[b], t_ref[b]); continue
        m = g_ref[b, :L] > 1e-30
        if not m.any(): print("  utt", b, "len", L, "no posterior above 1e-30; ttl", ttl[b], t_ref[b]); continue
        d = np.abs(np.log(np.maximum(gam[b, :L][m], 1e-300)) - np.log(g_ref[b, :L][m])) / np.maximum(np.abs(np.log(g_ref[b, :L][m])), 1)
        print("  utt", b, "len", L, "ttl", ttl[b], t_ref[b], "max rel dlog gamma %.3g" % d.max(), "at frame", np.unravel_index(d.argmax(), gam[b, :L][m].shape) if False else int(np.argwhere(m)[d.argmax()][0]), "nan", int(np.isnan(gam[b]).sum()))
