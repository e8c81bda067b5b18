"""Sharp emissions (log-softmax of 10 N(0,1)) on long utterances: the wide pair kernels (20-bit mantissas, float64 accumulation) and the
float64 pair kernels against the float64 oracle"""
import importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import torch
import graphs
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
o = importlib.import_module("oracle.mm_oracle"); oc = importlib.import_module("oracle.mm_oracle_c")
here = os.path.join(ROOT, "tests", "golden")
def err(gam, g_ref):
    m = g_ref > 1e-24
    rel = (np.abs(np.log(np.maximum(gam[m], 1e-300)) - np.log(g_ref[m])) / np.maximum(np.abs(np.log(g_ref[m])), 1)).max()
    return rel, np.abs(gam - g_ref).max()
for gname, g in (("config 3 graph", wl.lfmmi_denominator(2000, 84, seed=0)), ("WSJ denominator", wl.load_npz_graph(os.path.join(here, "den_fsm_wsj.npz")))):
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    for N in (700, 1500, 6000):
        B = 2
        rng = np.random.default_rng(N)
        x = 10.0 * rng.standard_normal((B, N, g.P))
        mx = x.max(-1, keepdims=True)
        V = (x - mx - np.log(np.exp(x - mx).sum(-1, keepdims=True))).astype(np.float32)
        lens = np.array([N, N - N // 5], dtype=np.int32)
        g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64, nthreads=2)
        line = f"{gname:16s} N {N:5d}:"
        for pol in ("f32_first", "f64_first"):
            bf = mm.batch(*([cf] * B)).set_exact_policy(pol)
            gam, ttl = bf.pdfposteriors(V, lens)
            r, a = err(gam.astype(np.float64), g_ref)
            line += f"  {pol}: {r:.1e} / {a:.1e} (redo {bf.last_redo_count()}, fallback {bf.last_fallback_count()}, ttl rel {np.abs((ttl - t_ref) / t_ref).max():.1e})"
        print(line, flush=True)
