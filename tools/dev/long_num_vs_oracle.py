"""Long utterances against the float64 oracle: how the float32 paths drift with N -- the wave kernel and the item kernel (log domain),
the pair / team kernels (linear domain), and the reference's own arithmetic (the C oracle in float32: its operation order)."""
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
for gname, g in (("WSJ numerator", wl.load_npz_graph(os.path.join(here, "num_fsm_wsj.npz"))), ("config 3 graph", wl.lfmmi_denominator(2000, 84, seed=0)),
                 ("WSJ denominator", wl.load_npz_graph(os.path.join(here, "den_fsm_wsj.npz")))):
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    for N in (700, 1500, 3000, 6000):
        B = 2
        rng = np.random.default_rng(N)
        V = rng.standard_normal((B, N, g.P)).astype(np.float32)
        lens = np.array([N, N - N // 5], dtype=np.int32)
        of = graphs.to_oracle(o, g)
        g_ref, t_ref = oc.batch_shared(of, g.state2pdf, g.P, V, lens, dtype=np.float64, nthreads=2)
        g_f32, t_f32 = oc.batch_shared(graphs.to_oracle(o, g, "log", np.float32), g.state2pdf, g.P, V, lens, dtype=np.float32, nthreads=2)
        os.environ.update({"MM_DEBUG": "1", "MM_KERNEL": "item"}); bi = mm.batch(*([cf] * B)); os.environ.pop("MM_KERNEL"); os.environ.pop("MM_DEBUG")
        bw = mm.batch(*([cf] * B))
        line = f"{gname:16s} N {N:5d}:"
        for name, bf in (("default", bw), ("item", bi)):
            gam, ttl = bf.pdfposteriors(V, lens)
            r, a = err(gam.astype(np.float64), g_ref)
            line += f"  {name} ({bf.kernels()[:14]}) {r:.1e} / {a:.1e}"
        r, a = err(g_f32.astype(np.float64), g_ref)
        line += f"  reference's order in float32 {r:.1e} / {a:.1e} (ttl rel {np.abs((t_f32 - t_ref) / t_ref).max():.1e})"
        print(line, flush=True)
