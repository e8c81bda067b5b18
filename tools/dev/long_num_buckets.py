"""WSJ numerator, N = 6000: where the wave kernel's error against float64 sits -- by the size of the posterior"""
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
g = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz"))
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
N, B = 6000, 1
rng = np.random.default_rng(N)
V = rng.standard_normal((B, N, g.P)).astype(np.float32)
lens = np.array([N], dtype=np.int32)
g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64, nthreads=2)
gam, ttl = mm.batch(cf).pdfposteriors(V, lens)
gam = gam.astype(np.float64)
for lo, hi in ((1e-1, 2), (1e-2, 1e-1), (1e-4, 1e-2), (1e-8, 1e-4), (1e-16, 1e-8), (1e-24, 1e-16)):
    m = (g_ref >= lo) & (g_ref < hi)
    if not m.any(): continue
    dl = np.abs(np.log(np.maximum(gam[m], 1e-300)) - np.log(g_ref[m]))
    fr = np.nonzero(m)[1]
    k = np.argmax(dl)
    print(f"gamma in [{lo:g}, {hi:g}): {m.sum():7d} entries, max |d log gamma| {dl.max():.2e} (frame {fr[k]}), mean {dl.mean():.2e}, max |d gamma| {np.abs(gam[m]-g_ref[m]).max():.2e}")
# error of log gamma against the frame index, for posteriors above 1e-4
m = g_ref >= 1e-4
dl = np.zeros_like(g_ref); dl[m] = np.abs(np.log(gam[m]) - np.log(g_ref[m]))
per = dl[0].max(axis=1)
print("max |d log gamma| over the posteriors above 1e-4, by thousand frames:", [f"{per[k:k+1000].max():.1e}" for k in range(0, N, 1000)])
