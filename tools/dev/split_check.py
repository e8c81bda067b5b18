"""Development aid: the fast path alone (MM_NO_REDO) against the oracle on a graph, with the redo count."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MM_DEBUG"] = "1"
os.environ["MM_NO_REDO"] = "1"
import __graft_entry__ as ge
import importlib, torch, graphs
mm = ge.load_package(); o, oc = ge.load_oracle()
wl = importlib.import_module(mm.__name__ + ".workloads")
name = sys.argv[1] if len(sys.argv) > 1 else "wsj"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 60
g = wl.load_npz_graph(os.path.join(ROOT, "tests/golden/den_fsm_wsj.npz")) if name == "wsj" else wl.lfmmi_denominator()
rng = np.random.default_rng(5)
V = rng.standard_normal((B, N, g.P)).astype(np.float32)
lens = np.full(B, N, np.int32); lens[1::3] = N - 7
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
print(bf.kernels())
gam, ttl = bf.pdfposteriors(torch.from_numpy(V).cuda(), torch.from_numpy(lens).cuda())
torch.cuda.synchronize()
print("redo", bf.last_redo_count())
gam, ttl = gam.cpu().numpy(), ttl.cpu().numpy()
nb = min(B, 4)
g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V[:nb], lens[:nb], dtype=np.float64, nthreads=4)
print("ttl", ttl[:nb], t_ref)
for b in range(nb):
    d = np.abs(gam[b] - g_ref[b])
    print(b, "max abs diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "nan", np.isnan(gam[b]).sum(), "sum/frame", gam[b].sum(1)[:5])
