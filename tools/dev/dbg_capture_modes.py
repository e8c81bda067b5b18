import importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
lf = importlib.import_module(mm.__name__ + ".lfmmi")
den = wl.lfmmi_denominator(1200, 30, seed=3)
P, N, B = den.P, 40, 6
gs = [wl.lexicon_fsm(150 + 20 * b, P, seed=b, hubs=1) for b in range(B)]
bden = mm.batch(*([mm.compile(wl.to_fsm(mm, den), mm.statemap(den.state2pdf, P))] * B))
bnum = mm.batch(*[mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, P)) for g in gs])
bden.reserve(N); bnum.reserve(N); bden.set_exact_policy("f32_first")
lens = torch.tensor([N, N - 7, 31, N, 12, N], dtype=torch.int32, device="cuda")
V = torch.randn(B, N, P, device="cuda")
for mode in ("serial", "concurrent", "fused"):
    grad = torch.empty(B, N, P, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        lf.posteriors_difference(V, bnum, bden, lens, mode, out=grad)
    torch.cuda.synchronize()
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            _, tn, td = lf.posteriors_difference(V, bnum, bden, lens, mode, out=grad)
        V.copy_(torch.randn(B, N, P, device="cuda"))
        grad.fill_(float("nan"))
        graph.replay(); torch.cuda.synchronize()
        gg = grad.clone()
        ge_, tne, tde = lf.posteriors_difference(V, bnum, bden, lens, mode)
        print(mode, "captured; replay == eager:", bool(torch.equal(gg, ge_)), float((gg - ge_).abs().max()))
    except Exception as e:
        print(mode, "capture FAILED:", type(e).__name__, str(e)[:200])
        torch.cuda.synchronize()
