import importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
g = wl.lfmmi_denominator(900, 40, seed=31)
B, N = 5, 33
S1 = g.S + 1
lens = torch.tensor([N, N - 4, 9, 1, N], dtype=torch.int32, device="cuda")
V = torch.randn(B, N, g.P, device="cuda")
def fresh():
    bf = mm.batch(*([mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * B))
    bf.reserve(N)
    return bf
def same(x, y): return [int((x[b*S1:(b+1)*S1] != y[b*S1:(b+1)*S1]).sum()) for b in range(B)]
bf = fresh()
a = [bf.alpharecursion(V, lens).clone() for _ in range(5)]
print("alpha x5, each against the first:", [same(a[0], x) for x in a[1:]])
bf = fresh()
a0 = bf.alpharecursion(V, lens).clone(); _ = bf.pdfposteriors(V, lens); a1 = bf.alpharecursion(V, lens).clone()
print("alpha, pdfposteriors, alpha:", same(a0, a1))
bf = fresh()
a0 = bf.alpharecursion(V, lens).clone(); _ = bf.betarecursion(V, lens); a1 = bf.alpharecursion(V, lens).clone(); a2 = bf.alpharecursion(V, lens).clone()
print("alpha, beta, alpha, alpha:", same(a0, a1), same(a1, a2))
bf = fresh()
b0 = bf.betarecursion(V, lens).clone(); _ = bf.alpharecursion(V, lens); b1 = bf.betarecursion(V, lens).clone()
print("beta, alpha, beta:", same(b0, b1))
# which one does the item kernel agree with more closely?
os.environ.update({"MM_DEBUG": "1", "MM_KERNEL": "item"})
bi = mm.batch(*([mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * B))
os.environ.pop("MM_KERNEL"); os.environ.pop("MM_DEBUG")
ai = bi.alpharecursion(V, lens)
for name, x in (("first", a0), ("later", a1)):
    d = (x - ai).abs().nan_to_num(nan=0.0, posinf=0.0)
    print(name, "vs item kernel, max abs per utterance:", [float(d[b*S1:(b+1)*S1].max()) for b in range(B)])
idx = (a0 != a1).nonzero()
cols = sorted(set(int(j) for _, j in idx.tolist()))
print("columns that differ:", cols[:40])
