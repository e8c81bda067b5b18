import os, sys, importlib, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
mm = ge.load_package(); wl = importlib.import_module(mm.__name__ + ".workloads"); lf = importlib.import_module(mm.__name__ + ".lfmmi")
gold = "/root/repo/tests/golden"
den = wl.load_npz_graph(os.path.join(gold, "den_fsm_wsj.npz")); num = wl.load_npz_graph(os.path.join(gold, "num_fsm_wsj.npz"))
P, N, B = den.P, 60, 6
gs = [num, wl.lexicon_fsm(300, P, seed=2, hubs=1), num, wl.lexicon_fsm(500, P, seed=5, hubs=2), num, wl.lexicon_fsm(150, P, seed=9, hubs=1)]
cden = mm.compile(wl.to_fsm(mm, den), mm.statemap(den.state2pdf, P)); bden = mm.batch(*([cden] * B))
bnum = mm.batch(*[mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, P)) for g in gs])
bden.reserve(N); bnum.reserve(N); bden.set_exact_policy("f32_first")
lens = torch.tensor([N, N - 7, 31, N, 12, N], dtype=torch.int32, device="cuda")
V = torch.randn(B, N, P, device="cuda"); grad = torch.empty(B, N, P, device="cuda")
a = [lf.posteriors_difference(V, bnum, bden, lens, "fused") for _ in range(3)]
print("eager vs eager grad equal", torch.equal(a[0][0], a[1][0]), torch.equal(a[1][0], a[2][0]), "tn", torch.equal(a[0][1], a[1][1]), "td", torch.equal(a[0][2], a[1][2]))
d = [bden.pdfposteriors(V, lens) for _ in range(3)]
print("den alone equal", torch.equal(d[0][0], d[1][0]), torch.equal(d[1][0], d[2][0]), "redo", bden.last_redo_count())
n = [bnum.pdfposteriors(V, lens) for _ in range(3)]
print("num alone equal", torch.equal(n[0][0], n[1][0]))
x = (a[0][0] - a[1][0]).abs(); print("max diff", x.max().item(), "nan", torch.isnan(a[0][0]).sum().item(), "where", (x > 0).nonzero()[:5].tolist())
ref = d[0][0] - n[0][0]; print("fused vs separate max diff", (a[0][0] - ref).abs().max().item())
buf = torch.full((B, N, P), float("nan"), device="cuda")
bden.pdfposteriors(V, lens, out=buf); torch.cuda.synchronize()
print("den into NaN buffer: NaNs left", torch.isnan(buf).sum().item(), torch.isnan(buf).nonzero()[:6].tolist())
buf = torch.full((B, N, P), float("nan"), device="cuda")
lf.posteriors_difference(V, bnum, bden, lens, "fused", out=buf); torch.cuda.synchronize()
print("fused into NaN buffer: NaNs left", torch.isnan(buf).sum().item(), torch.isnan(buf).nonzero()[:6].tolist(), "equal to eager", torch.equal(buf, a[0][0]))
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    lf.posteriors_difference(V, bnum, bden, lens, "fused", out=grad)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    _, tn, td = lf.posteriors_difference(V, bnum, bden, lens, "fused", out=grad)
for k in range(3):
    grad.fill_(float("nan")); graph.replay(); torch.cuda.synchronize()
    e = lf.posteriors_difference(V, bnum, bden, lens, "fused")
    x = (grad - e[0]).abs()
    print("replay", k, "NaNs", torch.isnan(grad).sum().item(), "equal", torch.equal(grad, e[0]), "max diff", x[~torch.isnan(x)].max().item(), (x > 0).nonzero()[:4].tolist(), "tn", torch.equal(tn, e[1]), "td", torch.equal(td, e[2]))
rng = np.random.default_rng(5)
for k in range(3):
    Vn = rng.standard_normal((B, N, P)).astype(np.float32)
    V.copy_(torch.from_numpy(Vn))
    grad.fill_(float("nan")); graph.replay(); torch.cuda.synchronize()
    gg, tng, tdg = grad.clone(), tn.clone(), td.clone()
    e = lf.posteriors_difference(V, bnum, bden, lens, "fused")
    redo = bden.last_redo_count()
    x = (gg - e[0]).abs()
    bad = (x > 0) | torch.isnan(x)
    print("newV replay", k, "NaNs", torch.isnan(gg).sum().item(), "equal", torch.equal(gg, e[0]), "n diff", bad.sum().item(), "utterances", sorted(set(bad.nonzero()[:, 0].tolist())),
          "frames", sorted(set(bad.nonzero()[:, 1].tolist()))[:10], "max", x[~torch.isnan(x)].max().item(), "tn", torch.equal(tng, e[1]), "td", torch.equal(tdg, e[2]), "redo", redo)
    e2 = lf.posteriors_difference(V, bnum, bden, lens, "fused")
    print("   eager twice equal", torch.equal(e[0], e2[0]))
    d1 = bden.pdfposteriors(V, lens)[0]; n1 = bnum.pdfposteriors(V, lens)[0]
    print("   eager fused vs separate", (e[0] - (d1 - n1)).abs().max().item(), " graph vs separate", (gg - (d1 - n1)).abs().max().item())
