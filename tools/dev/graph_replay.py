#!/usr/bin/env python3
"""config 3: one pdfposteriors call eager (back to back) against the same call captured in a hipGraph and replayed"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
g = wl.lfmmi_denominator(2000, 84, seed=0)
B, N = 256, 1500
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
V = torch.randn(B, N, g.P, device="cuda")
gam = torch.empty(B, N, g.P, device="cuda")
ttl = torch.empty(B, device="cuda")
for _ in range(5):
    bf.pdfposteriors(V, None, out=gam)
torch.cuda.synchronize()
def timed(f, n=50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
e1 = timed(lambda: bf.pdfposteriors(V, None, out=gam))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    bf.pdfposteriors(V, None, out=gam)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        bf.pdfposteriors(V, None, out=gam)
torch.cuda.synchronize()
gr.replay()
g1 = timed(gr.replay)
e2 = timed(lambda: bf.pdfposteriors(V, None, out=gam))
g2 = timed(gr.replay)
print(f"eager {e1:.3f} / {e2:.3f} ms, graph replay {g1:.3f} / {g2:.3f} ms")
