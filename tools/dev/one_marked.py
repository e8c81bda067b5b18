#!/usr/bin/env python3
"""What ONE hard utterance costs a call of config 3: 255 utterances of N(0,1) log-likelihoods and one of log-softmax(10 N(0,1)), under the
fixed policy "float32 first" (the policy a mixed batch ends up with) -- ms per call against the all-easy batch."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
g = wl.lfmmi_denominator(2000, 84, seed=0)
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
B, N = 256, 1500
bf = mm.batch(*([cf] * B))
bf.set_exact_policy("f32_first")
V = torch.randn(B, N, g.P, device="cuda")
gam = torch.empty(B, N, g.P, device="cuda")
def run(tag):
    for _ in range(3):
        bf.pdfposteriors(V, None, out=gam)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        bf.pdfposteriors(V, None, out=gam)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 100
    print(f"{tag}: {ms:.2f} ms per call, redone {bf.last_redo_count()}", flush=True)
    return ms
a = run("all easy")
x = 10.0 * torch.randn(N, g.P, device="cuda")
V[7] = torch.log_softmax(x, dim=-1)
b = run("one hard utterance")
print(f"one marked utterance: +{b - a:.2f} ms")
