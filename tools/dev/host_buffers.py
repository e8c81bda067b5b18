#!/usr/bin/env python3
"""config 3 with HOST buffers at the boundary (numpy in, numpy out: what the reference-style entry of the host mirror does):
the PCIe-inclusive time per call next to the device-resident one."""
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
Vh = np.random.default_rng(0).standard_normal((B, N, g.P)).astype(np.float32)
Vd = torch.from_numpy(Vh).cuda()
out = torch.empty(B, N, g.P, device="cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
dev = t(lambda: bf.pdfposteriors(Vd, None, out=out), 20)
host = t(lambda: bf.pdfposteriors(Vh, None))
Vp = torch.from_numpy(Vh).pin_memory(); gp = torch.empty(B, N, g.P).pin_memory()
def pinned():
    bf.pdfposteriors(Vp.cuda(non_blocking=True), None, out=out); gp.copy_(out, non_blocking=True)
pin = t(pinned)
mb = Vh.nbytes / 1e6
print(f"device-resident {dev:.2f} ms; pageable numpy in/out {host:.1f} ms; pinned host buffers, async copies {pin:.1f} ms ({mb:.0f} MB each way)")
print(f"frames/s: {B*N/dev*1e3:.3g} / {B*N/host*1e3:.3g} / {B*N/pin*1e3:.3g}")
