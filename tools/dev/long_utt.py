#!/usr/bin/env python3
"""long utterances on the pair kernels: redo count, time, and the difference to the item kernel"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
g = wl.lfmmi_denominator(2000, 84, seed=0)
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
B = 8
for N in (1500, 3000, 6000, 12000, 24000):
    V = torch.randn(B, N, g.P, device="cuda")
    bf = mm.batch(*([cf] * B))
    gam, ttl = bf.pdfposteriors(V)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gam, ttl = bf.pdfposteriors(V)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    os.environ["MM_DEBUG"] = "1"; os.environ["MM_KERNEL"] = "item"
    ref = mm.batch(*([cf] * B))
    os.environ.pop("MM_KERNEL")
    g_ref, t_ref = ref.pdfposteriors(V)
    torch.cuda.synchronize()
    print("T", N, "%.1f ms" % (dt * 1e3), "redone", bf.last_redo_count(), "of", B, "max |dgamma| %.2e" % (gam - g_ref).abs().max().item(),
          "ttl rel %.2e" % ((ttl - t_ref).abs() / t_ref.abs()).max().item(), bf.kernels()[:22], flush=True)
