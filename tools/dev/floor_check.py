#!/usr/bin/env python3
"""posterior floor 1e-12 on sharp inputs at full size: the fast kernels against the item kernel (log domain, exact)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
cases = [(wl.lfmmi_denominator(2000, 84, seed=0), 64, 1500), (wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz")), 32, 700)]
for g, B, N in cases:
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    for sigma in (4, 6, 8):
        V = torch.log_softmax(sigma * torch.randn(B, N, g.P, device="cuda"), dim=-1)
        os.environ["MM_DEBUG"] = "1"; os.environ["MM_KERNEL"] = "item"
        ref = mm.batch(*([cf] * B))
        g_ref, t_ref = ref.pdfposteriors(V)
        os.environ.pop("MM_KERNEL")
        bf = mm.batch(*([cf] * B)).set_posterior_floor(1e-12)
        g1, t1 = bf.pdfposteriors(V)
        torch.cuda.synchronize()
        d = (g1 - g_ref).abs()
        small = g_ref < 1e-12
        big = g_ref > 1e-10
        rel = ((g1[big].double().log() - g_ref[big].double().log()).abs() / g_ref[big].double().log().abs().clamp(min=1)).max().item()
        print(g.name, "sigma", sigma, "redone", bf.last_redo_count(), "max |dgamma| %.2e" % d.max().item(), "max |dgamma| where ref < 1e-12: %.2e" % d[small].max().item(),
              "rel log err where ref > 1e-10: %.2e" % rel, "ttl rel %.2e" % ((t1 - t_ref).abs() / t_ref.abs()).max().item(), flush=True)
