#!/usr/bin/env python3
"""How sharp may the emissions be before the linear-domain kernels hand utterances to the exact ones?  Two families on config 3
(B = 256, T = 1500) and on the reference's WSJ denominator (B = 128, T = 700):
  inconsistent   log-softmax(sigma * N(0,1))                                        (sharp, and not a path of the graph)
  consistent     log-softmax(sigma * (onehot(pdf of a sampled path) + 0.3 N(0,1)))  (sharp along a path: a trained acoustic model)
Per row: ms per call under the engine's own policy (auto), utterances on the exact kernels, whether the call skipped the float32
kernels; and -- fresh batch, policy f32_first -- how many utterances the float32 kernels mark (MARKED).
    [FLOOR=1e-12] python tools/sharpness.py       (GPU box; tools/measure_all.sh -> profiles/<tag>_sharpness.txt)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
cases = [(wl.lfmmi_denominator(2000, 84, seed=0), 256, 1500), (wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "den_fsm_wsj.npz")), 128, 700)]
verbose = os.environ.get("VERBOSE")
for g, B, N in cases:
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    gam = torch.empty(B, N, g.P, device="cuda")
    for family, sigmas in (("inconsistent", (1, 2, 3, 4, 5, 6, 8, 10)), ("consistent", (1, 3, 10, 30))):
        for sigma in sigmas:
            if family == "consistent":
                V = torch.from_numpy(wl.path_consistent_emissions(g, B, N, float(sigma), seed=11)).cuda()
            else:
                V = torch.log_softmax(sigma * torch.randn(B, N, g.P, device="cuda"), dim=-1)
            bf = mm.batch(*([cf] * B))
            if os.environ.get("FLOOR"):
                bf.set_posterior_floor(float(os.environ["FLOOR"]))
            for _ in range(3):
                bf.pdfposteriors(V, None, out=gam)
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                bf.pdfposteriors(V, None, out=gam)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 200
            redone, first = bf.last_redo_count(), bf.last_exact_first()
            if verbose:
                os.environ["MM_DEBUG"] = os.environ["MM_VERBOSE"] = "1"
            b2 = mm.batch(*([cf] * B)).set_exact_policy("f32_first")
            if os.environ.get("FLOOR"):
                b2.set_posterior_floor(float(os.environ["FLOOR"]))
            b2.pdfposteriors(V, None, out=gam)
            marked = b2.last_redo_count()
            os.environ.pop("MM_VERBOSE", None)
            print(f"{g.name} {family} sigma {sigma}: {ms:.2f} ms, on the exact kernels {redone} of {B}, exact first {first}; MARKED by the float32 kernels {marked}", flush=True)
            del bf, b2
