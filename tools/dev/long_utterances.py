"""Long utterances (N = 6000 frames; the tracked workloads stop at 1500) through the kernel families against the item kernel."""
import importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import torch
from fuzz_round3 import posterior_error
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
def batch_with(kernel, cfs, extra=None):
    env = {"MM_DEBUG": "1"}
    if kernel: env["MM_KERNEL"] = kernel
    env.update(extra or {})
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try: return mm.batch(*cfs)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
here = os.path.join(ROOT, "tests", "golden")
cases = [("config 3 graph (pair kernels)", wl.lfmmi_denominator(2000, 84, seed=0), None, None),
         ("WSJ denominator (teams of 2)", wl.load_npz_graph(os.path.join(here, "den_fsm_wsj.npz")), None, None),
         ("4000 states (teams of 4)", wl.lfmmi_denominator(4000, 84, seed=1), None, None),
         ("7000 states / 300 pdfs (stream, teams of 4)", wl.lfmmi_denominator(7000, 300, seed=2), "stream", {"MM_STREAM_H": "4"}),
         ("WSJ numerator x3 (wave kernel)", wl.load_npz_graph(os.path.join(here, "num_fsm_wsj.npz")), None, None)]
N, B = 6000, 3
gen = torch.Generator(device="cuda").manual_seed(1)
for name, g, kern, extra in cases:
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    V = torch.randn(B, N, g.P, device="cuda", generator=gen)
    lens = torch.tensor([N, N - 1234, 4001], dtype=torch.int32, device="cuda")
    ref = batch_with("item", [cf] * B)
    rg, rt = ref.pdfposteriors(V, lens)
    bf = batch_with(kern, [cf] * B, extra)
    a, t = bf.pdfposteriors(V, lens)
    et, eg = posterior_error(a.cpu().numpy().astype(np.float64), t.cpu().numpy().astype(np.float64), rg.cpu().numpy().astype(np.float64), rt.cpu().numpy().astype(np.float64))
    line = f"{name}: {bf.kernels()[:28]}  ttl {et:.1e} gamma {eg:.1e} redo {bf.last_redo_count()}"
    if "mm_pair_export_kernel" in bf.kernels("export"):
        for fn in ("alpharecursion", "betarecursion"):
            X, R = getattr(bf, fn)(V, lens), getattr(ref, fn)(V, lens)
            m = torch.isfinite(R)
            line += f" | {fn[:5]} -inf pattern {bool(torch.equal(torch.isfinite(X), m))} max abs {float((X[m] - R[m]).abs().max()):.1e} redo {bf.last_redo_count()}"
    print(line, flush=True)
