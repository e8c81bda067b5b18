#!/usr/bin/env python3
"""Diagnostic: where a step of the stream kernels goes, per wave of the first workgroup of every team (needs the -DMM_STAMPS build:
make -C markovmodels.jl_amd/csrc stamps).  S P B N from the environment; MM_DEBUG=1 MM_STREAM_H=n forces the team size."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_AMD_LIB", os.path.join(ROOT, "gpurun_stamps", "libmarkovmodels_amd_stamps.so"))
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402
mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
L = importlib.import_module(mm.__name__ + "._lib")
S, P, B, N = (int(os.environ.get(k, d)) for k, d in (("S", 10000), ("P", 1000), ("B", 64), ("N", 200)))
g = wl.lfmmi_denominator(S, P, seed=1)
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
V = torch.randn(B, N, g.P, device="cuda")
print(bf.kernels()[:160])
bf.pdfposteriors(V)
bf.pdfposteriors(V)
torch.cuda.synchronize()
n = B * 16 * 16
out = np.zeros(n, dtype=np.uint64)
L.lib.mm_debug_read_stamps.argtypes = [C.c_void_p, C.c_int64]
assert L.lib.mm_debug_read_stamps(out.ctypes.data, n) == 0
s = out.reshape(B, 2, 16, 8).astype(np.float64) / N
for d, dn in enumerate(("forward", "backward")):
    tot = s[:, d, :15, :3].sum(-1).mean()
    print(f"{dn}: cycles per step {tot:.0f} (100 MHz ticks x 24 if s_memtime counts the 100 MHz clock: see the ratio to the call's time)")
    for k, nm in enumerate(("arcs + finishes", "team's rows    ", "at the barrier ")):
        print("  ", nm, " ".join(f"{v:6.0f}" for v in s[:, d, :15, k].mean(0)))
    print("   service wave: staging %.0f, at the barrier %.0f" % (s[:, d, 15, 0].mean(), s[:, d, 15, 2].mean()))
