#!/usr/bin/env python3
"""Diagnostic: where a frame of the Viterbi kernel goes, per wave (needs the -DMM_STAMPS build: make -C markovmodels.jl_amd/csrc stamps)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MM_AMD_LIB", os.path.join(ROOT, "gpurun_stamps", "libmarkovmodels_amd_stamps.so"))
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")
L = importlib.import_module(mm.__name__ + "._lib")
g, B = wl.lexicon_fsm(5000, 84, seed=0), int(os.environ.get("B", 128))
N = int(os.environ.get("N", 1000))
cf = mm.compile(wl.to_fsm(mm, g, "tropical"), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
V = torch.randn(B, N, g.P, device="cuda")
print(bf.kernels("tropical"))
bf.viterbi(V)
bf.viterbi(V)
torch.cuda.synchronize()
n = B * 16 * 16
out = np.zeros(n, dtype=np.uint64)
L.lib.mm_debug_read_stamps.argtypes = [C.c_void_p, C.c_int64]
assert L.lib.mm_debug_read_stamps(out.ctypes.data, n) == 0
s = out[: B * 16 * 8].reshape(B, 16, 8).astype(np.float64) / N
print("cycles per frame, mean over the utterances: work (to the barrier) | in the barrier | (service: wait for the DMA)")
for wv in range(16):
    print("wave %2d  work %6.0f  barrier %6.0f  dma wait %6.0f  total %6.0f" % (wv, s[:, wv, 0].mean(), s[:, wv, 1].mean(), s[:, wv, 2].mean(), s[:, wv, :3].sum(-1).mean()))
