#!/usr/bin/env python3
"""the one mismatch of SEED=3 tools/fuzz_round3.py split (WSJ graph, B = 515, N = 9): which posterior, how far off"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fuzz_round3 as fz
import torch
rng = np.random.default_rng(3)
wl, mm = fz.wl, fz.mm
g = wl.load_npz_graph(os.path.join(fz.ROOT, "tests", "golden", "den_fsm_wsj.npz"))
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
for B, N in ((1, 1), (1, 6), (1, 61), (2, 1), (2, 2), (3, 3), (3, 4), (5, 7), (8, 40), (9, 101), (300, 12), (515, 9)):
    V = torch.from_numpy((1.5 * rng.standard_normal((B, N, g.P))).astype(np.float32)).cuda()
    sharp = rng.integers(0, 3) == 0
    if sharp:
        V = torch.log_softmax(8.0 * V, dim=-1)
    lens = fz.lens_pattern(rng, B, N)
    lt = torch.from_numpy(lens).cuda()
    if (B, N) != (515, 9):
        continue
    ref_g, ref_t, _, _ = fz.posteriors([cf] * B, V, lt, "item")
    a_g, a_t, names, bf = fz.posteriors([cf] * B, V, lt, None)
    m = ref_g > 1e-30
    err = np.zeros_like(ref_g)
    err[m] = np.abs(np.log(np.maximum(a_g[m], 1e-300)) - np.log(ref_g[m])) / np.maximum(np.abs(np.log(ref_g[m])), 1)
    idx = np.argsort(err.ravel())[-5:]
    print("sharp", sharp, "redo", bf.last_redo_count(), "exact first", bf.last_exact_first())
    for i in idx:
        b, n, p = np.unravel_index(i, ref_g.shape)
        print(f"utt {b} frame {n} pdf {p}: ref {ref_g[b, n, p]:.4e} got {a_g[b, n, p]:.4e} rel log err {err[b, n, p]:.2e}")
