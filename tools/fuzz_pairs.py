#!/usr/bin/env python3
"""Diagnostic: the pair kernels (and the row kernels) against the item kernel -- an independent implementation -- on
random graphs, batch sizes, frame counts and length patterns, including the degenerate ones (one or two frames, empty
utterances, an odd number of utterances).  GPU only.  The switches are read at batch creation (MM_DEBUG)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402

mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")


def run(cf, B, V, lens, kernel):
    os.environ["MM_DEBUG"] = "1"
    if kernel:
        os.environ["MM_KERNEL"] = kernel
    else:
        os.environ.pop("MM_KERNEL", None)
    bf = mm.batch(*([cf] * B))
    gam, ttl = bf.pdfposteriors(V, lens)
    torch.cuda.synchronize()
    return gam.cpu().numpy().astype(np.float64), ttl.cpu().numpy().astype(np.float64), bf.kernels("log")


def main(seed=0):
    """Returns the number of mismatches."""
    rng = np.random.default_rng(seed)
    saved = {k: os.environ.get(k) for k in ("MM_DEBUG", "MM_KERNEL")}
    bad = n = 0
    graphs = [lambda: wl.random_fsm(13, 3, 2.0, seed=1), lambda: wl.random_fsm(300, 9, 2.5, seed=2),
              lambda: wl.lfmmi_denominator(600, 40, seed=5), lambda: wl.lfmmi_denominator(2000, 84, seed=0),
              lambda: wl.lexicon_fsm(1500, 30, seed=3), lambda: wl.dense_ergodic(64, seed=0), lambda: wl.wide_row_fsm(),
              # (251 .. 506 pdfs: the NJ = 8 instances of the pair kernels; no row forms there -- "row" falls to the engine's choice)
              lambda: wl.lfmmi_denominator(1700, 2 * int(rng.integers(126, 253)), seed=int(rng.integers(1 << 30)))]
    for gi, mk in enumerate(graphs):
        g = mk()
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        for B in (1, 2, 3, 5, 8):
            for N in (1, 2, 3, 4, 7, 40, 101):
                V = torch.from_numpy((2.0 * rng.standard_normal((B, N, g.P))).astype(np.float32)).cuda()
                pat = rng.integers(0, 3)
                lens = np.full(B, N) if pat == 0 else rng.integers(0, N + 1, B) if pat == 1 else rng.integers(max(0, N - 2), N + 1, B)
                lt = torch.from_numpy(lens.astype(np.int32)).cuda()
                ref_g, ref_t, _ = run(cf, B, V, lt, "item")
                for kern in ("pair", "row", None):  # (None: the engine's own choice -- the wave kernel for the small graphs)
                    a_g, a_t, names = run(cf, B, V, lt, kern)
                    n += 1
                    same_inf = np.isinf(a_t) & np.isinf(ref_t) & (a_t == ref_t)
                    fin = ~same_inf
                    et = (np.abs(a_t[fin] - ref_t[fin]) / np.maximum(1.0, np.abs(ref_t[fin]))).max() if fin.any() else 0.0
                    m = ref_g > 1e-30
                    eg = np.abs(a_g - ref_g).max()
                    if m.any():
                        eg = max(eg, (np.abs(np.log(np.maximum(a_g[m], 1e-300)) - np.log(ref_g[m])) / np.maximum(np.abs(np.log(ref_g[m])), 1)).max())
                    ok = np.isfinite(et) and np.isfinite(eg) and et < 1e-4 and eg < 1e-4
                    if not ok:
                        bad += 1
                        print(f"MISMATCH graph {gi} B {B} N {N} lens {lens.tolist()} kernel {kern or 'auto'} ({names[:40]}): ttl {et:.2e} gamma {eg:.2e}")
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    print(f"{n} comparisons, {bad} mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(os.environ.get("SEED", 0))) else 0)
