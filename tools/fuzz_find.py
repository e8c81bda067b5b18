#!/usr/bin/env python3
"""Which cases of the team fuzzer's stream (tests/fuzz_cases.py) miss SURVEY 8(d)'s bar -- |d log gamma| <= 1e-4 max(|log gamma|, 1)
wherever gamma_ref > 1e-30 -- on the default path, and what the strict settings do with them (mark policy "keep"; exact policy
"f64_first").  Reference: the item kernel (log domain).      SEEDS=1,3 python tools/fuzz_find.py      (GPU box)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402
from fuzz_cases import split_cases  # noqa: E402

mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")


def strict_error(a, ref):
    m = ref > 1e-30
    if not m.any():
        return 0.0, None
    rel = np.zeros_like(ref)
    rel[m] = np.abs(np.log(np.maximum(a[m], 1e-300)) - np.log(ref[m])) / np.maximum(np.abs(np.log(ref[m])), 1)
    idx = np.unravel_index(np.argmax(rel), rel.shape)
    return float(rel[idx]), idx


def run(cf, B, V, lt, kernel=None, marks=None, exact=None):
    os.environ["MM_DEBUG"] = "1"
    if kernel:
        os.environ["MM_KERNEL"] = kernel
    else:
        os.environ.pop("MM_KERNEL", None)
    bf = mm.batch(*([cf] * B))
    if marks:
        bf.set_mark_policy(marks)
    if exact:
        bf.set_exact_policy(exact)
    g, t = bf.pdfposteriors(V, lt)
    torch.cuda.synchronize()
    return g.cpu().numpy().astype(np.float64), bf


for seed in [int(x) for x in os.environ.get("SEEDS", "1,3").split(",")]:
    last = None
    for gi, g, B, N, V0, sharp, lens in split_cases(wl, seed):
        if last != gi:
            cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
            last = gi
        V = torch.from_numpy(V0).cuda()
        if sharp:
            V = torch.log_softmax(8.0 * V, dim=-1)
        lt = torch.from_numpy(lens).cuda()
        ref, _ = run(cf, B, V, lt, "item")
        a, bf = run(cf, B, V, lt)
        e, idx = strict_error(a, ref)
        if e > 1e-4:
            print(f"seed {seed} graph {gi} ({g.name}, P={g.P}) B {B} N {N} sharp {sharp}: default misses the strict bar: {e:.3e} at {idx}: reference {ref[idx]:.6e} "
                  f"computed {a[idx]:.6e}; length {int(lens[idx[0]])}; redo {bf.last_redo_count()} {bf.kernels()[:30]}", flush=True)
            for marks, exact in (("keep", None), (None, "f64_first"), ("keep", "f32_first")):
                a2, b2 = run(cf, B, V, lt, None, marks, exact)
                e2, i2 = strict_error(a2, ref)
                print(f"      marks={marks} exact={exact}: {e2:.3e} (computed {a2[idx]:.6e}), redo {b2.last_redo_count()} fallback {b2.last_fallback_count()}", flush=True)
    print(f"seed {seed}: done", flush=True)
