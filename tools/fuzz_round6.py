#!/usr/bin/env python3
"""Diagnostic: what round 6 added, against the item kernel (an independent implementation in the log domain) on random graphs, batch
sizes, frame counts and length patterns, the degenerate ones included:

  * export   alpha-recursion / beta-recursion on the pair and team kernels (mm_fbx_kernel / mm_fbsx_kernel + mm_pair_export_kernel):
             the same -inf pattern, finite entries within 1e-5 relative / 2e-4 absolute (natural logarithms) -- mild and sharp emissions
             (sharp: range marks, the item kernel behind)
  * stream   the stream kernels as teams of 1, 2 and 4 workgroups (MM_STREAM_H): posteriors and log Z against the item kernel, two runs
             the same bits
  * prob     ProbSemiring batches on the log twins (mm_pdfposteriors_f32 on an MM_PROB batch) against the generic entry
  * gamma    mm_batch_set_gamma_mode on the wave kernel: X + s * gamma against the plain call

GPU only.  SEED=n python tools/fuzz_round6.py [export stream prob gamma]; the switches are read at batch creation (MM_DEBUG)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge  # noqa: E402
import torch  # noqa: E402
from fuzz_cases import lens_pattern  # noqa: E402
from fuzz_round3 import posterior_error  # noqa: E402

mm = ge.load_package()
wl = importlib.import_module(mm.__name__ + ".workloads")


def with_env(env, fn):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: v for k, v in env.items() if v is not None})
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
    try:
        return fn()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def batch_with(kernel, cfs, extra=None):
    env = {"MM_DEBUG": "1", "MM_KERNEL": kernel}
    env.update(extra or {})
    return with_env(env, lambda: mm.batch(*cfs))


def sharpen(rng, B, N, P, sigma):
    x = sigma * rng.standard_normal((B, N, P))
    mx = x.max(-1, keepdims=True)
    return (x - mx - np.log(np.exp(x - mx).sum(-1, keepdims=True))).astype(np.float32)


def fuzz_export(rng):
    bad = n = nfast = 0
    for trial in range(10):
        k = trial % 5
        P = int(rng.integers(5, 240))
        g = (wl.lfmmi_denominator(int(rng.integers(60, 1100)) * 2, P, seed=int(rng.integers(1 << 30))) if k < 2
             else wl.lfmmi_denominator(int(rng.integers(1300, 2100)) * 2, min(P, 120), seed=int(rng.integers(1 << 30))) if k == 2  # teams
             else wl.random_fsm(int(rng.integers(70, 900)), min(P, 60), float(rng.uniform(1.5, 5.0)), seed=int(rng.integers(1 << 30))) if k == 3
             else wl.lexicon_fsm(int(rng.integers(1100, 3000)), P, seed=int(rng.integers(1 << 30)), hubs=int(rng.integers(1, 4))))
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        S1 = g.S + 1
        for B, N in ((1, 1), (2, 2), (3, 7), (5, 40), (9, 130)):
            lens = lens_pattern(rng, B, N)
            for sigma in (0.0, 9.0):
                V = (1.3 * rng.standard_normal((B, N, g.P))).astype(np.float32) if sigma == 0.0 else sharpen(rng, B, N, g.P, sigma)
                ref = batch_with("item", [cf] * B)
                bf = batch_with(None, [cf] * B)
                fast = "mm_pair_export_kernel" in bf.kernels("export")
                nfast += fast
                for name in ("alpharecursion", "betarecursion"):
                    R = getattr(ref, name)(V, lens)
                    for rep in range(2):  # (the second call: the policy the first call's marks chose)
                        A = getattr(bf, name)(V, lens)
                        n += 1
                        same_inf = np.array_equal(np.isneginf(A), np.isneginf(R))
                        m = np.isfinite(R)
                        err = np.abs(A[m] - R[m]).max() if m.any() and same_inf else 0.0
                        ok = same_inf and A.shape == (B * S1, N + 1) and np.allclose(A[m], R[m], rtol=1e-5, atol=2e-4)
                        if not ok:
                            bad += 1
                            print(f"MISMATCH export {name} trial {trial} ({g.name}, S {g.S}, P {g.P}) B {B} N {N} sigma {sigma} rep {rep} lens {lens.tolist()[:10]} "
                                  f"fast {fast}: -inf pattern {same_inf}, max abs {err:.3e}, redo {bf.last_redo_count()}")
    print(f"  ({nfast} of the batches had the fast export)")
    return n, bad


def fuzz_stream(rng):
    bad = n = 0
    for trial in range(6):
        S = int(rng.integers(150, 4500)) * 2
        P = int(rng.integers(20, 900))
        g = wl.lfmmi_denominator(S, P, seed=int(rng.integers(1 << 30))) if trial % 3 else wl.lexicon_fsm(S, P, seed=int(rng.integers(1 << 30)), hubs=int(rng.integers(1, 5)))
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        for B, N in ((1, 1), (2, 3), (3, 33), (7, 60)):
            lens = lens_pattern(rng, B, N)
            lt = torch.from_numpy(lens).cuda()
            V = torch.from_numpy((1.2 * rng.standard_normal((B, N, g.P))).astype(np.float32)).cuda()
            ref = batch_with("item", [cf] * B)
            rg, rt = ref.pdfposteriors(V, lt)
            rg, rt = rg.cpu().numpy().astype(np.float64), rt.cpu().numpy().astype(np.float64)
            for H in (1, 2, 4):
                bf = batch_with("stream", [cf] * B, {"MM_STREAM_H": str(H)})
                names = bf.kernels()
                a, t = bf.pdfposteriors(V, lt)
                a2, t2 = bf.pdfposteriors(V, lt)
                n += 1
                et, eg = posterior_error(a.cpu().numpy().astype(np.float64), t.cpu().numpy().astype(np.float64), rg, rt)
                same = torch.equal(a, a2) and torch.equal(t, t2)
                ok = "mm_stream_kernel" in names and np.isfinite(et) and np.isfinite(eg) and et < 1e-4 and eg < 1e-4 and same
                if not ok:
                    bad += 1
                    print(f"MISMATCH stream trial {trial} ({g.name}, S {g.S}, P {g.P}) H {H} B {B} N {N} lens {lens.tolist()[:10]} ({names[:60]}): ttl {et:.2e} gamma {eg:.2e} "
                          f"same bits {same}, redo {bf.last_redo_count()} fallback {bf.last_fallback_count()}")
    return n, bad


def lin(wl, g):
    """a GraphSpec with probabilities in place of log-probabilities"""
    import copy

    h = copy.copy(g)
    h.w, h.final_w, h.init_w = np.exp(g.w), np.exp(g.final_w), np.exp(g.init_w)
    return h


def fuzz_prob(rng):
    bad = n = nfast = 0
    for trial in range(8):
        k = trial % 4
        P = int(rng.integers(3, 120))
        g = (wl.lfmmi_denominator(int(rng.integers(40, 1500)) * 2, P, seed=int(rng.integers(1 << 30))) if k == 0
             else wl.random_fsm(int(rng.integers(5, 500)), min(P, 50), float(rng.uniform(1.5, 4.0)), seed=int(rng.integers(1 << 30))) if k == 1
             else wl.lexicon_fsm(int(rng.integers(60, 900)), P, seed=int(rng.integers(1 << 30)), hubs=1) if k == 2
             else wl.dense_ergodic(int(rng.integers(2, 64)), seed=int(rng.integers(1 << 30))))
        cf = mm.compile(wl.to_fsm(mm, lin(wl, g), "prob", np.float32), mm.statemap(g.state2pdf, g.P))
        for B, N in ((1, 1), (2, 5), (4, 30), (6, 70)):
            lens = lens_pattern(rng, B, N)
            lt = torch.from_numpy(lens).cuda()
            # (likelihoods, not logarithms: exp of a mild N(0,1); the reference: the generic entry in the ProbSemiring itself, float64)
            lhs = [np.exp(0.7 * rng.standard_normal((g.P, N))) for _ in range(B)]
            if trial % 2:
                lhs[0][int(rng.integers(g.P)), int(rng.integers(N))] = 0.0  # zero(K): a pdf that cannot emit a frame
            V = torch.from_numpy(np.stack([x.T for x in lhs]).astype(np.float32)).cuda()
            bf = mm.batch(*([cf] * B))
            fast = bf.has_fast_entry()
            nfast += bool(fast)
            if not fast:
                continue
            a, t = bf.pdfposteriors(V, lt)
            rg, rt = bf.pdfposteriors_generic([mm.expand(lhs[b], int(lens[b]), "prob") for b in range(B)], dtype=np.float64)
            n += 1
            a, t = a.cpu().numpy().astype(np.float64).transpose(0, 2, 1), t.cpu().numpy().astype(np.float64)
            eg = np.abs(a - rg).max() if a.size else 0.0
            fin = np.isfinite(rt) & (rt > 0)
            et = (np.abs(np.log(np.maximum(t[fin], 1e-300)) - np.log(rt[fin])) / np.maximum(1.0, np.abs(np.log(rt[fin])))).max() if fin.any() else 0.0
            zero_ok = np.array_equal(t[~fin] == 0, rt[~fin] == 0) if (~fin).any() else True
            ok = np.isfinite(eg) and eg < 2e-4 and et < 1e-4 and zero_ok
            if not ok:
                bad += 1
                print(f"MISMATCH prob trial {trial} ({g.name}, S {g.S}, P {g.P}) B {B} N {N} lens {lens.tolist()[:10]} ({bf.kernels()[:40]}): ttl {et:.2e} gamma {eg:.2e} zeros {zero_ok}")
    print(f"  ({nfast} of the batches had log twins)")
    return n, bad


def fuzz_gamma(rng):
    bad = n = 0
    for trial in range(6):
        gs = [wl.lexicon_fsm(int(rng.integers(70, 900)), 30, seed=int(rng.integers(1 << 30)), hubs=1) for _ in range(int(rng.integers(1, 6)))]
        P = 30
        cfs = [mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, P)) for g in gs]
        for B, N in ((1, 1), (3, 9), (len(gs) * 3, 50)):
            cb = [cfs[b % len(cfs)] for b in range(B)]
            bf = batch_with("wave", cb)
            if "mm_wave_kernel" not in bf.kernels():
                continue
            lens = lens_pattern(rng, B, N)
            lt = torch.from_numpy(lens).cuda()
            V = torch.from_numpy((1.2 * rng.standard_normal((B, N, P))).astype(np.float32)).cuda()
            g0, t0 = bf.pdfposteriors(V, lt)
            X = torch.randn(B, N, P, device="cuda")
            s = float(rng.choice([-1.0, 0.5, 2.0]))
            buf = X.clone()
            bf.set_gamma_mode(True, s)
            g1, t1 = bf.pdfposteriors(V, lt, out=buf)
            bf.set_gamma_mode(False, 1.0)
            n += 1
            want = X + s * g0
            err = float((buf - want).abs().max())
            ok = torch.equal(t1, t0) and err <= 1e-6
            if not ok:
                bad += 1
                print(f"MISMATCH gamma trial {trial} B {B} N {N} scale {s} lens {lens.tolist()[:10]}: max abs {err:.3e}, ttl equal {torch.equal(t1, t0)}")
    return n, bad


def main(seed=0, which=("export", "stream", "prob", "gamma")):
    rng = np.random.default_rng(seed)
    total = bad = 0
    for name in which:
        n, b = {"export": fuzz_export, "stream": fuzz_stream, "prob": fuzz_prob, "gamma": fuzz_gamma}[name](rng)
        print(f"{name}: {n} comparisons, {b} mismatches", flush=True)
        total += n
        bad += b
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(os.environ.get("SEED", 0)), tuple(sys.argv[1:]) or ("export", "stream", "prob", "gamma")) else 0)
