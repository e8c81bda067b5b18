#!/usr/bin/env python3
"""Diagnostic: the kernels added in round 3 against the item kernel (an independent implementation in the log domain)
on random graphs, batch sizes, frame counts and length patterns, including the degenerate ones:

  * split pair kernels (teams of two workgroups): graphs beyond the pair kernels, odd and large batches (more teams
    than compute units), one or two frames, empty utterances;
  * wave kernel: batches of DIFFERENT small deep graphs, run twice (identical bits);
  * Viterbi on the row-lane form: paths and scores must be bit-identical to the item form's.

GPU only.  SEED=n python tools/fuzz_round3.py; the switches are read at batch creation (MM_DEBUG)."""
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


def set_kernel(kernel):
    os.environ["MM_DEBUG"] = "1"
    if kernel:
        os.environ["MM_KERNEL"] = kernel
    else:
        os.environ.pop("MM_KERNEL", None)


sys.path.insert(0, os.path.join(ROOT, "tests"))
from fuzz_cases import lens_pattern, split_cases  # noqa: E402  (the team fuzzer's inputs, replayable case by case: tests/fuzz_cases.py)


def posterior_error(a_g, a_t, ref_g, ref_t):
    """Largest relative error of log Z, and of the posteriors, against the contract the header documents for the default mark policy
    (include/markovmodels_amd.h, mm_batch_set_mark_policy; pinned by tests/test_gpu_exact.py on this fuzzer's findings): relative in
    the logarithm for posteriors above 1e-24; an ABSOLUTE error of at most 1e-27 below -- a term the float32 linear kernels drop is
    below 2^(-120 - L_n) (the overlap criterion) and a pdf sums a few dozen of them -- scaled here so that 1e-4 is
    the bar for both.  (MM_MARKS_KEEP holds the relative bound down to 1e-30: tools/fuzz_find.py.)"""
    same_inf = np.isinf(a_t) & np.isinf(ref_t) & (a_t == ref_t)
    fin = ~same_inf
    et = (np.abs(a_t[fin] - ref_t[fin]) / np.maximum(1.0, np.abs(ref_t[fin]))).max() if fin.any() else 0.0
    m = ref_g >= 1e-24
    eg = np.abs(a_g - ref_g).max() if a_g.size else 0.0
    if m.any():
        eg = max(eg, (np.abs(np.log(np.maximum(a_g[m], 1e-300)) - np.log(ref_g[m])) / np.maximum(np.abs(np.log(ref_g[m])), 1)).max())
    if (~m).any():
        eg = max(eg, 1e-4 * np.abs(a_g[~m] - ref_g[~m]).max() / 1e-27)
    return et, eg


def posteriors(cfs, V, lt, kernel):
    set_kernel(kernel)
    bf = mm.batch(*cfs)
    gam, ttl = bf.pdfposteriors(V, lt)
    torch.cuda.synchronize()
    return gam.cpu().numpy().astype(np.float64), ttl.cpu().numpy().astype(np.float64), bf.kernels("log"), bf


def fuzz_split(seed):
    bad = n = 0
    cfs = {}
    if True:
        for gi, g, B, N, V0, sharp, lens in split_cases(wl, seed):
            if gi not in cfs:
                cfs.clear()
                cfs[gi] = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
            cf = cfs[gi]
            V = torch.from_numpy(V0).cuda()
            if sharp:  # (sharp: the float64 team kernels behind the float32 ones)
                V = torch.log_softmax(8.0 * V, dim=-1)
            lt = torch.from_numpy(lens).cuda()
            ref_g, ref_t, _, _ = posteriors([cf] * B, V, lt, "item")
            a_g, a_t, names, bf = posteriors([cf] * B, V, lt, None)
            n += 1
            et, eg = posterior_error(a_g, a_t, ref_g, ref_t)
            ok = np.isfinite(et) and np.isfinite(eg) and et < 1e-4 and eg < 1e-4 and "mm_fbs_kernel" in names
            if not ok:
                bad += 1
                print(f"MISMATCH split graph {gi} B {B} N {N} lens {lens.tolist()[:12]} ({names[:50]}): ttl {et:.2e} gamma {eg:.2e}")
                if os.environ.get("FUZZ_VERBOSE"):  # where, and how large the reference's value is there
                    m = ref_g > float(os.environ.get("FUZZ_ABOVE", 1e-30))
                    rel = np.zeros_like(ref_g)
                    rel[m] = np.abs(np.log(np.maximum(a_g[m], 1e-300)) - np.log(ref_g[m])) / np.maximum(np.abs(np.log(ref_g[m])), 1)
                    idx = np.unravel_index(np.argmax(rel), rel.shape)
                    print(f"   worst relative log error {rel[idx]:.3e} at {idx}: reference {ref_g[idx]:.6e}, computed {a_g[idx]:.6e}; redo {bf.last_redo_count()}, "
                          f"fallback {bf.last_fallback_count()}, abs max {np.abs(a_g - ref_g).max():.3e}")
                    b0, n0, p0 = (int(x) for x in idx)
                    print(f"   utterance {b0}: length {int(lens[b0])}, frame {n0}, pdf {p0}; the reference's frame sums to {ref_g[b0, n0].sum():.9f}")
                    # the states of the pdf: log2 alpha / beta (item kernel, log domain) relative to the frame's maxima, and their products
                    set_kernel("item")
                    b1 = mm.batch(cf)
                    Vs, ls = V[[b0]].contiguous(), lt[[b0]].contiguous()
                    A = b1.alpharecursion(Vs, ls).cpu().numpy().astype(np.float64) / np.log(2.0)
                    Bt = b1.betarecursion(Vs, ls).cpu().numpy().astype(np.float64) / np.log(2.0)
                    fr = n0  # column of frame n0 + 1 in the (S+1) x (N+1) matrices
                    a, bb = A[:g.S, fr], Bt[:g.S, fr]
                    st = np.nonzero(np.asarray(g.state2pdf) == p0)[0]
                    fin = np.isfinite(a + bb)
                    print(f"   frame maxima: log2 alpha {a[np.isfinite(a)].max():.1f}, log2 beta {bb[np.isfinite(bb)].max():.1f}, log2 sum alpha*beta "
                          f"{np.log2(np.exp2((a + bb)[fin] - (a + bb)[fin].max()).sum()) + (a + bb)[fin].max():.1f}")
                    for sidx in st:
                        print(f"   state {sidx}: log2 alpha - max {a[sidx] - a[np.isfinite(a)].max():.1f}, log2 beta - max {bb[sidx] - bb[np.isfinite(bb)].max():.1f}")
                    for pol in ("f32_first", "f64_first"):  # the utterance alone, and with its neighbour
                        for sel in ([b0], [b0, (b0 + 1) % B]):
                            set_kernel(None)
                            os.environ["MM_VERBOSE"] = "1"
                            b2 = mm.batch(*([cf] * len(sel)))
                            os.environ.pop("MM_VERBOSE")
                            b2.set_exact_policy(pol)
                            g2, t2 = b2.pdfposteriors(V[sel].contiguous(), lt[sel].contiguous())
                            g2 = g2.cpu().numpy()
                            print(f"   {pol} batch {sel}: computed {g2[0, n0, p0]:.6e}, redo {b2.last_redo_count()}, fallback {b2.last_fallback_count()}, {b2.kernels()[:40]}")
    return n, bad


def fuzz_wave(rng):
    bad = n = nwave = 0
    num = wl.load_npz_graph(os.path.join(ROOT, "tests", "golden", "num_fsm_wsj.npz"))
    for trial in range(6):
        B = int(rng.choice([1, 2, 3, 7, 33, 130, 300, 515]))  # (more than 256: the kernel build of which two workgroups share a compute unit)
        gs = []
        for b in range(min(B, 6)):
            k = rng.integers(0, 7)
            gs.append(num if k == 0 else wl.lexicon_fsm(int(rng.integers(40, 900)), int(rng.integers(5, 200)), seed=int(rng.integers(1 << 30)), hubs=int(rng.integers(1, 3)))
                      if k < 3 else wl.l2r_hmm(int(rng.integers(3, 60))) if k == 3
                      else wl.random_fsm(int(rng.integers(5, 400)), int(rng.integers(2, 60)), float(rng.uniform(1.5, 5.0)), seed=int(rng.integers(1 << 30))) if k < 6
                      else wl.dense_ergodic(int(rng.integers(2, 33)), seed=int(rng.integers(1 << 30))))
        Pm = max(g.P for g in gs)
        cfs = [mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, Pm)) for g in gs]
        cfs = [cfs[b % len(cfs)] for b in range(B)]
        for N in (1, 2, 3, 5, 64, 150):
            V = torch.from_numpy((1.5 * rng.standard_normal((B, N, Pm))).astype(np.float32)).cuda()
            lens = lens_pattern(rng, B, N)
            lt = torch.from_numpy(lens).cuda()
            ref_g, ref_t, _, _ = posteriors(cfs, V, lt, "item")
            a_g, a_t, names, bf = posteriors(cfs, V, lt, "wave")
            g2, t2 = bf.pdfposteriors(V, lt)
            n += 1
            et, eg = posterior_error(a_g, a_t, ref_g, ref_t)
            same = np.array_equal(g2.cpu().numpy(), a_g.astype(np.float32)) and np.array_equal(t2.cpu().numpy(), a_t.astype(np.float32))
            ok = np.isfinite(et) and np.isfinite(eg) and et < 1e-4 and eg < 1e-4 and same
            nwave += "mm_wave_kernel" in names
            if not ok:
                bad += 1
                print(f"MISMATCH wave trial {trial} B {B} N {N} lens {lens.tolist()[:12]} ({names[:50]}): ttl {et:.2e} gamma {eg:.2e} same bits {same}")
    print(f"  ({nwave} of the {n} batches fitted the wave kernel)")
    return n, bad


def fuzz_viterbi(rng):
    bad = n = 0
    for trial in range(8):
        k = trial % 4
        g = (wl.lexicon_fsm(int(rng.integers(50, 6000)), int(rng.integers(5, 240)), seed=int(rng.integers(1 << 30)), hubs=int(rng.integers(1, 6))) if k < 2
             else wl.random_fsm(int(rng.integers(5, 900)), int(rng.integers(2, 60)), float(rng.uniform(1.5, 5.0)), seed=int(rng.integers(1 << 30))) if k == 2
             else wl.lfmmi_denominator(int(rng.integers(100, 1000)) * 2, 84, seed=int(rng.integers(1 << 30))))
        cf = mm.compile(wl.to_fsm(mm, g, "tropical"), mm.statemap(g.state2pdf, g.P))
        for B, N in ((1, 1), (2, 2), (3, 5), (5, 33), (4, 260), (131, 17)):
            V = torch.from_numpy((2.0 * rng.standard_normal((B, N, g.P))).astype(np.float32)).cuda()
            lens = lens_pattern(rng, B, N)
            lt = torch.from_numpy(lens).cuda()
            out = []
            for kern in ("item", None):
                set_kernel(kern)
                bf = mm.batch(*([cf] * B))
                path, score = bf.viterbi(V, lt)[:2]
                torch.cuda.synchronize()
                out.append((path.cpu().numpy(), score.cpu().numpy(), bf.kernels("tropical")))
            n += 1
            same = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
            if not same:
                bad += 1
                print(f"MISMATCH viterbi trial {trial} ({g.name}) B {B} N {N} lens {lens.tolist()[:12]} ({out[1][2][:50]})")
    return n, bad


def main(seed=0, which=("split", "wave", "viterbi")):
    rng = np.random.default_rng(seed)
    saved = {k: os.environ.get(k) for k in ("MM_DEBUG", "MM_KERNEL")}
    total = bad = 0
    for name in which:
        n, b = fuzz_split(seed) if name == "split" else {"wave": fuzz_wave, "viterbi": fuzz_viterbi}[name](rng)
        print(f"{name}: {n} comparisons, {b} mismatches", flush=True)
        total += n
        bad += b
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(os.environ.get("SEED", 0)), tuple(sys.argv[1:]) or ("split", "wave", "viterbi")) else 0)
