#!/usr/bin/env python3
"""Fixtures that pin the oracle's pdfposteriors on NON-TRIVIAL emissions.

The reference's own test suite carries an independent implementation of the quantity
`pdfposteriors` returns: dense log-domain forward / backward / forward_backward on the graph without
the phony final state (test/test_algorithms.jl:28-63), which its tests compare the sparse path with
(:200-248) -- there with constant emissions only.  This script evaluates that dense recursion
(restated as oracle.dense_forward_backward; it shares nothing with the sparse path: no CSC, no expand,
no state map, no rawunion) on seeded RANDOM emissions and writes inputs and outputs to
tests/golden/pin_*.npz.  tests/test_oracle_pin.py then checks the oracle's sparse path (NumPy and C)
against these numbers, so a frame shift in expand(), a transposed C_hat or a wrong ttl fails a test.

Cases (all float64):
  l2r3      the 3-state HMM of test/test_algorithms.jl:13-26, identity state map, N = 9
  rand30    random_fsm(30): several initial states, 30 % final states, random weights, identity map
  rand30m   the same graph with a many-to-one state map onto 7 pdfs: the pdf posterior is the sum of
            the state posteriors of its states (src/inference.jl:155 C_hat' * AB) and the state-level
            likelihood is the pdf's (src/inference.jl:150 C_hat * V_hat)
  every case twice: full length, and with seqlength < N (expand's padding, src/inference.jl:54-60:
  the dense recursion is simply run on the first `len` frames, test/test_algorithms.jl:226-239)

Run: python tests/golden/make_pin_fixtures.py      (needs nothing but the repo)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402


def dense_case(o, g, s2p, P, lhs_pdf, length):
    """gamma[P, N] (zeros beyond `length`) and ttl from the dense recursion on state-level likelihoods."""
    import graphs

    f = graphs.to_oracle(o, g)
    A_hat = f.T_hat.todense(o.LOG)
    lhs_state = lhs_pdf[np.asarray(s2p), :length]
    g_state, ttl = o.dense_forward_backward(A_hat, f.alpha_hat, lhs_state)
    gam = np.zeros((P, lhs_pdf.shape[1]))
    np.add.at(gam, np.asarray(s2p), np.pad(g_state, ((0, 0), (0, lhs_pdf.shape[1] - length))))
    return gam, ttl


def main():
    o, _ = ge.load_oracle()
    mm = ge.load_package()
    import importlib

    wl = importlib.import_module(mm.__name__ + ".workloads")
    rng = np.random.default_rng(31415)
    g3, g30 = wl.l2r_hmm(3), wl.random_fsm(30, 30, 3.0, seed=7)
    many = rng.integers(0, 7, 30).astype(np.int32)
    many[:7] = np.arange(7)  # every pdf used
    cases = {
        "l2r3": (g3, np.arange(3, dtype=np.int32), 3, 9, [9, 6]),
        "rand30": (g30, np.arange(30, dtype=np.int32), 30, 14, [14, 9]),
        "rand30m": (g30, many, 7, 14, [14, 11]),
    }
    for name, (g, s2p, P, N, lens) in cases.items():
        lhs = (2.0 * rng.standard_normal((len(lens), P, N))).astype(np.float64)  # [b][pdf][frame]
        gam = np.zeros((len(lens), P, N))
        ttl = np.zeros(len(lens))
        for b, L in enumerate(lens):
            gam[b], ttl[b] = dense_case(o, g, s2p, P, lhs[b], L)
        np.savez_compressed(os.path.join(HERE, f"pin_{name}.npz"), S=g.S, P=P, init_idx=g.init_idx, init_w=g.init_w,
                            src=g.src, dst=g.dst, w=g.w, final_idx=g.final_idx, final_w=g.final_w, state2pdf=s2p,
                            lhs=lhs, lens=np.asarray(lens, dtype=np.int32), gamma=gam, ttl=ttl)
        print(name, "S", g.S, "P", P, "N", N, "lens", lens, "ttl", ttl)


if __name__ == "__main__":
    main()
