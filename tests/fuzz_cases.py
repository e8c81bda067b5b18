"""The input stream of the team fuzzer (tools/fuzz_round3.py `split`), replayable case by case from its seed: the committed tests
that pin what the fuzzer found regenerate their inputs from (seed, graph number, B, N) instead of carrying arrays."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SPLIT_SIZES = ((1, 1), (1, 6), (1, 61), (2, 1), (2, 2), (3, 3), (3, 4), (5, 7), (8, 40), (9, 101), (300, 12), (515, 9))


def lens_pattern(rng, B, N):
    pat = rng.integers(0, 3)
    lens = np.full(B, N) if pat == 0 else rng.integers(0, N + 1, B) if pat == 1 else rng.integers(max(0, N - 2), N + 1, B)
    return lens.astype(np.int32)


def split_graph_makers(wl, rng):
    here = os.path.join(HERE, "golden", "den_fsm_wsj.npz")
    return [lambda: wl.load_npz_graph(here), lambda: wl.lfmmi_denominator(2900, 120, seed=int(rng.integers(1 << 30))),
            lambda: wl.lfmmi_denominator(2400, 200, seed=int(rng.integers(1 << 30))),
            lambda: wl.lfmmi_denominator(int(rng.integers(1600, 2040)) * 2, 100, seed=int(rng.integers(1 << 30))),  # (teams of 4)
            lambda: wl.lfmmi_denominator(int(rng.integers(2100, 3000)) * 2, 2 * int(rng.integers(20, 157)), seed=int(rng.integers(1 << 30))),  # (teams of 8)
            lambda: wl.lfmmi_denominator(2600, 2 * int(rng.integers(126, 253)), seed=int(rng.integers(1 << 30)))]  # (teams of 2, 251 .. 506 pdfs)


def split_cases(wl, seed, want=None):
    """Yields (graph number, graph, B, N, V0[B, N, P] float32, sharp, lens) in the fuzzer's order; the emissions of a case are
    V0, or log_softmax(8 V0) computed in float32 when `sharp`.  want: a set of (graph number, B, N) -- the graphs of the others are
    still built (their seeds come from the same stream) but their cases are skipped cheaply."""
    rng = np.random.default_rng(seed)
    for gi, mk in enumerate(split_graph_makers(wl, rng)):
        g = mk()
        for B, N in SPLIT_SIZES:
            V0 = (1.5 * rng.standard_normal((B, N, g.P))).astype(np.float32)
            sharp = rng.integers(0, 3) == 0
            lens = lens_pattern(rng, B, N)
            if want is None or (gi, B, N) in want:
                yield gi, g, B, N, V0, bool(sharp), lens
