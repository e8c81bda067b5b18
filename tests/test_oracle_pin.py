"""The oracle's sparse pdfposteriors path against the reference test suite's independent dense
forward/backward (test/test_algorithms.jl:28-63) on seeded RANDOM emissions: committed fixtures
tests/golden/pin_*.npz, written by tests/golden/make_pin_fixtures.py.

What these pin that the constant-emission vectors of the reference (demo notebook, :218-248) cannot:
the time alignment of the emissions (expand + the lhs[:, n] of both recursions), the C_hat gather and
the C_hat' reduction with a many-to-one state map, the padding for seqlength < N, and ttl.
The `mutant` tests show the teeth: the same comparison FAILS for an oracle whose emissions are shifted
by one frame, whose state map is transposed / permuted, or whose ttl takes another reduction."""
import os

import numpy as np
import pytest

import graphs

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["l2r3", "rand30", "rand30m"]


def load(wl, name):
    path = os.path.join(HERE, "golden", f"pin_{name}.npz")
    z = np.load(path)
    return wl.load_npz_graph(path), z["state2pdf"].astype(np.int32), int(z["P"]), z["lhs"], z["lens"], z["gamma"], z["ttl"]


def agree(gam, ttl, gam_ref, ttl_ref, rtol=1e-9):
    return bool(np.allclose(gam, gam_ref, rtol=rtol, atol=1e-12) and np.allclose(ttl, ttl_ref, rtol=rtol, atol=1e-9))


def run_numpy(o, g, s2p, P, lhs, lens, dtype=np.float64):
    f = graphs.to_oracle(o, g, "log", dtype)
    gam, ttl = o.pdfposteriors_batch(f, s2p.tolist(), P, [l.astype(dtype) for l in lhs], [int(x) for x in lens])
    return gam, ttl


@pytest.mark.parametrize("name", CASES)
def test_numpy_oracle_matches_dense_reference(oracle, wl, name):
    o, _ = oracle
    g, s2p, P, lhs, lens, gam_ref, ttl_ref = load(wl, name)
    gam, ttl = run_numpy(o, g, s2p, P, lhs, lens)
    assert agree(gam, ttl, gam_ref, ttl_ref)
    for b, L in enumerate(lens):
        assert (gam[b][:, L:] == 0).all()  # exact zeros beyond the sequence length (test/test_algorithms.jl:241)
    # float32 restatement: the reference's own tolerance is isapprox (sqrt(eps))
    gam32, ttl32 = run_numpy(o, g, s2p, P, lhs, lens, np.float32)
    assert agree(gam32, ttl32, gam_ref, ttl_ref, rtol=2e-4)


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_matches_dense_reference(oracle, wl, name):
    o, oc = oracle
    g, s2p, P, lhs, lens, gam_ref, ttl_ref = load(wl, name)
    f = graphs.to_oracle(o, g)
    gam, ttl = oc.batch_shared(f, s2p.tolist(), P, np.ascontiguousarray(lhs.transpose(0, 2, 1)), lens, dtype=np.float64)
    assert agree(gam.transpose(0, 2, 1), ttl, gam_ref, ttl_ref)
    gam32, ttl32 = oc.batch_shared(f, s2p.tolist(), P, np.ascontiguousarray(lhs.transpose(0, 2, 1)), lens, dtype=np.float32)
    assert agree(gam32.transpose(0, 2, 1), ttl32, gam_ref, ttl_ref, rtol=2e-4)
    # ... and with a thread per utterance (the cpu_baseline configuration of bench.py)
    gam2, ttl2 = oc.batch_shared(f, s2p.tolist(), P, np.ascontiguousarray(lhs.transpose(0, 2, 1)), lens, dtype=np.float64, nthreads=2)
    assert np.array_equal(gam2, gam) and np.array_equal(ttl2, ttl)


# ---- the teeth: mutated oracles must FAIL the comparison above

def _mutants(o):
    """name -> context manager that installs the mutation into the oracle module"""
    import contextlib

    @contextlib.contextmanager
    def patched(attr, fn):
        old = getattr(o, attr)
        setattr(o, attr, fn)
        try:
            yield
        finally:
            setattr(o, attr, old)

    def shifted_expand(lhs, seqlength, K, _orig=o.expand):  # emissions one frame late
        return _orig(np.roll(lhs, 1, axis=1), seqlength, K)

    def transposed_statemap(state2pdf, numpdf, K, dtype=np.float64, _orig=o.statemap):  # C_hat' in place of C_hat:
        s2p = list(state2pdf)                                                      # state s reads pdf (S-1-s) mod P
        return _orig([s2p[len(s2p) - 1 - i] for i in range(len(s2p))], numpdf, K, dtype=dtype)

    def late_padding(lhs, seqlength, K, _orig=o.expand):  # the final state may only be entered one frame later
        return _orig(lhs, None if seqlength is None else min(seqlength + 1, lhs.shape[1]), K)

    return {
        "frame_shift": lambda: patched("expand", shifted_expand),
        "statemap": lambda: patched("statemap", transposed_statemap),
        "padding": lambda: patched("expand", late_padding),
    }


@pytest.mark.parametrize("mutant", ["frame_shift", "statemap", "padding"])
def test_mutated_oracle_is_caught(oracle, wl, mutant):
    o, _ = oracle
    caught = 0
    for name in CASES:
        g, s2p, P, lhs, lens, gam_ref, ttl_ref = load(wl, name)
        with _mutants(o)[mutant]():
            gam, ttl = run_numpy(o, g, s2p, P, lhs, lens)
        caught += not agree(gam, ttl, gam_ref, ttl_ref, rtol=1e-6)
    # (the identity-map cases cannot see a permuted state map of a symmetric graph; every mutant must be
    # caught by at least two of the three cases, the frame shift by all)
    assert caught >= (3 if mutant == "frame_shift" else 2), (mutant, caught)
    # the unmutated oracle is back
    g, s2p, P, lhs, lens, gam_ref, ttl_ref = load(wl, "rand30m")
    assert agree(*run_numpy(o, g, s2p, P, lhs, lens), gam_ref, ttl_ref)


def test_ttl_is_the_minimum_over_frames_of_the_frame_normaliser(oracle, wl):
    """src/inference.jl:159 `minimum(sums)`: with float64 all frames give the same log Z up to rounding, so the
    fixture's ttl pins the VALUE (the dense reference's minimum(sums), :61), and a different total (e.g. the
    likelihood without the final weights) is caught."""
    o, _ = oracle
    g, s2p, P, lhs, lens, gam_ref, ttl_ref = load(wl, "rand30")
    f = graphs.to_oracle(o, g)
    _, ttl = run_numpy(o, g, s2p, P, lhs, lens)
    # log Z by brute force over the last frame's alpha and the final weights
    A_hat = f.T_hat.todense(o.LOG)
    for b, L in enumerate(lens):
        la = lhs[b][:, 0][s2p] + f.alpha_hat[:-1]
        for n in range(1, L):
            la = lhs[b][:, n][s2p] + o._logsumexp(A_hat[:-1, :-1] + la[:, None], axis=0)
        logz = o._logsumexp(la + A_hat[:-1, -1], axis=0)
        assert np.isclose(ttl[b], logz, rtol=1e-10) and np.isclose(ttl_ref[b], logz, rtol=1e-10)
