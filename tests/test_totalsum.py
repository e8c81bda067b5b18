"""Total-sum family (src/algorithms.jl:8-36): the oracle against brute-force path enumeration and
closed forms (CPU), the device entry point mm_totalsum_f32 against the oracle (GPU)."""
import itertools
import math

import numpy as np
import pytest

import graphs


def brute_force(o, fsm, n, cumulative):
    """Sum over every path of exactly n (or up to n) states of init * arcs * final, by enumeration."""
    K = fsm.K
    a, T, w = o.fsm_parts(fsm)
    S = fsm.nstates
    total = K.zero
    for length in (range(1, n + 1) if cumulative else [n]):
        for path in itertools.product(range(S), repeat=length):
            x = K.mul(a[path[0]], w[path[-1]])
            for i, j in zip(path[:-1], path[1:]):
                x = K.mul(x, T[i, j])
            total = K.add(total, x)
    return float(total)


@pytest.mark.parametrize("semiring", ["log", "tropical"])
def test_oracle_total_sums_enumerate_paths(oracle, wl, semiring):
    o, _ = oracle
    g = wl.random_fsm(4, 3, seed=7)
    fsm = graphs.to_oracle(o, g, semiring)
    a, T, w = o.fsm_parts(fsm)
    for n in (1, 2, 4):
        assert np.isclose(o.totalsum(a, T, w, n, fsm.K), brute_force(o, fsm, n, False), rtol=1e-12, atol=1e-12)
        assert np.isclose(o.totalcumsum(a, T, w, n, fsm.K), brute_force(o, fsm, n, True), rtol=1e-12, atol=1e-12)
    assert np.isclose(o.totalweightsum(fsm), o.totalcumsum(a, T, w, fsm.nstates, fsm.K))


def test_oracle_renormalised_fsm_sums_to_one(oracle, wl):
    """A renorm()ed FSM (src/fsmops.jl:71-79) is a probability distribution over paths: the cumulative
    total weight tends to one(K) = log 1; for the 3-state left-to-right HMM of the demo notebook the
    paths of exactly n states weigh C(n-1, 2) / 2^n."""
    o, _ = oracle
    fsm = graphs.to_oracle(o, wl.l2r_hmm(3))
    a, T, w = o.fsm_parts(fsm)
    for n in (3, 5, 9):
        assert np.isclose(o.totalsum(a, T, w, n, fsm.K), math.log(math.comb(n - 1, 2) / 2.0 ** n), atol=1e-12)
    assert o.totalsum(a, T, w, 2, fsm.K) == -math.inf
    assert abs(o.totalcumsum(a, T, w, 200, fsm.K)) < 1e-9


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("semiring", ["log", "tropical"])
def test_device_total_sums_match_oracle(mm, wl, oracle, semiring):
    o, _ = oracle
    specs = [wl.random_fsm(40, 4, seed=1), wl.l2r_hmm(3), wl.random_fsm(300, 6, seed=2), wl.lexicon_fsm(200, seed=3)]
    P = max(g.P for g in specs)
    cfs = [mm.compile(wl.to_fsm(mm, g, semiring), mm.statemap(g.state2pdf, P)) for g in specs]
    bf = mm.batch(*cfs)
    for n in (1, 2, 7, 60):
        for cumulative in (False, True):
            got = bf.totalsum(n, cumulative).cpu().numpy()
            for b, g in enumerate(specs):
                fsm = graphs.to_oracle(o, g, semiring)
                a, T, w = o.fsm_parts(fsm)
                ref = float((o.totalcumsum if cumulative else o.totalsum)(a, T, w, n, fsm.K))
                if ref == -math.inf:
                    assert got[b] == -math.inf
                else:
                    assert abs(got[b] - ref) <= 1e-4 * max(abs(ref), 1.0), (semiring, n, cumulative, b, got[b], ref)


@pytest.mark.gpu
def test_device_totalweightsum_api(mm, wl, oracle):
    o, _ = oracle
    g = wl.l2r_hmm(3)
    f = wl.to_fsm(mm, g)
    assert np.isclose(mm.totalsum(f, 5), math.log(6 / 32), atol=1e-6)  # = ttl of the demo notebook's lhs = zeros(3, 5)
    assert mm.totalsum(f, 2) == -math.inf
    assert abs(mm.totalcumsum(f, 200)) < 1e-5
    ref = float(o.totalweightsum(graphs.to_oracle(o, g)))
    assert np.isclose(mm.totalweightsum(f), ref, atol=1e-6)
    with pytest.raises(mm.DimensionMismatch):
        mm.totalsum(f, 0)


@pytest.mark.gpu
def test_device_total_sum_wsj_denominator(mm, wl, oracle):
    """The real denominator graph of the reference (3032 states): the C oracle's forward recursion with
    flat emissions gives the same totals (log Z of N frames of zeros = totalsum(N))."""
    import os

    o, oc = oracle
    g = wl.load_npz_graph(os.path.join(os.path.dirname(__file__), "golden", "den_fsm_wsj.npz"))
    n = 40
    V = np.zeros((1, n, g.P), dtype=np.float32)
    _, ttl = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, None, dtype=np.float64)
    got = mm.totalsum(wl.to_fsm(mm, g), n)
    assert abs(got - float(ttl[0])) <= 1e-4 * max(abs(float(ttl[0])), 1.0)
