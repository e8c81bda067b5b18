"""The CPU oracle against the reference's known-answer vectors (tests/golden/known_answers.json,
transcribed from the reference's tests/notebook) and against itself (NumPy vs C, float32 vs float64)."""
import json
import math
import os

import numpy as np
import pytest

import graphs

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))


def l2r3(o, wl, dtype=np.float64):
    return graphs.to_oracle(o, wl.l2r_hmm(3), "log", dtype)


def test_logaddexp_identities(oracle):
    o, _ = oracle
    for x, y, ref in KA["logaddexp"]["cases"]:
        assert math.isclose(o.logaddexp_scalar(x, y), ref, rel_tol=1e-12)
    assert o.logaddexp_scalar(-math.inf, -math.inf) == -math.inf
    assert o.LOG.add(np.float64(-np.inf), np.float64(-np.inf)) == -np.inf


def test_l2r_fsm_is_the_renormalised_demo_fsm(oracle, wl):
    """workloads.l2r_hmm builds the renormalised FSM directly; it must equal the
    reference's construction order: arc-list ctor, then renorm (src/fsmops.jl:71-79)."""
    o, _ = oracle
    K = o.LOG
    arcs = [((0, 0), K.one)]
    for s in range(1, 3):
        arcs += [((s - 1, s), K.one), ((s, s), K.one)]
    ref = o.renorm(o.make_fsm(K, [(0, K.one)], arcs, [(2, K.one)], [0, 1, 2]))
    got = l2r3(o, wl)
    assert np.allclose(ref.T_hat.todense(K), got.T_hat.todense(K), equal_nan=True)
    assert np.array_equal(ref.alpha_hat, got.alpha_hat)


def test_demo_notebook_gamma(oracle, wl):
    o, oc = oracle
    ka = KA["demo_notebook_gamma"]
    for dtype in (np.float64, np.float32):
        g, ttl = o.pdfposteriors_batch(l2r3(o, wl, dtype), [0, 1, 2], 3, [np.zeros((3, 5), dtype=dtype)], [5])
        assert np.allclose(g[0], ka["gamma"], atol=ka["atol"])
        assert np.allclose(g[0], ka["gamma_exact"], atol=1e-6)
        assert np.isclose(ttl[0], ka["ttl_derived"], atol=1e-6)
    gc, tc = oc.batch_shared(l2r3(o, wl), [0, 1, 2], 3, np.zeros((1, 5, 3)), None, dtype=np.float64)
    assert np.allclose(gc[0].T, ka["gamma_exact"], atol=1e-12)


def test_batch_varlen_expectations(oracle, wl):
    """test/test_algorithms.jl:218-248 expectations, with the dense logsumexp
    forward/backward of :28-63 as the independent cross-check."""
    o, oc = oracle
    f = l2r3(o, wl)
    lhs = np.ones((3, 7))
    g, ttl = o.pdfposteriors_batch(f, [0, 1, 2], 3, [lhs, lhs], KA["batch_varlen"]["seqlengths"])
    K = o.LOG
    g1, t1 = o.dense_forward_backward(f.T_hat.todense(K), f.alpha_hat, lhs[:, :5])
    g2, t2 = o.dense_forward_backward(f.T_hat.todense(K), f.alpha_hat, lhs)
    assert np.allclose(g[0][:, :5], g1) and np.isclose(ttl[0], t1)
    assert np.allclose(g[1], g2) and np.isclose(ttl[1], t2)
    assert (g[0][:, 5:] == 0).all()
    assert np.allclose(ttl, [3.3260236, 4.8560199], atol=1e-6)


def test_bestpath_chain(oracle):
    o, oc = oracle
    T = o.TROPICAL
    c = o.make_fsm(T, [(0, 0.0)], [((0, 1), 0.0), ((1, 2), 0.0), ((2, 3), 0.0)], [(3, 0.0)], list("abcd"), np.float32)
    for path, score in (o.viterbi(c, [0, 1, 2, 3], 4, np.ones((4, 4), np.float32))[:2],
                        oc.viterbi(c, [0, 1, 2, 3], 4, np.ones((4, 4), np.float32))[:2]):
        assert (path + 1).tolist() == KA["bestpath_chain"]["path_1based"]
        assert score == 4.0


def test_mul_known_answer(oracle):
    """test/test_linalg.jl:88-108: the semiring products are known by definition of (+), (*)."""
    o, _ = oracle
    ka = KA["mul_known_answer"]
    I, J = np.array(ka["I"]) - 1, np.array(ka["J"]) - 1
    V = np.array(ka["V"], dtype=np.float64)
    dv = np.array(ka["dv"], dtype=np.float64)
    dm = np.array(ka["dm_colmajor"], dtype=np.float64).reshape(4, 3).T  # reshape(1:12, 3, 4)
    for K in (o.LOG, o.TROPICAL, o.PROB):
        A = o.csc_from_coo(I, J, V, ka["shape"], K)
        Ad = A.todense(K)
        got_v, got_m = o.spmm_csc(A, dv, K), o.spmm_csc(A, dm, K)
        for r in range(4):
            terms = [K.mul(Ad[r, c], dv[c]) for c in range(3) if Ad[r, c] != K.zero]
            ref = K.zero
            for t in terms:
                ref = K.add(np.float64(ref), np.float64(t))
            assert np.isclose(got_v[r], ref) or (got_v[r] == ref)
        assert got_m.shape == (4, 4)
        assert np.allclose(got_m[:, 0], o.spmm_csc(A, dm[:, 0], K))


@pytest.mark.parametrize("gname", ["rand", "ergodic", "wide"])
def test_numpy_and_c_oracles_agree(oracle, wl, gname):
    o, oc = oracle
    g = {"rand": lambda: wl.random_fsm(30, 5, 3.0, seed=2), "ergodic": lambda: wl.dense_ergodic(16),
         "wide": lambda: wl.wide_row_fsm(300, 7)}[gname]()
    f = graphs.to_oracle(o, g)
    rng = np.random.default_rng(0)
    N, lens = 9, [9, 4]
    V = rng.standard_normal((2, N, g.P))
    gn, tn = o.pdfposteriors_batch(f, g.state2pdf, g.P, [V[0].T.copy(), V[1].T.copy()], lens)
    gc, tc = oc.batch_shared(f, g.state2pdf, g.P, V, lens, dtype=np.float64)
    assert np.allclose(gn.transpose(0, 2, 1), gc, atol=1e-12) and np.allclose(tn, tc)
    g32, t32 = oc.batch_shared(f, g.state2pdf, g.P, V, lens, dtype=np.float32, nthreads=2)
    assert np.allclose(g32, gc, atol=2e-5) and np.allclose(t32, tc, rtol=1e-5)
    # Viterbi: NumPy vs C, bit exact in float32
    ft = graphs.to_oracle(o, g, "tropical", np.float32)
    Vq = (np.round(V[0] * 2) / 2).astype(np.float32)
    p1, s1, _, bp1 = o.viterbi(ft, g.state2pdf, g.P, Vq.T.copy(), 7)
    p2, s2, bp2 = oc.viterbi(ft, g.state2pdf, g.P, Vq, 7, dtype=np.float32)
    assert np.array_equal(p1, p2[:7]) and s1 == s2 and np.array_equal(bp1, bp2)


def test_wsj_fixture_statistics(wl):
    """The converted reference graphs keep the statistics measured in SURVEY.md section 6."""
    den = wl.load_npz_graph(os.path.join(HERE, "golden", "den_fsm_wsj.npz"))
    assert (den.S, den.src.size, den.init_idx.size, den.final_idx.size, den.P) == (3032, 50984, 38, 942, 84)
    assert int((den.src == den.dst).sum()) == 1518
    num = wl.load_npz_graph(os.path.join(HERE, "golden", "num_fsm_wsj.npz"))
    assert num.S == 454


def test_wsj_oracle_golden_reproduces(oracle, wl):
    """The committed oracle outputs on the reference's numerator graph are reproducible."""
    o, oc = oracle
    g = wl.load_npz_graph(os.path.join(HERE, "golden", "num_fsm_wsj.npz"))
    z = np.load(os.path.join(HERE, "golden", "num_fsm_wsj_oracle.npz"))
    gam, ttl = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, z["V"], z["lens"], dtype=np.float64)
    assert np.allclose(gam, z["gamma"], atol=1e-6, equal_nan=True) and np.allclose(ttl, z["ttl"], equal_nan=True)
    assert np.isfinite(gam[:2]).all() and np.isnan(gam[2, 0]).all()  # utterance 2 (150 frames) has no accepting path


@pytest.mark.parametrize("gname", ["random", "lexicon", "l2r"])
def test_openfst_text_round_trip(mm, wl, gname):
    """FSM.to_openfst_text (the form misc/benchmark/generatefsm.jl:42-57 writes) read back by FSM.from_openfst_text
    gives the same FSM, state map and pdf count."""
    g = {"random": lambda: wl.random_fsm(60, 7, 3.0, seed=11), "lexicon": lambda: wl.lexicon_fsm(120, 9, seed=2),
         "l2r": lambda: wl.l2r_hmm(3)}[gname]()
    f = wl.to_fsm(mm, g)
    text = f.to_openfst_text(g.state2pdf)
    f2, s2p, P = mm.FSM.from_openfst_text(text)
    assert np.array_equal(s2p, g.state2pdf) and P == int(np.max(g.state2pdf)) + 1
    for a, b in zip(f.arc_lists(), f2.arc_lists()):
        assert np.array_equal(a, b)
    assert np.array_equal(f.colptr, f2.colptr) and np.array_equal(f.rowval, f2.rowval) and np.array_equal(f.nzval, f2.nzval)
    assert f2.to_openfst_text(s2p) == text


@pytest.mark.parametrize("name", ["den_fsm_wsj", "num_fsm_wsj"])
def test_product_reader_equals_oracle_parser_on_wsj_graphs(mm, wl, oracle, name):
    """The committed WSJ fixtures (tests/golden/*.npz, made by tests/golden/make_wsj_graphs.py through the PRODUCT
    reader): written as text by the product writer, the product reader and the oracle's parser give the same arrays
    -- those of the fixture.  In the build container the text is also compared with the reference's own file,
    byte for byte."""
    o, _ = oracle
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    g = wl.load_npz_graph(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    text = wl.to_fsm(mm, g).to_openfst_text(g.state2pdf)
    ref = os.path.join("/root/reference/misc/benchmark", name + ".txt")
    if os.path.exists(ref):
        assert open(ref).read() == text
    f, s2p, P = mm.FSM.from_openfst_text(text)
    ii, iw, src, dst, w, fi, fw = f.arc_lists()
    S, init, arcs, final, s2p_o, P_o = o.parse_openfst_text(text)
    assert (f.S1 - 1, P) == (S, P_o) == (int(z["S"]), int(z["P"]))
    assert np.array_equal(s2p, s2p_o) and np.array_equal(s2p, z["state2pdf"])
    assert np.array_equal(ii, [s for s, _ in init]) and np.array_equal(ii, z["init_idx"])
    assert np.array_equal(iw, np.array([x for _, x in init], dtype=np.float32)) and np.array_equal(iw, z["init_w"])
    assert np.array_equal(src, [a[0][0] for a in arcs]) and np.array_equal(src, z["src"])
    assert np.array_equal(dst, [a[0][1] for a in arcs]) and np.array_equal(dst, z["dst"])
    assert np.array_equal(w, np.array([a[1] for a in arcs], dtype=np.float32)) and np.array_equal(w, z["w"])
    assert np.array_equal(fi, [s for s, _ in final]) and np.array_equal(fi, z["final_idx"])
    assert np.array_equal(fw, np.array([x for _, x in final], dtype=np.float32)) and np.array_equal(fw, z["final_w"])
