"""GPU tests of the generic entry (mm_pdfposteriors_ex): the reference's pdfposteriors(fsm, V_hats, C_hats) over its whole
argument space -- three semirings x two float types (test/test_linalg.jl:88-108), FSM{LogSemiring{Float64}}
(test/test_fsms.jl:3-7), any sparse C_hat, any V_hat -- against the oracle's restatement of src/inference.jl:145-161."""
import os

import numpy as np
import pytest

import graphs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available()
    return torch


def lin(g):
    """a GraphSpec with probabilities in place of log-probabilities (ProbSemiring)"""
    import copy

    h = copy.copy(g)
    h.init_w, h.w, h.final_w = np.exp(g.init_w), np.exp(g.w), np.exp(g.final_w)
    return h


def oracle_run(o, g, semiring, dtype, lhs, lens, s2p, P):
    K = o.SEMIRINGS[semiring]
    f = graphs.to_oracle(o, lin(g) if semiring == "prob" else g, semiring, dtype)
    return o.pdfposteriors_batch(f, list(s2p), P, [l.astype(dtype) for l in lhs], [int(x) for x in lens]), K


@pytest.mark.parametrize("semiring", ["log", "tropical", "prob"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_three_semirings_two_float_types(mm, wl, oracle, torch, semiring, dtype):
    o, _ = oracle
    g = wl.random_fsm(40, 6, 3.0, seed=1)
    rng = np.random.default_rng(12)
    B, N = 3, 11
    lens = [11, 7, 4]
    lhs = [rng.standard_normal((g.P, N)) for _ in range(B)]
    if semiring == "prob":
        lhs = [np.exp(x) for x in lhs]
    (g_ref, t_ref), K = oracle_run(o, g, semiring, np.float64, lhs, lens, g.state2pdf, g.P)
    gg = lin(g) if semiring == "prob" else g
    fsm = wl.to_fsm(mm, gg, semiring, dtype)
    cf = mm.compile(fsm, mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    Vh = [mm.expand(x.astype(dtype), L, semiring) for x, L in zip(lhs, lens)]
    gam, ttl = bf.pdfposteriors_generic(Vh)
    assert gam.dtype == dtype and ttl.dtype == dtype
    tol = 1e-10 if dtype == np.float64 else 2e-5
    assert np.allclose(gam, g_ref, rtol=tol, atol=tol)
    assert np.allclose(ttl, t_ref, rtol=tol, atol=tol)
    for b, L in enumerate(lens):
        assert (gam[b][:, L:] == 0).all()
    # ... and through the reference's call shape, which picks the generic entry for float64 (and for a ProbSemiring{Float64}) by itself
    if dtype == np.float64:
        g2, t2 = mm.pdfposteriors(bf, Vh)
        assert np.array_equal(g2, gam) and np.array_equal(t2, ttl)
        assert not bf.has_fast_entry() or semiring == "log"
    elif semiring == "prob":
        # ProbSemiring{Float32} with the FSM's own one-hot map and V_hat of expand()'s form: the FAST kernels (round 6) -- the
        # library's log twins of the FSMs on log V_hat, ttl back as a probability -- asserted by name, against the float64 oracle
        assert bf.has_fast_entry() and ("mm_lane_kernel" in bf.kernels() or "mm_wave_kernel" in bf.kernels()), bf.kernels()
        g2, t2 = mm.pdfposteriors(bf, Vh)
        assert g2.dtype == np.float32 and np.allclose(g2, g_ref, rtol=1e-4, atol=2e-6) and np.allclose(t2, t_ref, rtol=1e-4)
        for b, L in enumerate(lens):
            assert (g2[b][:, L:] == 0).all()
        # a V_hat that expand() did not make still goes to the generic entry: the same numbers as above
        Vx = [v.copy() for v in Vh]
        Vx[0][-1, 0] = 0.25
        g3, _ = mm.pdfposteriors(bf, Vx)
        assert np.isfinite(g3).all()


def test_prob_semiring_on_the_pair_kernels(mm, wl, oracle, torch):
    """A ProbSemiring{Float32} denominator-sized graph on the pair kernels (mm_fbp_kernel by name: the log twin's), device-resident
    likelihoods, against the float64 oracle run in the ProbSemiring itself; zero likelihoods (zero(K)) included; the batch answers
    for redo counts and settings through its twin."""
    o, _ = oracle
    g = wl.lfmmi_denominator(600, 30, seed=4)
    rng = np.random.default_rng(2)
    B, N = 4, 25
    lens = [25, 25, 11, 19]
    lhs = [np.exp(rng.standard_normal((g.P, N))) for _ in range(B)]
    lhs[1][3, 5] = 0.0  # a pdf that cannot emit a frame
    (g_ref, t_ref), K = oracle_run(o, g, "prob", np.float64, lhs, lens, g.state2pdf, g.P)
    cf = mm.compile(wl.to_fsm(mm, lin(g), "prob", np.float32), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    assert bf.semiring == "prob" and "mm_fbp_kernel" in bf.kernels(), bf.kernels()
    V = torch.from_numpy(np.stack([x.T for x in lhs]).astype(np.float32)).cuda()  # [B, N, P] likelihoods
    lt = torch.tensor(lens, dtype=torch.int32, device="cuda")
    bf.reserve(N)
    gam, ttl = bf.pdfposteriors(V, lt)
    assert bf.last_redo_count() == 0
    gam, ttl = gam.cpu().numpy().transpose(0, 2, 1), ttl.cpu().numpy()
    assert np.allclose(gam, g_ref, rtol=1e-4, atol=2e-6) and np.allclose(ttl, t_ref, rtol=1e-4)
    bf.set_exact_policy("f64_first")  # (the wide-exponent kernels of the twin)
    g2, t2 = bf.pdfposteriors(V, lt)
    assert bf.last_exact_first() and np.allclose(g2.cpu().numpy().transpose(0, 2, 1), g_ref, rtol=1e-4, atol=2e-6) and np.allclose(t2.cpu().numpy(), t_ref, rtol=1e-4)


@pytest.mark.parametrize("name", ["l2r3", "rand30", "rand30m"])
def test_float64_against_the_dense_reference_fixtures(mm, wl, torch, name):
    """FSM{LogSemiring{Float64}} on the device against the committed dense forward/backward numbers
    (tests/golden/pin_*.npz: the reference test suite's own independent implementation, random emissions)."""
    path = os.path.join(HERE, "golden", f"pin_{name}.npz")
    z = np.load(path)
    g = wl.load_npz_graph(path)
    s2p, P = z["state2pdf"].astype(np.int32), int(z["P"])
    cf = mm.compile(wl.to_fsm(mm, g, "log", np.float64), mm.statemap(s2p, P))
    B = z["lhs"].shape[0]
    Vh = [mm.expand(z["lhs"][b], int(z["lens"][b]), "log") for b in range(B)]
    gam, ttl = mm.pdfposteriors(mm.batch(*([cf] * B)), Vh)
    assert gam.dtype == np.float64
    assert np.allclose(gam, z["gamma"], rtol=1e-9, atol=1e-12) and np.allclose(ttl, z["ttl"], rtol=1e-10)


def test_general_state_map_and_arbitrary_vhat(mm, wl, oracle, torch):
    """A C_hat with several weighted entries per row (a state that emits a mixture of pdfs) and matrices V_hat that
    expand() did not make (finite values in the phony row, no padding): src/inference.jl:145-161 takes both."""
    import scipy.sparse as sp

    o, _ = oracle
    K = o.LOG
    g = wl.random_fsm(25, 5, 2.5, seed=4)
    S1, P1, N1 = g.S + 1, g.P + 1, 9
    rng = np.random.default_rng(3)
    # C_hat: every state reads its own pdf with weight one and a second pdf with a random log weight; the final state the phony pdf
    rows, cols, vals = [], [], []
    for s in range(g.S):
        rows += [s, s]
        cols += [int(g.state2pdf[s]), int((g.state2pdf[s] + 1 + s % 3) % g.P)]
        vals += [0.0, float(-rng.random() - 0.3)]
        if cols[-1] == cols[-2]:
            rows.pop(), cols.pop(), vals.pop()
    rows.append(g.S), cols.append(g.P), vals.append(0.0)
    Vh = [rng.standard_normal((P1, N1)) for _ in range(2)]
    f = graphs.to_oracle(o, g)
    C_or = o.csc_from_coo(rows, cols, np.asarray(vals), (S1, P1), K)
    g_ref, t_ref = o.pdfposteriors(o.rawunion([f, f]), Vh, [C_or, C_or])
    Cm = mm.GeneralStateMap(sp.csr_matrix((vals, (rows, cols)), shape=(S1, P1)), "log")
    assert Cm.one_hot() is None
    for dtype, tol in ((np.float64, 1e-10), (np.float32, 3e-5)):
        fsm = wl.to_fsm(mm, g, "log", dtype)  # (the precision follows the FSM's K: src/inference.jl:147)
        gam, ttl = mm.pdfposteriors(mm.rawunion(fsm, fsm), Vh, [Cm, Cm])
        assert gam.dtype == dtype
        assert np.allclose(gam, g_ref, rtol=tol, atol=tol) and np.allclose(ttl, t_ref, rtol=tol, atol=tol)
    # a general map that is one-hot after all is recognised and takes the fast kernels (float32, expand()'s V_hat)
    one = mm.GeneralStateMap(sp.csr_matrix((np.zeros(S1), (np.arange(S1), np.append(g.state2pdf, g.P))), shape=(S1, P1)), "log")
    assert one.one_hot() is not None


def test_unexpanded_vhat_is_a_dimension_mismatch(mm, wl, torch):
    """A P x N matrix that expand() (src/inference.jl:54-60) was not applied to, or one with a wrong number of pdfs, is the
    reference's DimensionMismatch (vcat(V_hats...) against blockdiag(C_hats...)', :146-150) -- from the host mirror and
    from the C entry itself, which reads P1 rows per utterance: never an out-of-bounds read."""
    g = wl.random_fsm(30, 5, 2.5, seed=2)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(cf, cf)
    rng = np.random.default_rng(0)
    raw = [rng.standard_normal((g.P, 12)) for _ in range(2)]  # expand() forgotten
    with pytest.raises(mm.DimensionMismatch):
        mm.pdfposteriors(bf, raw)
    with pytest.raises(mm.DimensionMismatch):
        bf.pdfposteriors_generic([rng.standard_normal((g.P + 3, 13)) for _ in range(2)])
    # the C entry: V_hat with P rows handed over directly
    V = torch.randn(2, 13, g.P, device="cuda")
    with pytest.raises(mm.DimensionMismatch):
        bf.pdfposteriors_ex(V)


def test_precision_follows_the_fsm(mm, wl, oracle, torch):
    """src/inference.jl:147 converts V_hat to the FSM's K.  A Float32 FSM given NumPy's default float64 matrices of
    expand()'s form stays on the fast float32 kernels; a Float64 FSM given float32 matrices computes in float64."""
    o, oc = oracle
    g = wl.lfmmi_denominator(300, 20, seed=3)
    rng = np.random.default_rng(5)
    B, N = 3, 25
    lens = [25, 11, 18]
    lhs = [rng.standard_normal((g.P, N)) for _ in range(B)]  # float64
    V = np.stack([x.T for x in lhs]).astype(np.float32)
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, np.asarray(lens, np.int32), dtype=np.float64)
    Vh = [mm.expand(x, L) for x, L in zip(lhs, lens)]
    assert Vh[0].dtype == np.float64
    cf32 = mm.compile(wl.to_fsm(mm, g, "log", np.float32), mm.statemap(g.state2pdf, g.P))
    bf32 = mm.batch(*([cf32] * B))
    gam, ttl = mm.pdfposteriors(bf32, Vh)
    assert gam.dtype == np.float32
    assert np.allclose(gam.transpose(0, 2, 1), g_ref, rtol=1e-4, atol=2e-5) and np.allclose(ttl, t_ref, rtol=1e-5)
    cf64 = mm.compile(wl.to_fsm(mm, g, "log", np.float64), mm.statemap(g.state2pdf, g.P))
    gam64, ttl64 = mm.pdfposteriors(mm.batch(*([cf64] * B)), [v.astype(np.float32) for v in Vh])
    assert gam64.dtype == np.float64
    assert np.allclose(gam64.transpose(0, 2, 1), g_ref, rtol=1e-5, atol=1e-6) and np.allclose(ttl64, t_ref, rtol=1e-6)


def test_generic_entry_is_asynchronous_and_capturable(mm, wl, oracle, torch):
    """mm_pdfposteriors_ex keeps its workspace with the batch (mm_batch_reserve_ex): a steady-state call allocates nothing
    and does not wait -- it can be captured in a hipGraph; a replay gives the bits of the eager call, which match the
    oracle."""
    o, _ = oracle
    g = wl.random_fsm(40, 6, 3.0, seed=1)
    B, N = 4, 15
    rng = np.random.default_rng(1)
    lens = [15, 9, 15, 3]
    lhs = [rng.standard_normal((g.P, N)) for _ in range(B)]
    (g_ref, t_ref), _K = oracle_run(o, g, "log", np.float64, lhs, lens, g.state2pdf, g.P)
    cf = mm.compile(wl.to_fsm(mm, g, "log", np.float64), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    Vh = torch.from_numpy(np.ascontiguousarray(np.stack([mm.expand(x, L).T for x, L in zip(lhs, lens)]))).cuda()  # [B][N1][P1] float64
    bf.reserve_ex(np.float64, N + 1)
    gam0, ttl0 = (t.clone() for t in bf.pdfposteriors_ex(Vh))
    assert np.allclose(gam0.cpu().numpy().transpose(0, 2, 1), g_ref, rtol=1e-10, atol=1e-10)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        bf.pdfposteriors_ex(Vh)
    torch.cuda.synchronize()
    gam, ttl = torch.zeros_like(gam0), torch.zeros_like(ttl0)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        bf.pdfposteriors_ex(Vh, gamma=gam, ttl=ttl)
    for _ in range(2):
        gam.zero_()
        ttl.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(gam, gam0) and torch.equal(ttl, ttl0)


@pytest.mark.parametrize("S,P", [(25, 5), (130, 70)])
def test_prob_semiring_emission_gemm_on_the_matrix_cores(mm, wl, oracle, torch, S, P):
    """ProbSemiring FSMs with a general (mixture) state map in float32: lhs = C_hat * V_hat (src/inference.jl:150) is a plain
    GEMM there, and the generic entry computes it on the matrix cores (mm_prob_emission_mfma_kernel, v_mfma_f32_32x32x2f32)
    before the recursion kernel -- the one MFMA site BASELINE's north star names.  A DENSE C_hat (every state a mixture of all
    pdfs), matrices V_hat that expand() did not make, two utterances with different maps (one of them the FSM's own one-hot
    map: no GEMM for it); against the float64 oracle, and against the same call with the GEMM switched off."""
    import scipy.sparse as sp

    o, _ = oracle
    K = o.PROB
    g = wl.random_fsm(S, P, 3.0, seed=7)
    S1, P1, N1 = g.S + 1, g.P + 1, 45
    rng = np.random.default_rng(S)
    Cd = rng.random((S1, P1)) * (rng.random((S1, P1)) < 0.6)  # a dense mixture map, 60 % filled
    Cd[:g.S, g.P] = 0.0  # real states do not read the phony pdf,
    Cd[g.S, :] = 0.0     # the final state reads nothing else
    Cd[g.S, g.P] = 1.0
    Cd[np.arange(g.S), g.state2pdf] += 0.5  # (no empty row)
    Cd /= Cd.sum(1, keepdims=True)           # (mixture weights: unnormalised, 44 frames of 130 states overflow float32 -- in the reference too)
    rows, cols = np.nonzero(Cd)
    vals = Cd[rows, cols]
    Vh = [np.exp(0.5 * rng.standard_normal((P1, N1))) for _ in range(2)]
    f = graphs.to_oracle(o, lin(g), "prob", np.float64)
    C_or = o.csc_from_coo(list(rows), list(cols), vals, (S1, P1), K)
    own = o.csc_from_coo(list(range(S1)), list(g.state2pdf) + [g.P], np.ones(S1), (S1, P1), K)
    g_ref, t_ref = o.pdfposteriors(o.rawunion([f, f]), Vh, [C_or, own])
    Cm = mm.GeneralStateMap(sp.csr_matrix((vals, (rows, cols)), shape=(S1, P1)), "prob")
    fsm = wl.to_fsm(mm, lin(g), "prob", np.float32)
    cf = mm.compile(fsm, mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(cf, cf)
    gam, ttl = bf.pdfposteriors_generic(Vh, [Cm, None])
    assert "mm_prob_emission_mfma_kernel" in bf.kernels_generic() and "1 utterances" in bf.kernels_generic(), bf.kernels_generic()
    assert gam.dtype == np.float32
    assert np.allclose(gam, g_ref, rtol=3e-5, atol=3e-6) and np.allclose(ttl, t_ref, rtol=3e-5)
    # float64 FSMs: no float64 GEMM (the recursion kernel gathers), same answer
    bf64 = mm.batch(*([mm.compile(wl.to_fsm(mm, lin(g), "prob", np.float64), mm.statemap(g.state2pdf, g.P))] * 2))
    g64, t64 = bf64.pdfposteriors_generic(Vh, [Cm, None])
    assert "mfma" not in bf64.kernels_generic()
    assert np.allclose(g64, g_ref, rtol=1e-10, atol=1e-12) and np.allclose(t64, t_ref, rtol=1e-10)
