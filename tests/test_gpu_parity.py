"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle
on the same seeded inputs.  Tolerances follow BASELINE.json's north_star:
log-posteriors within 1e-4 relative, Viterbi back-pointers bit exact."""
import os

import numpy as np
import pytest

import graphs

pytestmark = pytest.mark.gpu

RTOL_LOGPOST = 1e-4  # |d log gamma| <= 1e-4 * max(|log gamma|, 1)
GAMMA_FLOOR = 1e-30


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available()
    return torch


def check_gamma(g, g_ref, lens):
    g = np.asarray(g, dtype=np.float64)
    assert np.isfinite(g).all()
    for b, L in enumerate(lens):
        assert (g[b, L:] == 0).all(), "frames beyond the sequence length must be exact zeros"
    assert np.abs(g - g_ref).max() <= 2e-5
    m = g_ref > GAMMA_FLOOR
    assert (g[m] > 0).all()
    lg, lr = np.log(g[m]), np.log(g_ref[m])
    assert (np.abs(lg - lr) <= RTOL_LOGPOST * np.maximum(np.abs(lr), 1.0)).all(), np.abs(lg - lr).max()
    for b, L in enumerate(lens):
        assert np.allclose(g[b, :L].sum(-1), 1.0, atol=1e-5)


def run_shared(mm, wl, oracle, torch, g, B, N, lens, seed=0, scale=1.0):
    o, oc = oracle
    rng = np.random.default_rng(seed)
    V = (scale * rng.standard_normal((B, N, g.P))).astype(np.float32)
    lens = np.asarray(lens, dtype=np.int32)
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    gam, ttl = bf.pdfposteriors(torch.from_numpy(V).cuda(), torch.from_numpy(lens).cuda())
    torch.cuda.synchronize()
    return gam.cpu().numpy(), ttl.cpu().numpy(), g_ref, t_ref, lens


def test_known_answer_demo_notebook(mm, wl, torch):
    """examples/demo.ipynb cell 13: 3-state left-to-right HMM, lhs = zeros(3, 5)."""
    g = wl.l2r_hmm(3)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    gam, ttl = mm.pdfposteriors(cf, [mm.expand(np.zeros((3, 5), dtype=np.float32))])
    ref = np.array([[1, .5, 1 / 6, 0, 0], [0, .5, 2 / 3, .5, 0], [0, 0, 1 / 6, .5, 1]])
    assert np.allclose(gam[0], ref, atol=1e-6)
    assert np.isclose(ttl[0], np.log(6 / 32), atol=1e-6)


def test_known_answer_batch_varlen(mm, wl, torch):
    """test/test_algorithms.jl:218-248: the same FSM twice, lhs = ones(3, 7), lengths [5, 7]."""
    g = wl.l2r_hmm(3)
    f = wl.to_fsm(mm, g)
    C = mm.statemap(g.state2pdf, g.P)
    lhs = np.ones((3, 7), dtype=np.float32)
    gam, ttl = mm.pdfposteriors(mm.rawunion(f, f), [mm.expand(lhs, 5), mm.expand(lhs, 7)], [C, C])
    r1 = np.array([[1, .5, 1 / 6, 0, 0], [0, .5, 2 / 3, .5, 0], [0, 0, 1 / 6, .5, 1]])
    r2 = np.array([[1, 2 / 3, .4, .2, 1 / 15, 0, 0], [0, 1 / 3, 8 / 15, .6, 8 / 15, 1 / 3, 0],
                   [0, 0, 1 / 15, .2, .4, 2 / 3, 1]])
    assert np.allclose(gam[0][:, :5], r1, atol=1e-6) and (gam[0][:, 5:] == 0).all()
    assert np.allclose(gam[1], r2, atol=1e-6)
    assert np.allclose(ttl, [3.3260236, 4.8560199], atol=1e-5)


@pytest.mark.parametrize("name,B,N", [("rand40", 5, 33), ("ergodic64", 4, 50), ("wide", 2, 9), ("lexicon", 2, 40),
                                      ("lfmmi", 3, 40)])
def test_pdfposteriors_vs_oracle(mm, wl, oracle, torch, name, B, N):
    g = {"rand40": lambda: wl.random_fsm(40, 6, 3.0, seed=1), "ergodic64": lambda: wl.dense_ergodic(64),
         "wide": lambda: wl.wide_row_fsm(), "lexicon": lambda: wl.lexicon_fsm(1200, 20),
         "lfmmi": lambda: wl.lfmmi_denominator(2000, 84)}[name]()
    lens = [N] + [max(1, N - 7 * (b + 1)) for b in range(B - 1)]
    gam, ttl, g_ref, t_ref, lens = run_shared(mm, wl, oracle, torch, g, B, N, lens)
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-4)


def test_long_sequence_drift(mm, wl, oracle, torch):
    """N = 1500 frames: the float32 engine must not drift (normalised recursion)."""
    g = wl.random_fsm(60, 8, 3.0, seed=3)
    gam, ttl, g_ref, t_ref, lens = run_shared(mm, wl, oracle, torch, g, 2, 1500, [1500, 1111], scale=2.0)
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5)


def test_edge_cases(mm, wl, oracle, torch):
    o, oc = oracle
    g = wl.random_fsm(12, 4, 2.0, seed=5)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    rng = np.random.default_rng(1)
    # length-1 and length-0 utterances next to a full one
    V = rng.standard_normal((3, 6, g.P)).astype(np.float32)
    lens = np.array([6, 1, 0], dtype=np.int32)
    gam, ttl = mm.batch(cf, cf, cf).pdfposteriors(V, lens)
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V[:2], lens[:2], dtype=np.float64)
    check_gamma(gam[:2], g_ref, lens[:2])
    assert np.allclose(ttl[:2], t_ref, rtol=1e-5, atol=1e-5)
    assert (gam[2] == 0).all() and ttl[2] == -np.inf  # no path of length 0: defined as gamma = 0, ttl = -inf
    # an utterance whose only paths are impossible (all emissions -inf on one frame)
    V2 = V[:1].copy()
    V2[0, 3, :] = -np.inf
    gam2, ttl2 = mm.batch(cf).pdfposteriors(V2, np.array([6], dtype=np.int32))
    assert (gam2 == 0).all() and ttl2[0] == -np.inf
    # emission spike: one pdf wins a frame by 60 nats
    V3 = V[:1].copy()
    V3[0, 2, 1] += 60.0
    g3, t3 = mm.batch(cf).pdfposteriors(V3, None)
    r3, rt3 = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V3, None, dtype=np.float64)
    check_gamma(g3, r3, [6])
    assert np.allclose(t3, rt3, rtol=1e-5)


def test_distinct_graphs_block_diagonal(mm, wl, oracle, torch):
    """Numerator-style batch: a different FSM per utterance (rawunion, src/fsmops.jl:28-36)."""
    o, oc = oracle
    P = 7
    gs = [wl.random_fsm(S, P, 2.5, seed=10 + S) for S in (9, 31, 70)]
    fs = [wl.to_fsm(mm, g) for g in gs]
    Cs = [mm.statemap(g.state2pdf, P) for g in gs]
    rng = np.random.default_rng(2)
    N = 21
    lens = [21, 13, 17]
    Vs = [rng.standard_normal((P, N)).astype(np.float32) for _ in gs]
    gam, ttl = mm.pdfposteriors(mm.rawunion(*fs), [mm.expand(v, L) for v, L in zip(Vs, lens)], Cs)
    for b, g in enumerate(gs):
        gr, tr = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, P, Vs[b].T[None], [lens[b]], dtype=np.float64)
        check_gamma(gam[b].T[None], gr, [lens[b]])
        # (log Z of these small graphs is near 0 while the per-frame terms it is the sum of are O(1..10): the
        # float32 resolution of those terms, ~1e-6 each, bounds the absolute error)
        assert np.isclose(ttl[b], tr[0], rtol=1e-5, atol=1e-5)


def test_alpha_beta_export(mm, wl, oracle, torch):
    """alpha-recursion / beta-recursion (src/inference.jl:62-74, 99-110) = state_A / state_B of pdfposteriors."""
    o, oc = oracle
    g = wl.random_fsm(25, 5, 3.0, seed=7)
    of = graphs.to_oracle(o, g)
    C = mm.statemap(g.state2pdf, g.P)
    f = wl.to_fsm(mm, g)
    rng = np.random.default_rng(3)
    N, lens = 14, [14, 9]
    Vs = [rng.standard_normal((g.P, N)).astype(np.float32) for _ in lens]
    Vh = [mm.expand(v, L) for v, L in zip(Vs, lens)]
    A = mm.αrecursion(mm.rawunion(f, f), Vh, [C, C])
    Bm = mm.βrecursion(mm.rawunion(f, f), Vh, [C, C])
    S1 = g.S + 1
    assert A.shape == (2 * S1, N + 1) and Bm.shape == (2 * S1, N + 1)
    for b in range(2):
        _, _, Ar, Br = oc.single(of, g.state2pdf, g.P, o.expand(Vs[b].astype(np.float64), lens[b], o.LOG), want_ab=True)
        for got, ref in ((A[b * S1:(b + 1) * S1], Ar), (Bm[b * S1:(b + 1) * S1], Br)):
            assert np.array_equal(np.isneginf(got), np.isneginf(ref))
            m = np.isfinite(ref)
            assert np.allclose(got[m], ref[m], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("name", ["chain", "rand", "lexicon", "wide"])
def test_viterbi_bit_exact(mm, wl, oracle, torch, name):
    """Tropical recursion + back-pointers: bit exact against the C oracle (float32, same adds)."""
    o, oc = oracle
    if name == "chain":  # test/test_algorithms.jl:262-284: a -> b -> c -> d, lhs = ones(4, 4) => path 1 2 3 4
        g = wl.GraphSpec("chain4", 4, np.array([0]), np.array([0.0]), np.arange(3), np.arange(1, 4), np.zeros(3),
                         np.array([3]), np.array([0.0]), np.arange(4, dtype=np.int32), 4)
        Vs, lens = np.ones((1, 4, 4), dtype=np.float32), [4]
    else:
        g = {"rand": lambda: wl.random_fsm(50, 6, 3.0, seed=4), "lexicon": lambda: wl.lexicon_fsm(900, 20),
             "wide": lambda: wl.wide_row_fsm()}[name]()
        rng = np.random.default_rng(5)
        N = 37
        lens = [37, 20, 1]
        # quantised scores make exact ties frequent, so the lowest-index rule is exercised
        Vs = (np.round(rng.standard_normal((3, N, g.P)) * 2) / 2).astype(np.float32)
    f = wl.to_fsm(mm, g, semiring="tropical")
    of = graphs.to_oracle(o, g, "tropical", np.float32)
    cf = mm.compile(f, mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * len(lens)))
    path, score, bp = bf.viterbi(Vs, np.asarray(lens, dtype=np.int32), return_backpointers=True)
    # ... and without the int32 table: the row-lane kernels with their one-byte back-pointers (mm_vit_kernel +
    # mm_vit_backtrace_kernel) where the graph fits them (`wide` has a row of more than 255 arcs: it does not)
    path2, score2 = bf.viterbi(Vs, np.asarray(lens, dtype=np.int32))
    assert ("mm_vit_kernel" in bf.kernels("tropical")) == (name != "wide")
    S1 = g.S + 1
    for b, L in enumerate(lens):
        pr, sr, bpr = oc.viterbi(of, g.state2pdf, g.P, Vs[b], L, dtype=np.float32)
        assert np.array_equal(path[b], pr), (b, path[b], pr)
        assert score[b] == sr
        assert np.array_equal(bp[:, b * S1:(b + 1) * S1], bpr)
        assert np.array_equal(path2[b], pr) and score2[b] == sr, (b, path2[b], pr)
    if name == "chain":
        assert path[0].tolist() == [0, 1, 2, 3]


def test_reference_output_layout(mm, wl, oracle, torch):
    """gamma written straight into the reference's B x P x N column-major layout (strides)."""
    g = wl.random_fsm(20, 5, 3.0, seed=8)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    B, N = 3, 11
    bf = mm.batch(*([cf] * B))
    V = torch.randn(B, N, g.P, device="cuda")
    g1, t1 = bf.pdfposteriors(V)
    out = torch.empty(N, g.P, B, device="cuda").permute(2, 0, 1)  # element (b, n, p): b fastest
    g2, t2 = bf.pdfposteriors(V, out=out)
    assert torch.allclose(g1, g2, atol=1e-6) and torch.allclose(t1, t2)


def test_errors(mm, wl, torch):
    g = wl.random_fsm(10, 4, 2.0, seed=9)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(cf, cf)
    with pytest.raises(mm.DimensionMismatch):
        bf.pdfposteriors(np.zeros((3, 5, g.P), dtype=np.float32))
    with pytest.raises(mm.DimensionMismatch):
        bf.pdfposteriors(np.zeros((2, 5, g.P + 1), dtype=np.float32))
    with pytest.raises(mm.MarkovModelsAMDError):
        bf.viterbi(np.zeros((2, 5, g.P), dtype=np.float32))  # log-semiring batch


@pytest.mark.parametrize("name", ["den_fsm_wsj", "num_fsm_wsj"])
def test_wsj_graphs_against_committed_golden(mm, wl, torch, name):
    """The reference's own benchmark graphs (misc/benchmark/*_fsm_wsj.txt, converted by
    tests/golden/make_wsj_graphs.py) against the committed float64 oracle outputs."""
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    g = wl.load_npz_graph(os.path.join(here, "golden", name + ".npz"))
    z = np.load(os.path.join(here, "golden", name + "_oracle.npz"))
    V, lens = z["V"], z["lens"]
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    gam, ttl = mm.batch(*([cf] * V.shape[0])).pdfposteriors(V, lens)
    ok = np.isfinite(z["ttl"])
    check_gamma(gam[ok], z["gamma"][ok].astype(np.float64), lens[ok])
    assert np.allclose(ttl[ok], z["ttl"][ok], rtol=1e-5, atol=5e-4)  # log Z is a sum over ~N frames of O(1..10) terms
    # utterances without an accepting path: the reference gives NaN (0/0), the engine gamma = 0, ttl = -inf
    assert (gam[~ok] == 0).all() and np.isneginf(ttl[~ok]).all()
    ct = mm.compile(wl.to_fsm(mm, g, semiring="tropical"), mm.statemap(g.state2pdf, g.P))
    path, score = mm.batch(ct).viterbi(V[:1], lens[:1])
    assert np.array_equal(path[0], z["path"]) and score[0] == z["score"]


def _with_env(env, fn):
    import os

    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("kernel", ["quad", "row", "item", "wave"])
def test_wsj_numerator_forced_kernels(mm, wl, torch, kernel):
    """The reference's numerator graph (left-to-right, depth 165: the values of one frame span far more
    than the float range) forced through each pdfposteriors kernel: on the linear-domain kernels most rows
    take the exact fallback (dead-row test, LDS-resident CSR / global CSR walk), which the default kernel
    selection never exercises at this scale.  Pinned to the committed float64 oracle output."""
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    g = wl.load_npz_graph(os.path.join(here, "golden", "num_fsm_wsj.npz"))
    z = np.load(os.path.join(here, "golden", "num_fsm_wsj_oracle.npz"))
    V, lens = z["V"], z["lens"]

    def run():
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        return mm.batch(*([cf] * V.shape[0])).pdfposteriors(V, lens)

    gam, ttl = _with_env({"MM_DEBUG": "1", "MM_KERNEL": kernel}, run)
    ok = np.isfinite(z["ttl"])
    check_gamma(gam[ok], z["gamma"][ok].astype(np.float64), lens[ok])
    assert np.allclose(ttl[ok], z["ttl"][ok], rtol=1e-5, atol=5e-4)
    assert (gam[~ok] == 0).all() and np.isneginf(ttl[~ok]).all()


@pytest.mark.parametrize("offset", [-300.0, 300.0])
def test_emission_scale(mm, wl, oracle, torch, offset):
    """Log-likelihoods far from 0 (GMM-style values around -300 nats, or +300): posteriors are invariant, log Z
    shifts by len * offset, and the linear-domain kernels must stay on their fast path (the per-frame emission
    maximum is part of the normaliser)."""
    g = wl.lfmmi_denominator(600, 40, seed=5)
    B, N = 3, 40
    lens = [40, 31, 9]
    o, oc = oracle
    rng = np.random.default_rng(17)
    V = (1.5 * rng.standard_normal((B, N, g.P)) + offset).astype(np.float32)  # both sides see the float32 values
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V.astype(np.float64),
                                   np.asarray(lens, dtype=np.int32), dtype=np.float64)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    gam, ttl = mm.batch(*([cf] * B)).pdfposteriors(V, np.asarray(lens, dtype=np.int32))
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=2e-6)


def test_lfmmi_loss_and_gradient(mm, wl, oracle, torch):
    """The caller's step (examples/test_cuda.jl:140-152): loss = -sum(ttl_num - ttl_den), gradient
    = gamma_den - gamma_num, checked against the oracle and against finite differences of the oracle."""
    o, oc = oracle
    P, N = 6, 15
    den = wl.random_fsm(30, P, 3.0, seed=21)
    nums = [wl.random_fsm(S, P, 2.0, seed=30 + S) for S in (8, 11, 9)]
    lens = np.array([15, 12, 9], dtype=np.int32)
    rng = np.random.default_rng(7)
    V = rng.standard_normal((3, N, P)).astype(np.float32)
    cden = mm.compile(wl.to_fsm(mm, den), mm.statemap(den.state2pdf, P))
    bden = mm.batch(cden, cden, cden)
    bnum = mm.batch(*[mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, P)) for g in nums])
    Vt = torch.from_numpy(V).cuda().requires_grad_(True)
    loss, tn, td = mm.lfmmi_loss(Vt, bnum, bden, torch.from_numpy(lens).cuda())
    loss.backward()

    def ref_loss(Vx):
        tot = 0.0
        gd, tdn = oc.batch_shared(graphs.to_oracle(o, den), den.state2pdf, P, Vx, lens, dtype=np.float64)
        gn = np.zeros_like(gd)
        for b, g in enumerate(nums):
            gb, tb = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, P, Vx[b:b + 1], lens[b:b + 1], dtype=np.float64)
            gn[b] = gb[0]
            tot -= tb[0] - tdn[b]
        return tot, gd - gn

    l_ref, g_ref = ref_loss(V.astype(np.float64))
    assert np.isclose(float(loss.detach()), l_ref, rtol=1e-5, atol=1e-4)
    assert np.allclose(Vt.grad.cpu().numpy(), g_ref, atol=2e-5)
    # the analytic gradient is the derivative of the oracle's loss (central differences, a few entries)
    for (b, n, p) in [(0, 3, 1), (1, 7, 4), (2, 0, 2)]:
        Vp, Vm = V.astype(np.float64).copy(), V.astype(np.float64).copy()
        Vp[b, n, p] += 1e-4
        Vm[b, n, p] -= 1e-4
        fd = (ref_loss(Vp)[0] - ref_loss(Vm)[0]) / 2e-4
        assert np.isclose(fd, g_ref[b, n, p], atol=1e-5)


@pytest.mark.parametrize("env", [{"MM_KERNEL": "item"}, {"MM_KERNEL": "quad"}, {"MM_KERNEL": "row"}, {"MM_KERNEL": "pair"},
                                 {"MM_KERNEL": "quad", "MM_KQ": "1"}, {"MM_KERNEL": "quad", "MM_KQ": "2", "MM_NWAVES": "3"},
                                 {"MM_KERNEL": "quad", "MM_KQ": "15"}, {"MM_KERNEL": "quad", "MM_KQ": "7"},
                                 {"MM_KERNEL": "item", "MM_BIGV": "1"}, {"MM_KERNEL": "item", "MM_BIGV": "1", "MM_NITEMS": "0"}])
def test_kernel_variants_agree(mm, wl, oracle, torch, env):
    """The general (item) kernel, the row kernel, the quad kernel with a streamed overflow (virtual lanes: KQ
    too small for the graph), an 8-wave geometry and a 16-wave one, and the item kernel with its state vectors in
    global memory (MM_BIGV: the path of FSMs beyond the LDS) all give the oracle's posteriors.  (The
    switches are test aids: read once at batch creation, and only under MM_DEBUG.)"""
    o, oc = oracle
    g = wl.lfmmi_denominator(600, 40, seed=5)
    rng = np.random.default_rng(11)
    B, N = 3, 25
    V = (1.5 * rng.standard_normal((B, N, g.P))).astype(np.float32)
    lens = np.array([25, 18, 7], dtype=np.int32)
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64)

    def run():
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        return mm.batch(cf, cf, cf).pdfposteriors(V, lens)

    gam, ttl = _with_env(dict(env, MM_DEBUG="1"), run)
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-4)


def test_more_utterances_than_cus_longest_first(mm, wl, oracle, torch):
    """More utterances than compute units with different lengths: the workgroups are handed out longest
    first (mm_length_order_kernel); every utterance must still land in its own output rows."""
    g = wl.random_fsm(30, 5, 3.0, seed=11)
    B, N = 333, 12
    rng = np.random.default_rng(5)
    lens = rng.integers(0, N + 1, size=B).astype(np.int32)
    gam, ttl, g_ref, t_ref, lens = run_shared(mm, wl, oracle, torch, g, B, N, lens, seed=6)
    ok = np.isfinite(t_ref)
    check_gamma(gam[ok], g_ref[ok], lens[ok])
    assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=1e-5)
    assert (gam[~ok] == 0).all() and np.isneginf(ttl[~ok]).all()


def test_call_is_capturable_in_a_hip_graph(mm, wl, torch):
    """Once the workspace has its size a pdfposteriors call only launches kernels on the caller's stream:
    it can be captured in a hipGraph and replayed."""
    g = wl.random_fsm(200, 10, 4.0, seed=4)
    B, N = 8, 30
    # (the pair kernels and the float64 exact kernels behind them: forced -- the engine prefers the wave kernel for a graph this small)
    bf = _with_env({"MM_DEBUG": "1", "MM_KERNEL": "pair"},
                   lambda: mm.batch(*([mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * B)))
    assert "mm_fbp_kernel" in bf.kernels(), bf.kernels()
    V = torch.randn(B, N, g.P, device="cuda")
    lens = torch.tensor([N, N - 3, 5, 1, N, 0, 17, N], dtype=torch.int32, device="cuda")
    gamma = torch.empty(B, N, g.P, device="cuda")
    _, t0 = bf.pdfposteriors(V, lens, out=gamma)
    g0 = gamma.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # (torch wants the warm-up on a side stream)
        bf.pdfposteriors(V, lens, out=gamma)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _, t1 = bf.pdfposteriors(V, lens, out=gamma)
    gamma.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(gamma, g0) and torch.equal(t0, t1)


@pytest.mark.parametrize("kind", ["split", "wave", "viterbi"])
def test_other_kernel_families_are_capturable(mm, wl, torch, kind):
    """The same for the kernels the test above does not reach: the teams of the split pair kernels (their exchange buffers
    are zeroed by a memset node before every replay), the wave kernel on a batch of different graphs, and the Viterbi
    kernel with its back-trace.  A replay must give the bits of the eager call."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if kind == "split":
        g = wl.load_npz_graph(os.path.join(here, "den_fsm_wsj.npz"))
        cfs = [mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * 6
        P, want = g.P, "mm_fbs_kernel"
    elif kind == "wave":
        gs = [wl.load_npz_graph(os.path.join(here, "num_fsm_wsj.npz")), wl.lexicon_fsm(300, 20, seed=2, hubs=1), wl.lexicon_fsm(700, 84, seed=5, hubs=2)]
        P = max(g.P for g in gs)
        cfs = [mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, P)) for g in gs] * 2
        want = "mm_wave_kernel"
    else:
        g = wl.lexicon_fsm(1500, 60, seed=5, hubs=2)
        cfs = [mm.compile(wl.to_fsm(mm, g, "tropical"), mm.statemap(g.state2pdf, g.P))] * 6
        P, want = g.P, "mm_vit_kernel"
    B, N = len(cfs), 40
    bf = mm.batch(*cfs)
    V = torch.randn(B, N, P, device="cuda")
    lens = torch.tensor([N, N - 3, 5, 1, N, 17], dtype=torch.int32, device="cuda")
    gamma = torch.empty(B, N, P, device="cuda")

    def call():
        if kind == "viterbi":
            return bf.viterbi(V, lens)
        return bf.pdfposteriors(V, lens, out=gamma)

    a0, b0 = (x.clone() for x in call())
    redo_eager = bf.last_redo_count() if kind != "viterbi" else 0
    assert want in bf.kernels("tropical" if kind == "viterbi" else "log"), bf.kernels()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        call()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        a1, b1 = call()
    for _ in range(2):
        a1.zero_()
        b1.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(a1, a0) and torch.equal(b1, b0)
    if kind != "viterbi":
        # the replay hands exactly the utterances to the exact kernels that the eager call did (the two shortest ones of the split
        # case have a frame whose forward and backward mass overlap below 2^-20; the wave kernel marks nothing) -- and no team timed
        # out in the replay (every utterance would be marked: the exchange areas are zeroed by the library's own kernel, round 6)
        assert bf.last_redo_count() == redo_eager and redo_eager <= (2 if kind == "split" else 0)


@pytest.mark.parametrize("kind", ["export_pairs", "export_teams", "stream_teams", "prob_twin"])
def test_round6_entries_are_capturable(mm, wl, torch, kind):
    """What round 6 added, captured in a hipGraph and replayed on new inputs -- the bits of the eager call: the alpha / beta export on
    the pair kernels and on the team kernels (their exchange areas zeroed by the library's own kernel, the mark count written to pinned
    memory by a kernel of the graph), the stream kernels as teams (a zeroed exchange area too), a ProbSemiring batch on its log twins
    (the logarithm pass and the ttl pass are nodes of the graph)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    N = 33
    if kind == "export_pairs":
        g = wl.lfmmi_denominator(900, 40, seed=31)
        bf = mm.batch(*([mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * 5))
        assert "mm_fbx_kernel" in bf.kernels("export")
    elif kind == "export_teams":
        g = wl.load_npz_graph(os.path.join(here, "den_fsm_wsj.npz"))
        bf = mm.batch(*([mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * 5))
        assert "mm_fbsx_kernel" in bf.kernels("export")
    elif kind == "stream_teams":
        g = wl.lfmmi_denominator(2400, 300, seed=32)
        bf = _with_env({"MM_DEBUG": "1", "MM_KERNEL": "stream", "MM_STREAM_H": "2"},
                       lambda: mm.batch(*([mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * 5)))
        assert "mm_stream_kernel" in bf.kernels() and "teams of 2" in bf.kernels(), bf.kernels()
    else:
        import copy

        g = wl.lfmmi_denominator(700, 30, seed=33)
        gl = copy.copy(g)
        gl.w, gl.final_w, gl.init_w = np.exp(g.w), np.exp(g.final_w), np.exp(g.init_w)
        bf = mm.batch(*([mm.compile(wl.to_fsm(mm, gl, "prob", np.float32), mm.statemap(g.state2pdf, g.P))] * 5))
        assert bf.has_fast_entry() and "mm_fbp_kernel" in bf.kernels(), bf.kernels()
    B = 5
    lens = torch.tensor([N, N - 4, 9, 1, N], dtype=torch.int32, device="cuda")
    V = torch.randn(B, N, g.P, device="cuda")
    if kind == "prob_twin":
        V = torch.exp(0.7 * V)  # likelihoods
    bf.reserve(N)
    # (the export hands the utterances whose alpha / beta leave float32's range to the item kernel, and -- under the default policy --
    # starts there when the export before handed over more than half: a capture never does.  Pinned, the eager call launches what the
    # captured one did, and the bits agree)
    bf.set_exact_policy("f32_first")

    def call():
        if kind.startswith("export"):
            return bf.alpharecursion(V, lens), bf.betarecursion(V, lens)
        return bf.pdfposteriors(V, lens)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        call()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        a1, b1 = call()
    rng = np.random.default_rng(34)
    for _ in range(3):  # new inputs in the captured buffer
        Vn = torch.from_numpy(rng.standard_normal((B, N, g.P)).astype(np.float32)).cuda()
        V.copy_(torch.exp(0.7 * Vn) if kind == "prob_twin" else Vn)
        graph.replay()
        torch.cuda.synchronize()
        ga, gb = a1.clone(), b1.clone()
        ea, eb = call()
        torch.cuda.synchronize()
        assert torch.equal(ga, ea) and torch.equal(gb, eb), (kind, float((ga - ea).abs().nan_to_num().max()))
        if not kind.startswith("export"):
            assert bf.last_redo_count() == 0


def test_tropical_beta_and_maxstateposteriors(mm, wl, oracle, torch):
    """beta-recursion with K = TropicalSemiring (src/inference.jl:99-110) and the max-marginals mu = alpha (*) beta
    (/) best (maxstateposteriors, docs/src/inference.md:5): against the NumPy oracle's generic recursions, and
    the defining properties -- mu <= 0, mu = 0 exactly along the Viterbi path, every frame's maximum is 0."""
    o, _ = oracle
    for g, N, lens in ((wl.random_fsm(30, 5, 3.0, seed=9), 12, [12, 7]), (wl.lexicon_fsm(80, seed=2), 25, [25, 25]),
                       (wl.wide_row_fsm(seed=1), 9, [9, 4])):
        K = o.TROPICAL
        of = graphs.to_oracle(o, g, "tropical")
        C = mm.statemap(g.state2pdf, g.P)
        f = wl.to_fsm(mm, g, "tropical")
        rng = np.random.default_rng(3)
        Vs = [rng.standard_normal((g.P, N)).astype(np.float32) for _ in lens]
        Vh = [mm.expand(v, L) for v, L in zip(Vs, lens)]
        fu = mm.rawunion(f, f)
        Bm = mm.βrecursion(fu, Vh, [C, C])
        A = mm.αrecursion(fu, Vh, [C, C])
        mu = mm.maxstateposteriors(fu, Vh, [C, C])
        paths, scores = mm.bestpath(fu, Vh, [C, C])
        S1 = g.S + 1
        Co = o.statemap(g.state2pdf, g.P, K)
        for b, L in enumerate(lens):
            lhs = o.spmm_csc(Co, o.expand(Vs[b].astype(np.float64), L, K), K)
            Ar = o.alpharecursion(of.alpha_hat, of.T_hat.transpose(), lhs, K)
            Br = o.betarecursion(of.T_hat, lhs, K)
            for got, ref in ((A[b * S1:(b + 1) * S1], Ar), (Bm[b * S1:(b + 1) * S1], Br)):
                assert np.array_equal(np.isneginf(got), np.isneginf(ref))
                m = np.isfinite(ref)
                assert np.allclose(got[m], ref[m], rtol=1e-5, atol=1e-4)
            m_b = mu[b * S1:(b + 1) * S1]
            if not np.isfinite(scores[b]):
                assert np.isneginf(m_b).all()
                continue
            assert (m_b <= 1e-4).all()
            assert np.allclose(m_b.max(axis=0), 0.0, atol=1e-4)  # some best path passes every frame
            for n, s in enumerate(paths[b]):  # ... and the Viterbi path is one of them
                assert abs(m_b[s, n]) <= 1e-4, (g.name, b, n, s, m_b[s, n])


@pytest.mark.gpu
def test_c_abi_collectives_one_rank(mm, torch):
    """mm_allreduce_logz / mm_allgather_ttl (include/markovmodels_amd.h) over an RCCL communicator of one rank: the
    float64 sum and the gather of the local ttl (several ranks need several GPUs: the driver's scaling run)."""
    uid = mm.dist.RcclComm.unique_id()
    comm = mm.dist.RcclComm(1, 0, uid)
    try:
        ttl = torch.tensor([-10.5, -3.25, float(np.float32(-1e-3)), -700.0], device="cuda")
        total = comm.allreduce_logz(ttl)
        allttl = comm.allgather_ttl(ttl, [4])
        torch.cuda.synchronize()
        assert total.dtype == torch.float64 and float(total) == float(ttl.double().sum())
        assert torch.equal(allttl, ttl)
    finally:
        comm.close()


@pytest.mark.gpu
def test_fsm_beyond_the_lds(mm, wl, oracle, torch):
    """An FSM whose state vectors do not fit the LDS (the reference has no size limit, src/linalg.jl:170-181): the item
    and tropical kernels keep the vectors in global memory.  pdfposteriors against the float64 oracle, Viterbi against
    the float32 one (bit exact), alpha / beta exports against each other through log Z."""
    o, oc = oracle
    g = wl.lexicon_fsm(24000, 50, seed=3)  # 16 B/state of LDS would need 384 KB
    rng = np.random.default_rng(5)
    B, N = 2, 12
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = np.array([12, 9], dtype=np.int32)
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(cf, cf)
    assert "mm_log_kernel" in bf.kernels("log")
    Vd, ld = torch.from_numpy(V).cuda(), torch.from_numpy(lens).cuda()
    gam, ttl = bf.pdfposteriors(Vd, ld)
    torch.cuda.synchronize()
    check_gamma(gam.cpu().numpy(), g_ref, lens)
    assert np.allclose(ttl.cpu().numpy(), t_ref, rtol=1e-5, atol=1e-4)
    A, Bm = bf.alpharecursion(Vd, ld).cpu().numpy(), bf.betarecursion(Vd, ld).cpu().numpy()
    S1 = g.S + 1
    for b in range(B):
        for n in (0, int(lens[b]) // 2, int(lens[b])):
            z = np.logaddexp.reduce((A[b * S1:(b + 1) * S1, n] + Bm[b * S1:(b + 1) * S1, n]).astype(np.float64))
            assert abs(z - t_ref[b]) <= 1e-3 * max(1.0, abs(t_ref[b])), (b, n, z, t_ref[b])
    ct = mm.compile(wl.to_fsm(mm, g, "tropical"), mm.statemap(g.state2pdf, g.P))
    bt = mm.batch(ct, ct)
    path, score = bt.viterbi(Vd, ld)
    torch.cuda.synchronize()
    for b in range(B):
        p_ref, s_ref, _ = oc.viterbi(graphs.to_oracle(o, g, "tropical", np.float32), g.state2pdf, g.P, V[b], int(lens[b]),
                                     dtype=np.float32)
        assert np.array_equal(path[b, : lens[b]].cpu().numpy(), p_ref[: lens[b]]) and float(score[b]) == float(np.ravel(s_ref)[0])


@pytest.mark.gpu
def test_deterministic_mode_of_the_item_kernel(mm, wl, torch):
    """mm_batch_set_deterministic: the general kernel -- the numerator path -- sums a pdf's state posteriors over a
    fixed list instead of LDS float atomics: identical bits on every run, and the committed oracle values."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "num_fsm_wsj_oracle.npz"))
    g = wl.load_npz_graph(os.path.join(os.path.dirname(__file__), "golden", "num_fsm_wsj.npz"))
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    V, lens = torch.from_numpy(z["V"]).cuda(), torch.from_numpy(z["lens"]).cuda()
    bf = _with_env({"MM_DEBUG": "1", "MM_KERNEL": "item"}, lambda: mm.batch(*([cf] * V.shape[0]))).set_deterministic(True)
    assert "mm_log_kernel" in bf.kernels("log")
    runs = [bf.pdfposteriors(V, lens) for _ in range(4)]
    torch.cuda.synchronize()
    for gam, ttl in runs[1:]:
        assert torch.equal(gam, runs[0][0]) and torch.equal(ttl, runs[0][1])
    ok = np.isfinite(z["ttl"])
    gam = runs[0][0].cpu().numpy()
    check_gamma(gam[ok], z["gamma"][ok].astype(np.float64), z["lens"][ok])
    assert (gam[~ok] == 0).all()


@pytest.mark.gpu
def test_wave_kernel_is_the_numerator_path_and_deterministic(mm, wl, oracle, torch):
    """Small deep graphs (LF-MMI numerators, examples/test_cuda.jl:78) run on the wave kernel by default: different
    graphs per utterance, different lengths, identical bits on every run (its per-pdf sums have a fixed order), the
    oracle's posteriors."""
    o, oc = oracle
    here = os.path.dirname(os.path.abspath(__file__))
    g0 = wl.load_npz_graph(os.path.join(here, "golden", "num_fsm_wsj.npz"))
    gs = [g0, wl.lexicon_fsm(300, 20, seed=2, hubs=1), g0, wl.lexicon_fsm(700, 84, seed=5, hubs=2)]
    rng = np.random.default_rng(77)
    N = 210
    lens = np.array([210, 60, 181, 140], dtype=np.int32)
    Pm = max(g.P for g in gs)
    cfs = [mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, Pm)) for g in gs]
    V = rng.standard_normal((len(gs), N, Pm)).astype(np.float32)
    bf = mm.batch(*cfs)
    assert "mm_wave_kernel" in bf.kernels("log"), bf.kernels("log")
    Vt, lt = torch.from_numpy(V).cuda(), torch.from_numpy(lens).cuda()
    runs = [bf.pdfposteriors(Vt, lt) for _ in range(4)]
    torch.cuda.synchronize()
    for gam, ttl in runs[1:]:
        assert torch.equal(gam, runs[0][0]) and torch.equal(ttl, runs[0][1])
    gam, ttl = runs[0][0].cpu().numpy(), runs[0][1].cpu().numpy()
    for b, g in enumerate(gs):
        g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, Pm, V[b : b + 1], lens[b : b + 1], dtype=np.float64)
        if np.isfinite(t_ref[0]):
            check_gamma(gam[b : b + 1], g_ref, lens[b : b + 1])
            assert np.allclose(ttl[b], t_ref[0], rtol=1e-5, atol=5e-4)
        else:
            assert (gam[b] == 0).all() and np.isneginf(ttl[b])


@pytest.mark.gpu
def test_wave_kernel_first_wherever_the_graphs_fit(mm, wl, oracle, torch):
    """The engine's choice (mm_batch_create): every graph of the batch within the wave form (up to 1023 states, 4096 arc
    slots, 250 pdfs) -> the wave kernel, for one shared graph as for different ones and for a single utterance; one shared
    DENSE graph that needs the 4-segment instance on more than two utterances per compute unit -> the pair kernels (more
    utterances than compute units on the 2-segment instance: the build of which two workgroups fit a compute unit); a
    graph beyond the form -> as before.  Batches of graphs of up to 64 states and pdfs come before all of that: the lane
    kernel.
    Results against the oracle in every case."""
    o, oc = oracle
    rng = np.random.default_rng(31)
    cases = [("shared lexicon", [wl.lexicon_fsm(300, 20, seed=2, hubs=1)] * 5, "mm_wave_kernel"),
             ("different random graphs", [wl.random_fsm(60 + 40 * b, 12, 3.0, seed=b) for b in range(5)], "mm_wave_kernel"),
             # graphs of up to 64 states: the lane kernel (one wave per utterance and direction), whatever the batch
             ("one utterance", [wl.l2r_hmm(3)], "mm_lane_kernel<8>"),
             ("dense, small batch", [wl.dense_ergodic(20, seed=3)] * 4, "mm_lane_kernel<32>"),
             ("dense, many utterances", [wl.dense_ergodic(16, seed=3)] * 520, "mm_lane_kernel<16>"),
             ("different tiny graphs", [wl.random_fsm(10 + 9 * b, 12, 3.0, seed=b) for b in range(6)], "mm_lane_kernel<64>"),
             ("sparse, many utterances, 2 segments per wave", [wl.lexicon_fsm(120, 20, seed=4, hubs=1)] * 520, "mm_wave_kernel<2,2,two per CU>"),
             ("dense beyond the lane kernel, many utterances, 4 segments per wave", [wl.random_fsm(100, 12, 20.0, seed=5)] * 520, "mm_fbp_kernel"),
             ("more arcs than the form holds", [wl.lfmmi_denominator(400, 20, seed=9)] * 4, "mm_fbp_kernel")]
    for name, gs, want in cases:
        B, N = len(gs), 23
        P = max(g.P for g in gs)
        uniq = {}
        for g in gs:
            if id(g) not in uniq:
                uniq[id(g)] = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, P))
        bf = mm.batch(*[uniq[id(g)] for g in gs])
        assert want in bf.kernels(), (name, bf.kernels())
        V = rng.standard_normal((B, N, P)).astype(np.float32)
        lens = rng.integers(1, N + 1, B).astype(np.int32)
        gam, ttl = bf.pdfposteriors(V, lens)
        for b in sorted({0, B // 2, B - 1}):
            g = gs[b]
            g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V[b : b + 1, :, : g.P], lens[b : b + 1], dtype=np.float64)
            if np.isfinite(t_ref[0]):
                check_gamma(gam[b : b + 1, :, : g.P], g_ref, lens[b : b + 1])
                assert np.allclose(ttl[b], t_ref[0], rtol=1e-5, atol=1e-4), name


@pytest.mark.parametrize("S,P", [(900, 7), (850, 120), (300, 3), (64, 200)])
def test_wave_kernel_pdf_segments(mm, wl, oracle, torch, S, P):
    """The per-pdf sums of the wave kernel are packed segments of their own (mm_engine.hip wave_pdf_table): a pdf with n
    states gets pow2(ceil(n / 4)) lanes.  900 states on 7 pdfs: groups of 64 lanes (all six butterfly levels, the last two
    across the 16-lane rows); 850 states on 120 pdfs: 307 lanes, two pdf segments per wave (the 4-segment instance); 300 states on
    3 pdfs; more pdfs than states (most pdfs empty)."""
    o, oc = oracle
    g = wl.lexicon_fsm(S, P, seed=S + P, hubs=1 if P < 4 else 2)
    rng = np.random.default_rng(S)
    B, N = 3, 33
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = np.array([33, 20, 9], dtype=np.int32)

    def run():
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        bf = mm.batch(*([cf] * B))
        gam, ttl = bf.pdfposteriors(V, lens)
        return gam, ttl, bf.kernels("log")

    gam, ttl, kernels = _with_env({"MM_DEBUG": "1", "MM_KERNEL": "wave"}, run)
    assert "mm_wave_kernel" in kernels and (S != 850 or "mm_wave_kernel<4" in kernels), kernels
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64)
    ok = np.isfinite(t_ref)
    assert ok.any()
    check_gamma(gam[ok], g_ref[ok], lens[ok])
    assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=5e-4)
    assert (gam[~ok] == 0).all() and np.isneginf(ttl[~ok]).all()


@pytest.mark.gpu
def test_wave_batch_of_many_new_graphs(mm, wl, oracle, torch):
    """An LF-MMI step brings a batch of numerator graphs nobody has seen before: mm_batch_create packs their wave forms on
    several host threads (24 distinct graphs here: 6 threads).  Every utterance against its own single-graph batch, whose
    forms were packed on the calling thread."""
    B, N = 24, 40
    gs = [wl.lexicon_fsm(120 + 17 * b, 30, seed=100 + b, hubs=1 + b % 2) for b in range(B)]
    rng = np.random.default_rng(8)
    V = rng.standard_normal((B, N, 30)).astype(np.float32)
    lens = rng.integers(N // 2, N + 1, B).astype(np.int32)
    sm = [mm.statemap(g.state2pdf, g.P) for g in gs]
    force = {"MM_DEBUG": "1", "MM_KERNEL": "wave"}  # (graphs of this size would take the row kernels)
    # (compile_many: mm_fsm_create for the graphs of a mini-batch on several host threads)
    bf = _with_env(force, lambda: mm.batch(*mm.compile_many([wl.to_fsm(mm, g) for g in gs], sm, threads=4)))
    assert "mm_wave_kernel" in bf.kernels("log"), bf.kernels("log")
    gam, ttl = bf.pdfposteriors(V, lens)
    for b in (0, 5, 11, 23):
        one = _with_env(force, lambda: mm.batch(mm.compile(wl.to_fsm(mm, gs[b]), sm[b])))
        g1, t1 = one.pdfposteriors(V[b : b + 1], lens[b : b + 1])
        assert np.array_equal(g1[0], gam[b]) and np.array_equal(t1, ttl[b : b + 1])
    o, oc = oracle
    for b in (3, 17):
        g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, gs[b]), gs[b].state2pdf, gs[b].P, V[b : b + 1], lens[b : b + 1], dtype=np.float64)
        if np.isfinite(t_ref[0]):
            check_gamma(gam[b : b + 1], g_ref, lens[b : b + 1])
            assert np.allclose(ttl[b], t_ref[0], rtol=1e-5, atol=5e-4)
