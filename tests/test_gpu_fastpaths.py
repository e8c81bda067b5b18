"""GPU tests of the fast (linear-domain) kernels ALONE and of the flag-and-redo machinery around them.

The parity tests elsewhere compare the final result of a call; a fast kernel that is wrong but marks its
utterances for the exact kernels would still pass them.  Here the exact kernels are switched off
(MM_NO_REDO under MM_DEBUG: what the fast path computed is what is compared), the kernel that ran is
asserted by name (BatchedFSM.kernels()), and the redo count is part of the contract."""
import os

import numpy as np
import pytest

import graphs
from test_gpu_parity import _with_env, check_gamma

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available()
    return torch


def oracle64(oracle, g, V, lens):
    o, oc = oracle
    return oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64, nthreads=4)


def run_fast_alone(mm, wl, g, V, lens, env=None):
    """pdfposteriors with the exact kernels switched off; returns (gamma, ttl, kernels, redo count)."""
    def run():
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        bf = mm.batch(*([cf] * V.shape[0]))
        gam, ttl = bf.pdfposteriors(V, lens)
        return gam, ttl, bf.kernels(), bf.last_redo_count()

    return _with_env(dict(env or {}, MM_DEBUG="1", MM_NO_REDO="1"), run)


def wsj_den(wl):
    return wl.load_npz_graph(os.path.join(HERE, "golden", "den_fsm_wsj.npz"))


@pytest.mark.parametrize("mode", ["auto", "write_through"])
def test_split_kernels_on_the_reference_wsj_denominator(mm, wl, oracle, torch, mode):
    """The reference's own benchmark graph (misc/benchmark/den_fsm_wsj.txt: 3032 states, 52 k arcs -- more than the
    registers of one compute unit hold) runs on the split pair kernels: teams of two workgroups that exchange their
    rows every step.  Odd batch, different lengths, an utterance of one frame; `write_through` forces the exchange
    form for teams that do not share an XCD (sc1 granules), `auto` lets the workgroups find out."""
    g = wsj_den(wl)
    rng = np.random.default_rng(17)
    B, N = 7, 70
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = np.array([70, 70, 41, 70, 1, 33, 64], dtype=np.int32)
    env = {"MM_SPLIT_SLEEP": str(8 | 0x800)} if mode == "write_through" else {}
    gam, ttl, kernels, redo = run_fast_alone(mm, wl, g, V, lens, env)
    assert "mm_fbs_kernel" in kernels and "teams of 2" in kernels
    assert redo == 0
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    ok = np.isfinite(t_ref)  # (the one-frame utterance has no accepting path: the reference's 0/0)
    assert ok.sum() == B - 1 and (gam[~ok] == 0).all() and np.isneginf(ttl[~ok]).all()
    check_gamma(gam[ok], g_ref[ok], lens[ok])
    assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("S,mode", [(3400, "auto"), (4000, "auto"), (4000, "write_through"), (4000, "apart"), (3600, "pdfs200")])
def test_split_kernels_teams_of_four(mm, wl, oracle, torch, S, mode):
    """Graphs beyond the teams of two (more than 3070 states) run on teams of FOUR workgroups per utterance pair and
    direction (up to 4094 states, 129 k arcs): every workgroup receives the rows of three others each step.
    `write_through`: the exchange form for teams that do not share an XCD; `apart`: a team that does not run together
    (the exact kernels -- here the item kernel: the graph is beyond the quad kernels too -- compute every utterance)."""
    g = wl.lfmmi_denominator(S, 200 if mode == "pdfs200" else 84, seed=S)  # (pdfs200: the instances for 129 .. 250 pdfs)
    rng = np.random.default_rng(S)
    B, N = 7, 48
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = np.array([48, 48, 31, 48, 2, 40, 17], dtype=np.int32)
    if mode == "apart":
        def run():
            cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
            bf = mm.batch(*([cf] * B))
            gam, ttl = bf.pdfposteriors(V, lens)
            return gam, ttl, bf.kernels(), bf.last_redo_count()

        gam, ttl, kernels, redo = _with_env({"MM_DEBUG": "1", "MM_SPLIT_SLEEP": str(8 | 0x200)}, run)
        assert redo == B
    else:
        env = {"MM_SPLIT_SLEEP": str(8 | 0x800)} if mode == "write_through" else {}
        gam, ttl, kernels, redo = run_fast_alone(mm, wl, g, V, lens, env)
        assert redo == 0
    assert "mm_fbs_kernel" in kernels and "teams of 4" in kernels, kernels
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    ok = np.isfinite(t_ref)
    assert ok.sum() >= B - 1
    check_gamma(gam[ok], g_ref[ok], lens[ok])
    assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=1e-4)
    assert (gam[~ok] == 0).all() and np.isneginf(ttl[~ok]).all()


def test_split_kernels_team_that_does_not_run_together(mm, wl, oracle, torch):
    """A team whose workgroups cannot see each other (a foreign kernel holding the compute units; here simulated: the
    exchange waves give up at once, MM_SPLIT_SLEEP bit 0x200) marks its utterances with 2; the finish kernel keeps such
    marks whatever the normalisers say and the exact kernels compute every utterance again: same results, B redone."""
    g = wsj_den(wl)
    rng = np.random.default_rng(18)
    B, N = 5, 24
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = np.array([24, 24, 13, 24, 20], dtype=np.int32)

    def run():
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        bf = mm.batch(*([cf] * B))
        gam, ttl = bf.pdfposteriors(V, lens)
        return gam, ttl, bf.kernels(), bf.last_redo_count()

    gam, ttl, kernels, redo = _with_env({"MM_DEBUG": "1", "MM_SPLIT_SLEEP": str(8 | 0x200)}, run)
    assert "mm_fbs_kernel" in kernels and redo == B
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-4)


def test_split_kernels_team_mate_that_never_arrives(mm, wl, oracle, torch):
    """The real thing: the workgroups of set 1 leave at once (MM_SPLIT_SLEEP bit 0x400 -- what a foreign kernel holding the compute
    units would do to them), the others WAIT for their rows.  Every poll gives up after the call's bound (10 us per frame, at least
    2 ms; it was 0.1 s flat), the utterances are marked, the float64 team kernels meet the same fate, the item kernel computes:
    the oracle's results, and the call comes back in tens of milliseconds, not in half a second."""
    import time

    g = wsj_den(wl)
    rng = np.random.default_rng(19)
    B, N = 3, 20
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = np.array([20, 14, 20], dtype=np.int32)

    def run():
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        bf = mm.batch(*([cf] * B))
        bf.pdfposteriors(V, lens)  # (workspace, code objects)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gam, ttl = bf.pdfposteriors(V, lens)
        torch.cuda.synchronize()
        return gam, ttl, bf.kernels(), bf.last_redo_count(), bf.last_fallback_count(), time.perf_counter() - t0

    gam, ttl, kernels, redo, fallback, dt = _with_env({"MM_DEBUG": "1", "MM_SPLIT_SLEEP": str(8 | 0x400), "MM_EXACT_FIRST": "0"}, run)
    assert "mm_fbs_kernel" in kernels and redo == B and fallback == B
    assert dt < 0.08, dt
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-4)


def test_split_kernels_long_utterances_keep_their_range_marks_harmless(mm, wl, oracle, torch):
    """On the WSJ graph the states of the initial contexts fall ~2 log2 per frame behind the rest: after ~55 frames they
    leave the float range of the linear path and the kernels mark the utterance.  The per-frame normalisers agree
    (nothing that matters was lost), so the marks are dropped: no utterance is computed twice, and the result is the
    oracle's."""
    g = wsj_den(wl)
    rng = np.random.default_rng(3)
    B, N = 4, 260
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = np.array([260, 260, 199, 120], dtype=np.int32)
    gam, ttl, kernels, redo = run_fast_alone(mm, wl, g, V, lens)
    assert "mm_fbs_kernel" in kernels and redo == 0
    g_ref, t_ref = oracle64(oracle, g, V[:2], lens[:2])
    check_gamma(gam[:2], g_ref, lens[:2])
    assert np.allclose(ttl[:2], t_ref, rtol=1e-5, atol=1e-3)
    # size-independent properties on all of them
    for b, L in enumerate(lens):
        assert np.allclose(gam[b, :L].sum(-1), 1.0, atol=1e-5) and (gam[b, L:] == 0).all()


def test_range_marks_that_matter_are_redone(mm, wl, torch):
    """A left-to-right graph forced onto the pair kernels (the engine would never pick them: depth rule): the values of
    one frame span far more than the float range and the states that fall out of it carry the paths that reach the end.
    The per-frame normalisers then disagree, the marks stay, and the exact kernels compute those utterances again:
    the final result is the committed float64 oracle's, and the redo count says what happened."""
    g = wl.load_npz_graph(os.path.join(HERE, "golden", "num_fsm_wsj.npz"))
    z = np.load(os.path.join(HERE, "golden", "num_fsm_wsj_oracle.npz"))
    V, lens = np.concatenate([z["V"], z["V"][:1]]), np.concatenate([z["lens"], z["lens"][:1]])

    def run():
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        bf = mm.batch(*([cf] * V.shape[0]))
        gam, ttl = bf.pdfposteriors(V, lens)
        return gam, ttl, bf.kernels(), bf.last_redo_count()

    gam, ttl, kernels, redo = _with_env({"MM_DEBUG": "1", "MM_KERNEL": "pair"}, run)
    assert "mm_fbp_kernel" in kernels
    assert redo >= 1
    ok = np.isfinite(z["ttl"])
    check_gamma(gam[:3][ok], z["gamma"][ok].astype(np.float64), lens[:3][ok])
    assert np.allclose(ttl[:3][ok], z["ttl"][ok], rtol=1e-5, atol=5e-4)
    assert np.array_equal(gam[3], gam[0]) and ttl[3] == ttl[0]  # the same utterance twice: the same bits


def test_redo_count_is_zero_on_the_benchmark_inputs_and_reported_on_peaky_ones(mm, wl, oracle, torch):
    """Config 3's graph: N(0,1) log-likelihoods never leave the fast path (what bench.py measures is the fast path);
    a sharp acoustic model (log-softmax of 10 N(0,1)) may: the count is reported, the result is
    the oracle's either way."""
    g = wl.lfmmi_denominator(2000, 84, seed=0)
    rng = np.random.default_rng(23)
    B, N = 6, 120
    lens = np.array([120, 120, 87, 120, 45, 120], dtype=np.int32)
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    assert bf.last_redo_count() == 0  # before the first call
    gam, ttl = bf.pdfposteriors(V, lens)
    assert "mm_fbp_kernel" in bf.kernels() and bf.last_redo_count() == 0
    Vp = 10.0 * V
    Vp = Vp - np.log(np.exp(Vp - Vp.max(-1, keepdims=True)).sum(-1, keepdims=True)) - Vp.max(-1, keepdims=True)
    gam, ttl = bf.pdfposteriors(Vp.astype(np.float32), lens)
    redo = bf.last_redo_count()
    assert 0 <= redo <= B
    g_ref, t_ref = oracle64(oracle, g, Vp.astype(np.float32), lens)
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-2)
    print(f"peaky emissions: {redo} of {B} utterances redone")


def test_posterior_floor_keeps_sharp_emissions_on_the_fast_kernels(mm, wl, oracle, torch):
    """mm_batch_set_posterior_floor: with sharp emissions (log-softmax of 5 N(0,1): the forward and the backward mass of a
    frame sit on different states, the overlap term L_n falls to -40 .. -60 log2) the default floor (1e-30) hands the
    utterances to the exact kernels; a caller who treats posteriors below 1e-12 as zero keeps them on the fast kernels:
    no utterance redone, every posterior within the floor of the oracle's, those above it within the usual tolerance,
    log Z as before."""
    g = wl.lfmmi_denominator(1000, 84, seed=3)
    rng = np.random.default_rng(5)
    B, N = 6, 300
    x = 5.0 * rng.standard_normal((B, N, g.P))
    V = (x - np.log(np.exp(x - x.max(-1, keepdims=True)).sum(-1, keepdims=True)) - x.max(-1, keepdims=True)).astype(np.float32)
    lens = np.array([300, 300, 211, 300, 150, 299], dtype=np.int32)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    assert "mm_fbp_kernel" in bf.kernels()
    g0, t0 = bf.pdfposteriors(V, lens)
    strict_redone = bf.last_redo_count()
    assert strict_redone > 0  # (what the default costs on such inputs)
    with pytest.raises(mm.MarkovModelsAMDError):
        bf.set_posterior_floor(1e-3)
    bf.set_posterior_floor(1e-12)
    g1, t1 = bf.pdfposteriors(V, lens)
    assert bf.last_redo_count() == 0
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    assert np.allclose(t1, t_ref, rtol=1e-5, atol=1e-4) and np.allclose(t0, t_ref, rtol=1e-5, atol=1e-4)
    check_gamma(g0, g_ref, lens)  # the default: the usual bar (everything above 1e-30)
    assert np.abs(g1 - g_ref).max() < 1e-5 and (np.abs(g1 - g_ref)[g_ref < 1e-12] < 1e-12).all()
    m = g_ref > 1e-10
    assert (np.abs(np.log(np.maximum(g1[m], 1e-300)) - np.log(g_ref[m])) / np.maximum(np.abs(np.log(g_ref[m])), 1.0)).max() < 1e-4
    bf.set_posterior_floor(1e-30)
    bf.pdfposteriors(V, lens)
    assert bf.last_redo_count() == strict_redone


@pytest.mark.parametrize("S,P", [(6000, 300), (5000, 100), (2900, 120), (1000, 640), (6100, 60)])
def test_graph_beyond_the_fast_paths(mm, wl, oracle, torch, S, P):
    """The reference's products have no size limit (src/linalg.jl:170-181).  1000 states x 640 pdfs and 6100 states are beyond
    the pair kernels (2047 states, 506 pdfs) and the teams (6014 states; 314 pdfs for teams of 8): they run on the stream kernels
    (mm_stream.hip; the quad / item kernels until round 5); a 2900-state graph of config 3's family takes the teams of two, 5000 and 6000 states (the latter with 300 pdfs:
    mm_fbs_kernel<5, ., 8>) teams of EIGHT workgroups per utterance pair and direction.  Same results either way."""
    g = wl.lfmmi_denominator(S, P, seed=S)
    rng = np.random.default_rng(S + P)
    B, N = 5, 36
    V = (1.3 * rng.standard_normal((B, N, g.P))).astype(np.float32)
    lens = np.array([36, 36, 20, 35, 7], dtype=np.int32)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    gam, ttl = bf.pdfposteriors(V, lens)
    kernels = bf.kernels()
    assert ("mm_fbs_kernel" in kernels) == (S in (2900, 5000, 6000)), kernels
    assert ("mm_stream_kernel" in kernels) == (S in (1000, 6100)), kernels  # (beyond every register-resident form: the stream kernels)
    if S in (5000, 6000):
        assert ("mm_fbs_kernel<5,A,8>" if P == 300 else "mm_fbs_kernel<2,A,8>") in kernels, kernels
    assert bf.last_redo_count() == 0 or "mm_fbs_kernel" not in kernels
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("which", ["pair", "split"])
def test_one_utterance_on_the_pair_kernels(mm, wl, oracle, torch, which):
    """A batch of ONE utterance runs on the pair / split pair kernels too (the pair computes the utterance twice, the copy
    writes to a workspace slot only): both directions at once instead of two passes of the row kernels (config 3's graph:
    4.1 -> 2.6 ms; the WSJ denominator: 6.5 -> 1.7 ms).  Full length, a shorter length, one frame, no frame."""
    g = wl.lfmmi_denominator(1100, 60, seed=2) if which == "pair" else wsj_den(wl)
    rng = np.random.default_rng(4)
    N = 57
    V = (1.3 * rng.standard_normal((1, N, g.P))).astype(np.float32)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(cf)
    assert ("mm_fbp_kernel" if which == "pair" else "mm_fbs_kernel") in bf.kernels(), bf.kernels()
    for L in (N, 31, 1, 0):
        lens = np.array([L], dtype=np.int32)
        gam, ttl = bf.pdfposteriors(V, lens)
        g_ref, t_ref = oracle64(oracle, g, V, lens)
        if L == 0 or not np.isfinite(t_ref[0]):  # (no frame, or no accepting path of that length: gamma = 0, ttl = zero(K))
            assert (gam == 0).all() and np.isneginf(ttl).all()
            continue
        assert bf.last_redo_count() == 0
        check_gamma(gam, g_ref, lens)
        assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("P", [130, 249])
def test_pair_kernels_with_many_pdfs(mm, wl, oracle, torch, P):
    """P + 1 in 129..250: the service wave of the pair kernels runs four 64-lane passes over the pdfs
    (mm_fbp_kernel<4, ...>), an instance no other test reaches."""
    g = wl.lfmmi_denominator(1200, P - (P % 2), seed=4)
    rng = np.random.default_rng(P)
    B, N = 5, 40
    V = (1.3 * rng.standard_normal((B, N, g.P))).astype(np.float32)
    lens = np.array([40, 40, 23, 40, 9], dtype=np.int32)
    gam, ttl, kernels, redo = run_fast_alone(mm, wl, g, V, lens)
    assert "mm_fbp_kernel<4" in kernels and redo == 0
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("S,P", [(1200, 300), (2000, 400), (1500, 505)])
def test_pair_kernels_with_251_to_506_pdfs(mm, wl, oracle, torch, S, P):
    """P + 1 in 251..506: eight 64-lane passes over the pdfs (mm_fbp_kernel<8, ...>), per-pdf LDS arrays of twice the size and
    a partner-row ring of TWO vectors (the row of a step is requested at the top of the step before, PairLay).  These graphs
    ran on the quad kernels until round 4 (2000 states, 400 pdfs, B = 256, T = 1500: 8.8 ms).  The float32 kernels alone
    against the float64 oracle; odd batch, lengths down to one frame."""
    g = wl.lfmmi_denominator(S, P - (P % 2), seed=P)
    rng = np.random.default_rng(P)
    B, N = 5, 61
    V = (1.3 * rng.standard_normal((B, N, g.P))).astype(np.float32)
    lens = np.array([61, 61, 23, 60, 1], dtype=np.int32)
    gam, ttl, kernels, redo = run_fast_alone(mm, wl, g, V, lens)
    assert "mm_fbp_kernel<8" in kernels and "mm_fbd_kernel<8" in kernels and redo == 0, kernels
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    ok = np.isfinite(t_ref)
    check_gamma(gam[ok], g_ref[ok], lens[ok])
    assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("S,deg", [(300, 2.0), (900, 12.0), (1500, 16.0), (1900, 16.5)])
def test_row_kernel_register_windows(mm, wl, oracle, torch, S, deg):
    """Per-utterance graphs of different sizes land on the different register windows of the row kernels
    (mm_fbr_kernel<24 | 40 | 42 | 44, ...>); the fast path alone against the oracle."""
    o, oc = oracle
    gs = [wl.random_fsm(S, 30, deg, seed=s, p_final=0.2) for s in (1, 2)]
    rng = np.random.default_rng(S)
    N = 30
    V = rng.standard_normal((2, N, 30)).astype(np.float32)
    lens = np.array([30, 17], dtype=np.int32)

    def run():
        cfs = [mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P)) for g in gs]
        bf = mm.batch(*cfs)
        gam, ttl = bf.pdfposteriors(V, lens)
        return gam, ttl, bf.kernels(), bf.last_redo_count()

    # (forced: the graphs of 300 states fit the wave kernel, which the engine prefers)
    gam, ttl, kernels, redo = _with_env({"MM_DEBUG": "1", "MM_NO_REDO": "1", "MM_KERNEL": "row"}, run)
    assert "mm_fbr_kernel" in kernels and redo == 0, kernels
    for b, g in enumerate(gs):
        g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V[b : b + 1], lens[b : b + 1], dtype=np.float64)
        check_gamma(gam[b : b + 1], g_ref, lens[b : b + 1])
        assert np.allclose(ttl[b], t_ref[0], rtol=1e-5, atol=1e-4)


def test_odd_batch_beyond_the_compute_units_on_the_pair_kernels(mm, wl, oracle, torch):
    """B = 515 utterances (odd, > 2 x 256 CUs) of different lengths on the pair kernels: pairing by length, the
    unpaired last utterance, several workgroups per compute unit in turn."""
    g = wl.lfmmi_denominator(400, 20, seed=9)
    rng = np.random.default_rng(8)
    B, N = 515, 14
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = rng.integers(0, N + 1, size=B).astype(np.int32)
    gam, ttl, kernels, redo = run_fast_alone(mm, wl, g, V, lens)
    # (a handful of the short utterances have a frame whose forward and backward mass overlap below 2^-20: marked since round 5 --
    # a product of the combine may have been flushed there --, whatever the exact kernels would find; here they are switched off)
    assert "mm_fbp_kernel" in kernels and redo <= B // 32
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    ok = np.isfinite(t_ref)
    check_gamma(gam[ok], g_ref[ok], lens[ok])
    assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=1e-4)
    assert (gam[~ok] == 0).all() and np.isneginf(ttl[~ok]).all()


@pytest.mark.parametrize("kernel", ["pair", "auto"])
def test_batch_beyond_the_ordered_limit(mm, wl, oracle, torch, kernel):
    """B = 8200 > 8192: the longest-first order is skipped (mm_engine.hip), the utterances run in batch order -- on the pair
    kernels (forced) and on the engine's own choice for a graph this small, the lane kernel."""
    g = wl.random_fsm(24, 4, 2.5, seed=3)
    rng = np.random.default_rng(2)
    B, N = 8200, 6
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = rng.integers(1, N + 1, size=B).astype(np.int32)

    def run():
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        bf = mm.batch(*([cf] * B))
        return bf.pdfposteriors(V, lens) + (bf.kernels(),)

    gam, ttl, kernels = _with_env({"MM_DEBUG": "1", "MM_KERNEL": kernel}, run)
    assert ("mm_fbp_kernel" if kernel == "pair" else "mm_lane_kernel") in kernels, kernels
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    ok = np.isfinite(t_ref)
    check_gamma(gam[ok], g_ref[ok], lens[ok])
    assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=1e-4)


def test_fuzz_pairs_one_seed(mm, wl, oracle, torch):
    """One seed of tools/fuzz_pairs.py: the pair and the row kernels against the item kernel (an independent
    implementation) on 7 graphs x batch sizes x frame counts x length patterns, the degenerate ones included."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("fuzz_pairs", os.path.join(os.path.dirname(HERE), "tools", "fuzz_pairs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(1) == 0


def test_fuzz_round3_one_seed(mm, wl, oracle, torch):
    """One seed of tools/fuzz_round3.py: the split pair kernels, the wave kernel (twice: identical bits) and Viterbi on the
    row-lane form against the item kernel, on random graphs, batch sizes (more teams than compute units), frame counts
    and length patterns."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("fuzz_round3", os.path.join(os.path.dirname(HERE), "tools", "fuzz_round3.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(2) == 0


def test_engine_under_a_live_rccl_process_group(mm, wl, oracle, torch):
    """The HIP engine inside a torch.distributed process ("nccl" = RCCL, one rank: what every rank of bench.py --gpus N
    is): with an RCCL communicator alive HIP maps streams to hardware queues differently, and the two agents of the
    pair kernels -- two kernels on two library streams until round 3 -- once shared a queue and took turns (5.7 instead of
    3.2 ms per call).  They are workgroups of ONE grid now; the step (pdfposteriors + the logZ all-reduce) must give the
    oracle's numbers.  Runs in a child process: the process group must not leak into the other tests."""
    import subprocess
    import sys

    root = os.path.dirname(HERE)
    code = r"""
import os, sys, importlib
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
import __graft_entry__ as ge, graphs
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(4, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
mm = ge.load_package(); o, oc = ge.load_oracle()
wl = importlib.import_module(mm.__name__ + ".workloads")
g = wl.lfmmi_denominator(600, 40, seed=5)
rng = np.random.default_rng(2)
B, N = 6, 50
V = rng.standard_normal((B, N, g.P)).astype(np.float32)
lens = np.array([50, 50, 31, 50, 12, 44], dtype=np.int32)
cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
bf = mm.batch(*([cf] * B))
k = bf.kernels()
assert "mm_fbp_kernel" in k and "in one grid" in k, k
gam, ttl = bf.pdfposteriors(torch.from_numpy(V).cuda(), torch.from_numpy(lens).cuda())
total = mm.dist.allreduce_logz(ttl)
torch.cuda.synchronize()
g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, lens, dtype=np.float64)
assert np.abs(gam.cpu().numpy() - g_ref).max() < 2e-5
assert np.allclose(ttl.cpu().numpy(), t_ref, rtol=1e-5, atol=1e-4)
assert abs(float(total) - float(t_ref.sum())) < 1e-3
dist.destroy_process_group()
print("OK-DIST")
""".replace("ROOT", repr(root))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK-DIST" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_two_batches_alive_on_two_streams(mm, wl, oracle, torch):
    """Two batches of the pair kernels alive at the same time, driven from two streams of the caller (each call is a chain
    of launches on its caller's stream, nothing shared but the FSM): same results as one after the other."""
    g = wl.lfmmi_denominator(600, 40, seed=5)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    B, N = 6, 80
    V = [torch.randn(B, N, g.P, device="cuda") for _ in range(2)]
    bfs = [mm.batch(*([cf] * B)) for _ in range(2)]
    assert all("mm_fbp_kernel" in bf.kernels() for bf in bfs)
    ref = [bf.pdfposteriors(v) for bf, v in zip(bfs, V)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    out = []
    for _ in range(3):
        out = []
        for bf, v, s in zip(bfs, V, streams):
            with torch.cuda.stream(s):
                out.append(bf.pdfposteriors(v))
    torch.cuda.synchronize()
    for (g0, t0), (g1, t1) in zip(ref, out):
        assert torch.equal(g0, g1) and torch.equal(t0, t1)
    del bfs
    bf3 = mm.batch(*([cf] * B))
    g3, t3 = bf3.pdfposteriors(V[0])
    assert torch.equal(g3, ref[0][0])


@pytest.mark.parametrize("case", ["wsj_numerators", "config3_x64"])
def test_reference_shaped_entry_at_engine_speed(mm, wl, oracle, torch, case):
    """`pdfposteriors(rawunion(fsms...), V_hats, C_hats)` (src/inference.jl:145, the call of examples/test_cuda.jl:128) on DEVICE
    tensors: the second call -- the graphs found in the compiled-graph cache, nothing crossing to the host -- within 1.3x of
    BatchedFSM.pdfposteriors on the same batch, with its results; 128 WSJ numerator graphs (one per utterance, each a separate
    FSM object with the same content here) and config 3's graph x 64."""
    import time

    if case == "wsj_numerators":
        g0 = wl.load_npz_graph(os.path.join(HERE, "golden", "num_fsm_wsj.npz"))
        B, N = 128, 700
        fsms = [wl.to_fsm(mm, g0) for _ in range(B)]
    else:
        g0 = wl.lfmmi_denominator(2000, 84, seed=0)
        B, N = 64, 1500
        f0 = wl.to_fsm(mm, g0)
        fsms = [f0] * B
    Cs = [mm.statemap(g0.state2pdf, g0.P)] * B
    rng = np.random.default_rng(0)
    lens = rng.integers(N // 2, N + 1, B).astype(np.int32)
    V = torch.randn(B, N, g0.P, device="cuda")
    ninf = torch.full((), -float("inf"), device="cuda")
    Vhat = torch.full((B, g0.P + 1, N + 1), -float("inf"), device="cuda")  # expand() (src/inference.jl:54-60) on the device
    t = torch.arange(N + 1, device="cuda")[None, :] < torch.from_numpy(lens).cuda()[:, None]
    Vhat[:, : g0.P, :N] = torch.where(t[:, None, :N], V.transpose(1, 2), ninf)
    Vhat[:, g0.P, :] = torch.where(t, ninf, torch.zeros((), device="cuda"))
    Vhats = [Vhat[b] for b in range(B)]
    u = mm.rawunion(*fsms)
    mm.compiled_cache_clear()
    mm.compiled_cache_stats(reset=True)
    g1, t1 = mm.pdfposteriors(u, Vhats, Cs)  # first call: compiles (one graph by content, found B - 1 times)
    st = mm.compiled_cache_stats(reset=True)
    assert st["misses"] == 1 and st["hits"] == B - 1
    assert isinstance(g1, torch.Tensor) and g1.is_cuda and tuple(g1.shape) == (B, g0.P, N)
    bf = mm.batch(*mm.compile_many(fsms, Cs)) if case == "wsj_numerators" else mm.batch(*([mm.compile(fsms[0], Cs[0])] * B))
    ld = torch.from_numpy(lens).cuda()
    g_ref, t_ref = bf.pdfposteriors(V, ld)
    assert torch.allclose(g1.transpose(1, 2), g_ref, rtol=1e-4, atol=1e-6) and torch.allclose(t1, t_ref, rtol=1e-5, atol=1e-4)

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    t_engine = timed(lambda: bf.pdfposteriors(V, ld))
    t_ref_shape = timed(lambda: mm.pdfposteriors(u, Vhats, Cs, seqlengths=ld))  # lengths given: nothing is read back
    t_checked = timed(lambda: mm.pdfposteriors(u, Vhats, Cs))                    # lengths from the phony row (a host read)
    st = mm.compiled_cache_stats()
    assert st["misses"] == 0 and st["memo_hits"] >= 20
    print(f"{case}: engine {1e3 * t_engine:.3f} ms, reference-shaped {1e3 * t_ref_shape:.3f} ms ({t_ref_shape / t_engine:.2f}x), "
          f"with the lengths read from the phony row {1e3 * t_checked:.3f} ms")
    assert t_ref_shape <= 1.3 * t_engine + 5e-5, (t_ref_shape, t_engine)
    # a fresh rawunion of the same graphs (what a training step builds): the graphs are found by content, no compile
    u2 = mm.rawunion(*fsms)
    g2, t2 = mm.pdfposteriors(u2, Vhats, Cs, seqlengths=ld)
    assert mm.compiled_cache_stats()["misses"] == 0 and torch.equal(t2, mm.pdfposteriors(u, Vhats, Cs, seqlengths=ld)[1])


@pytest.mark.parametrize("case", ["forced_small", "different_graphs", "wsj_den", "sharp", "big"])
def test_stream_kernels(mm, wl, oracle, torch, case):
    """The stream kernels (mm_stream.hip: the arcs streamed from L2 as 8-byte records, the vector in LDS as wide-exponent 32-bit
    values, float64 accumulation, one utterance per workgroup, forward launch then backward launch) against the float64 oracle:
    forced onto graphs the faster kernels would take (different graphs in one batch; the reference's WSJ denominator with its
    whole-wave final row; sharp emissions at -300 nats: no float32 anywhere on the path), and a 9000-state / 1000-pdf graph of
    config 3's family -- beyond every register-resident form -- against the item kernel, which ran such graphs until round 5."""
    rng = np.random.default_rng(3)
    env = {"MM_KERNEL": "stream"}
    if case == "forced_small":
        gs = [wl.random_fsm(300, 11, 4.0, seed=9)] * 4
        N, lens = 40, np.array([40, 1, 17, 0], dtype=np.int32)
    elif case == "different_graphs":
        gs = [wl.random_fsm(200 + 37 * i, 7, 3.0 + 0.3 * i, seed=i) for i in range(5)]
        N, lens = 33, np.array([33, 33, 12, 30, 2], dtype=np.int32)
    elif case == "wsj_den":
        gs = [wsj_den(wl)] * 3
        N, lens = 120, np.array([120, 77, 119], dtype=np.int32)
    elif case == "sharp":
        gs = [wl.lfmmi_denominator(1500, 84, seed=2)] * 5
        N, lens = 90, np.array([90, 41, 90, 3, 88], dtype=np.int32)
    else:
        gs = [wl.lfmmi_denominator(9000, 1000, seed=5)] * 3
        N, lens = 30, np.array([30, 19, 30], dtype=np.int32)
        env = {}
    B, P = len(gs), gs[0].P
    V = (1.3 * rng.standard_normal((B, N, P))).astype(np.float32)
    if case == "sharp":
        x = 10.0 * rng.standard_normal((B, N, P))
        V = (x - np.log(np.exp(x - x.max(-1, keepdims=True)).sum(-1, keepdims=True)) - x.max(-1, keepdims=True) - 300.0).astype(np.float32)
    cfs = {}
    for g in gs:
        if id(g) not in cfs:
            cfs[id(g)] = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = _with_env(dict(env, MM_DEBUG="1"), lambda: mm.batch(*[cfs[id(g)] for g in gs]))
    assert "mm_stream_kernel" in bf.kernels(), bf.kernels()
    gam, ttl = bf.pdfposteriors(V, lens)
    assert bf.last_redo_count() == 0
    if case == "big":  # (the item kernel is the independent implementation for the whole batch; the float64 oracle for one utterance)
        bi = _with_env({"MM_DEBUG": "1", "MM_KERNEL": "item"}, lambda: mm.batch(*[cfs[id(g)] for g in gs]))
        assert "mm_stream_kernel" not in bi.kernels()
        g_ref, t_ref = bi.pdfposteriors(V, lens)
        g_ref = g_ref.astype(np.float64)
        o, oc = oracle
        go, to = oc.batch_shared(graphs.to_oracle(o, gs[1]), gs[1].state2pdf, P, V[1:2], lens[1:2], dtype=np.float64, nthreads=4)
        check_gamma(gam[1:2], go, lens[1:2])
        assert np.allclose(ttl[1:2], to, rtol=1e-5, atol=1e-3)
    else:
        o, oc = oracle
        g_ref = np.zeros((B, N, P))
        t_ref = np.zeros(B)
        for b, g in enumerate(gs):
            gr, tr = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V[b : b + 1], lens[b : b + 1], dtype=np.float64)
            g_ref[b], t_ref[b] = gr[0], tr[0]
    ok = np.isfinite(t_ref)
    check_gamma(gam[ok], g_ref[ok], lens[ok])
    assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=1e-3)
    assert (gam[~ok] == 0).all() and np.isneginf(ttl[~ok]).all()
    g2, t2 = bf.pdfposteriors(V, lens)  # (every sum of the combine in a fixed order, float64: the same bits on every run)
    assert np.array_equal(g2, gam) and np.array_equal(t2, ttl)
    # teams (round 6): batches this small run as teams of 4 workgroups per utterance and direction; the lone workgroup and the teams of
    # 2 give the same posteriors (other sums: not the same bits)
    assert "teams of 4" in bf.kernels(), bf.kernels()
    if case != "big":
        for H in (1, 2):
            bh = _with_env(dict(env, MM_DEBUG="1", MM_STREAM_H=str(H)), lambda: mm.batch(*[cfs[id(g)] for g in gs]))
            assert ("teams of 2" in bh.kernels()) == (H == 2) and "teams of 4" not in bh.kernels()
            gh, th = bh.pdfposteriors(V, lens)
            assert bh.last_redo_count() == 0
            check_gamma(gh[ok], g_ref[ok], lens[ok])
            assert np.allclose(th[ok], t_ref[ok], rtol=1e-5, atol=1e-3) and np.allclose(gh, gam, rtol=2e-5, atol=1e-7)


def test_team_xcd_counter(mm, wl, torch):
    """mm_batch_team_xcd_stats: every workgroup of the team kernels' phase-A launch reports whether its whole team sits on one XCD
    (the plain-store exchange) -- counted since the last read, cleared by the read; no teams, no counts."""
    g = wsj_den(wl)
    B, N = 6, 40
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    V = torch.randn(B, N, g.P, device="cuda")
    assert bf.team_xcd_stats() == (0, 0)
    for _ in range(3):
        bf.pdfposteriors(V, None)
    assert "mm_fbs_kernel" in bf.kernels()
    same, total = bf.team_xcd_stats()
    assert total == 3 * 2 * 2 * ((B + 1) // 2) and 0 <= same <= total  # calls x directions x workgroups of a team x pairs
    assert bf.team_xcd_stats() == (0, 0)
    g3 = wl.lfmmi_denominator(2000, 84, seed=0)
    b3 = mm.batch(*([mm.compile(wl.to_fsm(mm, g3), mm.statemap(g3.state2pdf, g3.P))] * 4))
    b3.pdfposteriors(torch.randn(4, 30, g3.P, device="cuda"), None)
    assert b3.team_xcd_stats() == (0, 0)


def test_mark_policy_keep_keeps_every_range_mark(mm, wl, oracle, torch):
    """mm_batch_set_mark_policy(MM_MARKS_KEEP): a mark raised by a range check stays whatever the two criteria say -- the reference's
    WSJ denominator, whose initial-context states decay out of the float range in every utterance (marked and cleared by default:
    redo 0), is then computed by the exact kernels; the results are the float64 oracle's either way.  Per batch, switchable between
    calls, no environment variable involved."""
    g = wsj_den(wl)
    rng = np.random.default_rng(29)
    B, N = 6, 90
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    lens = np.array([90, 90, 71, 90, 84, 90], dtype=np.int32)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    bf.set_exact_policy("f32_first")
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    os.environ["MM_NEVER_CLEAR"] = "1"  # (round 5's switch: no longer read)
    try:
        for policy, want in (("decide", 0), ("keep", B), ("decide", 0)):
            bf.set_mark_policy(policy)
            gam, ttl = bf.pdfposteriors(V, lens)
            assert bf.last_redo_count() == want and bf.last_fallback_count() == 0
            check_gamma(gam, g_ref, lens)
            assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-4)
    finally:
        os.environ.pop("MM_NEVER_CLEAR")
    from importlib import import_module

    _lib = import_module(mm.__name__ + "._lib")
    assert _lib.lib.mm_batch_set_mark_policy(bf._h, 7) == -1  # MM_ERR_INVALID


def test_exact_policy_pins_the_export_path(mm, wl, torch):
    """A graph whose forward vectors leave float32's range in most utterances (here: 3 of 5 marked by the linear-domain export): under the
    default policy the SECOND alpha export starts on the item kernel -- the same numbers within the bar, other last bits --, under
    MM_EXACT_F32_FIRST every call launches the same kernels and gives the same bits, under MM_EXACT_F64_FIRST the item kernel alone
    (bit-identical to a batch forced onto it)."""
    g = wl.lfmmi_denominator(900, 40, seed=31)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    B, N = 5, 33
    lens = torch.tensor([N, N - 4, 9, 1, N], dtype=torch.int32, device="cuda")
    V = torch.randn(B, N, g.P, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    item = _with_env({"MM_DEBUG": "1", "MM_KERNEL": "item"}, lambda: mm.batch(*([cf] * B))).alpharecursion(V, lens).clone()

    def runs(policy, n=3):
        bf = mm.batch(*([cf] * B))
        assert "mm_fbx_kernel" in bf.kernels("export")
        if policy:
            bf.set_exact_policy(policy)
        out = []
        for _ in range(n):
            out.append(bf.alpharecursion(V, lens).clone())
            out.append(bf.last_redo_count())
        return out

    a0, r0, a1, r1, a2, r2 = runs(None)
    assert 2 * r0 > B  # (most utterances handed over: the premise of this test)
    assert torch.equal(a1, item) and torch.equal(a2, item) and not torch.equal(a0, item)  # the second call starts on the item kernel
    fin = torch.isfinite(item)
    assert torch.equal(torch.isfinite(a0), fin) and float((a0[fin] - item[fin]).abs().max()) <= 2e-4
    f0, q0, f1, q1, f2, q2 = runs("f32_first")
    assert torch.equal(f0, a0) and torch.equal(f1, f0) and torch.equal(f2, f0) and q0 == q1 == q2 == r0
    d0, _, d1, _, _, _ = runs("f64_first")
    assert torch.equal(d0, item) and torch.equal(d1, item)


def test_alpha_beta_export_on_the_team_kernels(mm, wl, oracle, torch):
    """The same export by TEAMS of workgroups (mm_fbsx_kernel: graphs beyond one compute unit) against the float64 oracle: a 3600-state
    graph of config 3's family (teams of 4: nothing leaves float32's range, the item kernel computes nothing) and the reference's WSJ
    denominator (teams of 2), whose initial-context states decay 2 log2 per frame against the rest -- beyond float32 after ~60
    frames: marked, computed by the item kernel, the reference's finite alpha where the linear domain holds zeros."""
    o, oc = oracle
    rng = np.random.default_rng(18)
    for g, N, want_redo in ((wl.lfmmi_denominator(3600, 84, seed=1), 21, False), (wsj_den(wl), 90, True)):
        of = graphs.to_oracle(o, g)
        S1 = g.S + 1
        B = 3
        lens = np.array([N, N - 5, 2], dtype=np.int32)
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        bf = mm.batch(*([cf] * B))
        assert "mm_fbs_kernel" in bf.kernels() and "mm_fbsx_kernel" in bf.kernels("export"), (bf.kernels(), bf.kernels("export"))
        V = rng.standard_normal((B, N, g.P)).astype(np.float32)
        A = bf.alpharecursion(V, lens)
        redo_a = bf.last_redo_count()
        Bm = bf.betarecursion(V, lens)
        redo_b = bf.last_redo_count()
        assert (redo_a + redo_b > 0) == want_redo, (g.name, redo_a, redo_b)
        for b in range(B):
            _, _, Ar, Br = oc.single(of, g.state2pdf, g.P, o.expand(V[b].T.astype(np.float64), int(lens[b]), o.LOG), want_ab=True)
            for got, ref, what in ((A[b * S1:(b + 1) * S1], Ar, "alpha"), (Bm[b * S1:(b + 1) * S1], Br, "beta")):
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)), (g.name, what, b)
                m = np.isfinite(ref)
                assert np.allclose(got[m], ref[m], rtol=1e-5, atol=3e-4), (g.name, what, b, np.abs(got[m] - ref[m]).max())


@pytest.mark.parametrize("P", [60, 200])
def test_alpha_beta_export_on_the_pair_kernels(mm, wl, oracle, torch, P):
    """alpha-recursion / beta-recursion (src/inference.jl:62-74, 99-110) of a shared graph in the pair form: phase A of the pair kernels
    over all N + 1 frames (mm_fbx_kernel: two utterances per workgroup, linear domain; the backward agent stores the sums BEFORE the
    frame's emission, as the reference's beta has them) + the layout pass -- asserted by name -- against the float64 oracle's state_A /
    state_B: odd batch, lengths 0 .. N (expand()'s padding beyond them: the final state keeps its value, beta of the frames beyond the
    length is omega).  Sharp emissions leave float32's range: marked, computed by the item kernel, the same results."""
    o, oc = oracle
    g = wl.lfmmi_denominator(1200, P, seed=6)
    of = graphs.to_oracle(o, g)
    S1 = g.S + 1
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    B, N = 5, 37
    lens = np.array([37, 20, 0, 1, 36], dtype=np.int32)
    rng = np.random.default_rng(8)
    bf = mm.batch(*([cf] * B))
    assert "mm_fbx_kernel" in bf.kernels("export") and "mm_pair_export_kernel" in bf.kernels("export"), bf.kernels("export")
    for sharp in (False, True):
        V = (1.2 * rng.standard_normal((B, N, g.P))).astype(np.float32)
        if sharp:  # log-softmax of 12 N(0,1)
            x = 12.0 * rng.standard_normal((B, N, g.P))
            V = (x - np.log(np.exp(x - x.max(-1, keepdims=True)).sum(-1, keepdims=True)) - x.max(-1, keepdims=True)).astype(np.float32)
        A = bf.alpharecursion(V, lens)
        redo_a = bf.last_redo_count()
        Bm = bf.betarecursion(V, lens)
        redo_b = bf.last_redo_count()
        assert A.shape == (B * S1, N + 1) and Bm.shape == (B * S1, N + 1)
        assert (redo_a, redo_b) == (0, 0) if not sharp else redo_a + redo_b > 0  # (sharp: utterances handed to the item kernel)
        for b in range(B):
            _, _, Ar, Br = oc.single(of, g.state2pdf, g.P, o.expand(V[b].T.astype(np.float64), int(lens[b]), o.LOG), want_ab=True)
            for got, ref, what in ((A[b * S1:(b + 1) * S1], Ar, "alpha"), (Bm[b * S1:(b + 1) * S1], Br, "beta")):
                assert np.array_equal(np.isneginf(got), np.isneginf(ref)), (what, b, sharp)
                m = np.isfinite(ref)
                assert np.allclose(got[m], ref[m], rtol=1e-5, atol=2e-4), (what, b, sharp, np.abs(got[m] - ref[m]).max())
    # a graph with states that cannot reach the final state: their alpha is not zero(K) -- the item kernel's business
    g2 = wl.random_fsm(300, 11, 2.0, seed=12)
    b2 = _with_env({"MM_DEBUG": "1", "MM_KERNEL": "pair"}, lambda: mm.batch(*([mm.compile(wl.to_fsm(mm, g2), mm.statemap(g2.state2pdf, g2.P))] * 2)))
    info = b2.kernels("export")
    V2 = rng.standard_normal((2, 9, g2.P)).astype(np.float32)
    A2 = b2.alpharecursion(V2, None)
    for b in range(2):
        _, _, Ar, _ = oc.single(graphs.to_oracle(o, g2), g2.state2pdf, g2.P, o.expand(V2[b].T.astype(np.float64), 9, o.LOG), want_ab=True)
        got = A2[b * (g2.S + 1):(b + 1) * (g2.S + 1)]
        assert np.array_equal(np.isneginf(got), np.isneginf(Ar)), info
        assert np.allclose(got[np.isfinite(Ar)], Ar[np.isfinite(Ar)], rtol=1e-5, atol=2e-4)
