"""GPU tests of the float64 exact pair kernels (mm_kernel_dpair.hip: mm_fbd_kernel) -- where the utterances go that the
float32 pair kernels mark (sharp emissions: a trained acoustic model), and what runs FIRST while the inputs stay sharp.

src/inference.jl:145-161 runs one algorithm, in the log domain, for every input; the engine's linear-domain kernels must
give its result whatever the data.  Here: the kernels are asserted by name, the log-domain kernels behind them are
switched off (MM_NO_FALLBACK: what the float64 kernels computed is what is compared), and the counts (mm_batch_last_redo_count,
mm_batch_last_fallback_count, mm_batch_last_exact_first) are part of the contract."""
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


def peaky(rng, shape, sigma):
    """log-softmax of sigma N(0,1): the outputs of a sharp acoustic model"""
    x = sigma * rng.standard_normal(shape)
    return (x - np.log(np.exp(x - x.max(-1, keepdims=True)).sum(-1, keepdims=True)) - x.max(-1, keepdims=True)).astype(np.float32)


def make_batch(mm, wl, g, B, env):
    return _with_env(dict(env, MM_DEBUG="1"), lambda: mm.batch(*([mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))] * B)))


@pytest.mark.parametrize("sigma", [10.0, 25.0])
def test_sharp_emissions_on_the_float64_kernels_alone(mm, wl, oracle, torch, sigma):
    """Config 3's graph, sharp emissions: every utterance is marked by the float32 kernels and computed by the float64
    kernels -- with nothing behind them (MM_NO_FALLBACK), against the float64 oracle.  Odd batch, different lengths (0 and 1
    included), twice: the second call skips the float32 kernels (the first one's utterances were hard) and must give the
    same bits."""
    g = wl.lfmmi_denominator(2000, 84, seed=0)
    rng = np.random.default_rng(int(sigma))
    B, N = 7, 130
    lens = np.array([130, 130, 87, 1, 45, 0, 129], dtype=np.int32)
    V = peaky(rng, (B, N, g.P), sigma)
    bf = make_batch(mm, wl, g, B, {"MM_NO_FALLBACK": "1"})
    assert "mm_fbd_kernel" in bf.kernels(), bf.kernels()
    g1, t1 = bf.pdfposteriors(V, lens)
    assert not bf.last_exact_first()
    redone = bf.last_redo_count()
    assert redone >= 4 and bf.last_fallback_count() == 0
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    ok = np.isfinite(t_ref)
    check_gamma(g1[ok], g_ref[ok], lens[ok])
    assert np.allclose(t1[ok], t_ref[ok], rtol=1e-5, atol=1e-3)
    assert (g1[~ok] == 0).all() and np.isneginf(t1[~ok]).all()
    g2, t2 = bf.pdfposteriors(V, lens)
    # (the whole batch on the wide pair kernels, two utterances per workgroup: an utterance WITHOUT any path whose partner raised
    # a range mark stays marked -- Z = 0 is what a total underflow would look like -- and would go to the log-domain kernels)
    assert bf.last_exact_first() and bf.last_redo_count() == B and bf.last_fallback_count() <= int((~ok & (lens >= 1)).sum())
    assert "mm_fbw_kernel" in bf.kernels()
    check_gamma(g2[ok], g_ref[ok], lens[ok])
    assert np.allclose(t2[ok], t_ref[ok], rtol=1e-5, atol=1e-3)
    m = lens >= 2  # (the utterances both calls computed on the float64 kernels: the same bits)
    for b in np.nonzero(m)[0]:
        if redone == B or np.array_equal(g1[b], g2[b]):
            continue
        # an utterance the float32 kernels kept in the first call: equal within the bar only
        check_gamma(g1[b : b + 1], g2[b : b + 1].astype(np.float64), lens[b : b + 1])


def test_the_choice_follows_the_data(mm, wl, oracle, torch):
    """randn -> sharp -> sharp -> randn -> randn: the float32 kernels run first until a call leaves utterances marked (and starting with the
    float64 kernels costs no more rounds of workgroups than redoing that many behind the float32 kernels: mm_engine.hip), the
    float64 kernels take whole batches while their overlap statistics say the float32 kernels
    would fail, and the engine goes back when the data does.  Results are the oracle's on every call."""
    g = wl.lfmmi_denominator(1500, 84, seed=2)
    rng = np.random.default_rng(3)
    B, N = 9, 90
    lens = rng.integers(40, N + 1, B).astype(np.int32)
    Vr = rng.standard_normal((B, N, g.P)).astype(np.float32)
    Vs = peaky(rng, (B, N, g.P), 10.0)
    bf = make_batch(mm, wl, g, B, {})
    want = [(Vr, False, 0), (Vs, False, None), (Vs, True, B), (Vr, True, B), (Vr, False, 0)]
    refs = {id(Vr): oracle64(oracle, g, Vr, lens), id(Vs): oracle64(oracle, g, Vs, lens)}
    for V, first, redo in want:
        gam, ttl = bf.pdfposteriors(V, lens)
        assert bf.last_exact_first() == first
        if redo is not None:
            assert bf.last_redo_count() == redo
        else:
            assert bf.last_redo_count() > B // 4
        assert bf.last_fallback_count() == 0
        g_ref, t_ref = refs[id(V)]
        check_gamma(gam, g_ref, lens)
        assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("wide", [True, False])
def test_float64_kernels_first_on_ordinary_inputs(mm, wl, oracle, torch, wide):
    """MM_EXACT_FIRST=1: the exact linear-domain kernels alone on N(0,1) log-likelihoods, B beyond the compute units' pairs (and
    odd), P + 1 > 128 (the 4-pass service wave), against the oracle and against the float32 kernels' result.  wide: the whole
    batch on the wide-exponent pair kernels (mm_kernel_wpair.hip: two utterances per workgroup, what the engine runs); else
    (MM_NO_WPAIR) on the one-utterance float64 kernels."""
    g = wl.lfmmi_denominator(1200, 150, seed=5)
    rng = np.random.default_rng(8)
    B, N = 11, 70
    lens = rng.integers(1, N + 1, B).astype(np.int32)
    V = rng.standard_normal((B, N, g.P)).astype(np.float32)
    bf = make_batch(mm, wl, g, B, dict({"MM_EXACT_FIRST": "1", "MM_NO_FALLBACK": "1"}, **({} if wide else {"MM_NO_WPAIR": "1"})))
    assert "mm_fbd_kernel<4" in bf.kernels() and ("mm_fbw_kernel<4" in bf.kernels()) == wide, bf.kernels()
    gam, ttl = bf.pdfposteriors(V, lens)
    assert bf.last_exact_first() and bf.last_fallback_count() == 0
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-4)
    bf32 = make_batch(mm, wl, g, B, {"MM_EXACT_FIRST": "0", "MM_NO_FALLBACK": "1"})
    g32, t32 = bf32.pdfposteriors(V, lens)
    assert not bf32.last_exact_first() and bf32.last_redo_count() == 0
    assert np.allclose(g32, gam, rtol=2e-4, atol=1e-6) and np.allclose(t32, ttl, rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize("wide", [True, False])
def test_emission_offsets_and_masked_pdfs(mm, wl, oracle, torch, wide):
    """GMM-like scores (-300 nats), and pdfs masked with -1e4 in some frames (states on them underflow even a double: the marks
    the float64 kernels raise carry no mass and are cleared by mm_dpair_finish_kernel)."""
    g = wl.lfmmi_denominator(800, 60, seed=7)
    rng = np.random.default_rng(4)
    B, N = 5, 80
    lens = np.array([80, 64, 80, 33, 80], dtype=np.int32)
    V = peaky(rng, (B, N, g.P), 8.0) - 300.0
    V[:, ::3, :7] = -1e4
    bf = make_batch(mm, wl, g, B, dict({"MM_EXACT_FIRST": "1", "MM_NO_FALLBACK": "1"}, **({} if wide else {"MM_NO_WPAIR": "1"})))
    gam, ttl = bf.pdfposteriors(V, lens)
    assert bf.last_fallback_count() == 0
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-2)


def test_capturable_in_a_hip_graph(mm, wl, torch):
    """The chain prologue -> float32 kernels -> finish -> float64 kernels -> finish -> item kernel is launches on the caller's
    stream: a captured call replays to the bits of the eager one, on sharp inputs (the float64 kernels do the work) as well."""
    g = wl.lfmmi_denominator(900, 40, seed=1)
    B, N = 6, 50
    bf = make_batch(mm, wl, g, B, {"MM_EXACT_FIRST": "0"})
    x = 10.0 * torch.randn(B, N, g.P, device="cuda")
    V = torch.log_softmax(x, dim=-1)
    lens = torch.tensor([N, N - 3, 5, 1, N, 17], dtype=torch.int32, device="cuda")
    gamma = torch.empty(B, N, g.P, device="cuda")
    _, t0 = bf.pdfposteriors(V, lens, out=gamma)
    assert bf.last_redo_count() >= 3
    g0, t0 = gamma.clone(), t0.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        bf.pdfposteriors(V, lens, out=gamma)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _, t1 = bf.pdfposteriors(V, lens, out=gamma)
    for _ in range(2):
        gamma.zero_()
        t1.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(gamma, g0) and torch.equal(t1, t0)


@pytest.mark.parametrize("which", ["wsj_den", "teams_of_4"])
def test_sharp_emissions_on_the_float64_team_kernels(mm, wl, oracle, torch, which):
    """Graphs beyond one compute unit (the reference's WSJ denominator: teams of 2; a 3600-state graph: teams of 4): the float64
    kernels run as teams too (mm_fbds_kernel: a granule of the exchange is one tagged double).  Sharp emissions, odd batch,
    different lengths; first behind the float32 team kernels (marked utterances only), then alone on the whole batch."""
    g = wl.load_npz_graph(os.path.join(HERE, "golden", "den_fsm_wsj.npz")) if which == "wsj_den" else wl.lfmmi_denominator(3600, 84, seed=1)
    rng = np.random.default_rng(11)
    B, N = 5, 60
    lens = np.array([60, 41, 60, 1, 33], dtype=np.int32)
    V = peaky(rng, (B, N, g.P), 10.0)
    bf = make_batch(mm, wl, g, B, {"MM_NO_FALLBACK": "1"})
    assert "mm_fbs_kernel" in bf.kernels() and "mm_fbds_kernel" in bf.kernels(), bf.kernels()
    assert ("teams of 2" if which == "wsj_den" else "teams of 4") in bf.kernels()
    # (teams of 2: a whole batch goes to the WIDE pair teams, mm_fbws_kernel -- two utterances per team; teams of 4 stay on mm_fbds)
    assert ("mm_fbws_kernel" in bf.kernels()) == (which == "wsj_den"), bf.kernels()
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    ok = np.isfinite(t_ref)  # (one frame is too short for a path through these graphs: Z = 0, the reference's 0 / 0)
    for call in range(2):
        gam, ttl = bf.pdfposteriors(V, lens)
        assert bf.last_exact_first() == (call == 1)
        # (an utterance without any path that shares a wide pair with a marked partner stays marked: Z = 0 is what a total underflow
        # would look like)
        assert bf.last_redo_count() >= 3 and bf.last_fallback_count() <= (int((~ok & (lens >= 1)).sum()) if call == 1 else 0)
        check_gamma(gam[ok], g_ref[ok], lens[ok])
        assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=1e-3)
        assert (gam[~ok] == 0).all() and np.isneginf(ttl[~ok]).all()


@pytest.mark.parametrize("which", ["config2", "many_to_one", "sharp"])
def test_lane_kernel(mm, wl, oracle, torch, which):
    """Graphs of up to 64 states (mm_kernel_lane.hip: one wave per utterance and direction, the graph in its registers,
    float64, no marks): BASELINE config 2 at full size (dense 64-state HMM, T = 500, B = 32, identity state map); a sparse
    graph whose states share pdfs (many-to-one map, pdfs without states), lengths 0 and 1 included; and emissions sharp
    enough (log-softmax of 25 N(0,1), then -300 nats) to overlap the forward and the backward mass at 2^-400."""
    rng = np.random.default_rng(7)
    if which == "config2":
        g, B, N = wl.dense_ergodic(64, seed=0), 32, 500
        V = rng.standard_normal((B, N, g.P)).astype(np.float32)
        lens = np.full(B, N, dtype=np.int32)
        lens[3], lens[17] = 371, 2
    elif which == "many_to_one":
        g, B, N = wl.random_fsm(50, 9, 3.0, seed=4), 7, 40
        V = rng.standard_normal((B, N, g.P)).astype(np.float32)
        lens = np.array([40, 0, 1, 33, 40, 2, 17], dtype=np.int32)
    else:
        g, B, N = wl.dense_ergodic(40, seed=2), 5, 120
        V = peaky(rng, (B, N, g.P), 25.0) - 300.0
        lens = np.array([120, 77, 120, 9, 120], dtype=np.int32)
    bf = make_batch(mm, wl, g, B, {})
    assert "mm_lane_kernel" in bf.kernels(), bf.kernels()
    gam, ttl = bf.pdfposteriors(V, lens)
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    ok = np.isfinite(t_ref)
    # (marks: only utterances without any path -- too short to reach a final state -- are looked at again by the log-domain
    # kernel behind the lane kernel: Z = 0 is what an underflow would look like too)
    assert bf.last_redo_count() == int((~ok & (lens >= 1)).sum())
    check_gamma(gam[ok], g_ref[ok], lens[ok])
    assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=1e-3)
    assert (gam[~ok] == 0).all() and np.isneginf(ttl[~ok]).all()
    g2, t2 = bf.pdfposteriors(V, lens)  # deterministic: the same bits
    assert np.array_equal(gam, g2) and np.array_equal(ttl, t2, equal_nan=True)


def test_float64_kernels_with_400_pdfs(mm, wl, oracle, torch):
    """mm_fbd_kernel<8, ...>: the float64 pair kernels of graphs with 251 .. 506 pdfs (one utterance per workgroup, per-pdf LDS
    arrays of twice the size, a partner-row ring of two vectors).  Sharp emissions, nothing behind the float64 kernels."""
    g = wl.lfmmi_denominator(2000, 400, seed=11)
    rng = np.random.default_rng(12)
    B, N = 5, 90
    lens = np.array([90, 90, 41, 1, 89], dtype=np.int32)
    V = peaky(rng, (B, N, g.P), 10.0)
    bf = make_batch(mm, wl, g, B, {"MM_NO_FALLBACK": "1", "MM_EXACT_FIRST": "1"})
    assert "mm_fbd_kernel<8" in bf.kernels(), bf.kernels()
    gam, ttl = bf.pdfposteriors(V, lens)
    assert bf.last_exact_first() and bf.last_fallback_count() == 0
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    ok = np.isfinite(t_ref)
    check_gamma(gam[ok], g_ref[ok], lens[ok])
    assert np.allclose(ttl[ok], t_ref[ok], rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("S", [3, 24, 64])
def test_lane_kernel_left_to_right_with_sharp_emissions(mm, wl, oracle, torch, S):
    """A left-to-right HMM (the reference's 3-state demo graph, test/test_algorithms.jl:13-26, and longer chains) has ONE way
    through: when a frame's emission for the state the path must be in lies hundreds of nats below the frame's best pdf, that
    state's linear emission 2^(e - E) must still be a positive number -- a float32 v_exp_f32 gave 0 there (alpha = 0, ttl =
    -inf, gamma = 0 where the reference is finite).  log-softmax of 40 N(0,1): emissions down to ~-250 nats below the best."""
    g = wl.l2r_hmm(S)
    rng = np.random.default_rng(S)
    B, N = 5, max(2 * S, 12)
    V = peaky(rng, (B, N, g.P), 40.0)
    V[0] -= 300.0
    lens = np.array([N, N - 1, N, max(S, 2), N], dtype=np.int32)
    bf = make_batch(mm, wl, g, B, {})
    assert "mm_lane_kernel" in bf.kernels(), bf.kernels()
    gam, ttl = bf.pdfposteriors(V, lens)
    g_ref, t_ref = oracle64(oracle, g, V, lens)
    assert np.isfinite(t_ref).all()
    assert (V.max(-1) - V.min(-1)).max() > 150.0  # (beyond what a float32 exponent can hold below the frame's best)
    assert np.isfinite(ttl).all(), ttl
    check_gamma(gam, g_ref, lens)
    assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("policy", ["f32_first", "f64_first"])
def test_fixed_exact_policy_is_reproducible(mm, wl, oracle, torch, policy):
    """mm_batch_set_exact_policy: under a FIXED policy the launches of a call are a function of the call alone, so two identical
    PIPELINED call sequences (no synchronisation between the calls: what a training loop does) give identical bits -- under
    "auto" the second call's kernels depend on whether the host saw the first call's marks in time.  randn, sharp, sharp, randn
    on one stream, twice; results also against the float64 oracle."""
    g = wl.lfmmi_denominator(1500, 84, seed=2)
    rng = np.random.default_rng(21)
    B, N = 9, 80
    lens = torch.from_numpy(rng.integers(30, N + 1, B).astype(np.int32)).cuda()
    Vr = torch.from_numpy(rng.standard_normal((B, N, g.P)).astype(np.float32)).cuda()
    Vs = torch.from_numpy(peaky(rng, (B, N, g.P), 10.0)).cuda()
    seq = [Vr, Vs, Vs, Vr]

    def run():
        bf = make_batch(mm, wl, g, B, {}).set_exact_policy(policy)
        outs = [torch.empty(B, N, g.P, device="cuda") for _ in seq]
        ttls, firsts = [], []
        for V, o in zip(seq, outs):  # (no synchronisation in here)
            ttls.append(bf.pdfposteriors(V, lens, out=o)[1].clone())
            firsts.append(bf.last_exact_first())
        torch.cuda.synchronize()
        return [o.cpu().numpy() for o in outs], [t.cpu().numpy() for t in ttls], firsts

    g1, t1, f1 = run()
    g2, t2, f2 = run()
    assert f1 == f2 == [policy == "f64_first"] * 4
    for a, b in zip(g1 + t1, g2 + t2):
        assert np.array_equal(a, b, equal_nan=True)
    ln = lens.cpu().numpy()
    for V, gam, ttl in zip(seq, g1, t1):
        g_ref, t_ref = oracle64(oracle, g, V.cpu().numpy(), ln)
        check_gamma(gam, g_ref, ln)
        assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-3)
    with pytest.raises(mm.MarkovModelsAMDError):
        mm._lib.check(mm._lib.lib.mm_batch_set_exact_policy(make_batch(mm, wl, g, 2, {})._h, 7))


def test_wide_pair_kernels_at_full_length(mm, wl, oracle, torch):
    """Config 3's graph, T = 1500, sharp emissions, the whole batch on the wide-exponent pair kernels: the 20 mantissa bits of
    their operands must not add up over 1500 frames -- four utterances against the float64 oracle (posteriors AND the spread of
    the per-frame log Z, which mm_dpair_finish_kernel reads as "no mass lost": nothing may be handed on), and every utterance's
    posteriors summing to 1 in every frame."""
    g = wl.lfmmi_denominator(2000, 84, seed=0)
    rng = np.random.default_rng(31)
    B, N = 6, 1500
    lens = np.array([1500, 1500, 1211, 1500, 977, 1500], dtype=np.int32)
    V = peaky(rng, (B, N, g.P), 10.0)
    bf = make_batch(mm, wl, g, B, {"MM_EXACT_FIRST": "1", "MM_NO_FALLBACK": "1"})
    assert "mm_fbw_kernel<2" in bf.kernels()
    gam, ttl = bf.pdfposteriors(V, lens)
    assert bf.last_exact_first() and bf.last_redo_count() == B and bf.last_fallback_count() == 0
    sel = [0, 1, 2, 4]
    g_ref, t_ref = oracle64(oracle, g, V[sel], lens[sel])
    check_gamma(gam[sel], g_ref, lens[sel])
    assert np.allclose(ttl[sel], t_ref, rtol=1e-6, atol=1e-3)
    for b in range(B):
        assert np.allclose(gam[b, : lens[b]].sum(-1), 1.0, atol=2e-5) and (gam[b, lens[b]:] == 0).all()


@pytest.mark.parametrize("which", ["config3", "wsj_den"])
def test_medium_sharp_emissions_under_every_policy(mm, wl, oracle, torch, which):
    """log-softmax of 2 .. 6 N(0,1): between the benchmark's inputs and a sharp acoustic model, where the float32 kernels' range check
    (round 5: the smallest non-zero SUM of a step against a threshold the service wave derives from the step's emission factors --
    the finishes are linear, src/inference.jl:70-71 as one multiplication) decides utterance by utterance.  Whatever it decides, and
    whichever kernels run first, the result is the float64 oracle's; the float32-first call of the mildest input keeps every
    utterance on the float32 kernels."""
    g = wl.lfmmi_denominator(2000, 84, seed=0) if which == "config3" else wl.load_npz_graph(os.path.join(HERE, "golden", "den_fsm_wsj.npz"))
    rng = np.random.default_rng(41)
    B, N = 6, 160
    lens = np.array([N, N, 97, N, 31, N], dtype=np.int32)
    for sigma in (2.0, 3.0, 4.0, 6.0):
        V = peaky(rng, (B, N, g.P), sigma)
        g_ref, t_ref = oracle64(oracle, g, V, lens)
        for policy in ("f32_first", "f64_first"):
            bf = make_batch(mm, wl, g, B, {})
            bf.set_exact_policy(policy)
            gam, ttl = bf.pdfposteriors(V, lens)
            assert bf.last_fallback_count() == 0
            if policy == "f32_first" and sigma == 2.0 and which == "config3":
                assert bf.last_redo_count() <= B // 2 and "mm_fbp_kernel" in bf.kernels()  # (most utterances stay on the float32 kernels)
            check_gamma(gam, g_ref, lens)
            assert np.allclose(ttl, t_ref, rtol=1e-5, atol=1e-3)


RTOL = 1e-4  # |d log gamma| <= 1e-4 max(|log gamma|, 1) wherever gamma_ref > 1e-30


# ---- the parity contract, pinned on what the team fuzzer found (tools/fuzz_round3.py `split`, inputs replayed from the seed) ----
def _documented_bound_ok(a, ref):
    """The DEFAULT mark policy's contract (include/markovmodels_amd.h, mm_batch_set_mark_policy): |d log gamma| <= 1e-4 max(|log gamma|, 1)
    for posteriors above 1e-24, an absolute error below 1e-27 for smaller ones, 2e-5 absolute overall."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    if not (np.isfinite(a).all() and np.abs(a - ref).max() <= 2e-5):
        return False, "absolute"
    hi = ref >= 1e-24
    if hi.any():
        lr = np.log(ref[hi])
        err = np.abs(np.log(np.maximum(a[hi], 1e-300)) - lr) / np.maximum(np.abs(lr), 1.0)
        if err.max() > 1e-4:
            return False, f"relative {err.max():.3e}"
    lo = ~hi
    if lo.any() and np.abs(a[lo] - ref[lo]).max() > 1e-27:
        return False, f"absolute below 1e-24: {np.abs(a[lo] - ref[lo]).max():.3e}"
    return True, ""


def _strict_miss(a, ref):
    """largest |d log gamma| / max(|log gamma|, 1) over gamma_ref > 1e-30: SURVEY 8(d)'s bar is 1e-4"""
    m = ref > 1e-30
    if not m.any():
        return 0.0
    lr = np.log(ref[m])
    return float((np.abs(np.log(np.maximum(a[m], 1e-300)) - lr) / np.maximum(np.abs(lr), 1.0)).max())


@pytest.mark.parametrize("seed", [1, 3])
def test_fuzzer_findings_pin_the_parity_contract(mm, wl, oracle, torch, seed):
    """Round 4 / 5's fuzzer findings as tests: SEED=1 of `tools/fuzz_round3.py split` holds a 5-frame utterance
    on a 2600-state / 472-pdf graph, sharp emissions, whose float32 range marks are raised and CLEARED and one posterior of 8.49e-29
    comes out as 0 (its only state sits 105 log2 below its frame's maximum, behind flushed predecessors); SEED=3 held a posterior
    of 1.436e-30 computed 1 % low in round 4 (the kernels have changed since: its stream is replayed whole).  Every case of both
    streams, regenerated from the seed (tests/fuzz_cases.py), against the item kernel (log domain):
      * default policy: the DOCUMENTED bound -- relative above 1e-24, absolute 1e-27 below -- an explicit assertion;
      * every case that misses SURVEY 8(d)'s bar (relative down to 1e-30) on the default path, and the pinned case of seed 1 in
        any event: under mm_batch_set_mark_policy(MM_MARKS_KEEP) check_gamma passes UNCHANGED, the pinned case against the float64
        oracle as well."""
    from fuzz_cases import split_cases

    pinned = {(5, 300, 12)} if seed == 1 else set()
    misses, n = [], 0
    cf, last = None, None
    for gi, g, B, N, V0, sharp, lens in split_cases(wl, seed):
        if gi != last:
            cf, last = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P)), gi
        V = torch.from_numpy(V0).cuda()
        if sharp:
            V = torch.log_softmax(8.0 * V, dim=-1)
        lt = torch.from_numpy(lens).cuda()
        ref = _with_env({"MM_DEBUG": "1", "MM_KERNEL": "item"}, lambda: mm.batch(*([cf] * B))).pdfposteriors(V, lt)[0].cpu().numpy().astype(np.float64)
        bf = mm.batch(*([cf] * B))
        assert "mm_fbs_kernel" in bf.kernels(), bf.kernels()
        a = bf.pdfposteriors(V, lt)[0].cpu().numpy().astype(np.float64)
        ok, why = _documented_bound_ok(a, ref)
        assert ok, (seed, gi, B, N, why)
        n += 1
        miss = _strict_miss(a, ref)
        if miss > RTOL or (gi, B, N) in pinned:
            misses.append((gi, B, N, miss))
            bk = mm.batch(*([cf] * B)).set_mark_policy("keep")
            ak, tk = bk.pdfposteriors(V, lt)
            ak, tk = ak.cpu().numpy(), tk.cpu().numpy()
            if (gi, B, N) in pinned:  # the float64 oracle itself
                ref, t_ref = oracle64(oracle, g, V.cpu().numpy(), lens)
                okp = np.isfinite(t_ref)
                assert np.allclose(tk[okp], t_ref[okp], rtol=1e-5, atol=1e-4)
                assert _strict_miss(a, ref) > 1.0, "the pinned finding no longer misses on the default path: the contract in the header can be tightened"
            path = ak.reshape(B, -1).sum(1) > 0  # (utterances without a path: gamma = 0, nothing to compare)
            check_gamma(ak[path], ref[path], lens[path])
            assert (ak[~path] == 0).all() and _strict_miss(ak, ref) <= RTOL
    assert n == 72 and all(m[:3] in pinned or m[3] > RTOL for m in misses)
    if seed == 1:
        assert any(m[:3] == (5, 300, 12) and m[3] > 1.0 for m in misses), misses  # 8.49e-29 computed as 0: |d log| / |log| > 1
