"""GPU tests of the caller's step behind the path (examples/test_cuda.jl:128-152): numerator + denominator posteriors and their
difference.  mm_batch_set_gamma_mode (the numerator kernel accumulates -gamma_num into the buffer the denominator call wrote:
no third pass), the three ways lfmmi.py assembles the step, and the whole step captured in ONE hipGraph -- all against the
float64 oracle."""
import os

import numpy as np
import pytest

import graphs
from test_gpu_parity import check_gamma

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available()
    return torch


def oracle_batch(oracle, gs, V, lens):
    """float64 oracle posteriors / ttl of a batch of (possibly different) graphs"""
    o, oc = oracle
    B, N, P = V.shape
    gam, ttl = np.zeros((B, N, P)), np.zeros(B)
    for b, g in enumerate(gs):
        gb, tb = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, P, V[b : b + 1].astype(np.float64), lens[b : b + 1], dtype=np.float64)
        gam[b], ttl[b] = gb[0], tb[0]
    return gam, ttl


def numerator_graphs(wl, P, kind):
    if kind == "lane":  # (<= 64 states: the lane kernel -- no accumulate mode there, "auto" subtracts in a pass of its own)
        return [wl.random_fsm(S, P, 2.0, seed=30 + S) for S in (8, 11, 9, 14, 10)]
    # left-to-right word chains of 90 .. 300 states: the wave kernel, like LF-MMI numerators
    return [wl.lexicon_fsm(S, P, seed=40 + S, hubs=1) for S in (90, 140, 300, 120, 201)]


def test_gamma_mode_accumulates_into_the_callers_buffer(mm, wl, oracle, torch):
    """mm_batch_set_gamma_mode on a batch of the wave kernel: scale alone (gamma_out = s * gamma), accumulate (gamma_out += s * gamma,
    the frames beyond the lengths untouched), twice the same bits; batches of other kernels refuse it (MM_ERR_UNSUPPORTED)."""
    P, N = 12, 33
    gs = numerator_graphs(wl, P, "wave")
    B = len(gs)
    lens = np.array([33, 20, 33, 1, 0], dtype=np.int32)
    rng = np.random.default_rng(3)
    V = rng.standard_normal((B, N, P)).astype(np.float32)
    g_ref, t_ref = oracle_batch(oracle, gs, V, lens)
    bf = mm.batch(*[mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, P)) for g in gs])
    assert "mm_wave_kernel" in bf.kernels(), bf.kernels()
    Vt, lt = torch.from_numpy(V).cuda(), torch.from_numpy(lens).cuda()
    g0, t0 = bf.pdfposteriors(Vt, lt)
    check_gamma(g0.cpu().numpy()[:4], g_ref[:4], lens[:4])
    ok = np.isfinite(t_ref)
    # scale alone
    bf.set_gamma_mode(False, -0.5)
    g1, t1 = bf.pdfposteriors(Vt, lt)
    assert torch.equal(g1, -0.5 * g0) and torch.equal(t1, t0)
    # accumulate: X - gamma, X untouched beyond the lengths
    X = torch.from_numpy(rng.standard_normal((B, N, P)).astype(np.float32)).cuda()
    outs = []
    for _ in range(2):
        buf = X.clone()
        bf.set_gamma_mode(True, -1.0)
        g2, t2 = bf.pdfposteriors(Vt, lt, out=buf)
        assert g2.data_ptr() == buf.data_ptr() and torch.equal(t2, t0)
        outs.append(buf.clone())
    assert torch.equal(outs[0], outs[1])  # (one add per element: no order to depend on)
    want = X.cpu().numpy().astype(np.float64) - g_ref
    got = outs[0].cpu().numpy()
    assert np.abs(got - want)[ok].max() <= 3e-5
    for b, L in enumerate(lens):
        assert np.array_equal(got[b, L:], X.cpu().numpy()[b, L:])
    # back to the default: the plain posteriors again
    bf.set_gamma_mode(False, 1.0)
    g3, _ = bf.pdfposteriors(Vt, lt)
    assert torch.equal(g3, g0)
    # a batch of the pair kernels may compute an utterance twice: no accumulate mode
    den = wl.lfmmi_denominator(1200, P, seed=3)
    bd = mm.batch(*([mm.compile(wl.to_fsm(mm, den), mm.statemap(den.state2pdf, P))] * 3))
    assert "mm_fbp_kernel" in bd.kernels()
    with pytest.raises(mm.MarkovModelsAMDError) as e:
        bd.set_gamma_mode(True, -1.0)
    assert e.value.code == -4
    bd.set_gamma_mode(False, 1.0)  # (the default is always accepted)
    # a scale that is not a finite number, a NULL batch: MM_ERR_INVALID, the mode left as it was
    for bad in (float("nan"), float("inf")):
        with pytest.raises(mm.MarkovModelsAMDError) as e:
            bf.set_gamma_mode(False, bad)
        assert e.value.code == -1
    from importlib import import_module

    lib = import_module(mm.__name__ + "._lib").lib
    assert lib.mm_batch_set_gamma_mode(None, 0, 1.0) == -1
    g4, _ = bf.pdfposteriors(Vt, lt)
    assert torch.equal(g4, g0)


@pytest.mark.parametrize("nums", ["lane", "wave"])
@pytest.mark.parametrize("mode", ["auto", "serial", "concurrent", "fused"])
def test_lfmmi_step_every_mode(mm, wl, oracle, torch, mode, nums):
    """loss = -sum(ttl_num - ttl_den), gradient = gamma_den - gamma_num (examples/test_cuda.jl:140-152) through every assembly of
    the step, numerators on the lane kernel (tiny graphs) and on the wave kernel, against the float64 oracle."""
    P, N = 10, 40
    den = wl.lfmmi_denominator(700, P, seed=21)
    gs = numerator_graphs(wl, P, nums)
    B = len(gs)
    lens = np.array([40, 31, 40, 7, 22], dtype=np.int32)
    rng = np.random.default_rng(11)
    V = rng.standard_normal((B, N, P)).astype(np.float32)
    cden = mm.compile(wl.to_fsm(mm, den), mm.statemap(den.state2pdf, P))
    bden = mm.batch(*([cden] * B))
    bnum = mm.batch(*[mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, P)) for g in gs])
    assert ("mm_wave_kernel" if nums == "wave" else "mm_lane_kernel") in bnum.kernels(), bnum.kernels()
    Vt = torch.from_numpy(V).cuda().requires_grad_(True)
    lt = torch.from_numpy(lens).cuda()
    if mode == "fused" and nums == "lane":
        with pytest.raises(mm.MarkovModelsAMDError):
            mm.lfmmi_loss(Vt, bnum, bden, lt, mode=mode)
        return
    loss, tn, td = mm.lfmmi_loss(Vt, bnum, bden, lt, mode=mode)
    (2.0 * loss).backward()
    gn, tnr = oracle_batch(oracle, gs, V, lens)
    gd, tdr = oracle_batch(oracle, [den] * B, V, lens)
    assert np.allclose(tn.cpu().numpy(), tnr, rtol=1e-5, atol=1e-4) and np.allclose(td.cpu().numpy(), tdr, rtol=1e-5, atol=1e-4)
    assert np.isclose(float(loss.detach()), -(tnr - tdr).sum(), rtol=1e-5, atol=1e-3)
    assert np.abs(Vt.grad.cpu().numpy() - 2.0 * (gd - gn)).max() <= 6e-5
    # the numerator batch is left in its default mode: a plain call gives plain posteriors
    g_plain, _ = bnum.pdfposteriors(Vt.detach(), lt)
    check_gamma(g_plain.cpu().numpy(), gn, lens)


def test_lfmmi_step_in_one_hip_graph(mm, wl, oracle, torch):
    """The whole step -- denominator call (split pair kernels on the reference's WSJ denominator: teams, their memset node, the
    exact launches behind), numerator call accumulating into the same buffer (wave kernel, 6 different graphs) -- captured in ONE
    hipGraph and replayed on new inputs: the bits of the eager step, the oracle's gradient."""
    gold = os.path.join(HERE, "golden")
    den = wl.load_npz_graph(os.path.join(gold, "den_fsm_wsj.npz"))
    num = wl.load_npz_graph(os.path.join(gold, "num_fsm_wsj.npz"))
    P, N, B = den.P, 60, 6
    # (lexicon graphs: the WSJ numerator itself is 165 arcs deep, no path of 60 frames accepts it -- the first and the fourth stand
    # for utterances without a numerator path: Z = 0, gamma_num = 0, ttl_num = -inf)
    gs = [num, wl.lexicon_fsm(300, P, seed=2, hubs=1), wl.lexicon_fsm(220, P, seed=3, hubs=1), wl.lexicon_fsm(500, P, seed=5, hubs=2), num, wl.lexicon_fsm(150, P, seed=9, hubs=1)]
    cden = mm.compile(wl.to_fsm(mm, den), mm.statemap(den.state2pdf, P))
    bden = mm.batch(*([cden] * B))
    bnum = mm.batch(*[mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, P)) for g in gs])
    assert "mm_fbs_kernel" in bden.kernels() and "mm_wave_kernel" in bnum.kernels()
    bden.reserve(N)
    bnum.reserve(N)
    bden.set_deterministic(True)  # (should a short utterance ever reach the item kernel: no float atomics there)
    bden.set_exact_policy("f32_first")  # (the launches of a call a function of the call alone: the eager step below must match the captured one bit for bit)
    lens = torch.tensor([N, N - 7, 31, N, 12, N], dtype=torch.int32, device="cuda")
    V = torch.randn(B, N, P, device="cuda")
    grad = torch.empty(B, N, P, device="cuda")
    from importlib import import_module

    lf = import_module(mm.__name__ + ".lfmmi")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):  # (warm-up on a side stream, as torch wants it before a capture)
        lf.posteriors_difference(V, bnum, bden, lens, "fused", out=grad)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _, tn, td = lf.posteriors_difference(V, bnum, bden, lens, "fused", out=grad)
    rng = np.random.default_rng(5)
    for _ in range(6):  # new inputs in the captured buffers
        Vn = rng.standard_normal((B, N, P)).astype(np.float32)
        V.copy_(torch.from_numpy(Vn))
        grad.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        g_graph, tn_g, td_g = grad.clone(), tn.clone(), td.clone()
        marks_graph = (bden.last_redo_count(), bden.last_fallback_count())
        g_eager, tn_e, td_e = lf.posteriors_difference(V, bnum, bden, lens, "fused")
        # (the replay marks what the eager call marks -- no team timed out in it: see mm_zero_kernel in mm_engine.hip)
        assert marks_graph == (bden.last_redo_count(), bden.last_fallback_count()) and marks_graph[0] < B
        assert torch.equal(g_graph, g_eager), float((g_graph - g_eager).abs().max())
        assert torch.equal(tn_g, tn_e) and torch.equal(td_g, td_e)
    ln = lens.cpu().numpy()
    gn, tnr = oracle_batch(oracle, gs, Vn, ln)
    gd, tdr = oracle_batch(oracle, [den] * B, Vn, ln)
    nopath = ~np.isfinite(tnr)  # (the oracle's 0 / 0, src/inference.jl:158; the engine: gamma = 0, ttl = -inf)
    assert nopath.tolist() == [True, False, False, False, True, False]
    gn[nopath] = 0.0
    assert np.abs(g_graph.cpu().numpy() - (gd - gn)).max() <= 6e-5
    assert np.isneginf(tn_g.cpu().numpy()[nopath]).all()
    assert np.allclose(tn_g.cpu().numpy()[~nopath], tnr[~nopath], rtol=1e-5, atol=1e-3) and np.allclose(td_g.cpu().numpy(), tdr, rtol=1e-5, atol=1e-3)
