"""BASELINE.json's full-size configurations on the GPU.  The whole batch is checked through
size-independent properties (per-frame normalisation, exact zeros beyond the sequence length,
invariance of the posteriors / shift of log Z under a per-frame emission offset, agreement of the
independent kernels, Viterbi path consistency); a few utterances of every configuration -- the
longest, the shortest and two others -- are compared with the CPU oracle at full size (one
1500-frame utterance of the config-3 graph takes the C oracle ~3 s)."""
import math
import os

import numpy as np
import pytest

import graphs
from test_gpu_parity import check_gamma

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available()
    return torch


def pick_utterances(lens):
    """longest, shortest and two others"""
    lens = np.asarray(lens)
    return sorted({int(np.argmax(lens)), int(np.argmin(lens)), len(lens) // 3, (2 * len(lens)) // 3})


def test_config1_l2r_hmm_T100(mm, wl, oracle, torch):
    """3-state left-to-right HMM, T = 100, B = 1 (BASELINE configs[0]; the reference runs it on the Julia CPU
    path, here it goes through the same HIP engine).  lhs = 0: every accepting path has weight 2^-100 and there
    are C(99, 2) of them (test/test_algorithms.jl:13-26's FSM, examples/demo.ipynb cell 13 at N = 5)."""
    o, oc = oracle
    g = wl.l2r_hmm(3)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    N = 100
    gam, ttl = mm.pdfposteriors(cf, [mm.expand(np.zeros((3, N), dtype=np.float32))])
    assert np.isclose(ttl[0], math.log(math.comb(N - 1, 2)) - N * math.log(2.0), rtol=1e-6)
    # closed form of the state posteriors: state 1 occupied at frame n (0-based) iff the first jump comes later
    tot = math.comb(N - 1, 2)
    p1 = np.array([math.comb(N - 1 - n, 2) / tot for n in range(N)])       # both jumps after frame n
    p3 = np.array([math.comb(n, 2) / tot for n in range(N)])               # both jumps before or at frame n
    assert np.allclose(gam[0][0], p1, atol=2e-6) and np.allclose(gam[0][2], p3, atol=2e-6)
    rng = np.random.default_rng(100)
    V = rng.standard_normal((1, N, g.P)).astype(np.float32)
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, None, dtype=np.float64)
    gam, ttl = mm.batch(cf).pdfposteriors(V)
    check_gamma(gam, g_ref, [N])
    assert np.allclose(ttl, t_ref, rtol=1e-6)


def test_config3_lfmmi_denominator_full_size(mm, wl, oracle, torch):
    """S = 2000, T = 1500, B = 256 (BASELINE configs[2])."""
    g = wl.lfmmi_denominator(2000, 84, seed=0)
    B, N = 256, 1500
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    gen = torch.Generator(device="cuda").manual_seed(3)
    V = torch.randn(B, N, g.P, device="cuda", generator=gen)
    lens = torch.randint(N // 2, N + 1, (B,), device="cuda", generator=gen, dtype=torch.int32)
    lens[0] = N
    gam, ttl = bf.pdfposteriors(V, lens)
    assert torch.isfinite(gam).all() and torch.isfinite(ttl).all() and (gam >= 0).all()
    frame = torch.arange(N, device="cuda")[None, :]
    valid = frame < lens[:, None]
    sums = gam.sum(-1)
    assert torch.allclose(sums[valid], torch.ones_like(sums[valid]), atol=2e-5)
    assert (gam[~valid] == 0).all()
    # offset c_n added to every pdf of frame n: posteriors unchanged, log Z shifted by sum_n c_n
    c = torch.randn(B, N, 1, device="cuda", generator=gen)
    gam2, ttl2 = bf.pdfposteriors(V + c, lens)
    shift = (c[:, :, 0] * valid).double().sum(1)
    assert torch.allclose(gam2, gam, atol=2e-5)
    assert torch.allclose(ttl2.double(), ttl.double() + shift, rtol=2e-6, atol=5e-3)
    # the general (item) kernel is an independent implementation of the same path
    os.environ.update({"MM_DEBUG": "1", "MM_KERNEL": "item"})
    try:
        sub = slice(0, 32)
        g_item, t_item = mm.batch(*([cf] * 32)).pdfposteriors(V[sub].contiguous(), lens[sub].contiguous())
    finally:
        os.environ.pop("MM_KERNEL", None)
        os.environ.pop("MM_DEBUG", None)
    assert torch.allclose(g_item, gam[sub], atol=2e-5)
    assert torch.allclose(t_item, ttl[sub], rtol=1e-5, atol=5e-3)
    # the oracle at full size on four utterances (float64)
    o, oc = oracle
    sel = pick_utterances(lens.cpu().numpy())
    Vs, Ls = V[sel].cpu().numpy(), lens[sel].cpu().numpy()
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, Vs, Ls, dtype=np.float64, nthreads=4)
    check_gamma(gam[sel].cpu().numpy(), g_ref, Ls)
    assert np.allclose(ttl[sel].cpu().numpy(), t_ref, rtol=1e-6)


def test_config5_viterbi_full_size(mm, wl, oracle, torch):
    """5000-state lexicon FSM, T = 1000, B = 128, tropical (BASELINE configs[4])."""
    g = wl.lexicon_fsm(5000, 84, seed=0)
    B, N = 128, 1000
    ct = mm.compile(wl.to_fsm(mm, g, semiring="tropical"), mm.statemap(g.state2pdf, g.P))
    bt = mm.batch(*([ct] * B))
    gen = torch.Generator(device="cuda").manual_seed(4)
    V = torch.randn(B, N, g.P, device="cuda", generator=gen)
    lens = torch.randint(N // 2, N + 1, (B,), device="cuda", generator=gen, dtype=torch.int32)
    path, score, bp = bt.viterbi(V, lens, return_backpointers=True)
    # the same call without the int32 back-pointer table runs on the row-lane kernels (what bench.py measures): every path
    # and score of the 128 utterances must be the same bits
    assert "mm_vit_kernel" in bt.kernels("tropical")
    path2, score2 = bt.viterbi(V, lens)
    assert torch.equal(path2, path) and torch.equal(score2, score)
    path, score, Vh, L = path.cpu().numpy(), score.cpu().numpy(), V.cpu().numpy(), lens.cpu().numpy()
    assert np.isfinite(score).all()
    # the oracle at full size on four utterances: paths, scores and back-pointers bit exact (float32, same adds)
    o, oc = oracle
    of = graphs.to_oracle(o, g, "tropical", np.float32)
    S1 = g.S + 1
    for b in pick_utterances(L):
        pr, sr, bpr = oc.viterbi(of, g.state2pdf, g.P, Vh[b], int(L[b]), dtype=np.float32)
        assert np.array_equal(path[b], pr) and score[b] == sr
        assert np.array_equal(bp[: L[b] + 1, b * S1:(b + 1) * S1].cpu().numpy(), bpr[: L[b] + 1])
    del bp
    # dense lookup of the arc weights (float32, like the engine)
    W = np.full((g.S, g.S), -np.inf, dtype=np.float32)
    W[g.src, g.dst] = g.w.astype(np.float32)
    init = np.full(g.S, -np.inf, dtype=np.float32)
    init[g.init_idx] = g.init_w.astype(np.float32)
    fin = np.full(g.S, -np.inf, dtype=np.float32)
    fin[g.final_idx] = g.final_w.astype(np.float32)
    for b in range(0, B, 9):
        p = path[b, : L[b]]
        assert (p >= 0).all() and (p < g.S).all() and (path[b, L[b]:] == -1).all()
        # the reported score is the weight of the reported path, accumulated in the engine's order
        acc = np.float32(init[p[0]] + Vh[b, 0, g.state2pdf[p[0]]])
        for n in range(1, L[b]):
            acc = np.float32(np.float32(W[p[n - 1], p[n]] + acc) + Vh[b, n, g.state2pdf[p[n]]])
        acc = np.float32(np.float32(fin[p[-1]] + acc) + np.float32(0))
        assert np.isfinite(acc) and acc == score[b], (b, acc, score[b])
    # max-plus <= log-sum-exp: the best path cannot beat the total
    cl = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    _, ttl = mm.batch(*([cl] * B)).pdfposteriors(V, lens)
    assert (score <= ttl.cpu().numpy() + 1e-3).all()


def test_config2_dense_ergodic(mm, wl, oracle, torch):
    """Dense 64-state ergodic HMM, T = 500, B = 32 (BASELINE configs[1]) against the oracle."""
    import graphs

    o, oc = oracle
    g = wl.dense_ergodic(64, seed=0)
    rng = np.random.default_rng(1)
    V = rng.standard_normal((32, 500, g.P)).astype(np.float32)
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, V, None, dtype=np.float64, nthreads=8)
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    gam, ttl = mm.batch(*([cf] * 32)).pdfposteriors(V)
    assert np.abs(gam - g_ref).max() <= 2e-5
    m = g_ref > 1e-30
    assert (np.abs(np.log(gam[m]) - np.log(g_ref[m])) <= 1e-4 * np.maximum(np.abs(np.log(g_ref[m])), 1)).all()
    assert np.allclose(ttl, t_ref, rtol=1e-5)


def test_config3_alpha_beta_export_full_size(mm, wl, oracle, torch):
    """alpha-recursion / beta-recursion (src/inference.jl:62-74, 99-110) at config 3's full size -- S = 2000, T = 1500, B = 256, lengths
    750 .. 1500: two (512 256 x 1501) matrices of 3.1 GB -- on the pair kernels (by name).  The property that holds for the whole batch
    whatever its size: (+)_s A[s, n] (*) B[s, n] is the utterance's total for EVERY column n, the frames beyond the length included
    (expand()'s padding carries the mass on the phony state) -- compared with the ttl of pdfposteriors, an independent set of kernels;
    the longest and the shortest utterance against the float64 oracle's state_A / state_B."""
    o, oc = oracle
    g = wl.lfmmi_denominator(2000, 84, seed=0)
    B, N = 256, 1500
    S1 = g.S + 1
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    assert "mm_fbx_kernel" in bf.kernels("export") and "mm_pair_export_kernel" in bf.kernels("export"), bf.kernels("export")
    gen = torch.Generator(device="cuda").manual_seed(13)
    V = torch.randn(B, N, g.P, device="cuda", generator=gen)
    lens = torch.randint(N // 2, N + 1, (B,), device="cuda", generator=gen, dtype=torch.int32)
    lens[0] = N
    _, ttl = bf.pdfposteriors(V, lens)
    A = bf.alpharecursion(V, lens)
    assert bf.last_redo_count() == 0  # (N(0,1) emissions: nothing leaves float32's range, nothing is handed to the item kernel)
    Bm = bf.betarecursion(V, lens)
    assert bf.last_redo_count() == 0
    assert tuple(A.shape) == (B * S1, N + 1) and tuple(Bm.shape) == (B * S1, N + 1)
    tot = torch.logsumexp((A + Bm).t().reshape(N + 1, B, S1), dim=2)  # [N + 1, B]
    assert torch.isfinite(tot).all()
    assert torch.allclose(tot, ttl[None, :].expand_as(tot), rtol=1e-5, atol=5e-3), float((tot - ttl[None, :]).abs().max())
    del tot
    L = lens.cpu().numpy()
    for b in (int(np.argmax(L)), int(np.argmin(L))):
        Vb = V[b].cpu().numpy().T.astype(np.float64)
        _, _, Ar, Br = oc.single(graphs.to_oracle(o, g), g.state2pdf, g.P, o.expand(Vb, int(L[b]), o.LOG), want_ab=True)
        for got, ref, what in ((A[b * S1:(b + 1) * S1].cpu().numpy(), Ar, "alpha"), (Bm[b * S1:(b + 1) * S1].cpu().numpy(), Br, "beta")):
            assert np.array_equal(np.isneginf(got), np.isneginf(ref)), (what, b)
            m = np.isfinite(ref)
            assert np.allclose(got[m], ref[m], rtol=1e-5, atol=5e-3), (what, b, np.abs(got[m] - ref[m]).max())


def test_lfmmi_step_full_size_gradient_rows_sum_to_zero(mm, wl, torch):
    """The caller's step at the size bench.py times it (`--workload lfmmi_step`): the reference's WSJ denominator x 128 and 128
    different numerator graphs, T = 700, the fused assembly.  gamma_den and gamma_num are both distributions over the pdfs of a
    frame, so every live row of the gradient sums to zero, the rows beyond the lengths are exactly zero, and the loss is
    -(sum ttl_num - sum ttl_den) of the two calls made on their own."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    den = wl.load_npz_graph(os.path.join(here, "den_fsm_wsj.npz"))
    P, N, B = den.P, 700, 128
    gs = [wl.lexicon_fsm(150 + 3 * b, P, seed=100 + b, hubs=1 + b % 2) for b in range(B)]
    cden = mm.compile(wl.to_fsm(mm, den), mm.statemap(den.state2pdf, P))
    bden = mm.batch(*([cden] * B))
    bnum = mm.batch(*mm.compile_many([wl.to_fsm(mm, g) for g in gs], [mm.statemap(g.state2pdf, P) for g in gs]))
    assert "mm_fbs_kernel" in bden.kernels() and "mm_wave_kernel" in bnum.kernels()
    gen = torch.Generator(device="cuda").manual_seed(17)
    V = torch.randn(B, N, P, device="cuda", generator=gen).requires_grad_(True)
    lens = torch.randint(N // 2, N + 1, (B,), device="cuda", generator=gen, dtype=torch.int32)
    loss, tn, td = mm.lfmmi_loss(V, bnum, bden, lens, mode="fused")
    loss.backward()
    grad = V.grad
    assert torch.isfinite(grad).all() and torch.isfinite(tn).all() and torch.isfinite(td).all()
    valid = torch.arange(N, device="cuda")[None, :] < lens[:, None]
    assert (grad[~valid] == 0).all()
    assert float(grad.sum(-1)[valid].abs().max()) <= 3e-5
    assert float(grad.abs().max()) <= 1.0 + 1e-5
    _, tn2 = bnum.pdfposteriors(V.detach(), lens)
    _, td2 = bden.pdfposteriors(V.detach(), lens)
    assert torch.equal(tn2, tn) and torch.allclose(td2, td, rtol=1e-6)
    assert np.isclose(float(loss.detach()), -float((tn.double() - td.double()).sum()), rtol=1e-6)


@pytest.mark.parametrize("emissions", ["randn", "sharp"])
def test_reference_wsj_denominator_full_size(mm, wl, oracle, torch, emissions):
    """The reference's own benchmark (README: den_fsm_wsj.txt, batch 128, 700 frames -- SURVEY 8d runs it beside config 3) at full
    size on the team kernels: N(0,1) log-likelihoods (the float32 teams) and log-softmax(10 N(0,1)) (`sharp`: every utterance leaves
    float32's range and the exact kernels take the batch -- the second call on the wide teams alone).  Whole batch: per-frame
    normalisation, exact zeros beyond the lengths, the beta export's column totals against ttl; four utterances against the float64
    oracle at full length."""
    o, oc = oracle
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = wl.load_npz_graph(os.path.join(here, "den_fsm_wsj.npz"))
    B, N, S1 = 128, 700, g.S + 1
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    assert "mm_fbs_kernel" in bf.kernels() and "teams of 2" in bf.kernels()
    gen = torch.Generator(device="cuda").manual_seed(23)
    V = torch.randn(B, N, g.P, device="cuda", generator=gen)
    if emissions == "sharp":
        V = torch.log_softmax(10.0 * V, dim=-1)
    lens = torch.randint(N // 2, N + 1, (B,), device="cuda", generator=gen, dtype=torch.int32)
    lens[0] = N
    gam, ttl = bf.pdfposteriors(V, lens)
    redo1 = bf.last_redo_count()
    gam2, ttl2 = bf.pdfposteriors(V, lens)  # (sharp: exact first now -- the wide team kernels alone)
    if emissions == "sharp":
        assert redo1 == B and bf.last_exact_first() and bf.last_fallback_count() == 0
        assert torch.allclose(gam2, gam, atol=2e-5) and torch.allclose(ttl2, ttl, rtol=1e-5)
    else:
        assert redo1 == 0 and torch.equal(gam2, gam) and torch.equal(ttl2, ttl)
    valid = torch.arange(N, device="cuda")[None, :] < lens[:, None]
    assert torch.isfinite(gam2).all() and torch.isfinite(ttl2).all() and (gam2 >= 0).all()
    sums = gam2.sum(-1)
    assert torch.allclose(sums[valid], torch.ones_like(sums[valid]), atol=2e-5)
    assert (gam2[~valid] == 0).all()
    if emissions == "randn":  # the beta export on the team kernels: every column's (+)_s alpha_hat-weighted total ... = ttl through column 1
        assert "mm_fbsx_kernel" in bf.kernels("export")
        Bm = bf.betarecursion(V, lens)  # [B * S1, N + 1]
        assert bf.last_redo_count() == 0
        init = torch.full((S1,), float("-inf"), device="cuda")
        init[torch.from_numpy(g.init_idx.astype(np.int64)).cuda()] = torch.from_numpy(g.init_w.astype(np.float32)).cuda()
        pdf = torch.from_numpy(np.concatenate([g.state2pdf, [0]]).astype(np.int64)).cuda()
        # A[:, 1] = alpha_hat (*) lhs[:, 1] (src/inference.jl:68): total = (+)_s A[s, 1] (*) B[s, 1]
        e1 = V[:, 0, :][:, pdf]          # [B, S1] emission of every state's pdf at frame 1 (the phony state's entry is masked by init = -inf)
        tot = torch.logsumexp(init[None, :] + e1 + Bm[:, 0].reshape(B, S1), dim=1)
        assert torch.allclose(tot, ttl, rtol=1e-5, atol=5e-3), float((tot - ttl).abs().max())
    sel = pick_utterances(lens.cpu().numpy())
    Vs, Ls = V[sel].cpu().numpy(), lens[sel].cpu().numpy()
    g_ref, t_ref = oc.batch_shared(graphs.to_oracle(o, g), g.state2pdf, g.P, Vs, Ls, dtype=np.float64, nthreads=4)
    check_gamma(gam2[sel].cpu().numpy(), g_ref, Ls)
    assert np.allclose(ttl2[sel].cpu().numpy(), t_ref, rtol=1e-5, atol=5e-4)


def test_stream_kernels_at_the_benchmarked_size(mm, wl, torch):
    """The stream kernels at the size tools/bench_big.py times them: 10 000 states, 162 k arcs, 1000 pdfs, B = 64, T = 700 -- teams of 2
    workgroups per utterance and direction.  Whole batch: normalisation, zeros beyond the lengths, a second run the same bits, no
    utterance handed to the exact kernels; four utterances against the item kernel (the log-domain implementation)."""
    g = wl.lfmmi_denominator(10000, 1000, seed=0)
    B, N = 64, 700
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    bf = mm.batch(*([cf] * B))
    assert "mm_stream_kernel" in bf.kernels() and "teams of 2" in bf.kernels(), bf.kernels()
    gen = torch.Generator(device="cuda").manual_seed(29)
    V = torch.randn(B, N, g.P, device="cuda", generator=gen)
    lens = torch.randint(N // 2, N + 1, (B,), device="cuda", generator=gen, dtype=torch.int32)
    lens[0] = N
    gam, ttl = bf.pdfposteriors(V, lens)
    assert bf.last_redo_count() == 0
    gam2, ttl2 = bf.pdfposteriors(V, lens)
    assert torch.equal(gam2, gam) and torch.equal(ttl2, ttl)
    valid = torch.arange(N, device="cuda")[None, :] < lens[:, None]
    assert torch.isfinite(gam).all() and torch.isfinite(ttl).all() and (gam >= 0).all()
    sums = gam.sum(-1)
    assert torch.allclose(sums[valid], torch.ones_like(sums[valid]), atol=2e-5)
    assert (gam[~valid] == 0).all()
    sel = pick_utterances(lens.cpu().numpy())
    os.environ.update({"MM_DEBUG": "1", "MM_KERNEL": "item"})
    try:
        g_item, t_item = mm.batch(*([cf] * len(sel))).pdfposteriors(V[sel].contiguous(), lens[sel].contiguous())
    finally:
        os.environ.pop("MM_KERNEL", None)
        os.environ.pop("MM_DEBUG", None)
    check_gamma(gam[sel].cpu().numpy(), g_item.cpu().numpy().astype(np.float64), lens[sel].cpu().numpy())
    assert torch.allclose(t_item, ttl[sel], rtol=1e-5, atol=5e-3)
