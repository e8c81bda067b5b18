"""Host-side logic of the product that needs no GPU: FSM construction, rawunion,
expand, state maps, the CSR -> packed-item compile step (checked through the
host evaluator mm_debug_packed_product), the C ABI's argument validation."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import graphs


def test_fsm_constructor_matches_oracle(mm, wl, oracle):
    o, _ = oracle
    g = wl.random_fsm(23, 5, 3.0, seed=11)
    f = wl.to_fsm(mm, g)
    of = graphs.to_oracle(o, g, dtype=np.float32)
    assert np.array_equal(f.colptr, of.T_hat.colptr) and np.array_equal(f.rowval, of.T_hat.rowval)
    assert np.allclose(f.nzval, of.T_hat.nzval)
    assert np.array_equal(f.alpha_hat_dense(), of.alpha_hat)
    assert mm.nstates(f) == g.S
    # the arc-list constructor gives the same FSM, duplicates combined with (+)
    f2 = mm.FSM(list(zip(g.init_idx.tolist(), g.init_w.tolist())),
                [((int(i), int(j)), float(w)) for i, j, w in zip(g.src, g.dst, g.w)],
                list(zip(g.final_idx.tolist(), g.final_w.tolist())), list(range(g.S)))
    assert np.array_equal(f2.colptr, f.colptr) and np.allclose(f2.nzval, f.nzval)
    dup = mm.FSM([(0, 0.0)], [((0, 1), np.log(0.25)), ((0, 1), np.log(0.25))], [(1, 0.0)], ["a", "b"])
    assert np.isclose(dup.T_hat_dense()[0, 1], np.log(0.5))
    # extended system: last row = [zero ... zero one]
    Td = f.T_hat_dense()
    assert Td[-1, -1] == 0 and np.isneginf(Td[-1, :-1]).all()


def test_json_and_openfst_readers(mm):
    s = json.dumps({"semiring": "LogSemiring{Float32}", "initstates": [[1, 0.0]],
                    "arcs": [[1, 1, -0.5], [1, 2, -1.0], [2, 2, -0.25]], "finalstates": [[2, -0.125]],
                    "labels": ["a", "b"]})
    f = mm.FSM.from_json(s)
    Td = f.T_hat_dense()
    assert f.semiring == "log" and f.S1 == 3
    assert np.isclose(Td[0, 1], -1.0) and np.isclose(Td[1, 2], -0.125) and Td[2, 2] == 0
    txt = "0 1 3 3 0.5\n1 1 3 3 0.25\n1 2 7 7 1.5\n2 2 7 7 0.1\n2 0.75\n"
    f, s2p, P = mm.FSM.from_openfst_text(txt)
    assert f.S1 == 3 and P == 7 and s2p.tolist() == [2, 6]
    assert np.isclose(f.T_hat_dense()[0, 1], -1.5) and np.isclose(f.alpha_hat_dense()[0], -0.5)


def test_rawunion_and_split(mm, wl):
    gs = [wl.random_fsm(S, 4, 2.0, seed=S) for S in (5, 9, 3)]
    fs = [wl.to_fsm(mm, g) for g in gs]
    u = mm.rawunion(*fs)
    assert u.S1 == sum(f.S1 for f in fs)
    D = u.T_hat_dense()
    o = 0
    for f in fs:
        assert np.array_equal(D[o:o + f.S1, o:o + f.S1], f.T_hat_dense())
        o += f.S1
    assert np.isneginf(D[: fs[0].S1, fs[0].S1:]).all()
    # split a union that carries no provenance
    u._parts = None
    from importlib import import_module
    fsm_mod = import_module(mm.__name__ + ".fsm")
    parts = fsm_mod.split_blocks(u, [f.S1 for f in fs])
    for p, f in zip(parts, fs):
        assert np.array_equal(p.T_hat_dense(), f.T_hat_dense())
        assert np.array_equal(p.alpha_hat_dense(), f.alpha_hat_dense())
    with pytest.raises(mm.DimensionMismatch):
        fsm_mod.split_blocks(u, [3, 4])


def test_expand_semantics(mm, oracle):
    """expand (src/inference.jl:54-60) against the oracle restatement."""
    o, _ = oracle
    lhs = np.arange(12, dtype=np.float32).reshape(3, 4)
    for L in (None, 4, 2, 0):
        assert np.array_equal(mm.expand(lhs, L), o.expand(lhs, L, o.LOG))
    e = mm.expand(lhs, 2)
    assert e.shape == (4, 5) and np.isneginf(e[:3, 2:]).all() and (e[3, 2:] == 0).all() and np.isneginf(e[3, :2]).all()


def test_statemap(mm):
    C1 = mm.statemap([2, 0, 1], 3)
    assert C1.shape == (4, 4) and C1.state2pdf.tolist() == [2, 0, 1, 3]
    M = np.full((4, 4), -np.inf)
    M[[0, 1, 2, 3], [2, 0, 1, 3]] = 0
    assert mm.StateMap.from_matrix(M).state2pdf.tolist() == [2, 0, 1, 3]
    with pytest.raises(ValueError):
        M[0, 0] = 0
        mm.StateMap.from_matrix(M)


@pytest.mark.parametrize("gname", ["l2r", "rand", "ergodic", "lfmmi", "lexicon", "wide", "wsj_num"])
def test_packed_form_products(mm, wl, gname):
    """The compile step (CSR -> wave items) preserves both semiring products:
    host evaluation THROUGH the packed form == dense reference, log and tropical."""
    here = os.path.dirname(os.path.abspath(__file__))
    g = {"l2r": lambda: wl.l2r_hmm(3), "rand": lambda: wl.random_fsm(40, 6, 3.0, seed=1),
         "ergodic": lambda: wl.dense_ergodic(64), "lfmmi": lambda: wl.lfmmi_denominator(2000, 84),
         "lexicon": lambda: wl.lexicon_fsm(1500, 30), "wide": lambda: wl.wide_row_fsm(),
         "wsj_num": lambda: wl.load_npz_graph(os.path.join(here, "golden", "num_fsm_wsj.npz"))}[gname]()
    rng = np.random.default_rng(0)
    for semiring in ("log", "tropical"):
        f = wl.to_fsm(mm, g, semiring=semiring)
        cf = mm.compile(f, mm.statemap(g.state2pdf, g.P))
        info = cf.info()
        assert info["S1"] == g.S + 1 and info["nnz"] == g.n_arcs
        assert min(info["packed_slots"]) >= g.n_arcs
        x = rng.standard_normal(f.S1).astype(np.float32)
        x[rng.random(f.S1) < 0.2] = -np.inf
        Td = f.T_hat_dense().astype(np.float64)
        for d in (0, 1):
            out, arg = cf.packed_product(x, d)
            M = Td.T if d == 0 else Td
            Z = M + x[None, :].astype(np.float64)
            if semiring == "log":
                mx = Z.max(1)
                with np.errstate(invalid="ignore", divide="ignore"):
                    ref = np.where(np.isfinite(mx), mx + np.log(np.exp(Z - np.where(np.isfinite(mx), mx, 0)[:, None]).sum(1)), -np.inf)
                assert np.array_equal(np.isfinite(out), np.isfinite(ref))
                m = np.isfinite(ref)
                assert np.allclose(out[m], ref[m], rtol=1e-5, atol=1e-5)
            else:
                Zf = (M.astype(np.float32) + x[None, :])
                ref = Zf.max(1)
                assert np.array_equal(out, ref)
                ra = np.where(np.isfinite(ref), Zf.argmax(1), -1)  # argmax = first (lowest) index among maxima
                assert np.array_equal(arg, ra)


def test_create_validation(mm):
    lib = mm._lib.lib if hasattr(mm, "_lib") else None
    from importlib import import_module
    L = import_module(mm.__name__ + "._lib")
    lib = L.lib
    h = C.c_void_p()
    colptr = np.array([0, 1, 2], dtype=np.int64)
    rowval = np.array([0, 1], dtype=np.int64)
    nz = np.zeros(2, dtype=np.float32)
    s2p = np.array([0, 1], dtype=np.int32)
    ai = np.array([0], dtype=np.int64)
    av = np.zeros(1, dtype=np.float32)

    def create(**kw):
        a = dict(semiring=0, S1=2, nnz=2, layout=0, ib=8, base=0, vb=4, colptr=colptr, rowval=rowval, s2p=s2p, P1=2)
        a.update(kw)
        return lib.mm_fsm_create(a["semiring"], a["S1"], a["nnz"], a["layout"], a["ib"], a["base"], a["vb"],
                                 a["colptr"].ctypes.data, a["rowval"].ctypes.data, nz.ctypes.data, 1, ai.ctypes.data,
                                 av.ctypes.data, a["s2p"].ctypes.data, a["P1"], C.byref(h))

    assert create() == 0
    assert lib.mm_fsm_destroy(h) == 0
    assert create(semiring=7) == -1 and b"semiring" in lib.mm_last_error()
    assert create(rowval=np.array([0, 5], dtype=np.int64)) == -2  # DimensionMismatch-class error
    assert create(s2p=np.array([1, 0], dtype=np.int32)) == -2  # final state must map to the phony pdf
    assert create(ib=2) == -1
    assert create(colptr=np.array([0, 2, 1], dtype=np.int64)) == -2
    # 1-based Int32 CSR input (the reference's GPU containers) gives the same packed sizes as 0-based Int64 CSC
    g_rowptr = np.array([1, 2, 3], dtype=np.int32)
    g_col = np.array([1, 2], dtype=np.int32)
    ai32 = np.array([1], dtype=np.int32)
    s2p1 = np.array([1, 2], dtype=np.int32)
    rc = lib.mm_fsm_create(0, 2, 2, 1, 4, 1, 4, g_rowptr.ctypes.data, g_col.ctypes.data, nz.ctypes.data, 1,
                           ai32.ctypes.data, av.ctypes.data, s2p1.ctypes.data, 2, C.byref(h))
    assert rc == 0, lib.mm_last_error()
    lib.mm_fsm_destroy(h)


@pytest.mark.parametrize("gname,kqs", [("l2r", (1, 2)), ("rand", (1, 3, 5)), ("ergodic", (1, 2, 7)), ("lfmmi", (9, 10, 13)),
                                       ("lexicon", (5, 6)), ("wide", (1, 2, 11)), ("wsj_den", (13,))])
def test_quad_form_products(mm, wl, gname, kqs):
    """The quad form of the fast kernel (internal renumbering, quads of 4 arcs, bank-aware arc
    placement, per-lane running sums, row totals from lane partials) preserves both products for
    every lane geometry KQ: host evaluation through it == evaluation through the item form."""
    here = os.path.dirname(os.path.abspath(__file__))
    g = {"l2r": lambda: wl.l2r_hmm(3), "rand": lambda: wl.random_fsm(40, 6, 3.0, seed=1),
         "ergodic": lambda: wl.dense_ergodic(64), "lfmmi": lambda: wl.lfmmi_denominator(2000, 84),
         "lexicon": lambda: wl.lexicon_fsm(1500, 30), "wide": lambda: wl.wide_row_fsm(),
         "wsj_den": lambda: wl.load_npz_graph(os.path.join(here, "golden", "den_fsm_wsj.npz"))}[gname]()
    f = wl.to_fsm(mm, g)
    cf = mm.compile(f, mm.statemap(g.state2pdf, g.P))
    rng = np.random.default_rng(3)
    x = (3 * rng.standard_normal(f.S1)).astype(np.float32)
    x[rng.random(f.S1) < 0.1] = -np.inf
    for d in (0, 1):
        ref, _ = cf.packed_product(x, d)
        for KQ in kqs:
            out, stats = cf.quad_product(x, d, KQ)
            assert np.array_equal(np.isfinite(out), np.isfinite(ref))
            m = np.isfinite(ref)
            assert np.allclose(out[m], ref[m], rtol=1e-5, atol=2e-5), (gname, d, KQ)
            assert stats[0] >= g.n_arcs / 4 and stats[1] == np.ceil(stats[0] / KQ)
            # the bank-aware placement must not be worse than CSR order by more than noise
            assert stats[3] <= stats[2] * 1.15 + 0.05


@pytest.mark.parametrize("gname", ["l2r", "rand", "ergodic", "lfmmi", "lfmmi600", "lexicon", "wide", "num_wsj"])
def test_row_form_products(mm, wl, gname):
    """The row-lane form of the row kernels (rows sorted by size and dealt to the compute waves as segments of
    equally long rows, arcs in per-lane register slots with bank-aware placement, lane-group sums for long rows,
    internal numbering = finishing order) preserves both products: host evaluation through it == evaluation
    through the item form.  The schedule must be balanced and the placement no worse than CSR order."""
    here = os.path.dirname(os.path.abspath(__file__))
    g = {"l2r": lambda: wl.l2r_hmm(3), "rand": lambda: wl.random_fsm(40, 6, 3.0, seed=1),
         "ergodic": lambda: wl.dense_ergodic(64), "lfmmi": lambda: wl.lfmmi_denominator(2000, 84),
         "lfmmi600": lambda: wl.lfmmi_denominator(600, 40, seed=5),
         "lexicon": lambda: wl.lexicon_fsm(1500, 30), "wide": lambda: wl.wide_row_fsm(),
         "num_wsj": lambda: wl.load_npz_graph(os.path.join(here, "golden", "num_fsm_wsj.npz"))}[gname]()
    f = wl.to_fsm(mm, g)
    cf = mm.compile(f, mm.statemap(g.state2pdf, g.P))
    rng = np.random.default_rng(3)
    x = (3 * rng.standard_normal(f.S1)).astype(np.float32)
    x[rng.random(f.S1) < 0.1] = -np.inf
    for d in (0, 1):
        ref, _ = cf.packed_product(x, d)
        out, stats = cf.row_product(x, d)
        assert np.array_equal(np.isfinite(out), np.isfinite(ref)), (gname, d)
        m = np.isfinite(ref)
        assert np.allclose(out[m], ref[m], rtol=1e-5, atol=2e-5), (gname, d)
        ka, nwc, nseg, eff, cmax, cmin = stats[:6]
        assert ka % 2 == 0 and 2 <= ka <= 48 and 1 <= nwc <= 15 and nseg >= nwc
        assert eff <= 1.0 and (g.n_arcs < 5000 or eff > 0.75), (gname, d, eff)   # padding of the equal-length segments
        # the waves reach the barrier together (small graphs have fewer segments than waves: nothing to balance)
        assert g.n_arcs < 20000 or cmax <= 1.25 * cmin + 4, (gname, d, cmax, cmin)
        assert stats[7] <= stats[6] * 1.15 + 0.05                                 # bank-aware placement


def test_reach_distance_bounds_the_support_of_alpha_and_beta(mm, wl, oracle):
    """mm_debug_reach_distance: BFS distances on the pruned graph, checked against the oracle's recursions:
    alpha_n[s] is zero(K) for n - 1 < d_f[s], beta_n[s] for (N + 1) - n < d_b[s], and both bounds are tight
    (some frame reaches every useful state exactly at its distance when all emissions are finite)."""
    o, _ = oracle
    for g in (wl.l2r_hmm(5), wl.random_fsm(40, 4, 2.0, seed=3), wl.lexicon_fsm(60, seed=1)):
        cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
        df, db = cf.reach_distance(0), cf.reach_distance(1)
        S1, N = g.S + 1, 2 * (g.S + 1)
        fsm = graphs.to_oracle(o, g)
        lhs = o.expand(np.zeros((g.P, N)), N, fsm.K)
        C = o.statemap(g.state2pdf, g.P, fsm.K)
        state_lhs = o.spmm_csc(C, lhs, fsm.K)
        A = o.alpharecursion(fsm.alpha_hat, fsm.T_hat.transpose(), state_lhs, fsm.K)
        Bm = o.betarecursion(fsm.T_hat, state_lhs, fsm.K)
        useful = (df >= 0) & (db >= 0)
        assert useful[S1 - 1] and df[S1 - 1] > 0 and db[S1 - 1] == 0
        for s in np.nonzero(useful)[0]:
            # frames are 1-based n = 1 .. N+1 in the reference; column n-1 here
            alive_f = np.nonzero(np.isfinite(A[s]))[0]
            if s != S1 - 1:  # (the final state only lives in the last frame: its emission is zero(K) before)
                assert alive_f.size and alive_f[0] >= df[s], (g.name, s)
            if s == S1 - 1:
                continue
            # (the reference initialises beta's last column to one(K) for every state: src/inference.jl:104)
            alive_b = np.nonzero(np.isfinite(Bm[s, :N]))[0]
            assert alive_b.size and (N - alive_b[-1]) >= db[s], (g.name, s)
            assert (N - alive_b[-1]) == db[s], (g.name, s)  # tight: beta reaches s exactly db arcs before the end


@pytest.mark.parametrize("pair,copies,scrambled", [(True, 0, False), (True, 2, False), (True, 2, True), (False, 1, False),
                                                   (False, 2, True)])
def test_row_form_variants(mm, wl, pair, copies, scrambled):
    """The pair variant of the row-lane form (what the pair kernels load) and the placement's copy options: the
    products are those of the item form whatever the layout; a second, scrambled copy of the vector lowers the
    modelled bank conflicts of the gathers below those of a single copy."""
    g = wl.lfmmi_denominator(2000, 84)
    f = wl.to_fsm(mm, g)
    cf = mm.compile(f, mm.statemap(g.state2pdf, g.P))
    rng = np.random.default_rng(8)
    x = (3 * rng.standard_normal(f.S1)).astype(np.float32)
    x[rng.random(f.S1) < 0.1] = -np.inf
    for d in (0, 1):
        ref, _ = cf.packed_product(x, d)
        out, stats = cf.row_product(x, d, pair=pair, copies=copies, scrambled=scrambled)
        m = np.isfinite(ref)
        assert np.array_equal(np.isfinite(out), m) and np.allclose(out[m], ref[m], rtol=1e-5, atol=2e-5)
        _, one = cf.row_product(x, d, pair=pair, copies=1)
        assert stats[7] <= one[7] + 1e-9 if copies != 1 else stats[7] == one[7]
        if copies == 2 and scrambled:
            assert stats[7] < 1.2 and stats[7] < one[7] - 0.1, (d, stats[7], one[7])


@pytest.mark.parametrize("gname,H", [("den_wsj", 2), ("den_wsj", 4), ("lfmmi", 2), ("lfmmi600", 2), ("wide", 2)])
def test_split_form_products(mm, wl, gname, H):
    """The split pair forms (mm_rows.h make_rows_split: the rows cut into H sets, the same sets in both directions, one
    row-lane form per set whose arcs read the whole team's vector at its positions) preserve both products, fit the
    registers of the split kernels on the reference's WSJ denominator graph, and cut the arcs about evenly."""
    here = os.path.dirname(os.path.abspath(__file__))
    g = {"den_wsj": lambda: wl.load_npz_graph(os.path.join(here, "golden", "den_fsm_wsj.npz")),
         "lfmmi": lambda: wl.lfmmi_denominator(2000, 84), "lfmmi600": lambda: wl.lfmmi_denominator(600, 40, seed=5),
         "wide": lambda: wl.wide_row_fsm()}[gname]()
    f = wl.to_fsm(mm, g)
    cf = mm.compile(f, mm.statemap(g.state2pdf, g.P))
    rng = np.random.default_rng(4)
    x = (3 * rng.standard_normal(f.S1)).astype(np.float32)
    x[rng.random(f.S1) < 0.1] = -np.inf
    for d in (0, 1):
        ref, _ = cf.packed_product(x, d)
        out, stats = cf.split_product(x, d, H)
        m = np.isfinite(ref)
        assert np.array_equal(np.isfinite(out), m), (gname, d)
        assert np.allclose(out[m], ref[m], rtol=1e-5, atol=2e-5), (gname, d)
        ka, total, nseg, eff, cmax, cmin = stats[:6]
        assert ka <= 36 and f.S1 <= total <= f.S1 + H
        assert stats[7] <= stats[6] * 1.15 + 0.05
        if gname == "den_wsj" and H == 2:  # (the kernels run teams of 2)
            assert eff > 0.8 and cmax <= 1.4 * cmin + 8, (d, eff, cmax, cmin)


@pytest.mark.parametrize("gname", ["l2r", "rand", "num_wsj", "lfmmi600", "wide"])
def test_wave_form_products(mm, wl, gname):
    """The wave form (one wave per direction: segments of 64 / g rows, at most 4 arcs per lane and segment, log2
    weights, lane-group log-sum-exp) preserves both products -- also rows of more than 256 arcs (`wide`)."""
    here = os.path.dirname(os.path.abspath(__file__))
    g = {"l2r": lambda: wl.l2r_hmm(3), "rand": lambda: wl.random_fsm(40, 6, 3.0, seed=1),
         "lfmmi600": lambda: wl.lfmmi_denominator(600, 40, seed=5), "wide": lambda: wl.wide_row_fsm(),
         "num_wsj": lambda: wl.load_npz_graph(os.path.join(here, "golden", "num_fsm_wsj.npz"))}[gname]()
    f = wl.to_fsm(mm, g)
    cf = mm.compile(f, mm.statemap(g.state2pdf, g.P))
    rng = np.random.default_rng(5)
    x = (3 * rng.standard_normal(f.S1)).astype(np.float32)
    x[rng.random(f.S1) < 0.1] = -np.inf
    for d in (0, 1):
        ref, _ = cf.packed_product(x, d)
        try:
            out, stats = cf.wave_product(x, d)
        except mm.MarkovModelsAMDError:
            assert gname in ("lfmmi600", "wide")  # more than 16 segments of 64 rows, or rows beyond 64 x 4 arcs: not this kernel's graphs
            continue
        m = np.isfinite(ref)
        assert np.array_equal(np.isfinite(out), m), (gname, d)
        assert np.allclose(out[m], ref[m], rtol=1e-5, atol=2e-5), (gname, d)
        assert stats[0] <= 16 and stats[1] <= 16  # (at most 4 segments of 4 slots per wave, 4 waves)


def test_create_many_equals_single_creates(mm, wl):
    """mm_fsm_create_many (compile_many: a mini-batch of new numerator graphs in one call, packed on host threads) against
    mm_fsm_create (compile) graph by graph: 128 different graphs -- the reference's WSJ numerator, lexicon graphs of
    150..500 states, random sparse graphs, one too large for the wave form -- must give the same handles: sizes, the
    semiring product through the item form, and through the wave form (positions, weights, schedules), bit for bit."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    gs = [wl.load_npz_graph(os.path.join(here, "num_fsm_wsj.npz"))]
    for k in range(90):
        gs.append(wl.lexicon_fsm(150 + 4 * k, 40, seed=k, hubs=1 + k % 3))
    for k in range(36):
        gs.append(wl.random_fsm(60 + 7 * k, 40, 2.0 + 0.05 * k, seed=100 + k))
    gs.append(wl.lfmmi_denominator(1500, 40, seed=1))  # beyond the wave form
    assert len(gs) == 128
    P = 40
    fsms, maps = [], []
    for g in gs:
        fsms.append(wl.to_fsm(mm, g))
        maps.append(mm.statemap(g.state2pdf, max(P, g.P)))
    many = mm.compile_many(fsms, maps, threads=4)
    rng = np.random.default_rng(0)
    for g, f, m, cm in zip(gs, fsms, maps, many):
        cs = mm.compile(f, m)
        assert cm.info() == cs.info()
        x = rng.standard_normal(g.S + 1).astype(np.float32)
        for d in (0, 1):
            a, b = cm.packed_product(x, d), cs.packed_product(x, d)
            assert np.array_equal(a[0], b[0], equal_nan=True)
            if g.S + 1 <= 1023:
                wa, wb = cm.wave_product(x, d), cs.wave_product(x, d)
                assert np.array_equal(wa[0], wb[0], equal_nan=True) and np.array_equal(wa[1], wb[1])
    with pytest.raises(mm.MarkovModelsAMDError):  # an invalid graph among them: none is created
        bad = wl.to_fsm(mm, gs[1])
        bad.rowval = bad.rowval.copy()
        bad.rowval[0] = 10 ** 6
        mm.compile_many([fsms[0], bad], [maps[0], maps[1]])


def test_compiled_graph_cache_is_keyed_by_content(mm, wl):
    """The reference-shaped entries (pdfposteriors(fsm::FSM, V_hats, C_hats), src/inference.jl:145) find a graph they have compiled
    before by the CONTENT of its fields: equal graphs built twice share one CompiledFSM, a changed weight or state map does not,
    the misses of one call are compiled together, and a repeated object is hashed once."""
    import importlib

    inf = importlib.import_module(mm.__name__ + ".inference")
    inf.compiled_cache_clear()
    inf.compiled_cache_stats(reset=True)
    gs = [wl.random_fsm(20 + i, 5, 3.0, seed=i) for i in range(4)]
    fsms = [wl.to_fsm(mm, g) for g in gs]
    maps = [mm.statemap(g.state2pdf, g.P) for g in gs]
    a = inf._compiled_for(fsms + [fsms[0]], maps + [maps[0]])
    st = inf.compiled_cache_stats(reset=True)
    assert st["misses"] == 4 and st["hits"] == 1 and a[0] is a[4] and len({id(x) for x in a}) == 4
    # the same graphs built again from scratch: every one found
    fsms2 = [wl.to_fsm(mm, wl.random_fsm(20 + i, 5, 3.0, seed=i)) for i in range(4)]
    maps2 = [mm.statemap(g.state2pdf, g.P) for g in gs]
    b = inf._compiled_for(fsms2, maps2)
    st = inf.compiled_cache_stats(reset=True)
    assert st["misses"] == 0 and st["hits"] == 4 and all(x is y for x, y in zip(a, b))
    # one weight changed / another state map: new entries
    f3 = wl.to_fsm(mm, gs[1])
    f3.nzval = f3.nzval.copy()
    f3.nzval[0] += 0.25
    s2p = gs[2].state2pdf.copy()
    s2p[0] = (s2p[0] + 1) % gs[2].P
    c = inf._compiled_for([f3, fsms[2]], [maps[1], mm.statemap(s2p, gs[2].P)])
    st = inf.compiled_cache_stats()
    assert st["misses"] == 2 and c[0] is not a[1] and c[1] is not a[2] and st["entries"] == 6
    assert inf._content_key(fsms[0], maps[0]) == inf._content_key(fsms2[0], maps2[0]) != inf._content_key(fsms[1], maps[1])
    # the identity memo of `_as_batch` (the same FSM object with the same map objects: no hashing at all) carries a fingerprint of the
    # object's arrays: an FSM edited IN PLACE between two calls is not served the batch of what it used to hold
    f4 = wl.to_fsm(mm, gs[3])
    fp = inf._fingerprint(f4, [maps[3]])
    assert fp == inf._fingerprint(f4, [maps[3]])
    f4.nzval[0] += 0.5
    assert fp != inf._fingerprint(f4, [maps[3]])
    f4.nzval[0] -= 0.5
    f4.alpha_val[0] -= 1.0
    assert fp != inf._fingerprint(f4, [maps[3]])
    # the table is bounded by entries AND by an estimate of the device bytes its graphs hold
    assert inf.compiled_cache_stats()["bytes_estimate"] > 0
    old = inf._COMPILED_LRU_MAX_BYTES
    inf._COMPILED_LRU_MAX_BYTES = 1
    try:
        d = inf._compiled_for(fsms, maps)  # (everything but the newest entry falls out; the call still gets its four graphs)
        assert len(d) == 4 and inf.compiled_cache_stats()["entries"] == 1
    finally:
        inf._COMPILED_LRU_MAX_BYTES = old
    inf.compiled_cache_clear()
    assert inf.compiled_cache_stats()["bytes_estimate"] == 0


@pytest.mark.parametrize("gname", ["rand300", "wide", "den_wsj", "big"])
def test_stream_form_product(mm, wl, gname):
    """The stream form (mm_stream.hip: arcs as 8-byte records streamed per frame, rows in segments of 64 sorted by length, rows of
    more than 128 arcs on a whole wave) evaluates the same semiring products as the packed item form, in both directions: the
    reference's WSJ denominator (its final state has ~1000 incoming arcs: a whole-wave row), a graph with wide rows, and a
    7000-state / 900-pdf graph of config 3's family that no register-resident form takes."""
    g = {"rand300": lambda: wl.random_fsm(300, 11, 4.0, seed=9), "wide": lambda: wl.wide_row_fsm(700, 11),
         "den_wsj": lambda: wl.load_npz_graph(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "den_fsm_wsj.npz")),
         "big": lambda: wl.lfmmi_denominator(7000, 900, seed=3)}[gname]()
    cf = mm.compile(wl.to_fsm(mm, g), mm.statemap(g.state2pdf, g.P))
    rng = np.random.default_rng(1)
    x = rng.standard_normal(cf.S1).astype(np.float32)
    x[rng.integers(0, cf.S1, 5)] = -np.inf
    for d in (0, 1):
        ref, _ = cf.packed_product(x, d)
        slots1 = None
        for H in (1, 2, 4):  # (teams of H workgroups: the rows dealt to H sets, one record stream per set and wave -- the same product)
            got, stats = cf.stream_product(x, d, H)
            fin = np.isfinite(ref)
            assert (got[~fin] == ref[~fin]).all()
            assert np.allclose(got[fin], ref[fin], rtol=0, atol=5e-6 * np.maximum(1, np.abs(ref[fin])).max() + 2e-6 * 44), (gname, d, H)
            assert (0.6 if cf.S1 > 2000 and H == 1 else 0.1) < stats[2] <= 1.0 and stats[1] >= cf.S1 / 64
            if H == 1:
                slots1 = stats[3]
            elif gname == "big":  # the most loaded wave of a team carries less than the lone workgroup's (bounded below by the longest
                # segment: 64 rows of up to 128 arcs go to ONE wave whatever the team)
                assert stats[3] <= (0.85 if H == 2 else 0.82) * slots1, (gname, d, H, stats[3], slots1)
