"""GPU tests of the reference's `mul!` at the C boundary (mm_spmv / mm_spmm / mm_svdv through linalg.py): the reference's
OWN live test at this seam -- test/test_linalg.jl:88-108, a rectangular 4 x 3 sparse matrix times a dense matrix and a dense
vector for LogSemiring / ProbSemiring / TropicalSemiring x Float32 / Float64 -- run through the HIP kernels and compared with
the values of tests/golden/known_answers.json (the semirings' definitions, tests/golden/make_mul_fixture.py); then larger
random systems against the oracle's restatement of Julia's generic CSC `mul!` (oracle/mm_oracle.py spmm_csc)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "known_answers.json")))["mul_known_answer"]


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available()
    return torch


def rtol_of(dtype):
    return 1e-12 if dtype == np.float64 else 2e-6


@pytest.mark.parametrize("semiring", ["log", "prob", "tropical"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_reference_mul_test_through_hip(mm, torch, semiring, dtype):
    """test/test_linalg.jl:93-106: `sm`, `dm`, `dv` as the reference builds them, `mul!(similar(dm, 4, 4), sm, dm)` and
    `mul!(similar(dv, 4), sm, dv)` on the device."""
    sm = mm.SparseCSR.from_coo(KA["I"], KA["J"], KA["V"], KA["shape"], semiring, dtype)  # 1-based like Julia
    assert sm.index_base == 1 and sm.rowptr.cpu().tolist() == [1, 2, 4, 5, 6]
    dv = torch.tensor(KA["dv"], dtype=torch.from_numpy(np.zeros(1, dtype)).dtype).cuda()
    dm = torch.tensor(KA["dm_colmajor"], dtype=dv.dtype).cuda().reshape(4, 3).t()  # reshape(1:12, 3, 4), column-major
    assert dm.shape == (3, 4) and dm.stride(0) == 1
    exp = KA["expected"][semiring]
    # garbage in the outputs first: `similar` is uninitialised memory
    cv = torch.full((4,), 123.0, dtype=dv.dtype).cuda()
    r = mm.mul_(cv, sm, dv)
    assert r is cv
    assert np.allclose(cv.cpu().numpy(), exp["spmv"], rtol=rtol_of(dtype), atol=0)
    cm = mm.linalg.colmajor(torch.full((4, 4), -7.0, dtype=dv.dtype).cuda())
    mm.mul_(cm, sm, dm)
    assert np.allclose(cm.cpu().numpy(), np.array(exp["spmm_rows"]), rtol=rtol_of(dtype), atol=0)
    # the first column of the matrix product is the vector product (dm[:, 1] == dv)
    assert np.allclose(cm[:, 0].cpu().numpy(), cv.cpu().numpy(), rtol=rtol_of(dtype))


def random_system(o, rng, rows, cols, mean_nnz, K, dtype, empty_rows=True):
    nnz_r = rng.poisson(mean_nnz, rows)
    if empty_rows:
        nnz_r[rng.integers(0, rows, max(1, rows // 10))] = 0
    nnz_r = np.minimum(nnz_r, cols)
    I = np.repeat(np.arange(rows), nnz_r)
    J = np.concatenate([rng.choice(cols, n, replace=False) for n in nnz_r]) if nnz_r.sum() else np.zeros(0, np.int64)
    V = rng.standard_normal(I.shape[0]) * 3
    if K.name == "prob":
        V = np.exp(V)
    return I, J, V.astype(dtype)


@pytest.mark.parametrize("semiring", ["log", "tropical", "prob"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(257, 131, 3.0), (64, 2000, 150.0), (1500, 1500, 17.0), (40, 9, 1.0)])
def test_spmv_spmm_against_the_oracle(mm, torch, oracle, semiring, dtype, shape):
    """Rectangular systems with empty rows, rows of one entry (lane groups of 1) up to rows longer than a wave (groups of
    64 striding the row), zero(K) entries in b: every lane-group width of the SpMV kernel, beta = 0 and beta = 1 of the SpMM."""
    o, _ = oracle
    K = o.SEMIRINGS[semiring]
    rows, cols, mean = shape
    rng = np.random.default_rng(sum(map(ord, semiring)) + rows)
    I, J, V = random_system(o, rng, rows, cols, mean, K, dtype)
    A = mm.SparseCSR.from_coo(I + 1, J + 1, V, (rows, cols), semiring, dtype)
    Ao = o.csc_from_coo(I, J, V.astype(np.float64), (rows, cols), K)
    tdt = A.dtype
    b = rng.standard_normal(cols) * 4
    Bm = rng.standard_normal((cols, 5)) * 4
    if semiring == "prob":
        b, Bm = np.exp(b), np.exp(Bm)
    b[rng.integers(0, cols, 3)] = K.zero
    Bm[rng.integers(0, cols, 3), 2] = K.zero
    tol = dict(rtol=1e-11, atol=1e-11) if dtype == np.float64 else dict(rtol=3e-5, atol=3e-5)

    def close(got, ref):
        got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
        fin = np.isfinite(ref)
        return (got[~fin] == ref[~fin]).all() and np.allclose(got[fin], ref[fin], **tol)

    c = torch.full((rows,), 55.0, dtype=tdt).cuda()
    mm.mul_(c, A, torch.from_numpy(b.astype(dtype)).cuda())
    ref_v = o.spmm_csc(Ao, b.astype(dtype).astype(np.float64), K)
    assert close(c.cpu().numpy(), ref_v)
    Bd = mm.linalg.colmajor(torch.from_numpy(Bm.astype(dtype)).cuda())
    Cd = mm.linalg.colmajor(torch.full((rows, 5), 9.0, dtype=tdt).cuda())
    mm.mul_(Cd, A, Bd)
    ref_m = o.spmm_csc(Ao, Bm.astype(dtype).astype(np.float64), K)
    assert close(Cd.cpu().numpy(), ref_m)
    # beta = 1 (src/linalg.jl:246: C is kept and accumulated into)
    C0 = rng.standard_normal((rows, 5))
    if semiring == "prob":
        C0 = np.exp(C0)
    Cd = mm.linalg.colmajor(torch.from_numpy(C0.astype(dtype)).cuda())
    mm.mul_(Cd, A, Bd, True, True)
    acc = K.add(C0.astype(dtype).astype(np.float64), ref_m)
    assert close(Cd.cpu().numpy(), acc)
    # a leading dimension larger than the rows (a view into a taller matrix)
    tall = torch.full((5, rows + 3), 1.0, dtype=tdt).cuda()
    view = tall.t()[:rows]
    assert view.stride(0) == 1 and view.stride(1) == rows + 3
    mm.mul_(view, A, Bd)
    assert close(view.cpu().numpy(), ref_m) and (tall[:, rows:] == 1.0).all()


@pytest.mark.parametrize("semiring", ["log", "prob"])
def test_sparse_vector_broadcast(mm, torch, semiring):
    """elmul! / eldiv! of a sparse vector with a dense one (src/linalg.jl:287-328): what alpha-recursion's first frame uses
    (`alpha_hat (*) lhs[:, 1]`, src/inference.jl:68)."""
    rng = np.random.default_rng(5)
    n, idx = 1000, np.sort(np.random.default_rng(6).choice(1000, 38, replace=False))
    xv = rng.standard_normal(38).astype(np.float32)
    y = rng.standard_normal(n).astype(np.float32)
    if semiring == "prob":
        xv, y = np.exp(xv), np.exp(y)
    x = mm.SparseVector(idx + 1, xv, n, semiring)
    zero = 0.0 if semiring == "prob" else -np.inf
    for fn, op in ((mm.elmul_, (lambda a, b: a * b) if semiring == "prob" else (lambda a, b: a + b)),
                   (mm.eldiv_, (lambda a, b: a / b) if semiring == "prob" else (lambda a, b: a - b))):
        out = torch.full((n,), 3.0).cuda()
        fn(out, x, torch.from_numpy(y).cuda())
        ref = np.full(n, zero, np.float32)
        ref[idx] = op(xv, y[idx])
        got = out.cpu().numpy()
        assert (got[np.isinf(ref)] == ref[np.isinf(ref)]).all() and (got[ref == 0] == 0).all()
        assert np.allclose(got[idx], ref[idx], rtol=1e-6)


def test_dimension_mismatch_and_empty(mm, torch):
    """@boundscheck (src/linalg.jl:166-167, 242-244) -> DimensionMismatch; an A without stored entries leaves c alone (:169)."""
    A = mm.SparseCSR.from_coo(KA["I"], KA["J"], KA["V"], KA["shape"], "log", np.float32)
    with pytest.raises(mm.DimensionMismatch):
        mm.mul_(torch.zeros(4).cuda(), A, torch.zeros(4).cuda())
    with pytest.raises(mm.DimensionMismatch):
        mm.mul_(torch.zeros(3).cuda(), A, torch.zeros(3).cuda())
    with pytest.raises(mm.DimensionMismatch):
        mm.mul_(mm.linalg.colmajor(torch.zeros(4, 2).cuda()), A, mm.linalg.colmajor(torch.zeros(3, 5).cuda()))
    with pytest.raises(TypeError):
        mm.mul_(torch.zeros(4, dtype=torch.float64).cuda(), A, torch.zeros(3).cuda())
    E = mm.SparseCSR([1, 1, 1], np.zeros(0, np.int32), np.zeros(0, np.float32), (2, 3), "log")
    c = torch.full((2,), 7.0).cuda()
    mm.mul_(c, E, torch.zeros(3).cuda())
    assert (c == 7.0).all()
    Cm = mm.linalg.colmajor(torch.full((2, 2), 7.0).cuda())
    mm.mul_(Cm, E, mm.linalg.colmajor(torch.zeros(3, 2).cuda()))
    assert torch.isinf(Cm).all() and (Cm < 0).all()  # fill!(C, zero(K)) happens before the nnz test (:246-249)


def test_spmv_is_the_recursions_product(mm, wl, torch, oracle):
    """`mul!(buffer, T_hat', A[:, n-1])` of alpha-recursion (src/inference.jl:70) on the reference's WSJ denominator graph:
    the stand-alone product against the packed forms' host evaluation (mm_debug_packed_product) and the oracle."""
    g = wl.load_npz_graph(os.path.join(HERE, "golden", "den_fsm_wsj.npz"))
    fsm = wl.to_fsm(mm, g)
    S1 = fsm.S1
    # CSR of T_hat' = CSC of T_hat with rows and columns swapped: colptr is the rowptr of the transpose
    A = mm.SparseCSR(np.asarray(fsm.colptr) + 1, np.asarray(fsm.rowval) + 1, np.asarray(fsm.nzval, np.float32), (S1, S1), "log")
    x = np.random.default_rng(0).standard_normal(S1).astype(np.float32)
    c = torch.empty(S1).cuda()
    mm.mul_(c, A, torch.from_numpy(x).cuda())
    cf = mm.compile(fsm, mm.statemap(g.state2pdf, g.P))
    ref, _ = cf.packed_product(x, 0)
    o, _ = oracle
    Ao = o.csc_from_coo(np.repeat(np.arange(S1), np.diff(np.asarray(fsm.colptr))), np.asarray(fsm.rowval), np.asarray(fsm.nzval, np.float64), (S1, S1), o.LOG)
    ref_o = o.spmm_csc(Ao, x.astype(np.float64), o.LOG)
    assert np.allclose(ref[np.isfinite(ref_o)], ref_o[np.isfinite(ref_o)], rtol=2e-5, atol=2e-5)
    got = c.cpu().numpy()
    fin = np.isfinite(ref)
    assert (got[~fin] == ref[~fin]).all() and np.allclose(got[fin], ref[fin], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("semiring", ["log", "tropical", "prob"])
def test_nan_terms_reach_the_result(mm, torch, semiring):
    """A NaN in b (a diverged network's output) must reach every row of c that reads it -- the reference's logaddexp / max / +
    propagate it (Semirings.jl) -- and no other row; rows of one entry and rows whose lane groups are merged by the DPP butterflies
    alike.  A column index outside the matrix gives NaN as well (never a read out of bounds)."""
    rng = np.random.default_rng(4)
    rows, cols = 70, 50
    for mean in (1.5, 12.0, 40.0):  # lane groups of 2, 16 and 64 lanes
        nnz_r = np.clip(rng.poisson(mean, rows), 1, cols)
        I = np.repeat(np.arange(rows), nnz_r) + 1
        J = np.concatenate([rng.choice(cols, n, replace=False) for n in nnz_r]) + 1
        V = rng.standard_normal(I.size).astype(np.float32)
        if semiring == "prob":
            V = np.exp(V)
        A = mm.SparseCSR.from_coo(I, J, V, (rows, cols), semiring, np.float32)
        b = rng.standard_normal(cols).astype(np.float32)
        if semiring == "prob":
            b = np.exp(b)
        bad = 7
        b[bad] = np.nan
        c = mm.mul_(torch.zeros(rows, device="cuda"), A, torch.from_numpy(b).cuda()).cpu().numpy()
        reads = np.zeros(rows, bool)
        reads[I[J == bad + 1] - 1] = True
        assert np.isnan(c[reads]).all() and np.isfinite(c[~reads]).all(), (semiring, mean)
        Bm = mm.linalg.colmajor(torch.from_numpy(np.stack([b, np.where(np.isnan(b), 0.5, b)], 1)).cuda())
        C = mm.mul_(mm.linalg.colmajor(torch.zeros(rows, 2, device="cuda")), A, Bm).cpu().numpy()
        assert np.isnan(C[reads, 0]).all() and np.isfinite(C[~reads, 0]).all() and np.isfinite(C[:, 1]).all()
    # an index beyond the columns
    A2 = mm.SparseCSR(np.array([1, 2, 3], np.int32), np.array([2, 9], np.int32), np.array([0.5, 0.25], np.float32), (2, 3), semiring)
    c2 = mm.mul_(torch.zeros(2, device="cuda"), A2, torch.ones(3, device="cuda")).cpu().numpy()
    assert np.isfinite(c2[0]) and np.isnan(c2[1])
