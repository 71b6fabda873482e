"""Parity of the HIP path (through the C ABI, via ctypes) against the CPU oracle and the
golden vectors recorded from the real reference.  Needs an MI355X: run with -m gpu.

Tolerances (fp64): the reference sums rows with NumPy/BLAS and SciPy, the kernels with
sequential fused multiply-adds in the same stored order, so agreement is to rounding,
not bitwise: element-wise rtol 1e-11 on single operations, and BASELINE.json's gate —
relative difference of the final residual norm <= 1e-10 — on whole cycles.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import openmg_amd
from openmg_amd import _hip, operators, solvers, tools
from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu

OP = dict(rtol=1e-11, atol=1e-13)      # one operation
CYC = dict(rtol=1e-9, atol=1e-11)      # iterate after several cycles
NORM_RTOL = 1e-10                      # BASELINE.json parity gate


def csr_from(d, prefix):
    return sp.csr_matrix((d[prefix + "_data"], d[prefix + "_indices"], d[prefix + "_indptr"]),
                         shape=tuple(d[prefix + "_shape"]))


def random_csr(n, m, rng, density=0.02, empty_rows=True, long_row=None, unsorted=True):
    A = sp.random(n, m, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))),
                  format="lil")
    if empty_rows and n > 6:
        A[3, :] = 0
        A[n - 2, :] = 0
    if long_row is not None:
        A[long_row, :] = rng.standard_normal(m)
    A = sp.csr_matrix(A)
    A.eliminate_zeros()
    if unsorted:
        for i in range(n):
            s, e = A.indptr[i], A.indptr[i + 1]
            q = rng.permutation(e - s)
            A.indices[s:e] = A.indices[s:e][q]
            A.data[s:e] = A.data[s:e][q]
        A.has_sorted_indices = False
    return A


# ---------------------------------------------------------------------------- kernels --
def test_spmv_and_residual_irregular_rows():
    rng = np.random.default_rng(1)
    for n, m, dens, long_row in ((1, 1, 1.0, None), (37, 53, 0.2, None), (700, 700, 0.01, None),
                                 (3000, 5000, 0.002, 17), (5000, 5000, 0.0005, 4999)):
        A = random_csr(n, m, rng, density=dens, long_row=long_row)
        x = rng.standard_normal(m)
        y = _hip.spmv(A, x)
        ip, ix, dv = orc._csr(A)
        want = np.empty(n)
        orc._clib().oracle_spmv(n, ip, ix, dv, np.ascontiguousarray(x), want)
        np.testing.assert_allclose(y, want, rtol=1e-11, atol=1e-11)
        if n == m:
            b = rng.standard_normal(n)
            r, norm = _hip.residual(A, b, x, want_norm=True)
            want_r = np.empty(n)
            orc._clib().oracle_residual(n, ip, ix, dv, b, np.ascontiguousarray(x), want_r)
            np.testing.assert_allclose(r, want_r, rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(norm, np.linalg.norm(want_r), rtol=1e-12)


def test_spmv_stencils_and_reference_helpers():
    rng = np.random.default_rng(2)
    for shape in ((4096,), (96, 70), (24, 20, 28), (64, 64, 64)):
        A = operators.stencil_poisson(shape)
        x = rng.random(A.shape[0])
        np.testing.assert_allclose(_hip.spmv(A, x), A @ x, **OP)
        b = rng.random(A.shape[0])
        N = A.shape[0]
        got = tools.getresidual(b, A, x, N)                       # (N, 1) like the reference
        assert got.shape == (N, 1)
        np.testing.assert_allclose(got, orc.get_residual(b, A, x, N), **OP)
        col = tools.flexibleMmult(A, x.reshape(N, 1))
        assert col.shape == (N, 1)
        np.testing.assert_allclose(col.ravel(), A @ x, **OP)


def test_empty_and_degenerate_inputs():
    A = sp.csr_matrix((5, 5))
    np.testing.assert_array_equal(_hip.spmv(A, np.ones(5)), np.zeros(5))
    D = sp.identity(3, format="csr") * 2.0
    x = np.array([1.0, 2.0, 3.0])
    np.testing.assert_array_equal(_hip.residual(D, np.ones(3), x), np.ones(3) - 2 * x)
    with pytest.raises(_hip.HipError) as e:                      # no diagonal -> reference divides by 0
        _hip.gauss_seidel(sp.csr_matrix(np.array([[0.0, 1.0], [1.0, 2.0]])), np.ones(2), np.zeros(2))
    assert e.value.code == _hip.ERR_NO_DIAGONAL
    with pytest.raises(_hip.HipError) as e:
        _hip.direct_solve(sp.csr_matrix(np.array([[1.0, 2.0], [2.0, 4.0]])), np.ones(2))
    assert e.value.code == _hip.ERR_SINGULAR


# --------------------------------------------------------------------------- smoother --
def test_gauss_seidel_reference_fixtures(golden):
    """solvers.gaussSeidel / smooth / smoothToThreshold against the real reference's output."""
    d = golden("g7_stop_rules_misc")
    A1 = operators.poisson(64, sparse=True)
    x = d["gs_x0"].copy()
    out = solvers.gaussSeidel(A1, d["gs_b"], x)
    assert out is x                                               # in place, same object (Q2)
    np.testing.assert_allclose(x, d["gs_x_it1"], **OP)
    np.testing.assert_allclose(solvers.smooth(A1, d["gs_b"], d["gs_x0"].copy(), 3), d["gs_x_it3"], **OP)
    np.testing.assert_allclose(solvers.smoothToThreshold(A1, d["gs_b"], d["gs_x0"].copy(), 1e-6),
                               d["gs_x_thr"], **OP)
    # (N,1) iterate, dense quirky 2-D operator, threshold stop (tests.py:342-356)
    A2 = operators.poisson((12, 12))
    x2 = np.zeros((144, 1))
    solvers.smoothToThreshold(A2, d["gs_thresh_b"], x2, 1e-4)
    np.testing.assert_allclose(x2.ravel(), d["gs_thresh_x"], rtol=1e-9, atol=1e-11)
    # unsorted columns: sums in stored order
    Au = csr_from(d, "unsorted_A")
    np.testing.assert_allclose(solvers.gaussSeidel(Au, d["unsorted_b"], np.zeros(36), iterations=2),
                               d["unsorted_x"], **OP)
    np.testing.assert_allclose(solvers.coarseSolve(A1, d["gs_b"].reshape(-1, 1)), d["coarse_x"], rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("shape", [(5000,), (70, 90), (20, 24, 28), (48, 48, 48)])
def test_lexicographic_sweep_is_the_reference_iterate(shape):
    """Level-scheduled device sweep == sequential lexicographic sweep (oracle C loop)."""
    rng = np.random.default_rng(3)
    A = operators.stencil_poisson(shape)
    n = A.shape[0]
    b, x0 = rng.random(n), rng.random(n)
    x = x0.copy()
    assert _hip.gauss_seidel(A, b, x, smoother="gs", iterations=2) == 2
    want = orc.gauss_seidel(A, b, x0.copy(), iterations=2)
    np.testing.assert_allclose(x, want, **OP)


def test_lexicographic_sweep_unsymmetric_pattern_and_unsorted_columns():
    rng = np.random.default_rng(4)
    n = 900
    A = random_csr(n, n, rng, density=0.01, empty_rows=False)
    A = sp.csr_matrix(A + sp.diags(np.full(n, 8.0)))              # diagonally dominant, unsymmetric pattern
    A = random_csr_shuffle(A, rng)
    b, x0 = rng.random(n), rng.random(n)
    x = x0.copy()
    _hip.gauss_seidel(A, b, x, smoother="gs", iterations=3)
    np.testing.assert_allclose(x, orc.gauss_seidel(A, b, x0.copy(), iterations=3), rtol=1e-10, atol=1e-12)


def random_csr_shuffle(A, rng):
    A = sp.csr_matrix(A)
    for i in range(A.shape[0]):
        s, e = A.indptr[i], A.indptr[i + 1]
        q = rng.permutation(e - s)
        A.indices[s:e] = A.indices[s:e][q]
        A.data[s:e] = A.data[s:e][q]
    A.has_sorted_indices = False
    return A


def test_colour_sweep_matches_reference_pin(golden):
    """g4: the real reference's sweep on red-first permuted operators."""
    d = golden("g4_redblack_pin")
    for tag in ("p5", "p7"):
        shape = tuple(int(s) for s in d[tag + "_shape"])
        A = operators.stencil_poisson(shape)
        x = d[tag + "_x0"].copy()
        _hip.gauss_seidel(A, d[tag + "_b"], x, smoother="colour", iterations=2)
        np.testing.assert_allclose(x, d[tag + "_x_after2"], **OP)


def test_colour_and_jacobi_against_oracle_27_point():
    rng = np.random.default_rng(5)
    n1 = 12
    T = sp.diags([np.ones(n1 - 1), np.ones(n1), np.ones(n1 - 1)], [-1, 0, 1])
    A = -sp.kron(sp.kron(T, T), T).tocsr()
    A = sp.csr_matrix(A + sp.diags(np.full(A.shape[0], 28.0)))   # 27-point, diagonally dominant
    n = A.shape[0]
    b, x0 = rng.random(n), rng.random(n)
    colour = orc.greedy_colouring(A)
    assert colour.max() == 7                                     # 2x2x2 colouring
    x = x0.copy()
    _hip.gauss_seidel(A, b, x, smoother="colour", iterations=2)
    np.testing.assert_allclose(x, orc.gs_ordered(A, b, x0.copy(), orc.colour_order(colour), 2), **OP)
    x = x0.copy()
    _hip.gauss_seidel(A, b, x, smoother="jacobi", omega=0.7, iterations=3)
    np.testing.assert_allclose(x, orc.jacobi(A, b, x0.copy(), 3, 0.7), **OP)     # UNPINNED by the reference


def test_smooth_to_threshold_sweep_count():
    A = operators.stencil_poisson((40, 40))
    rng = np.random.default_rng(6)
    b = rng.random(1600)
    x = np.zeros(1600)
    sweeps = _hip.gauss_seidel(A, b, x, smoother="gs", threshold=1e-3)
    want = np.zeros(1600)
    k = 0
    while np.linalg.norm(b - A @ want) >= 1e-3:                  # solvers.py:43-54
        orc.gauss_seidel(A, b, want, iterations=1)
        k += 1
    assert sweeps == k
    np.testing.assert_allclose(x, want, rtol=1e-9, atol=1e-12)
    # already converged: zero sweeps (the test runs before the first sweep)
    assert _hip.gauss_seidel(A, b, x, smoother="gs", threshold=1.0) == 0


# ------------------------------------------------------------------------------ setup --
def test_restriction_matches_reference(golden):
    d = golden("g5_restriction")
    for tag in d["cases"]:
        tag = str(tag)
        shape = tuple(int(s) for s in tag[1:].split("x"))
        if tag + "_error" in d.files:
            with pytest.raises(IndexError):
                operators.restriction(shape)
            continue
        G = csr_from(d, tag)
        R = operators.restriction(shape)
        assert R.shape == G.shape and R.nnz == G.nnz
        assert abs(R - G).max() == 0
    for row in d["restrictionList_cases"]:
        dim, coarsest, minsize, nR, last0, last1 = (int(v) for v in row[:6])
        shape = tuple(int(v) for v in row[6:6 + dim])
        R = operators.restrictionList(shape, coarsest, minsize)
        assert len(R) == nR and R[-1].shape == (last0, last1)
    assert isinstance(operators.restriction((4, 4), dense=True), np.ndarray)    # tests.py:538-542


def test_galerkin_product_matches_reference(golden):
    d = golden("g3_poisson3d_16")
    A0 = operators.stencil_poisson((16, 16, 16))
    R = operators.restrictionList((16, 16, 16), 1, 8)
    A = operators.coeffecientList(A0, R)
    assert len(A) == int(d["n_levels"])
    for l, M in enumerate(A):
        G = csr_from(d, "A%d" % l)
        assert M.shape == G.shape
        assert abs(sp.csr_matrix(M) - G).max() == 0               # exact: power-of-two weights
    assert abs(A[1] - operators.stencil_poisson((8, 8, 8)) / 16.0).max() == 0


def test_spgemm_bitwise_against_scipy_order():
    """Products are accumulated in SciPy's csr_matmat order without FMA contraction."""
    rng = np.random.default_rng(7)
    X = random_csr(300, 200, rng, density=0.05)
    Y = random_csr(200, 250, rng, density=0.05)
    C = _hip.spgemm(X, Y)
    W = sp.csr_matrix(X @ Y)
    W.sort_indices()
    assert all(np.all(np.diff(C.indices[C.indptr[i]:C.indptr[i + 1]]) > 0) for i in range(C.shape[0]))
    assert np.array_equal(C.indptr, W.indptr) and np.array_equal(C.indices, W.indices)
    np.testing.assert_allclose(C.data, W.data, rtol=1e-14, atol=0)
    Rr = random_csr(60, 300, rng, density=0.05, empty_rows=False)
    Aq = random_csr(300, 300, rng, density=0.03)
    got = _hip.rap(Rr, Aq)
    want = sp.csr_matrix((Rr @ Aq) @ Rr.T)
    assert abs(got - want).max() < 1e-13
    both = tools.flexibleMmult(X.toarray(), Y.toarray())          # dense x dense still on the device
    assert isinstance(both, np.ndarray)
    np.testing.assert_allclose(both, X.toarray() @ Y.toarray(), rtol=1e-12, atol=1e-13)


def _identical_csr(a, b):
    return (a.shape == b.shape and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
            and np.array_equal(a.data, b.data))


def test_fused_galerkin_kernel_has_the_bits_of_the_two_products(monkeypatch):
    """omg_rap on an aggregation restriction (every column of R owned by one row) runs the fused
    one-wave-per-coarse-row kernel; OMG_RAP_FUSED=0 takes the two Gustavson products.  Same structure, same
    bits: 7-point, 27-point variable-coefficient (216 products per coarse row), unsymmetric random values on a
    stencil pattern, irregular aggregates with cancellation, a second level; operands that do not qualify
    (a column owned twice, unsorted rows) fall back by themselves."""
    rng = np.random.default_rng(23)
    cases = []
    for shape in ((16, 16, 16), (12, 12, 12)):
        cases.append((operators.restriction(shape), operators.stencil_poisson(shape)))
    shape = (8, 8, 8)
    A27 = operators.stencil27_variable(shape)
    cases.append((operators.restriction(shape), A27))
    U = A27.copy()
    U.data = rng.standard_normal(U.nnz)                                   # unsymmetric values, same pattern
    cases.append((operators.restriction(shape), U))
    n = 3000                                                              # irregular aggregates of 1..7 unknowns
    cuts = np.unique(np.concatenate([[0, n], rng.integers(1, n, 900)]))
    agg = np.repeat(np.arange(cuts.size - 1), np.diff(cuts))
    Rirr = sp.csr_matrix((rng.choice([0.5, 1.0, -1.0, 0.25], n), (agg, np.arange(n))), shape=(cuts.size - 1, n))
    Airr = random_csr(n, n, rng, density=0.004, unsorted=False)
    Airr.data = rng.choice([1.0, -1.0, 2.0, 0.5], Airr.nnz)               # exact cancellations happen
    Airr.sort_indices()
    cases.append((Rirr, Airr))
    for Rl, Al in cases:
        Rl, Al = sp.csr_matrix(Rl), sp.csr_matrix(Al)
        monkeypatch.setenv("OMG_RAP_FUSED", "1")
        fused = _hip.rap(Rl, Al)
        monkeypatch.setenv("OMG_RAP_FUSED", "0")
        two = _hip.rap(Rl, Al)
        assert _identical_csr(fused, two), (Rl.shape, abs(fused - two).max())
        assert all(np.all(np.diff(fused.indices[fused.indptr[i]:fused.indptr[i + 1]]) > 0) for i in range(fused.shape[0]))
        want = sp.csr_matrix((Rl @ Al) @ Rl.T)
        assert abs(fused - want).max() <= 1e-13 * max(abs(want).max(), 1.0)
        # the next level of the hierarchy: a product of a product
        if Rl.shape[0] % 8 == 0 and Rl.shape[0] >= 64:
            m = Rl.shape[0]
            R2 = sp.csr_matrix((np.full(m, 0.5), (np.arange(m) // 2, np.arange(m))), shape=(m // 2, m))
            monkeypatch.setenv("OMG_RAP_FUSED", "1")
            f2 = _hip.rap(R2, fused)
            monkeypatch.setenv("OMG_RAP_FUSED", "0")
            assert _identical_csr(f2, _hip.rap(R2, fused))
    monkeypatch.setenv("OMG_RAP_FUSED", "1")
    # not an aggregation: a column in two rows -> the two products, silently
    Rdup = sp.csr_matrix(np.array([[1.0, 0.5, 0.0, 0.0], [0.0, 0.5, 1.0, 1.0]]))
    T = sp.diags([-np.ones(3), 2.0 * np.ones(4), -np.ones(3)], [-1, 0, 1], format="csr")
    assert abs(_hip.rap(Rdup, T) - sp.csr_matrix((Rdup @ T) @ Rdup.T)).max() < 1e-14
    # rows of A out of order -> the two products (SciPy's order follows the STORED order)
    Au = sp.csr_matrix(operators.stencil_poisson((4, 4, 4)))
    Au.indices[0:2] = Au.indices[0:2][::-1].copy()
    Au.data[0:2] = Au.data[0:2][::-1].copy()
    Rl = sp.csr_matrix(operators.restriction((4, 4, 4)))
    got = _hip.rap(Rl, Au)
    monkeypatch.setenv("OMG_RAP_FUSED", "0")
    assert _identical_csr(got, _hip.rap(Rl, Au))


def test_spgemm_drops_cancelled_entries_and_dense_operands_give_dense_results():
    """SciPy's csr_matmat stores an accumulated entry only when it is != 0, so sums
    that cancel exactly must not appear in the device product either (structure == SciPy's);
    and `sparse * dense-2-D` is a dense array in SciPy (openmg/tools.py:26), so
    flexibleMmult returns an ndarray unless BOTH operands are sparse."""
    X = sp.csr_matrix(np.array([[1.0, -1.0, 0.0], [2.0, 0.0, 1.0], [0.0, 0.0, 0.0], [1.0, 1.0, -2.0]]))
    Y = sp.csr_matrix(np.array([[1.0, 3.0], [1.0, 0.0], [1.0, 1.5]]))
    C = _hip.spgemm(X, Y)
    W = sp.csr_matrix(X @ Y)
    W.sort_indices()
    assert W.nnz == 3                                             # (0,0), (3,0) and (3,1) cancel: 1-1, 1+1-2, 3-3
    assert np.array_equal(C.indptr, W.indptr) and np.array_equal(C.indices, W.indices)
    assert np.array_equal(C.data, W.data) and not (C.data == 0).any()
    # Galerkin product with cancellation: a constant vector is in the null space of the graph Laplacian
    n = 64
    L = sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="lil")
    L[0, 0] = L[n - 1, n - 1] = 1.0
    Rr = sp.csr_matrix(np.ones((1, n)))
    got = _hip.rap(Rr, sp.csr_matrix(L))
    assert got.nnz == 0 and sp.csr_matrix((Rr @ sp.csr_matrix(L)) @ Rr.T).nnz == 0
    rng = np.random.default_rng(17)
    S = random_csr(40, 30, rng, density=0.2)
    D = rng.standard_normal((30, 5))
    out = tools.flexibleMmult(S, D)
    assert isinstance(out, np.ndarray) and out.shape == (40, 5)
    np.testing.assert_allclose(out, S @ D, rtol=1e-12, atol=1e-13)
    out = tools.flexibleMmult(D.T, sp.csr_matrix(S.T))
    assert isinstance(out, np.ndarray) and out.shape == (5, 40)
    assert sp.issparse(tools.flexibleMmult(S, sp.csr_matrix(S.T)))


# -------------------------------------------------------------------- level operations --
def test_level_operations_against_oracle(golden):
    d = golden("g3_poisson3d_16")
    A = [csr_from(d, "A%d" % l) for l in range(3)]               # SciPy's unsorted RAP output
    R = [csr_from(d, "R%d" % l) for l in range(2)]
    rng = np.random.default_rng(8)
    for smoother in ("gs", "colour", "jacobi"):
        with _hip.Hierarchy(A, R, smoother=smoother, omega=0.8) as h:
            assert h.n_levels == 3
            for l in (0, 1):
                n = A[l].shape[0]
                b, x0 = rng.random(n), rng.random(n)
                x = x0.copy()
                h.smooth(l, b, x, 2)
                if smoother == "gs":
                    want = orc.gauss_seidel(A[l], b, x0.copy(), iterations=2)
                elif smoother == "colour":
                    want = orc.gs_ordered(A[l], b, x0.copy(), orc.colour_order(orc.greedy_colouring(A[l])), 2)
                else:
                    want = orc.jacobi(A[l], b, x0.copy(), 2, 0.8)
                np.testing.assert_allclose(x, want, **OP)
                r, norm = h.residual(l, b, x0, want_norm=True)
                wr = orc.get_residual(b, A[l], x0, n).ravel()
                np.testing.assert_allclose(r, wr, **OP)
                np.testing.assert_allclose(norm, np.linalg.norm(wr), rtol=1e-12)
                np.testing.assert_allclose(h.restrict(l, r), R[l] @ wr, **OP)
                e = rng.random(R[l].shape[0])
                np.testing.assert_allclose(h.prolong_add(l, e, x0), x0 + R[l].T @ e, **OP)
            bc = rng.random(A[2].shape[0])
            np.testing.assert_allclose(h.coarse_solve(bc), orc.coarse_solve(A[2], bc.reshape(-1, 1)),
                                       rtol=1e-10, atol=1e-12)


# ------------------------------------------------------------------- whole-path parity --
def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def test_simple_demo_known_answer(golden):
    """openmg_usage_demo.py:27-67 through openmg_amd.mgSolve."""
    d = golden("g1_simple_demo")
    N = 100
    u_true = np.array([np.sin(x / 10.0) for x in np.linspace(0, 20, N)])
    A = openmg_amd.operators.poisson(N, sparse=True)
    b = openmg_amd.tools.flexibleMmult(A, u_true)
    np.testing.assert_allclose(b, d["b"], rtol=1e-14, atol=1e-15)
    for gl, dense in ((2, True), (2, False), (3, True), (3, False)):
        tag = "gl%d_%s" % (gl, "dense" if dense else "sparse")
        params = {"problemShape": (N,), "gridLevels": gl, "cycles": 10, "iterations": 2,
                  "verbose": False, "dense": dense, "threshold": 1e-2, "giveInfo": True}
        u, info = openmg_amd.mgSolve(A, b, params)
        assert info["cycle"] == int(d[tag + "_cycle"]) == 4
        assert rel(info["norm"], d[tag + "_norms"][-1]) < NORM_RTOL
        np.testing.assert_allclose(u, d[tag + "_u"], **CYC)
    p2 = {"problemShape": (N,), "gridLevels": 2, "cycles": 10, "threshold": 1e-2, "giveInfo": True}
    _, info2 = openmg_amd.mgSolve(A, b, p2)
    assert abs(info2["norm"] - 0.003405) < 5e-7                  # openmg_usage_demo.py:63-66


@pytest.mark.parametrize("post", [0, 1])
def test_config1_poisson1d_4096_cycles(golden, post):
    """BASELINE config 1: 1-D N=4096, 3 grids — iterates and norms after 1/2/5 cycles."""
    d = golden("g2_poisson1d_4096")
    A = operators.poisson(4096, sparse=True)
    R = operators.restrictionList((4096,), 1, 8)
    Al = operators.coeffecientList(A, R)
    p = {"preIterations": 1, "postIterations": post, "coarsestLevel": len(R)}
    x = None
    norms = []
    for c in range(1, 6):
        x, info = openmg_amd.mgCycle(Al, d["b"], 0, R, p, initial=x)
        norms.append(info["norm"])
        if c in (1, 2, 5):
            np.testing.assert_allclose(x, d["v1%d_x_c%d" % (post, c)], **CYC)
    for got, want in zip(norms, d["v1%d_norms" % post]):
        assert rel(got, want) < NORM_RTOL
    openmg_amd.clear_cache()


def test_poisson3d_reference_traces(golden):
    for n in (16, 32):
        d = golden("g3_poisson3d_%d" % n)
        A0 = operators.stencil_poisson((n, n, n))
        for pre, post in (((1, 0), (1, 1)) if n == 16 else ((1, 1),)):
            p = {"problemShape": (n, n, n), "gridLevels": 2, "preIterations": pre, "postIterations": post,
                 "cycles": 3, "threshold": 0, "giveInfo": True, "minSize": 8}
            x, info = openmg_amd.mgSolve(A0, d["b"], p)
            assert info["cycle"] == 3 and len(info["A"]) == int(d["n_levels"])
            np.testing.assert_allclose(x, d["v%d%d_x_c3" % (pre, post)], **CYC)
            assert rel(info["norm"], d["v%d%d_norms" % (pre, post)][2]) < NORM_RTOL


def test_redblack_vcycle_pinned_by_reference(golden):
    d = golden("g4_redblack_pin")
    shape = tuple(int(s) for s in d["vc_shape"])
    p = {"problemShape": shape, "gridLevels": 2, "preIterations": 1, "postIterations": 1,
         "cycles": 3, "threshold": 0, "giveInfo": True, "minSize": 8, "smoother": "colour"}
    x, info = openmg_amd.mgSolve(operators.stencil_poisson(shape), d["vc_b"], p)
    np.testing.assert_allclose(x, d["vc_x_c3"], **CYC)
    assert rel(info["norm"], d["vc_norms"][2]) < NORM_RTOL


def test_stop_rules_dict_mutation_and_errors(golden):
    d = golden("g7_stop_rules_misc")
    A, b = d["stop_A"], d["stop_b"]
    p = {"problemShape": (36,), "gridLevels": 2, "threshold": 8e-3, "giveInfo": True}
    u, info = openmg_amd.mgSolve(A, b, p)                         # tests.py:502-515 (dense A_in)
    assert info["cycle"] == int(d["thresh_cycle"])
    assert rel(info["norm"], float(d["thresh_norm"])) < 1e-9
    np.testing.assert_allclose(u, d["thresh_u"], **CYC)
    assert sorted(p.keys()) == [str(k) for k in d["thresh_keys_after"]]          # Q1
    assert p["coarsestLevel"] == int(d["thresh_coarsestLevel_after"])
    assert openmg_amd.defaults["coarsestLevel"] == 1
    assert np.linalg.norm(A @ u - b) < 8e-3
    p = {"problemShape": (36,), "gridLevels": 2, "cycles": 3, "threshold": 1e-10, "giveInfo": True}
    u, info = openmg_amd.mgSolve(A, b, p)                         # tests.py:517-531
    assert info["cycle"] == 3
    np.testing.assert_allclose(u, d["cyc_u"], **CYC)
    p = {"problemShape": (1024,), "gridLevels": 24, "iterations": 1, "verbose": False,
         "threshold": 4, "giveInfo": True, "minSize": 23}
    soln, info = openmg_amd.mgSolve(operators.poisson((1024,)), d["minsize_b"], p)   # tests.py:558-570
    assert [list(r.shape) for r in info["R"]] == d["minsize_R_shapes"].tolist()
    assert min(info["R"][-1].shape) > 23 and info["cycle"] == int(d["minsize_cycle"])
    np.testing.assert_allclose(soln, d["minsize_soln"], **CYC)
    with pytest.raises(ValueError):                               # tests.py:550-556
        openmg_amd.mgSolve(operators.poisson((64,)), np.ones(64),
                           {"problemShape": (64,), "gridLevels": 2, "cycles": 0, "threshold": 0})
    only_u = openmg_amd.mgSolve(A, b, {"problemShape": (36,), "gridLevels": 2, "cycles": 1})
    assert isinstance(only_u, np.ndarray) and only_u.shape == (36,)              # giveInfo False


def test_test_a_1d_operator_with_3d_shape(golden):
    d = golden("g7_stop_rules_misc")                              # tests.py:58-81
    p = {"coarsestLevel": 3, "problemShape": (12, 12, 12), "gridLevels": 4, "threshold": 8e-3,
         "giveInfo": True}
    u, info = openmg_amd.mgSolve(operators.poisson((1728,)), d["testa_b"], p)
    assert info["cycle"] == int(d["testa_cycle"])
    assert rel(info["norm"], float(d["testa_norm"])) < 1e-8
    np.testing.assert_allclose(u, d["testa_u"], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("smoother", ["gs", "colour", "jacobi"])
def test_config2_2d_five_point_against_oracle(smoother):
    """BASELINE config 2 shape (2-D 5-point, 4 grids) at 128^2 against the CPU oracle."""
    shape = (128, 128)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    p = {"problemShape": shape, "gridLevels": 3, "preIterations": 1, "postIterations": 1,
         "cycles": 4, "threshold": 0, "giveInfo": True, "smoother": smoother, "omega": 0.8}
    x, info = openmg_amd.mgSolve(A0, b, dict(p))
    assert len(info["A"]) == 4
    po = dict(p)
    R = orc.restriction_list(shape, 2, 8)
    Ao = orc.coefficient_list(A0, R)
    po["coarsestLevel"] = len(R)
    sm = orc.make_smoother(smoother, Ao, omega=0.8)
    xo = None
    for _ in range(4):
        xo, inf = orc.mg_cycle(Ao, b, 0, R, po, initial=xo, smoother=sm)
    np.testing.assert_allclose(x, xo, **CYC)
    assert rel(info["norm"], inf["norm"]) < NORM_RTOL


def test_hipgraph_replay_is_identical():
    shape = (32, 32, 32)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(1).random(A0.shape[0])
    R = operators.restrictionList(shape, 2, 8)
    A = operators.coeffecientList(A0, R)
    out = []
    for graph in (False, True):
        with _hip.Hierarchy(A, R, smoother="colour") as h:
            h.use_graph(graph)
            h.resident_load(b)
            norms = [h.resident_cycle(1, 1) for _ in range(4)]
            out.append((norms, h.resident_fetch()))
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1])


# -------------------------------------------------- full-size, size-independent checks --
@pytest.mark.parametrize("n,levels", [(128, 3), (256, 4)])
def test_full_size_properties(n, levels):
    """BASELINE config 3 (256^3, 5 grids, red-black GS) is far beyond what the sequential
    oracle finishes in seconds, so check properties that do not need it:
      * the device-reported norm equals ||b - A x|| recomputed on the host with SciPy;
      * exact linearity: a V-cycle is linear in b and scaling by 2 is exact in binary
        floating point, so cycle(2 b) == 2 cycle(b) BITWISE;
      * the norm falls monotonically at the known rate of this method;
      * the Galerkin operator of the constant stencil is lap3(n/2)/16 exactly."""
    shape = (n, n, n)
    A0 = operators.stencil_poisson(shape)
    N = A0.shape[0]
    u_true = np.random.default_rng(12345).random(N)
    b = A0 @ u_true
    R = operators.restrictionList(shape, levels - 1, 8)
    A = operators.coeffecientList(A0, R)
    assert len(A) == levels + 1
    assert A[1].nnz == 7 * (n // 2) ** 3 - 6 * (n // 2) ** 2
    probe = np.random.default_rng(3).random(A[1].shape[0])
    want = (operators.stencil_poisson((n // 2,) * 3) / 16.0) @ probe
    np.testing.assert_allclose(_hip.spmv(A[1], probe), want, **OP)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        h.resident_load(b)
        norms = [h.resident_cycle(1, 1) for _ in range(3)]
        x = h.resident_fetch()
        assert rel(norms[-1], np.linalg.norm(b - A0 @ x)) < 1e-10
        assert norms[0] > norms[1] > norms[2] and norms[2] / norms[1] < 0.8
        h.resident_load(2.0 * b)
        norms2 = [h.resident_cycle(1, 1) for _ in range(3)]
        x2 = h.resident_fetch()
        assert np.array_equal(x2, 2.0 * x)
        assert norms2 == [2.0 * v for v in norms]


# ------------------------------------------------ fused last-set sweep + residual / norm --
@pytest.mark.parametrize("smoother", ["colour", "gs"])
@pytest.mark.parametrize("pre,post", [(1, 1), (2, 1), (1, 0), (0, 2)])
def test_fused_last_set_is_bit_identical(monkeypatch, smoother, pre, post):
    """The smoother's last set launch also emits that set's residual / norm share
    (ROW_GS_RES / ROW_GS_NORM); OMG_NO_FUSE=1 runs the plain passes.  Same bits either way,
    for 7-entry rows (register path) and 27-entry rows (LDS re-walk path)."""
    n1 = 16
    T = sp.diags([np.ones(n1 - 1), np.ones(n1), np.ones(n1 - 1)], [-1, 0, 1])
    A27 = sp.csr_matrix(-sp.kron(sp.kron(T, T), T) + sp.diags(np.full(n1 ** 3, 28.0)))
    for A0, shape in ((operators.stencil_poisson((32, 32, 32)), (32, 32, 32)), (A27, (n1,) * 3)):
        R = operators.restrictionList(shape, 1, 8)
        A = operators.coeffecientList(A0, R)
        b = A0 @ np.random.default_rng(21).random(A0.shape[0])
        out = []
        for no_fuse in ("1", "0"):
            monkeypatch.setenv("OMG_NO_FUSE", no_fuse)
            with _hip.Hierarchy(A, R, smoother=smoother) as h:
                if no_fuse == "0" and smoother == "colour":
                    assert h.level_fused(0)
                h.resident_load(b)
                norms = [h.resident_cycle(pre, post) for _ in range(3)]
                out.append((norms, h.resident_fetch()))
        assert out[0][0] == out[1][0]
        assert np.array_equal(out[0][1], out[1][1])


# ------------------------------------------------------ irregular (non-stencil) hierarchies --
def irregular_problem(rng, n=2600, long_row=True):
    """Symmetric, strictly diagonally dominant, irregular sparsity, unsorted columns; one row
    (and column) is denser than the LDS block budget so the whole-workgroup row path runs."""
    S = sp.random(n, n, density=0.004, random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csr")
    S = S + S.T
    if long_row:
        dense = sp.csr_matrix((rng.standard_normal(n) * 0.01, (np.full(n, 7), np.arange(n))), shape=(n, n))
        S = S + dense + dense.T
    S = sp.csr_matrix(S)
    S.setdiag(0)
    S.eliminate_zeros()
    d = np.asarray(abs(S).sum(axis=1)).ravel() + 1.0
    A = sp.csr_matrix(S + sp.diags(d))
    A = random_csr_shuffle(A, rng)
    # aggregation of consecutive pairs / triples with unequal weights
    sizes = rng.integers(2, 4, size=n)
    ends = np.cumsum(sizes)
    ends = ends[ends < n]
    starts = np.concatenate([[0], ends])
    ends = np.concatenate([ends, [n]])
    rows = np.repeat(np.arange(starts.size), ends - starts)
    w = rng.random(n) + 0.5
    R = sp.csr_matrix((w, (rows, np.arange(n))), shape=(starts.size, n))
    return A, R


@pytest.mark.parametrize("smoother", ["gs", "colour", "jacobi"])
def test_irregular_two_level_cycle_against_oracle(smoother):
    rng = np.random.default_rng(31)
    A0, R0 = irregular_problem(rng)
    assert np.diff(A0.indptr).max() > 2048                       # long-row path is exercised
    A1 = _hip.rap(R0, A0)
    np.testing.assert_allclose(A1.toarray(), (R0 @ A0 @ R0.T).toarray(), rtol=1e-12, atol=1e-13)
    A, R = [A0, A1], [R0]
    b = rng.standard_normal(A0.shape[0])
    p = {"preIterations": 2, "postIterations": 1, "coarsestLevel": 1, "smoother": smoother, "omega": 0.6}
    sm = orc.make_smoother(smoother, A, omega=0.6)
    x, xo = None, None
    for _ in range(3):
        x, info = openmg_amd.mgCycle(A, b, 0, R, p, initial=x)
        xo, inf = orc.mg_cycle(A, b, 0, R, p, initial=xo, smoother=sm)
        assert rel(info["norm"], inf["norm"]) < 1e-9
    np.testing.assert_allclose(x, xo, rtol=1e-8, atol=1e-11)
    openmg_amd.clear_cache()


def test_omg_solve_entry_point_and_mgcycle_cache():
    """omg_solve (the C loop of mgSolve) and the device-hierarchy cache of repeated mgCycle calls."""
    shape = (16, 16, 16)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(2).random(4096)
    R = operators.restrictionList(shape, 1, 8)
    A = operators.coeffecientList(A0, R)
    with _hip.Hierarchy(A, R, smoother="gs") as h:
        x = np.zeros(4096)
        cycles, norm = h.solve(b, x, 1, 1, 0, 1e-3)               # threshold stop
        assert norm < 1e-3 and cycles > 1
        x2 = np.zeros(4096)
        c2, n2 = h.solve(b, x2, 1, 1, cycles, 0.0)                # cycle-count stop, same work
        assert c2 == cycles and n2 == norm and np.array_equal(x, x2)
        with pytest.raises(_hip.HipError):
            h.solve(b, x2, 1, 1, 0, 0.0)                          # both stop rules off
    p = {"preIterations": 1, "postIterations": 1, "coarsestLevel": 2}
    u = None
    for _ in range(cycles):
        u, info = openmg_amd.mg_cycle(A, b, 0, R, p, initial=u)
    assert len(openmg_amd._cache) == 1                            # one upload for all calls
    np.testing.assert_allclose(u, x, rtol=1e-12, atol=1e-14)
    assert rel(info["norm"], norm) < 1e-12
    # entering below the top: mgCycle(A, b_c, 1, R, ...) works on the sub-hierarchy
    bc = np.random.default_rng(4).random(A[1].shape[0])
    uc, ic = openmg_amd.mgCycle(A, bc, 1, R, p)
    po = dict(p)
    want, iw = orc.mg_cycle(A, bc, 1, R, po)
    np.testing.assert_allclose(uc, want, rtol=1e-10, atol=1e-12)
    assert rel(ic["norm"], iw["norm"]) < 1e-9
    top, it = openmg_amd.mgCycle(A, np.ones(A[2].shape[0]), 2, R, p)   # at the coarsest: direct solve, norm 0
    assert it["norm"] == 0
    np.testing.assert_allclose(A[2] @ top, np.ones(A[2].shape[0]), rtol=1e-10)
    openmg_amd.clear_cache()


def test_converged_regime_tracks_the_oracle():
    """Far beyond the 3-5 cycles of the fixtures: 40 V(1,1) cycles on 16^3 drive the residual
    down ~8 orders of magnitude; the GPU norm trace must follow the CPU oracle's all the way
    (relative difference grows only with the rounding noise floor ~1e-16 * ||b|| / ||r||)."""
    shape = (16, 16, 16)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(4096)
    R = operators.restrictionList(shape, 1, 8)
    A = operators.coeffecientList(A0, R)
    Ro = orc.restriction_list(shape, 1, 8)
    Ao = orc.coefficient_list(A0, Ro)
    p = {"preIterations": 1, "postIterations": 1, "coarsestLevel": len(R)}
    xo, norms_o = None, []
    for _ in range(40):
        xo, inf = orc.mg_cycle(Ao, b, 0, Ro, p, initial=xo)
        norms_o.append(inf["norm"])
    with _hip.Hierarchy(A, R, smoother="gs") as h:
        h.resident_load(b)
        norms = [h.resident_cycle(1, 1) for _ in range(40)]
        x = h.resident_fetch()
    assert norms_o[-1] < 1e-7 * norms_o[0]
    bnorm = np.linalg.norm(b)
    for k, (g, o) in enumerate(zip(norms, norms_o)):
        assert abs(g - o) <= 1e-10 * o + 1e-14 * bnorm, (k, g, o)
    np.testing.assert_allclose(x, xo, rtol=1e-10, atol=1e-12)


# ----------------------------------------------------------- device format (lossless recoding) --
def _variable_coefficient_7pt(shape, rng):
    """-div(k grad u) on a box, k random per cell face: the 7-point sparsity with no two equal values."""
    n = int(np.prod(shape))
    idx = np.arange(n).reshape(shape)
    rows, cols, vals = [], [], []
    diag = np.zeros(n)
    for ax in range(3):
        lo = np.take(idx, np.arange(shape[ax] - 1), axis=ax).ravel()
        hi = np.take(idx, np.arange(1, shape[ax]), axis=ax).ravel()
        k = rng.random(lo.size) + 0.5
        rows += [lo, hi]
        cols += [hi, lo]
        vals += [-k, -k]
        np.add.at(diag, lo, k)
        np.add.at(diag, hi, k)
    rows.append(np.arange(n)); cols.append(np.arange(n)); vals.append(diag + 0.1)
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))


@pytest.mark.parametrize("case", ["poisson7", "poisson7_lex", "galerkin_golden", "variable7", "stencil27", "irregular",
                                  "poisson7_f32", "jacobi2d"])
def test_device_format_modes_are_bit_identical(monkeypatch, golden, case):
    """The device format (csrc/common.h: row patterns, per-entry column / value dictionaries,
    plain CSR; OMG_COMPRESS bit mask) and the kernel that walks it (OMG_PATTERN_KERNEL) are
    speed choices only: every combination must give the same bits as plain CSR."""
    rng = np.random.default_rng(77)
    dtype, smoother = "float64", "colour"
    monkeypatch.setenv("OMG_STENCIL27", "0")       # this test is about the codings of the operator AS STORED (not padded to 27 slots)
    if case in ("poisson7", "poisson7_lex", "poisson7_f32"):
        shape = (24, 20, 28)
        A0 = operators.stencil_poisson(shape)
        R = operators.restrictionList(shape, 1, 4)
        A = operators.coeffecientList(A0, R)
        smoother = "gs" if case == "poisson7_lex" else "colour"
        dtype = "float32" if case == "poisson7_f32" else "float64"
    elif case == "galerkin_golden":
        d = golden("g3_poisson3d_16")
        A = [csr_from(d, "A%d" % l) for l in range(3)]
        R = [csr_from(d, "R%d" % l) for l in range(2)]
    elif case == "variable7":
        shape = (20, 16, 24)
        A0 = _variable_coefficient_7pt(shape, rng)
        R = operators.restrictionList(shape, 0, 4)
        A = operators.coeffecientList(A0, R)
    elif case == "stencil27":
        n1 = 16
        T = sp.diags([np.ones(n1 - 1), np.ones(n1), np.ones(n1 - 1)], [-1, 0, 1])
        A0 = sp.csr_matrix(-sp.kron(sp.kron(T, T), T) + sp.diags(np.full(n1 ** 3, 28.0)))
        R = operators.restrictionList((n1,) * 3, 0, 4)
        A = operators.coeffecientList(A0, R)
    elif case == "jacobi2d":
        shape = (64, 64)
        A0 = operators.stencil_poisson(shape)
        R = operators.restrictionList(shape, 1, 4)
        A = operators.coeffecientList(A0, R)
        smoother = "jacobi"
    else:
        A0, R0 = irregular_problem(rng)
        A, R = [A0, sp.csr_matrix(R0 @ A0 @ R0.T)], [R0]
    b = A[0] @ rng.random(A[0].shape[0])
    runs = {}
    # pk: "1" rows_pattern_kernel where the dictionaries fit a wave, "0" rows_kernel walking them through LDS
    # ur: rows per thread of the union walk (rows_union_kernel, blocks of ur x 256 rows); "0" = that kernel off
    for mode, pk, ur in (("0", "1", "4"), ("1", "1", "4"), ("2", "1", "4"), ("3", "1", "4"), ("4", "1", "0"), ("4", "0", "4"),
                         ("7", "1", "0"), ("7", "0", "4"), ("15", "1", "0"), ("15", "0", "4"), ("12", "1", "0"),
                         ("4", "1", "1"), ("15", "1", "1"), ("15", "1", "2"), ("15", "1", "4"), ("7", "1", "4")):
        monkeypatch.setenv("OMG_COMPRESS", mode)
        monkeypatch.setenv("OMG_PATTERN_KERNEL", pk)
        monkeypatch.setenv("OMG_UNION_KERNEL", "0" if ur == "0" else "1")
        monkeypatch.setenv("OMG_UNION_ROWS", ur if ur != "0" else "1")
        with _hip.Hierarchy(A, R, smoother=smoother, dtype=dtype) as h:
            info = h.format_info(0)
            h.resident_load(b)
            norms = [h.resident_cycle(2, 1) for _ in range(3)]
            x = h.resident_fetch()
            # the single-operation entry points go through the same operators
            r, nr = h.residual(0, b, x, want_norm=True)
            rc = h.restrict(0, r)
            xp = h.prolong_add(0, rc, x)
        runs[(mode, pk, ur)] = (norms, x, r, nr, rc, xp, info)
    base = runs[("0", "1", "4")]
    w = 4 if dtype == "float32" else 8
    assert base[6]["pattern_rows"] == 0 and base[6]["coldict_nnz"] == 0 and base[6]["valdict_nnz"] == 0
    assert base[6]["format_bytes"] == base[6]["csr_bytes"] + 32 * base[6]["blocks"]
    assert base[6]["csr_bytes"] == A[0].nnz * (4 + w) + 4 * A[0].shape[0]
    for key, run in runs.items():
        # vectors: to the bit.  Norms are sums over row blocks, and a coding may come with another
        # block partition (256-row blocks for long rows in the pattern kernel): same terms, other grouping.
        np.testing.assert_allclose(run[0], base[0], rtol=1e-13, err_msg=str(key))
        np.testing.assert_allclose(run[3], base[3], rtol=1e-13, err_msg=str(key))
        for got, want in zip((run[1], run[2], run[4], run[5]), (base[1], base[2], base[4], base[5])):
            assert np.array_equal(got, want), key
    full = runs[("7", "1", "0")][6]
    if case in ("poisson7", "poisson7_f32", "jacobi2d"):
        # constant stencils, short rows: every block carries a union -> blocks of 4 x 256 rows
        wide, std = runs[("15", "1", "4")][6], runs[("15", "1", "1")][6]
        assert wide["pattern_rows"] == wide["rows"] and wide["blocks"] < std["blocks"]
    if case == "variable7":                                      # offsets repeat, values do not: offset patterns + ELL values
        ell = runs[("15", "1", "0")][6]
        assert ell["pattern_rows"] == ell["rows"] and ell["format_bytes"] < 0.75 * ell["csr_bytes"]
    if case in ("poisson7", "poisson7_f32", "stencil27", "jacobi2d", "galerkin_golden"):
        assert full["pattern_rows"] == full["rows"]              # constant stencils: a byte per row
        assert full["format_bytes"] < 0.2 * full["csr_bytes"]
    if case == "variable7":
        assert full["pattern_rows"] == 0 and full["valdict_nnz"] == 0
        assert full["coldict_nnz"] == full["nnz"]                # offsets repeat, values do not
    if case == "irregular":
        assert full["pattern_rows"] == 0 and full["coldict_nnz"] < full["nnz"]


def test_prolongation_as_a_scatter_over_restriction_rows_is_bit_identical(monkeypatch):
    """ROW_SCATTER: x += R^T e applied from R's row patterns (aggregation: every fine unknown
    has one parent) instead of from the explicit transpose; OMG_PROLONG_SCATTER=0 switches it
    off.  Same two roundings per entry -> same bits, in both precisions; an R whose rows share
    columns (linear interpolation) keeps the transpose."""
    shape = (128, 128, 128)       # long grid lines: R's blocks then hold a handful of row patterns
    A0 = operators.stencil_poisson(shape)
    R = operators.restrictionList(shape, 2, 4)
    A = operators.coeffecientList(A0, R)
    rng = np.random.default_rng(90)
    b = A0 @ rng.random(A0.shape[0])
    e0, x0 = rng.random(R[0].shape[0]), rng.random(A0.shape[0])
    for dtype in ("float64", "float32"):
        out = {}
        for sw in ("0", "1"):
            monkeypatch.setenv("OMG_PROLONG_SCATTER", sw)
            with _hip.Hierarchy(A, R, smoother="colour", dtype=dtype) as h:
                assert h.level_flags(0)["scatter_prolong"] == (sw == "1")
                p = h.prolong_add(0, e0, x0)
                h.resident_load(b)
                norms = [h.resident_cycle(1, 1) for _ in range(3)]
                out[sw] = (p, norms, h.resident_fetch())
        assert np.array_equal(out["0"][0], out["1"][0])
        assert out["0"][1] == out["1"][1] and np.array_equal(out["0"][2], out["1"][2])
    monkeypatch.setenv("OMG_PROLONG_SCATTER", "1")
    # overlapping aggregates: columns shared between rows -> no scatter
    n = 600
    Rl = sp.diags([0.25 * np.ones(n - 1), 0.5 * np.ones(n), 0.25 * np.ones(n - 1)], [-1, 0, 1], format="csr")[::2]
    Al = sp.diags([-np.ones(n - 1), 2.5 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")
    Ac = sp.csr_matrix(Rl @ Al @ Rl.T)
    with _hip.Hierarchy([Al, Ac], [sp.csr_matrix(Rl)], smoother="colour") as h:
        assert not h.level_flags(0)["scatter_prolong"]
        e, x = rng.random(Rl.shape[0]), rng.random(n)
        np.testing.assert_allclose(h.prolong_add(0, e, x), x + Rl.T @ e, **OP)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_long_rows_pattern_kernel_matches_four_lanes_per_row(monkeypatch, dtype):
    """27-point variable-coefficient operator with long grid lines: by default every set runs the
    LDS-free pattern kernel with 256-row blocks, offset patterns and ELL values (one thread per
    row, four accumulators); with OMG_COMPRESS=0 rows_kernel gives each row four lanes.  The
    association of a row's sum is fixed by its length (csrc/common.h ASSOC_LEN), so both give
    the same bits — sweeps, fused last colour, residual, norm."""
    shape = (4, 6, 256)
    A0 = operators.stencil27_variable(shape)
    n = A0.shape[0]
    pairs = np.arange(n) // 2
    R = sp.csr_matrix((np.full(n, 0.5), (pairs, np.arange(n))), shape=(n // 2, n))
    # keep the coarse level small enough for the dense inverse: aggregate 8 more times
    Rs = [R]
    while Rs[-1].shape[0] > 1500:
        m = Rs[-1].shape[0]
        Rs.append(sp.csr_matrix((np.full(m, 0.5), (np.arange(m) // 2, np.arange(m))), shape=((m + 1) // 2, m)))
    A = [A0]
    for Rl in Rs:
        A.append(sp.csr_matrix(Rl @ A[-1] @ Rl.T))
    rng = np.random.default_rng(55)
    b, x0 = A0 @ rng.random(n), rng.random(n)
    out = {}
    for mode in ("0", "15"):
        monkeypatch.setenv("OMG_COMPRESS", mode)
        with _hip.Hierarchy(A, Rs, smoother="colour", dtype=dtype) as h:
            info = h.format_info(0)
            r, nr = h.residual(0, b, x0, want_norm=True)
            h.resident_load(b, x0)
            norms = [h.resident_cycle(2, 2) for _ in range(2)]
            out[mode] = (r, h.resident_fetch(), norms, nr, info, h.level_sets(0))
    assert out["15"][5] == out["0"][5] >= 8
    wide, plain = out["15"][4], out["0"][4]
    assert wide["pattern_rows"] == wide["rows"] and wide["blocks"] * 4 <= plain["blocks"] + 32    # 256-row blocks vs 64-row
    # the kernel configs[4] runs at full size — offset patterns + block-transposed (ELL) values on EVERY block
    assert wide["ell_blocks"] == wide["blocks"] and wide["ell_nnz"] == wide["nnz"] and plain["ell_blocks"] == 0
    if dtype == "float64":
        # ... directly against the CPU oracle on the same A / R lists (openmg/__init__.py:151-236 with the
        # reference's sweep on the greedy colours): 1e-10 on every norm, rtol 1e-9 on the iterate
        sm = orc.make_smoother("colour", A)
        p = {"preIterations": 2, "postIterations": 2, "coarsestLevel": len(Rs)}
        xo = x0.copy()
        for k in range(2):
            xo, inf = orc.mg_cycle(A, b, 0, Rs, p, initial=xo, smoother=sm)
            assert rel(out["15"][2][k], inf["norm"]) < 1e-10, k
        np.testing.assert_allclose(out["15"][1], xo, rtol=1e-9, atol=1e-12)
    assert np.array_equal(out["15"][0], out["0"][0])
    assert np.array_equal(out["15"][1], out["0"][1])
    np.testing.assert_allclose(out["15"][2], out["0"][2], rtol=1e-13)
    np.testing.assert_allclose(out["15"][3], out["0"][3], rtol=1e-13)
    if dtype == "float64":
        np.testing.assert_allclose(out["0"][0], b - A0 @ x0, rtol=0, atol=1e-12 * (abs(A0).sum(axis=1).max() + np.abs(b).max()))


# ------------------------------------------------------------ union walk on random stencil-like operators --
def _random_stencil_operator(rng, n, n_offsets, n_values, drop=0.15, shuffle=False):
    """n x n operator whose rows repeat: a random set of (column - row) offsets with values drawn
    from a few constants, entries that would leave the matrix dropped (boundary rows = subsequences
    of the interior row), a random subset of further entries dropped in runs (more row patterns),
    a dominant diagonal.  Optionally the stored order inside the rows is shuffled per PATTERN."""
    offs = np.unique(np.concatenate([[0], rng.integers(-n // 3, n // 3, size=n_offsets)]))
    vals = rng.choice(rng.standard_normal(n_values), size=offs.size)
    order = rng.permutation(offs.size) if shuffle else np.arange(offs.size)
    rows, cols, data = [], [], []
    i = np.arange(n)
    for k in order:
        o, v = int(offs[k]), float(vals[k])
        keep = (i + o >= 0) & (i + o < n)
        if o != 0:
            gate = np.repeat(rng.random(n // 97 + 1) >= drop, 97)[:n]          # runs of 97 rows lose this entry together
            keep &= gate
        rows.append(i[keep]); cols.append(i[keep] + o); data.append(np.full(int(keep.sum()), v if o else 0.0))
    rows, cols, data = np.concatenate(rows), np.concatenate(cols), np.concatenate(data)
    # CSR with the stored order of `order` inside every row: stable sort by row only
    perm = np.argsort(rows, kind="stable")
    indptr = np.searchsorted(rows[perm], np.arange(n + 1)).astype(np.int32)
    A = sp.csr_matrix((data[perm], cols[perm].astype(np.int32), indptr), shape=(n, n))
    d = np.asarray(abs(A).sum(axis=1)).ravel() + 1.0
    diag_pos = A.indices == np.repeat(np.arange(n), np.diff(A.indptr))
    A.data[diag_pos] = d
    A.has_sorted_indices = False
    return A


@pytest.mark.parametrize("seed", range(6))
def test_union_walk_on_random_stencil_like_operators_is_bit_identical(monkeypatch, seed):
    """rows_union_kernel against plain CSR on operators made to have many row patterns per block
    that are subsequences of one union (and, seed permitting, a shuffled stored order): SpMV,
    residual + norm, colour / Jacobi sweeps incl. the fused last set, in both precisions."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(3000, 9000))
    A0 = _random_stencil_operator(rng, n, n_offsets=int(rng.integers(6, 14)), n_values=3, shuffle=bool(seed % 2))
    pairs = np.arange(n) // 2
    R0 = sp.csr_matrix((np.full(n, 0.5), (pairs, np.arange(n))), shape=((n + 1) // 2, n))
    A = [A0, sp.csr_matrix(R0 @ A0 @ R0.T)]
    b, x0 = rng.standard_normal(n), rng.standard_normal(n)
    for dtype in ("float64", "float32"):
        for smoother in ("colour", "jacobi"):
            out = {}
            for mode, uk, ur in (("0", "1", "2"), ("15", "1", "1"), ("15", "1", "2"), ("15", "1", "4"), ("15", "0", "1")):
                monkeypatch.setenv("OMG_COMPRESS", mode)
                monkeypatch.setenv("OMG_UNION_KERNEL", uk)
                monkeypatch.setenv("OMG_UNION_ROWS", ur)
                with _hip.Hierarchy(A, [R0], smoother=smoother, omega=0.7, dtype=dtype) as h:
                    flags = h.level_flags(0)
                    r, nr = h.residual(0, b, x0, want_norm=True)
                    xs = x0.copy()
                    h.smooth(0, b, xs, 2)
                    h.resident_load(b, x0)
                    norms = [h.resident_cycle(1, 1) for _ in range(2)]
                    out[(mode, uk, ur)] = (r, xs, h.resident_fetch(), nr, norms, flags["union_walk"])
            base = out[("0", "1", "2")]
            assert not base[5]
            assert not out[("15", "0", "1")][5]
            if seed >= 2 and smoother == "jacobi":               # natural order: these operators get 512-row union blocks (host check)
                assert out[("15", "1", "2")][5]                  # the union walk really ran
            for key, run in out.items():
                for got, want in zip(run[:3], base[:3]):
                    assert np.array_equal(got, want), (key, dtype, smoother)
                np.testing.assert_allclose(run[3], base[3], rtol=1e-13 if dtype == "float64" else 1e-6)
                np.testing.assert_allclose(run[4], base[4], rtol=1e-13 if dtype == "float64" else 1e-6)


# -------------------------------------------- batched cycles: the norm finished by the next cycle --
@pytest.mark.parametrize("smoother,dtype", [("colour", "float64"), ("jacobi", "float64"), ("gs", "float64"),
                                            ("colour", "float32"), ("jacobi", "float32")])
def test_batched_cycles_return_every_norm_bit_for_bit(monkeypatch, smoother, dtype):
    """omg_resident_cycles(n): n cycles, n norms.  With two colour sets (or Jacobi) the norm of cycle
    k is finished inside cycle k + 1's first launch (ROW_GS_PRENORM / ROW_JACOBI_PRENORM: the launch
    forms b - A x for its rows with the iterate cycle k left) — the norms and the iterate must be
    the bits of n single omg_resident_cycle calls, and of the batch with OMG_NO_PRENORM=1."""
    runs = {}
    for shape, grids in (((32, 32, 32), 3), ((96, 96), 3)):
        A0 = operators.stencil_poisson(shape)
        R = operators.restrictionList(shape, grids - 2, 4)
        A = operators.coeffecientList(A0, R)
        b = A0 @ np.random.default_rng(5).random(A0.shape[0])
        for pre, post in ((1, 1), (2, 1), (1, 0), (0, 1)):
            out = []
            for how in ("single", "batch", "batch_noprenorm"):
                monkeypatch.setenv("OMG_NO_PRENORM", "1" if how == "batch_noprenorm" else "0")
                with _hip.Hierarchy(A, R, smoother=smoother, omega=0.7, dtype=dtype) as h:
                    h.resident_load(b)
                    if how == "single":
                        norms = [h.resident_cycle(pre, post) for _ in range(5)]
                    else:
                        norms = h.resident_cycles(pre, post, 2) + h.resident_cycles(pre, post, 3)     # batches chain
                    out.append((norms, h.resident_fetch()))
            for other in out[1:]:
                assert other[0] == out[0][0], (shape, pre, post)
                assert np.array_equal(other[1], out[0][1])
    # through mgSolve: a cycle-count stop rule takes the batched path
    shape = (32, 32, 32)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(6).random(A0.shape[0])
    p = {"problemShape": shape, "gridLevels": 2, "preIterations": 1, "postIterations": 1, "cycles": 6, "threshold": 0,
         "giveInfo": True, "smoother": smoother, "dtype": dtype}
    u, info = openmg_amd.mgSolve(A0, b, dict(p))
    with _hip.Hierarchy(info["A"], info["R"], smoother=smoother, omega=2.0 / 3.0, dtype=dtype) as h:
        h.resident_load(b)
        norms = [h.resident_cycle(1, 1) for _ in range(6)]
        assert info["cycle"] == 6 and info["norm"] == norms[-1] and np.array_equal(u, h.resident_fetch())


@pytest.mark.parametrize("smoother", ["colour", "jacobi", "gs"])
def test_first_relaxation_applied_by_the_restriction_is_bit_identical(monkeypatch, smoother):
    """A cycle enters every coarse level with a zero iterate (openmg/__init__.py:191-192), so the
    first smoothing launch there computes x_i = 0 + (b_i - 0) / a_ii for its rows; the restriction
    launch that produces b_i writes that instead of the zero and the launch is skipped
    (hierarchy.hip first_sweep_in_restrict).  OMG_NO_FIRST_SWEEP=1 runs the launch: same bits, for
    7-entry and 27-entry rows, both precisions, several sweep counts, under hipGraph replay."""
    n1 = 16
    T = sp.diags([np.ones(n1 - 1), np.ones(n1), np.ones(n1 - 1)], [-1, 0, 1])
    A27 = sp.csr_matrix(-sp.kron(sp.kron(T, T), T) + sp.diags(np.full(n1 ** 3, 28.0)))
    for A0, shape, grids in ((operators.stencil_poisson((32, 32, 32)), (32, 32, 32), 4), (A27, (n1,) * 3, 3),
                             (operators.stencil_poisson((128, 128)), (128, 128), 4)):
        R = operators.restrictionList(shape, grids - 2, 4)
        A = operators.coeffecientList(A0, R)
        b = A0 @ np.random.default_rng(3).random(A0.shape[0])
        for dtype in ("float64", "float32"):
            for pre, post, graph in ((1, 1, False), (2, 0, False), (1, 2, False), (1, 1, True)):
                out = []
                for off in ("1", "0"):
                    monkeypatch.setenv("OMG_NO_FIRST_SWEEP", off)
                    with _hip.Hierarchy(A, R, smoother=smoother, omega=0.7, dtype=dtype) as h:
                        h.use_graph(graph)
                        h.resident_load(b)
                        norms = [h.resident_cycle(pre, post) for _ in range(3)] + h.resident_cycles(pre, post, 2)
                        out.append((norms, h.resident_fetch()))
                assert out[0][0] == out[1][0], (shape, dtype, pre, post, graph)
                assert np.array_equal(out[0][1], out[1][1])


def test_mgcycle_overwrites_initial_with_the_presmoothed_iterate():
    """Q2: the reference's smoother works in place (openmg/solvers.py:68,75), so the array passed as
    `initial` holds the PRE-SMOOTHED iterate after mgCycle (openmg/__init__.py:201) and uOut is another
    array.  Checked against the oracle's sequential sweep and, bit for bit, the device's own smoother."""
    shape = (12, 12, 12)
    A0 = operators.stencil_poisson(shape)
    rng = np.random.default_rng(77)
    b, start = rng.standard_normal(A0.shape[0]), rng.standard_normal(A0.shape[0])
    R = operators.restrictionList(shape, 1, 8)
    A = operators.coeffecientList(A0, R)
    for smoother, pre in (("gs", 1), ("gs", 2), ("colour", 1)):
        p = {"coarsestLevel": len(R), "preIterations": pre, "postIterations": 1, "smoother": smoother}
        x0 = start.copy()
        u, info = openmg_amd.mgCycle(A, b, 0, R, p, initial=x0)
        assert u is not x0 and not np.array_equal(x0, start)
        want = start.copy()
        _hip.gauss_seidel(A[0], b, want, smoother=smoother, iterations=pre)
        assert np.array_equal(x0, want)
        if smoother == "gs":
            np.testing.assert_allclose(x0, orc.gauss_seidel(A[0], b, start.copy(), iterations=pre), rtol=1e-12, atol=1e-14)
        # the cycle itself is unchanged by that
        u2, info2 = openmg_amd.mgCycle(A, b, 0, R, p, initial=start.copy())
        assert np.array_equal(u, u2) and info["norm"] == info2["norm"]
        # pre = 0: nothing is relaxed in place
        x1 = start.copy()
        openmg_amd.mgCycle(A, b, 0, R, dict(p, preIterations=0), initial=x1)
        assert np.array_equal(x1, start)
