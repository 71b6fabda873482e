"""fp32 hierarchies (BASELINE.json configs[4]: "fp32") through the C ABI.  Needs an MI355X.

The reference is fp64 throughout, so there is no fp32 golden vector: the statement tested is
"the fp32 path computes the same algorithm, to single-precision rounding".  The oracle runs
in fp64; tolerances are multiples of eps32 = 6e-8 scaled by the operator's row sums (one
operation: a row sum of <= 27 terms of magnitude <= ROW * max|x|), written next to each check.
Where the domain offers an exact statement (fused vs unfused launches, hipGraph replay,
operators whose entries and inputs are exactly representable) the comparison is bitwise.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import openmg_amd
from openmg_amd import _hip, operators
from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu

EPS32 = float(np.finfo(np.float32).eps)


def csr_from(d, prefix):
    return sp.csr_matrix((d[prefix + "_data"], d[prefix + "_indices"], d[prefix + "_indptr"]),
                         shape=tuple(d[prefix + "_shape"]))


def close32(got, want, scale, ulps):
    """|got - want| <= ulps * eps32 * scale element-wise."""
    np.testing.assert_allclose(got, want, rtol=0, atol=ulps * EPS32 * scale)


def test_dtype_is_reported_and_validated():
    A0 = operators.stencil_poisson((16, 16))
    R = operators.restrictionList((16, 16), 0, 4)
    A = operators.coeffecientList(A0, R)
    for name, want in (("float64", np.float64), ("float32", np.float32), (np.float32, np.float32)):
        with _hip.Hierarchy(A, R, smoother="colour", dtype=name) as h:
            assert h.device_dtype() == np.dtype(want)
    with pytest.raises(ValueError):
        _hip.Hierarchy(A, R, dtype="float16")
    out = _hip.ctypes.c_void_p()
    arrA = (_hip.CsrView * 2)(*[_hip.csr_view(_hip.as_csr(M)) for M in A])
    arrR = (_hip.CsrView * 1)(*[_hip.csr_view(_hip.as_csr(M)) for M in R])
    code = _hip.lib().omg_hierarchy_create_ex(2, arrA, arrR, 0, 1.0, 7, _hip.ctypes.byref(out))
    assert code != 0 and not out.value and b"dtype" in _hip.lib().omg_last_error()


def test_fp32_exact_on_representable_data():
    """Integer-valued operators and vectors small enough for a 24-bit significand: every fp32
    product and partial sum is exact, so the fp32 kernels must return the fp64 result bit for
    bit (restriction, prolongation, SpMV-type residual)."""
    shape = (24, 20, 28)
    A0 = operators.stencil_poisson(shape)                   # entries -1, 6
    R = operators.restrictionList(shape, 0, 4)               # piecewise-constant aggregation
    A = operators.coeffecientList(A0, R)
    rng = np.random.default_rng(3)
    n = A0.shape[0]
    x = rng.integers(-100, 100, n).astype(np.float64)
    b = rng.integers(-1000, 1000, n).astype(np.float64)
    with _hip.Hierarchy(A, R, smoother="colour", dtype="float32") as h:
        r, norm = h.residual(0, b, x, want_norm=True)
        want = b - A0 @ x
        assert np.array_equal(r, want)
        # the norm is accumulated in double from the (exact) fp32 residuals
        np.testing.assert_allclose(norm, np.linalg.norm(want), rtol=1e-13)
        assert np.all(R[0].data == 0.125)                   # a power of two: products stay exact
        assert np.array_equal(h.restrict(0, want), R[0] @ want)
        e = rng.integers(-50, 50, R[0].shape[0]).astype(np.float64) * 8
        assert np.array_equal(h.prolong_add(0, e, x), x + R[0].T @ e)


@pytest.mark.parametrize("smoother", ["gs", "colour", "jacobi"])
def test_fp32_level_operations_against_oracle(golden, smoother):
    d = golden("g3_poisson3d_16")
    A = [csr_from(d, "A%d" % l) for l in range(3)]
    R = [csr_from(d, "R%d" % l) for l in range(2)]
    rng = np.random.default_rng(8)
    with _hip.Hierarchy(A, R, smoother=smoother, omega=0.8, dtype="float32") as h:
        for l in (0, 1):
            n = A[l].shape[0]
            row = float(abs(A[l]).sum(axis=1).max())          # |row sum| bound
            dmin = float(abs(A[l].diagonal()).min())
            b, x0 = rng.random(n), rng.random(n)
            x = x0.copy()
            h.smooth(l, b, x, 2)
            if smoother == "gs":
                want = orc.gauss_seidel(A[l], b, x0.copy(), iterations=2)
            elif smoother == "colour":
                want = orc.gs_ordered(A[l], b, x0.copy(), orc.colour_order(orc.greedy_colouring(A[l])), 2)
            else:
                want = orc.jacobi(A[l], b, x0.copy(), 2, 0.8)
            # two sweeps; each update divides a row-sum error (~ nnz/row * eps32 * row) by the
            # diagonal, and lexicographic sweeps carry it along the ordering: 200 * eps32 * row/dmin
            close32(x, want, row / dmin * max(1.0, np.abs(want).max()), 200)
            r, norm = h.residual(l, b, x0, want_norm=True)
            wr = orc.get_residual(b, A[l], x0, n).ravel()
            close32(r, wr, row, 32)                          # inputs rounded to fp32 + <= 27-term fma chain
            np.testing.assert_allclose(norm, np.linalg.norm(wr), rtol=64 * EPS32)
            rsum = float(abs(R[l]).sum(axis=1).max())
            close32(h.restrict(l, r), R[l] @ r, rsum * np.abs(r).max(), 16)
            e = rng.random(R[l].shape[0])
            close32(h.prolong_add(l, e, x0), x0 + R[l].T @ e, float(abs(R[l]).max()) + 1.0, 8)
        bc = rng.random(A[2].shape[0])
        want = orc.coarse_solve(A[2], bc.reshape(-1, 1)).ravel()
        # inverse computed in fp64, rounded once, applied in fp32: cond(A_c) * eps32
        cond = np.linalg.cond(A[2].toarray())
        close32(h.coarse_solve(bc), want, cond * np.abs(want).max(), 16)


@pytest.mark.parametrize("smoother,shape,grids", [("colour", (32, 32, 32), 3), ("jacobi", (64, 64), 3),
                                                  ("gs", (16, 16, 16), 2)])
def test_fp32_vcycles_track_the_fp64_path(smoother, shape, grids):
    """Whole cycles: while the residual is above the fp32 floor the fp32 norms follow the fp64
    ones to 1e-3 relative, and the fp32 iterate ends with a true residual (evaluated in fp64 on
    the returned vector) no worse than the fp64 path's plus floor = 64 * eps32 * ||A||_inf *
    ||x||_inf * sqrt(n): single-precision backward stability."""
    A0 = operators.stencil_poisson(shape)
    R = operators.restrictionList(shape, grids - 2, 4)
    A = operators.coeffecientList(A0, R)
    n = A0.shape[0]
    xt = np.random.default_rng(5).random(n)
    b = A0 @ xt
    runs = {}
    for dtype in ("float64", "float32"):
        with _hip.Hierarchy(A, R, smoother=smoother, dtype=dtype) as h:
            h.resident_load(b)
            norms = [h.resident_cycle(1, 1) for _ in range(25)]
            runs[dtype] = (np.array(norms), h.resident_fetch())
    n64, x64 = runs["float64"]
    n32, x32 = runs["float32"]
    floor = 64 * EPS32 * float(abs(A0).sum(axis=1).max()) * np.abs(xt).max() * np.sqrt(n)
    above = n64 > 50 * floor
    assert above.sum() >= 2
    np.testing.assert_allclose(n32[above], n64[above], rtol=1e-3)
    true_r = np.linalg.norm(b - A0 @ x32)
    # as converged as the fp64 path, down to the fp32 floor
    assert true_r <= 1.001 * n64[-1] + floor, (true_r, n64[-1], floor)
    # the reported fp32 norm is honest about that residual (not the fp64 path's value)
    assert abs(n32[-1] - true_r) <= floor, (n32[-1], true_r, floor)
    assert np.abs(x32 - x64).max() <= 1e-3 * np.abs(x64).max()


@pytest.mark.parametrize("smoother", ["colour", "gs"])
def test_fp32_fused_last_set_and_graph_replay_are_bit_identical(monkeypatch, smoother):
    shape = (32, 32, 32)
    A0 = operators.stencil_poisson(shape)
    R = operators.restrictionList(shape, 1, 8)
    A = operators.coeffecientList(A0, R)
    b = A0 @ np.random.default_rng(21).random(A0.shape[0])
    out = []
    for no_fuse, graph in (("1", False), ("0", False), ("0", True)):
        monkeypatch.setenv("OMG_NO_FUSE", no_fuse)
        with _hip.Hierarchy(A, R, smoother=smoother, dtype="float32") as h:
            h.use_graph(graph)
            h.resident_load(b)
            norms = [h.resident_cycle(1, 1) for _ in range(4)]
            out.append((norms, h.resident_fetch()))
    for other in out[1:]:
        assert out[0][0] == other[0]
        assert np.array_equal(out[0][1], other[1])


def test_fp32_through_the_public_drivers():
    shape = (24, 24, 24)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(9).random(A0.shape[0])
    base = {"problemShape": shape, "gridLevels": 2, "cycles": 12, "threshold": 0, "preIterations": 1,
            "postIterations": 1, "smoother": "colour", "giveInfo": True, "minSize": 4}
    u64, i64 = openmg_amd.mgSolve(A0, b, dict(base))
    u32, i32 = openmg_amd.mgSolve(A0, b, dict(base, dtype="float32"))
    assert i32["cycle"] == i64["cycle"] == 12
    assert u32.dtype == np.float64
    assert np.abs(u32 - u64).max() <= 1e-3 * np.abs(u64).max()
    assert not np.array_equal(u32, u64)                      # it really ran in another precision
    # mgCycle: the cache keeps one device hierarchy per dtype
    p = dict(base, coarsestLevel=len(i64["R"]))
    x64, _ = openmg_amd.mgCycle(i64["A"], b, 0, i64["R"], p)
    x32, _ = openmg_amd.mgCycle(i64["A"], b, 0, i64["R"], dict(p, dtype="float32"))
    again, _ = openmg_amd.mgCycle(i64["A"], b, 0, i64["R"], p)
    assert np.array_equal(again, x64)
    assert 0 < np.abs(x32 - x64).max() <= 1e-4 * np.abs(x64).max()
    openmg_amd.clear_cache()
