"""7-point operators with PER-ROW coefficients — the ordinary variable-coefficient input of openmg.mgSolve
(openmg/__init__.py:28, operators.py:178; VERDICT r3 / r4 / r5: "a fused pass for 7-point per-row coefficients") —
through the fused passes of csrc/var7.hip: each half of the cycle over such a level is one launch.  Bit for bit the
set-by-set schedule of the same hierarchy (omg_hierarchy_use_plane(0): the path the oracle comparisons at full size were
made on) for every sweep-count pair, both precisions, the symmetric (four arrays) and the general (seven arrays) storage,
ragged tiles; and against the CPU oracle's cycle (openmg/__init__.py:151-236) at BASELINE's 1e-10."""
import numpy as np
import pytest
import scipy.sparse as sp

import openmg_amd
from openmg_amd import _hip, operators
from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu


def hierarchy(shape, grids, seed=2024, unsymmetric=False):
    from test_gpu_plane import aggregation
    A0 = operators.stencil7_variable(shape, seed)
    if unsymmetric:                                       # same pattern, values no longer symmetric (a convection-like skew)
        A0 = sp.csr_matrix(A0)
        skew = 1.0 + 0.05 * np.sin(np.arange(A0.nnz))
        rows = np.repeat(np.arange(A0.shape[0]), np.diff(A0.indptr))
        A0.data = np.where(A0.indices == rows, A0.data, A0.data * skew)
    A, R, sh = [sp.csr_matrix(A0)], [], tuple(shape)
    for _ in range(grids - 1):
        R.append(aggregation(sh))
        Ac = sp.csr_matrix((R[-1] @ A[-1]) @ R[-1].T)
        Ac.sort_indices()
        A.append(Ac)
        sh = tuple(s // 2 for s in sh)
    return A, R


def cycles(h, b, x0, pre, post, n=3):
    h.resident_load(b, x0)
    norms = h.resident_cycles(pre, post, n)
    return norms, h.resident_fetch()


CASES = [((16, 16, 16), 2), ((32, 32, 32), 3), ((16, 20, 36), 2), ((40, 24, 72), 2), ((64, 32, 96), 3), ((20, 16, 132), 2)]


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,grids", CASES)
def test_fused_passes_have_the_bits_of_the_set_schedule(monkeypatch, shape, grids, dtype):
    monkeypatch.setenv("OMG_VAR7_MIN", "4096")            # (by default only levels of 128^3 and more take the passes)
    A, R = hierarchy(shape, grids)
    rng = np.random.default_rng(5)
    b = A[0] @ rng.random(A[0].shape[0])
    x0 = rng.standard_normal(b.size)
    if dtype == "float32":
        b, x0 = b.astype(np.float32).astype(np.float64), x0.astype(np.float32).astype(np.float64)
    for sym in ("1", "0"):
        monkeypatch.setenv("OMG_VAR7_SYM", sym)
        with _hip.Hierarchy(A, R, smoother="colour", dtype=dtype) as h:
            assert h.level_flags(0)["var7"] and not h.level_flags(0)["plane"]
            for pre, post in ((1, 1), (1, 0), (0, 1), (0, 0), (2, 1), (1, 2), (2, 2)):
                for start in (x0, None):
                    h.use_plane(True)
                    got = cycles(h, b, start, pre, post)
                    h.use_plane(False)
                    want = cycles(h, b, start, pre, post)
                    assert np.array_equal(got[1], want[1]), (shape, dtype, sym, pre, post, start is None, int(np.sum(got[1] != want[1])))
                    np.testing.assert_allclose(got[0], want[0], rtol=1e-12 if dtype == "float64" else 1e-6)


def test_general_storage_for_an_unsymmetric_operator(monkeypatch):
    monkeypatch.setenv("OMG_VAR7_MIN", "4096")
    A, R = hierarchy((32, 16, 32), 2, unsymmetric=True)
    b = A[0] @ np.random.default_rng(2).random(A[0].shape[0])
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        assert h.level_flags(0)["var7"]
        got = cycles(h, b, None, 1, 1)
        h.use_plane(False)
        want = cycles(h, b, None, 1, 1)
    assert np.array_equal(got[1], want[1])
    np.testing.assert_allclose(got[0], want[0], rtol=1e-12)


@pytest.mark.parametrize("size,grid_levels,pre,post", [(16, 1, 1, 1), (32, 2, 1, 1), (32, 2, 1, 0)])
def test_variable_coefficient_cycle_against_the_oracle(monkeypatch, size, grid_levels, pre, post):
    monkeypatch.setenv("OMG_VAR7_MIN", "4096")
    """kappa spans two decades; the reference's own lists (operators.restrictionList / coeffecientList), the oracle's
    mg_cycle with the red-black sweep; every cycle's norm within 1e-10, the iterate to rtol 1e-9."""
    shape = (size,) * 3
    A0 = operators.stencil7_variable(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    p = {"problemShape": shape, "gridLevels": grid_levels, "preIterations": pre, "postIterations": post, "cycles": 3, "threshold": 0,
         "giveInfo": True, "smoother": "colour", "minSize": 8}
    Ro = orc.restriction_list(shape, grid_levels - 1, 8)
    Ao = orc.coefficient_list(A0, Ro)
    sm = orc.make_smoother("colour", Ao)
    with _hip.Hierarchy(Ao, Ro, smoother="colour") as h:
        assert h.level_flags(0)["var7"]
        h.resident_load(b)
        norms = h.resident_cycles(pre, post, 3)
        x = h.resident_fetch()
    xo = None
    for k in range(3):
        xo, inf = orc.mg_cycle(Ao, b, 0, Ro, dict(p, coarsestLevel=len(Ro)), initial=xo, smoother=sm)
        assert abs(norms[k] - inf["norm"]) <= 1e-10 * inf["norm"], (k, norms[k], inf["norm"])
    np.testing.assert_allclose(x, xo, rtol=1e-9, atol=1e-12 * np.abs(xo).max())
    xs, info = openmg_amd.mgSolve(A0, b, dict(p))
    assert abs(info["norm"] - inf["norm"]) <= 1e-10 * inf["norm"]
    np.testing.assert_allclose(xs, xo, rtol=1e-9, atol=1e-12 * np.abs(xo).max())
    # without giveInfo mgSolve sets up on the device (omg_hierarchy_create_from_fine): the Galerkin chain, the check of every
    # large level's rows and its coefficient arrays in HBM; the small levels below the fused ones are fetched and coded by the
    # host.  The same hierarchy, the same bits — with every level fused and with the last smoothed level left to the host
    assert np.array_equal(openmg_amd.mgSolve(A0, b, dict(p, giveInfo=False)), xs)
    monkeypatch.setenv("OMG_VAR7_MIN", str(size ** 3))
    assert np.array_equal(openmg_amd.mgSolve(A0, b, dict(p, giveInfo=False)), xs)
    assert np.array_equal(openmg_amd.mgSolve(A0, b, dict(p))[0], xs)


def test_mgcycle_drop_in_q2_on_a_variable_coefficient_level(monkeypatch):
    monkeypatch.setenv("OMG_VAR7_MIN", "4096")
    """mgCycle with `initial`: the reference's pre-smoother works in place on the caller's array (Q2) — the pre-smoothed
    iterate comes back from the same fused down pass."""
    shape = (32, 32, 32)
    A0 = operators.stencil7_variable(shape)
    R = operators.restrictionList(shape, 0, 8)
    A = operators.coeffecientList(A0, R)
    b = A0 @ np.random.default_rng(1).random(A0.shape[0])
    x0 = np.random.default_rng(2).standard_normal(b.size)
    p = {"coarsestLevel": 1, "preIterations": 1, "postIterations": 1, "smoother": "colour"}
    init = x0.copy()
    x, info = openmg_amd.mgCycle(A, b, 0, R, p, initial=init)
    sm = orc.make_smoother("colour", A)
    want_init = x0.copy()
    xo, inf = orc.mg_cycle(A, b, 0, R, p, initial=want_init, smoother=sm)
    np.testing.assert_allclose(x, xo, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(init, want_init, rtol=1e-9, atol=1e-12)          # Q2: pre-smoothed in place
    assert abs(info["norm"] - inf["norm"]) <= 1e-10 * inf["norm"]
    openmg_amd.clear_cache()


def test_device_setup_in_single_precision_and_lexicographic_on_the_same_input(monkeypatch):
    """mgSolve with dtype float32 on both setup routes (the device route scatters the rows into float arrays), and the
    reference's default smoother on the same operator: per-row wavefront levels (march.hip), against the oracle."""
    monkeypatch.setenv("OMG_VAR7_MIN", "4096")
    shape = (32, 32, 32)
    A0 = operators.stencil7_variable(shape)
    b = A0 @ np.random.default_rng(5).random(A0.shape[0])
    p = {"problemShape": shape, "gridLevels": 2, "preIterations": 1, "postIterations": 1, "cycles": 3, "threshold": 0,
         "smoother": "colour", "dtype": "float32", "minSize": 8}
    x_dev = openmg_amd.mgSolve(A0, b, dict(p))
    x_host, info = openmg_amd.mgSolve(A0, b, dict(p, giveInfo=True))
    assert np.array_equal(x_dev, x_host)
    ref, iref = openmg_amd.mgSolve(A0, b, dict(p, dtype="float64", giveInfo=True))
    assert abs(info["norm"] - iref["norm"]) <= 5e-4 * iref["norm"]
    # the reference's own smoother ('gs', the default) on the variable-coefficient input
    pg = {"problemShape": shape, "gridLevels": 2, "preIterations": 1, "postIterations": 1, "cycles": 2, "threshold": 0, "giveInfo": True, "minSize": 8}
    xg, ig = openmg_amd.mgSolve(A0, b, dict(pg))
    with _hip.Hierarchy(ig["A"], ig["R"], smoother="gs") as h:
        assert all(h.level_flags(l)["march"] for l in range(2))
    xo = None
    for _ in range(2):
        xo, io = orc.mg_cycle(ig["A"], b, 0, ig["R"], dict(pg, coarsestLevel=2), initial=xo)
    assert abs(ig["norm"] - io["norm"]) <= 1e-10 * io["norm"]
    np.testing.assert_allclose(xg, xo, rtol=1e-9, atol=1e-12 * np.abs(xo).max())


def test_full_size_256_cubed_through_size_independent_properties():
    """The bench's `var7` workload itself — 256^3, 5 grids, kappa over two decades — where the sequential oracle is out of
    reach: the fused passes give the bits of the set-by-set schedule, the device's norm is SciPy's ||b - A x|| of the fetched
    iterate (1e-10), the cycle is linear bit for bit (cycle(2 b) = 2 cycle(b): every operation of the passes is), it contracts;
    and the reference's own smoother on the same hierarchy (per-row wavefront levels) gives the level schedule's bits at 128^3."""
    shape = (256, 256, 256)
    A0 = operators.stencil7_variable(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, 3, 8)
    A = operators.coeffecientList(A0, R)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        assert [h.level_flags(l)["var7"] for l in range(4)] == [True, True, False, False]
        h.resident_load(b)
        norms = h.resident_cycles(1, 1, 3)
        x = h.resident_fetch()
        host = float(np.linalg.norm(b - A0 @ x))
        assert abs(norms[-1] - host) <= 1e-10 * host
        assert norms[2] < norms[1] < norms[0]
        h.resident_load(2.0 * b)
        h.resident_cycles(1, 1, 3)
        assert np.array_equal(h.resident_fetch(), 2.0 * x)
        h.use_plane(False)
        h.resident_load(b)
        ns = h.resident_cycles(1, 1, 3)
        assert np.array_equal(h.resident_fetch(), x)
        np.testing.assert_allclose(ns, norms, rtol=1e-12)
    # the reference's lexicographic sweep on per-row coefficients: wavefront launches against the level schedule, 128^3
    import os
    As, Rs = A[1:], R[1:]
    bs = np.random.default_rng(5).random(As[0].shape[0])
    out = {}
    for march in ("1", "0"):
        os.environ["OMG_MARCH"] = march
        try:
            with _hip.Hierarchy(As, Rs, smoother="gs") as h:
                assert bool(h.level_flags(0)["march"]) == (march == "1")
                h.resident_load(bs)
                out[march] = (h.resident_cycles(1, 1, 2), h.resident_fetch())
        finally:
            del os.environ["OMG_MARCH"]
    assert np.array_equal(out["1"][1], out["0"][1])
    np.testing.assert_allclose(out["1"][0], out["0"][0], rtol=1e-12)
