"""BASELINE configs[4]'s operator family on its own kernels (openmg_amd/csrc/stencil27.hip): a 27-point grid stencil
with per-row coefficients under the 2x2x2 aggregation, 8-colour Gauss-Seidel.  Against the set-by-set schedule of the
same hierarchy — same bits in the iterate — and against the CPU oracle (openmg/__init__.py:151-236 with the
reference's sweep, openmg/solvers.py:56-68, on the colour-permuted system)."""
import numpy as np
import pytest
import scipy.sparse as sp

from openmg_amd import _hip, operators
from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu


def aggregation(shape):
    """2x2x2 cell aggregation with weight 1/8 for any even shape (sorted columns)."""
    mats = []
    for s in shape:
        m = sp.lil_matrix((s // 2, s))
        for i in range(s // 2):
            m[i, 2 * i] = 0.5
            m[i, 2 * i + 1] = 0.5
        mats.append(sp.csr_matrix(m))
    R = mats[0]
    for m in mats[1:]:
        R = sp.kron(R, m, format="csr")
    R = sp.csr_matrix(R)
    R.sort_indices()
    return R


def hierarchy(shape, grids, seed=2024):
    A = [operators.stencil27_variable(shape, seed=seed)]
    R = []
    sh = tuple(shape)
    for _ in range(grids - 1):
        R.append(aggregation(sh))
        Ac = sp.csr_matrix((R[-1] @ A[-1]) @ R[-1].T)
        Ac.sort_indices()
        A.append(Ac)
        sh = tuple(s // 2 for s in sh)
    return A, R


def run(h, b, pre, post, cycles, x0=None):
    h.resident_load(b, x0)
    norms = [h.resident_cycle(pre, post) for _ in range(cycles)]
    return norms, h.resident_fetch()


def close(a, b, tol=1e-12):
    return all(abs(u - v) <= tol * abs(v) for u, v in zip(a, b))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,grids", [((16, 16, 16), 3), ((8, 12, 20), 2), ((4, 4, 4), 2), ((6, 10, 14), 2), ((32, 32, 32), 4),
                                         ((12, 8, 36), 2)])
def test_stencil27_kernels_have_the_bits_of_the_set_schedule(shape, grids, dtype):
    """Same hierarchy object, the 27-point kernels on / off (omg_hierarchy_use_plane): the iterate bit for bit for every
    sweep count, the norm to rounding (its partial sums are associated per workgroup instead of per row block)."""
    A, R = hierarchy(shape, grids)
    rng = np.random.default_rng(7)
    b = A[0] @ rng.random(A[0].shape[0])
    x0 = rng.standard_normal(A[0].shape[0])
    if dtype == "float32":
        b, x0 = b.astype(np.float32).astype(np.float64), x0.astype(np.float32).astype(np.float64)
    with _hip.Hierarchy(A, R, smoother="colour", dtype=dtype) as h:
        assert h.level_flags(0)["stencil27"] and not h.level_flags(0)["plane"], shape
        assert h.level_sets(0) == 8
        for pre, post in ((1, 1), (2, 1), (1, 2), (1, 0), (0, 1), (0, 0), (3, 2)):
            h.use_plane(True)
            assert h.level_flags(0)["stencil27"]
            got = run(h, b, pre, post, 3, x0)
            h.resident_load(b, x0)
            batch = h.resident_cycles(pre, post, 3)
            xb = h.resident_fetch()
            h.use_plane(False)
            assert not h.level_flags(0)["stencil27"]
            ref = run(h, b, pre, post, 3, x0)
            assert np.array_equal(got[1], ref[1]), (shape, dtype, pre, post, int(np.sum(got[1] != ref[1])))
            assert close(got[0], ref[0]), (pre, post, got[0], ref[0])
            # batched cycles: the norm of cycle k formed inside cycle k + 1's first sweep
            assert np.array_equal(xb, got[1]) and close(batch, got[0]), (pre, post, batch, got[0])


def test_octant_ordering_is_the_greedy_colouring(monkeypatch):
    monkeypatch.setenv("OMG_PLANE_CHECK_ORDER", "1")
    for shape, grids in (((8, 8, 8), 2), ((8, 12, 20), 2), ((16, 16, 16), 3)):
        A, R = hierarchy(shape, grids)
        with _hip.Hierarchy(A, R, smoother="colour") as h:
            assert h.level_flags(0)["stencil27"] and h.level_sets(0) == 8


@pytest.mark.parametrize("pre,post", [(1, 1), (1, 0), (2, 2)])
def test_stencil27_cycle_against_the_oracle(pre, post):
    """16^3, 3 grids, fp64: every cycle's norm within BASELINE's 1e-10 of the oracle's, the iterate rtol 1e-9."""
    shape = (16, 16, 16)
    A0 = operators.stencil27_variable(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    Ro = orc.restriction_list(shape, 1, 1)
    Ao = orc.coefficient_list(A0, Ro)
    sm = orc.make_smoother("colour", Ao)
    R = operators.restrictionList(shape, 1, 1)
    A = operators.coeffecientList(A0, R)
    p = {"preIterations": pre, "postIterations": post, "coarsestLevel": len(Ro)}
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        assert all(h.level_flags(l)["stencil27"] for l in range(len(R)))
        h.resident_load(b)
        xo = None
        for _ in range(3):
            norm = h.resident_cycle(pre, post)
            xo, info = orc.mg_cycle(Ao, b, 0, Ro, p, initial=xo, smoother=sm)
            assert abs(norm - info["norm"]) <= 1e-10 * info["norm"]
        np.testing.assert_allclose(h.resident_fetch(), xo, rtol=1e-9, atol=1e-12)


def test_stencil27_entry_points_agree():
    """omg_vcycle, omg_solve, hipGraph replay and the device-pointer cycle give the bits of omg_resident_cycle."""
    import ctypes
    shape = (16, 16, 16)
    A, R = hierarchy(shape, 3)
    b = A[0] @ np.random.default_rng(3).random(A[0].shape[0])
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        single = run(h, b, 1, 1, 4)
        x = np.zeros(b.size)
        norms = [h.vcycle(b, x, 1, 1) for _ in range(4)]
        assert norms == single[0] and np.array_equal(x, single[1])
        x = np.zeros(b.size)
        assert h.solve(b, x, 1, 1, 4, 0.0) == (4, single[0][-1]) and np.array_equal(x, single[1])
        h.use_graph(True)
        graph = run(h, b, 1, 1, 4)
        h.use_graph(False)
        assert graph[0] == single[0] and np.array_equal(graph[1], single[1])
        first = run(h, b, 1, 1, 1)[1]
        hip = ctypes.CDLL("libamdhip64.so.7")
        hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipFree.argtypes = [ctypes.c_void_p]
        bd, xd = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(bd), b.nbytes) == 0 and hip.hipMalloc(ctypes.byref(xd), b.nbytes) == 0
        try:
            got = np.full(b.size, np.nan)
            assert hip.hipMemcpy(bd, b.ctypes.data, b.nbytes, 1) == 0 and hip.hipMemcpy(xd, got.ctypes.data, b.nbytes, 1) == 0
            h.cycle_dev(bd.value, xd.value, 1, 1)
            h.sync()
            assert hip.hipMemcpy(got.ctypes.data, xd, b.nbytes, 2) == 0
        finally:
            hip.hipFree(bd)
            hip.hipFree(xd)
        assert np.array_equal(got, first)


def test_operators_that_are_not_such_a_stencil_keep_the_set_schedule():
    shape = (8, 8, 8)
    A, R = hierarchy(shape, 2)
    Ad = sp.csr_matrix(A[0])
    Ad.data = Ad.data.copy()
    lil = Ad.tolil()
    lil[100, 101] = 0.0                                      # one coupling missing: not the full 27-point pattern
    Am = sp.csr_matrix(lil)
    Am.eliminate_zeros()
    Am.sort_indices()
    with _hip.Hierarchy([Am, A[1]], R, smoother="colour") as h:
        assert not h.level_flags(0)["stencil27"]
    for smoother in ("gs", "jacobi"):
        with _hip.Hierarchy(A, R, smoother=smoother) as h:
            assert not h.level_flags(0)["stencil27"]
    A7 = [operators.stencil_poisson(shape)]
    A7.append(sp.csr_matrix((R[0] @ A7[0]) @ R[0].T))
    with _hip.Hierarchy(A7, R, smoother="colour") as h:
        assert h.level_flags(0)["plane"] and not h.level_flags(0)["stencil27"]
