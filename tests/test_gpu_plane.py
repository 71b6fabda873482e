"""The plane-pipelined halves of a red-black V-cycle (openmg_amd/csrc/plane.hip: sweep + residual +
restriction in one launch, prolongation + sweep + norm in another) against the set-by-set schedule
they replace — the iterate must have the same bits — and against the CPU oracle
(openmg/__init__.py:201-227 on the colour-permuted system)."""
import numpy as np
import pytest
import scipy.sparse as sp

from openmg_amd import _hip, operators
from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu


def aggregation(shape):
    """2x2x2 cell aggregation with weight 1/8 for ANY even shape (openmg/operators.py:73-84 builds this
    for cubes; for other shapes its second-axis stride quirk Q6 gives something else), sorted columns."""
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


def hierarchy(shape, grids, scale=1.0):
    """Galerkin hierarchy of the 7-point operator over `grids` grids with the plain aggregation."""
    A = [sp.csr_matrix(operators.stencil_poisson(shape) * scale)]
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


def close(a, b, tol=1e-13):
    return all(abs(u - v) <= tol * abs(v) for u, v in zip(a, b))


@pytest.mark.parametrize("shape,grids", [((16, 16, 16), 3), ((32, 32, 32), 4), ((8, 12, 20), 2), ((10, 8, 6), 2),
                                         ((4, 4, 4), 2), ((12, 20, 34), 2), ((40, 24, 72), 3), ((64, 64, 64), 4)])
def test_plane_passes_have_the_bits_of_the_set_schedule(shape, grids):
    """Same hierarchy object, plane passes on / off: the iterate bit for bit, the norm to rounding
    (its partial sums are associated per workgroup tile instead of per 512-row block)."""
    A, R = hierarchy(shape, grids)
    rng = np.random.default_rng(7)
    b = A[0] @ rng.random(A[0].shape[0])
    x0 = rng.standard_normal(A[0].shape[0])
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        assert h.level_flags(0)["plane"], shape
        info = h.plane_info(0)
        assert (info["nz"], info["ny"], info["nx"]) == tuple(shape)
        for pre, post in ((1, 1), (2, 1), (1, 2), (3, 2)):
            h.use_plane(True)
            got = run(h, b, pre, post, 3, x0)
            h.use_plane(False)
            assert not h.level_flags(0)["plane"]
            ref = run(h, b, pre, post, 3, x0)
            assert np.array_equal(got[1], ref[1]), (shape, pre, post, int(np.sum(got[1] != ref[1])))
            assert close(got[0], ref[0]), (got[0], ref[0])
        # cycles without pre- or post-smoothing — V(1,0) is the reference's default (openmg/__init__.py:22-23) —
        # stay on the plane passes: the same launches without their relaxation stages
        for pre, post in ((1, 0), (0, 1), (2, 0), (0, 2), (0, 0)):
            h.use_plane(True)
            assert h.level_flags(0)["plane"]
            got = run(h, b, pre, post, 3, x0)
            batch = None
            if post <= 1:
                h.resident_load(b, x0)
                batch = h.resident_cycles(pre, post, 3)
                assert batch == got[0] and np.array_equal(h.resident_fetch(), got[1])
            h.use_plane(False)
            ref = run(h, b, pre, post, 3, x0)
            assert np.array_equal(got[1], ref[1]), (shape, pre, post, int(np.sum(got[1] != ref[1])))
            assert close(got[0], ref[0]), (pre, post, got[0], ref[0])


def test_default_cycle_against_the_oracle():
    """The reference's own parameter dict — preIterations 1, postIterations 0 (openmg/__init__.py:22-23) — and
    V(0,1), on the plane passes against the CPU restatement of mgCycle with the red-black ordering."""
    shape = (32, 32, 32)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, 2, 8)
    A = operators.coeffecientList(A0, R)
    Ro = orc.restriction_list(shape, 2, 8)
    Ao = orc.coefficient_list(A0, Ro)
    sm = orc.make_smoother("colour", Ao)
    for pre, post in ((1, 0), (0, 1)):
        p = {"preIterations": pre, "postIterations": post, "coarsestLevel": len(Ro)}
        with _hip.Hierarchy(A, R, smoother="colour") as h:
            assert all(h.level_flags(l)["plane"] for l in range(len(R)))
            h.resident_load(b)
            xo = None
            for _ in range(3):
                norm = h.resident_cycle(pre, post)
                xo, info = orc.mg_cycle(Ao, b, 0, Ro, p, initial=xo, smoother=sm)
                assert abs(norm - info["norm"]) <= 1e-10 * info["norm"]
            np.testing.assert_allclose(h.resident_fetch(), xo, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("pre,post", [(1, 1), (2, 1), (1, 0), (2, 2), (0, 1), (3, 0)])
def test_device_pointer_cycle_on_a_plane_level(pre, post):
    """omg_hierarchy_cycle_dev (the multi-GPU runner's replicated tail) on a plane-qualified hierarchy whose
    row-kernel format is still pending: every sweep count gives what omg_vcycle gives from a zero iterate
    (the diagonal of the first relaxation was formed before the format existed)."""
    import ctypes
    # device vectors from the HIP runtime libopenmg_hip.so itself is linked against (NOT torch: it ships a second
    # copy of the runtime, and two of them in one process end in a double free at interpreter exit)
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    shape = (16, 16, 16)
    A, R = hierarchy(shape, 3)
    b = A[0] @ np.random.default_rng(11).random(A[0].shape[0])
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        x = np.zeros(b.size)
        h.vcycle(b, x, pre, post)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        assert h.level_flags(0)["plane"]
        bd, xd = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(bd), b.nbytes) == 0 and hip.hipMalloc(ctypes.byref(xd), b.nbytes) == 0
        try:
            got = np.full(b.size, np.nan)
            assert hip.hipMemcpy(bd, b.ctypes.data, b.nbytes, 1) == 0 and hip.hipMemcpy(xd, got.ctypes.data, b.nbytes, 1) == 0
            h.cycle_dev(bd.value, xd.value, pre, post)
            h.sync()
            assert hip.hipMemcpy(got.ctypes.data, xd, b.nbytes, 2) == 0
        finally:
            hip.hipFree(bd)
            hip.hipFree(xd)
        assert np.array_equal(got, x), (pre, post)


@pytest.mark.parametrize("shape", [(16, 16, 16), (8, 12, 20), (10, 8, 6), (12, 20, 34), (4, 4, 4)])
def test_matrix_free_spmv_of_a_plane_level(shape):
    """omg_level_spmv on a plane level (plane_spmv_kernel) against the row kernels on the same operator — same bits —
    and against SciPy's csr_matvec (openmg/tools.py:26)."""
    A, R = hierarchy(shape, 2, scale=0.77)
    x = np.random.default_rng(4).standard_normal(A[0].shape[0])
    for dtype in ("float64", "float32"):
        xx = x if dtype == "float64" else x.astype(np.float32).astype(np.float64)
        with _hip.Hierarchy(A, R, smoother="colour", dtype=dtype) as h:
            assert h.level_flags(0)["plane"]
            got = h.spmv(0, xx)
            h.use_plane(False)
            ref = h.spmv(0, xx)
        assert np.array_equal(got, ref), (shape, dtype, int(np.sum(got != ref)))
        np.testing.assert_allclose(got, A[0] @ xx, rtol=1e-13 if dtype == "float64" else 1e-5, atol=1e-12 if dtype == "float64" else 1e-5)


def test_parity_ordering_is_the_greedy_colouring(monkeypatch):
    """A level that qualifies gets its red-black ordering in closed form instead of the sequential greedy
    pass; OMG_PLANE_CHECK_ORDER=1 makes the library compare the two (sets, perm, inv) at creation."""
    monkeypatch.setenv("OMG_PLANE_CHECK_ORDER", "1")
    for shape, grids in (((16, 16, 16), 3), ((8, 12, 20), 2), ((10, 8, 6), 2), ((40, 24, 72), 3)):
        A, R = hierarchy(shape, grids)
        with _hip.Hierarchy(A, R, smoother="colour") as h:
            assert h.level_flags(0)["plane"] and h.level_sets(0) == 2


@pytest.mark.parametrize("tile", ["4,2,2", "8,4,2", "8,2,4", "12,6,4", "16,8,8", "64,32,32", "32,8,2",
                                  "256,2,4", "128,22,26"])     # 66 threads per row: a wave's LDS neighbours are two waves away
def test_plane_tilings_agree(monkeypatch, tile):
    """Every tiling — rings, partial tiles at the grid's edges, chunks of two planes — is the same
    arithmetic: same bits as the set schedule."""
    shape = (24, 20, 36)                       # nz, ny, nx: nx % 4 == 0, ny % 4 == 0, tiles do not divide them
    A, R = hierarchy(shape, 2, scale=0.37)
    rng = np.random.default_rng(8)
    b = rng.standard_normal(A[0].shape[0])
    x0 = rng.standard_normal(A[0].shape[0])
    monkeypatch.setenv("OMG_PLANE", "0")
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        assert not h.level_flags(0)["plane"]
        ref = run(h, b, 1, 1, 3, x0)
    monkeypatch.setenv("OMG_PLANE", "1")
    monkeypatch.setenv("OMG_PLANE_TILE", tile)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        info = h.plane_info(0)
        assert [info["tile_x"], info["tile_y"], info["tile_z"]] == [int(v) for v in tile.split(",")]
        got = run(h, b, 1, 1, 3, x0)
    assert np.array_equal(got[1], ref[1]), (tile, int(np.sum(got[1] != ref[1])))
    assert close(got[0], ref[0])


def test_plane_cycle_against_the_oracle():
    """The fused passes against the CPU restatement of mgCycle (openmg/__init__.py:151-236) with the
    red-black ordering: BASELINE's 1e-10 gate on every cycle's norm, rtol 1e-9 on the iterate."""
    shape = (32, 32, 32)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, 2, 8)
    A = operators.coeffecientList(A0, R)
    Ro = orc.restriction_list(shape, 2, 8)
    Ao = orc.coefficient_list(A0, Ro)
    sm = orc.make_smoother("colour", Ao)
    for pre, post in ((1, 1), (2, 2)):
        p = {"preIterations": pre, "postIterations": post, "coarsestLevel": len(Ro)}
        with _hip.Hierarchy(A, R, smoother="colour") as h:
            assert all(h.level_flags(l)["plane"] for l in range(len(R)))
            h.resident_load(b)
            xo = None
            for _ in range(3):
                norm = h.resident_cycle(pre, post)
                xo, info = orc.mg_cycle(Ao, b, 0, Ro, p, initial=xo, smoother=sm)
                assert abs(norm - info["norm"]) <= 1e-10 * info["norm"]
            np.testing.assert_allclose(h.resident_fetch(), xo, rtol=1e-9, atol=1e-12)


def test_plane_batched_cycles_graph_and_entry_points():
    """omg_resident_cycles (one sum launch per chunk), hipGraph replay, omg_vcycle / omg_solve and
    mgSolve give the bits of single omg_resident_cycle calls."""
    import openmg_amd
    shape = (32, 32, 32)
    A, R = hierarchy(shape, 3)
    b = A[0] @ np.random.default_rng(3).random(A[0].shape[0])
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        single = run(h, b, 1, 1, 5)
        h.resident_load(b)
        batch = h.resident_cycles(1, 1, 2) + h.resident_cycles(1, 1, 3)
        assert batch == single[0] and np.array_equal(h.resident_fetch(), single[1])
        h.use_graph(True)
        graph = run(h, b, 1, 1, 5)
        h.use_graph(False)
        assert graph[0] == single[0] and np.array_equal(graph[1], single[1])
        x = np.zeros(A[0].shape[0])
        norms = [h.vcycle(b, x, 1, 1) for _ in range(5)]
        assert norms == single[0] and np.array_equal(x, single[1])
        x = np.zeros(A[0].shape[0])
        assert h.solve(b, x, 1, 1, 5, 0.0) == (5, single[0][-1]) and np.array_equal(x, single[1])
    A0 = operators.stencil_poisson(shape)
    p = {"problemShape": shape, "gridLevels": 2, "preIterations": 1, "postIterations": 1, "cycles": 4, "threshold": 0,
         "giveInfo": True, "smoother": "colour"}
    u, info = openmg_amd.mgSolve(A0, b, dict(p))
    with _hip.Hierarchy(info["A"], info["R"], smoother="colour") as h:
        ref = run(h, b, 1, 1, 4)
        assert info["norm"] == ref[0][-1] and np.array_equal(u, ref[1])


def test_plane_fp32_levels():
    shape = (32, 32, 32)
    A, R = hierarchy(shape, 3)
    rng = np.random.default_rng(5)
    b = (A[0] @ rng.random(A[0].shape[0])).astype(np.float32).astype(np.float64)
    with _hip.Hierarchy(A, R, smoother="colour", dtype="float32") as h:
        assert h.level_flags(0)["plane"]
        got = run(h, b, 1, 1, 4)
        h.use_plane(False)
        ref = run(h, b, 1, 1, 4)
        assert np.array_equal(got[1], ref[1])
        assert close(got[0], ref[0], 1e-12)


def test_levels_that_do_not_qualify_keep_the_set_schedule():
    """Variable coefficients, unsorted columns, the reference's quirky restriction of a non-cube, odd
    extents, other smoothers: no plane passes, results as before."""
    shape = (8, 8, 8)
    A, R = hierarchy(shape, 2)
    rng = np.random.default_rng(9)
    Av = sp.csr_matrix(A[0].multiply(sp.csr_matrix(1.0 + 0.1 * rng.random(A[0].shape))) + sp.diags(np.full(A[0].shape[0], 3.0)))
    Av.sort_indices()
    Ac = sp.csr_matrix((R[0] @ Av) @ R[0].T)
    with _hip.Hierarchy([Av, Ac], R, smoother="colour") as h:
        assert not h.level_flags(0)["plane"]
    Au = sp.csr_matrix(A[0])
    for i in range(0, Au.shape[0], 7):          # reverse the stored order of some rows
        s, e = Au.indptr[i], Au.indptr[i + 1]
        Au.indices[s:e] = Au.indices[s:e][::-1].copy()
        Au.data[s:e] = Au.data[s:e][::-1].copy()
    with _hip.Hierarchy([Au, A[1]], R, smoother="colour") as h:
        assert not h.level_flags(0)["plane"]
    for smoother in ("gs", "jacobi"):
        with _hip.Hierarchy(A, R, smoother=smoother) as h:
            assert not h.level_flags(0)["plane"]
    shape = (8, 12, 16)                         # openmg/operators.py:46,78 (Q6): not the plain aggregation
    A0 = operators.stencil_poisson(shape)
    Rq = operators.restrictionList(shape, 1, 8)
    Aq = operators.coeffecientList(A0, Rq)
    with _hip.Hierarchy(Aq, Rq, smoother="colour") as h:
        assert not h.level_flags(0)["plane"]
    shape = (6, 6, 7)
    A0 = operators.stencil_poisson(shape)
    Ro = operators.restrictionList(shape, 1, 8)
    Ao = operators.coeffecientList(A0, Ro)
    with _hip.Hierarchy(Ao, Ro, smoother="colour") as h:
        assert not h.level_flags(0)["plane"]


def test_plane_full_size_properties():
    """BASELINE configs[2] (256^3, 5 grids): the device norm equals SciPy's ||b - A x|| of the fetched
    iterate, exact linearity (cycle(2b) == 2 cycle(b) bitwise), and plane on / off give one iterate."""
    n = 256
    shape = (n, n, n)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, 3, 8)
    A = operators.coeffecientList(A0, R)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        assert all(h.level_flags(l)["plane"] for l in range(4))
        norms, x = run(h, b, 1, 1, 3)
        assert abs(norms[-1] - np.linalg.norm(b - A0 @ x)) <= 1e-10 * norms[-1]
        assert norms[0] > norms[1] > norms[2]
        norms2, x2 = run(h, 2.0 * b, 1, 1, 3)
        assert np.array_equal(x2, 2.0 * x) and norms2 == [2.0 * v for v in norms]
        h.use_plane(False)
        norms3, x3 = run(h, b, 1, 1, 3)
        assert np.array_equal(x3, x) and close(norms3, norms)


@pytest.mark.parametrize("shape,grids", [((48, 40, 72), 4), ((64, 64, 64), 5), ((16, 24, 20), 3)])
def test_small_level_kernels_are_interchangeable_bit_for_bit(monkeypatch, shape, grids):
    """Levels of <= 64^3 cells below the finest: the block-in-LDS kernel (default), the marching kernel with two
    steps of lookahead, the marching kernel with one — and the set schedule: one iterate."""
    A, R = hierarchy(shape, grids, scale=0.61)
    rng = np.random.default_rng(21)
    b = rng.standard_normal(A[0].shape[0])
    x0 = rng.standard_normal(A[0].shape[0])
    results = {}
    for name, env in (("sets", {"OMG_PLANE": "0"}), ("block", {}), ("march la2", {"OMG_PLANE_BLOCK": "0"}),
                      ("march la1", {"OMG_PLANE_BLOCK": "0", "OMG_PLANE_LA2": "0"})):
        for k in ("OMG_PLANE", "OMG_PLANE_BLOCK", "OMG_PLANE_LA2"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with _hip.Hierarchy(A, R, smoother="colour") as h:
            results[name] = run(h, b, 1, 1, 3, x0)
    for name in ("block", "march la2", "march la1"):
        assert np.array_equal(results[name][1], results["sets"][1]), (name, int(np.sum(results[name][1] != results["sets"][1])))
        assert close(results[name][0], results["sets"][0])


def test_up_pass_marching_down_gives_the_bits_of_the_one_marching_up(monkeypatch):
    """The up pass of a whole grid marches from the last plane to the first (plane_kernel MIRROR: it starts where the down
    pass ended); OMG_PLANE_MIRROR=0 marches it upwards like the down pass.  A colour's sweep does not depend on the order of
    its cells: the same iterate bit for bit for every sweep count, the norm to rounding (its squares are added in another
    order).  Shapes with odd hx, several chunks in z and tiles that overhang the grid."""
    for shape, grids in (((64, 64, 64), 3), ((20, 12, 24), 2), ((40, 24, 72), 3), ((128, 128, 128), 4)):
        A, R = hierarchy(shape, grids)
        rng = np.random.default_rng(3)
        b = A[0] @ rng.random(A[0].shape[0])
        x0 = rng.standard_normal(A[0].shape[0])
        for dtype in ("float64", "float32"):
            with _hip.Hierarchy(A, R, smoother="colour", dtype=dtype) as h:
                assert h.level_flags(0)["plane"]
                for pre, post in ((1, 1), (1, 0), (2, 2)):
                    monkeypatch.setenv("OMG_PLANE_MIRROR", "1")
                    down = run(h, b, pre, post, 3, x0)
                    monkeypatch.setenv("OMG_PLANE_MIRROR", "0")
                    up = run(h, b, pre, post, 3, x0)
                    assert np.array_equal(down[1], up[1]), (shape, dtype, pre, post)
                    assert close(down[0], up[0], 1e-6 if dtype == "float32" else 1e-12), (shape, dtype, down[0], up[0])
