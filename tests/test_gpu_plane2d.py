"""2-D five-point levels (BASELINE configs[1]'s grids) on the fused tile passes of plane.hip (tile2d_kernel) against
the set-by-set schedule — same bits — and against the CPU oracle with the red-black ordering."""
import numpy as np
import pytest
import scipy.sparse as sp

from openmg_amd import _hip, operators
from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu


def aggregation2(shape):
    mats = []
    for s in shape:
        m = sp.lil_matrix((s // 2, s))
        for i in range(s // 2):
            m[i, 2 * i] = 0.5
            m[i, 2 * i + 1] = 0.5
        mats.append(sp.csr_matrix(m))
    R = sp.csr_matrix(sp.kron(mats[0], mats[1], format="csr"))
    R.sort_indices()
    return R


def hierarchy(shape, grids, scale=1.0):
    A = [sp.csr_matrix(operators.stencil_poisson(shape) * scale)]
    R = []
    sh = tuple(shape)
    for _ in range(grids - 1):
        R.append(aggregation2(sh))
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


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,grids", [((16, 16), 3), ((64, 64), 4), ((10, 22), 2), ((36, 6), 2), ((128, 256), 4), ((70, 98), 2)])
def test_2d_tile_passes_have_the_bits_of_the_set_schedule(shape, grids, dtype):
    A, R = hierarchy(shape, grids, scale=0.83)
    rng = np.random.default_rng(17)
    b = rng.standard_normal(A[0].shape[0])
    x0 = rng.standard_normal(A[0].shape[0])
    if dtype == "float32":
        b, x0 = b.astype(np.float32).astype(np.float64), x0.astype(np.float32).astype(np.float64)
    with _hip.Hierarchy(A, R, smoother="colour", dtype=dtype) as h:
        assert h.level_flags(0)["plane"], shape
        info = h.plane_info(0)
        assert (info["nz"], info["ny"], info["nx"]) == (1,) + tuple(shape)
        for pre, post in ((1, 1), (2, 1), (1, 2), (1, 0), (0, 1), (0, 0)):
            h.use_plane(True)
            got = run(h, b, pre, post, 3, x0)
            h.resident_load(b, x0)
            batch = h.resident_cycles(pre, post, 3)
            xb = h.resident_fetch()
            h.use_plane(False)
            ref = run(h, b, pre, post, 3, x0)
            assert np.array_equal(got[1], ref[1]), (shape, dtype, pre, post, int(np.sum(got[1] != ref[1])))
            assert close(got[0], ref[0], 1e-12), (pre, post, got[0], ref[0])
            assert np.array_equal(xb, got[1]) and batch == got[0]


def test_config1_red_black_on_the_tile_passes_against_the_oracle():
    """1024^2 / 4 grids is configs[1]; here 128^2, 4 grids, red-black, V(1,1) and the reference's default V(1,0)
    against the oracle: 1e-10 on every norm, rtol 1e-9 on the iterate."""
    shape = (128, 128)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, 2, 8)
    A = operators.coeffecientList(A0, R)
    Ro = orc.restriction_list(shape, 2, 8)
    Ao = orc.coefficient_list(A0, Ro)
    sm = orc.make_smoother("colour", Ao)
    for pre, post in ((1, 1), (1, 0)):
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


def test_config1_full_size_properties():
    """configs[1]'s grid itself, 1024^2 / 4 grids, red-black: every level on the tile passes, the device norm equals
    SciPy's of the fetched iterate, exact linearity, and the set schedule gives the same iterate."""
    shape = (1024, 1024)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, 2, 8)
    A = operators.coeffecientList(A0, R)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        assert all(h.level_flags(l)["plane"] for l in range(3))
        norms, x = run(h, b, 1, 1, 3)
        assert abs(norms[-1] - np.linalg.norm(b - A0 @ x)) <= 1e-10 * norms[-1]
        norms2, x2 = run(h, 2.0 * b, 1, 1, 3)
        assert np.array_equal(x2, 2.0 * x) and norms2 == [2.0 * v for v in norms]
        h.use_plane(False)
        norms3, x3 = run(h, b, 1, 1, 3)
        assert np.array_equal(x3, x) and close(norms3, norms)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,grids", [((16, 16), 3), ((64, 64), 4), ((10, 22), 2), ((128, 256), 4), ((70, 98), 2)])
def test_2d_weighted_jacobi_tile_passes_have_the_bits_of_the_set_schedule(shape, grids, dtype):
    """Weighted Jacobi (the smoother BASELINE configs[1] names) on the same fused tile passes, natural ordering."""
    A, R = hierarchy(shape, grids, scale=0.83)
    rng = np.random.default_rng(19)
    b = rng.standard_normal(A[0].shape[0])
    x0 = rng.standard_normal(A[0].shape[0])
    if dtype == "float32":
        b, x0 = b.astype(np.float32).astype(np.float64), x0.astype(np.float32).astype(np.float64)
    with _hip.Hierarchy(A, R, smoother="jacobi", omega=2.0 / 3.0, dtype=dtype) as h:
        assert h.level_flags(0)["plane"] and h.level_sets(0) == 1, shape
        for pre, post in ((1, 1), (2, 1), (1, 2), (1, 0), (0, 1)):
            h.use_plane(True)
            got = run(h, b, pre, post, 3, x0)
            h.resident_load(b, x0)
            batch = h.resident_cycles(pre, post, 3)
            xb = h.resident_fetch()
            h.use_plane(False)
            ref = run(h, b, pre, post, 3, x0)
            assert np.array_equal(got[1], ref[1]), (shape, dtype, pre, post, int(np.sum(got[1] != ref[1])))
            assert close(got[0], ref[0], 1e-12), (pre, post, got[0], ref[0])
            assert np.array_equal(xb, got[1]) and batch == got[0]


def test_3d_jacobi_keeps_the_set_schedule():
    A0 = operators.stencil_poisson((8, 8, 8))
    R = operators.restrictionList((8, 8, 8), 0, 4)
    A = operators.coeffecientList(A0, R)
    with _hip.Hierarchy(A, R, smoother="jacobi", omega=0.7) as h:
        assert not h.level_flags(0)["plane"]
