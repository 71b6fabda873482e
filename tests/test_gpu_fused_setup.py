"""mgSolve's setup kept on the device (omg_hierarchy_create_from_fine: restrictionList + coeffecientList + the levels'
qualification in HBM, openmg/__init__.py:103-109) against the ordinary route through host lists: the same hierarchy,
the same iterate bit for bit."""
import numpy as np
import pytest
import scipy.sparse as sp

import openmg_amd
from openmg_amd import _hip, operators

pytestmark = pytest.mark.gpu


def variable7(shape, rng):
    A = sp.csr_matrix(operators.stencil_poisson(shape))
    A = sp.csr_matrix(A.multiply(sp.csr_matrix(1.0 + 0.1 * rng.random(A.shape))) + sp.diags(np.full(A.shape[0], 3.0)))
    A.sort_indices()
    return A


CASES = [
    ("poisson7 32^3 colour", lambda rng: operators.stencil_poisson((32, 32, 32)), (32, 32, 32), 3, "colour", "float64"),
    ("poisson7 32^3 colour f32", lambda rng: operators.stencil_poisson((32, 32, 32)), (32, 32, 32), 3, "colour", "float32"),
    ("poisson5 64^2 colour", lambda rng: operators.stencil_poisson((64, 64)), (64, 64), 3, "colour", "float64"),
    ("poisson5 64^2 jacobi", lambda rng: operators.stencil_poisson((64, 64)), (64, 64), 3, "jacobi", "float64"),
    ("stencil27 16^3 colour", lambda rng: operators.stencil27_variable((16, 16, 16)), (16, 16, 16), 2, "colour", "float64"),
    ("stencil27 16^3 colour f32", lambda rng: operators.stencil27_variable((16, 16, 16)), (16, 16, 16), 2, "colour", "float32"),
    ("poisson7 16^3 gs (ordinary route inside)", lambda rng: operators.stencil_poisson((16, 16, 16)), (16, 16, 16), 2, "gs", "float64"),
    ("variable7 16^3 colour (ordinary route inside)", lambda rng: variable7((16, 16, 16), rng), (16, 16, 16), 2, "colour", "float64"),
    ("poisson7 16x32x16 colour (non-cube)", lambda rng: operators.stencil_poisson((16, 32, 16)), (16, 32, 16), 2, "colour", "float64"),
]


@pytest.mark.parametrize("name,make,shape,levels,smoother,dtype", CASES, ids=[c[0] for c in CASES])
def test_device_setup_gives_the_hierarchy_of_the_host_route(name, make, shape, levels, smoother, dtype):
    rng = np.random.default_rng(5)
    A0 = sp.csr_matrix(make(rng))
    b = A0 @ rng.random(A0.shape[0])
    p = {"problemShape": shape, "gridLevels": levels, "preIterations": 1, "postIterations": 1, "cycles": 3, "threshold": 0,
         "smoother": smoother, "dtype": dtype, "minSize": 1}
    if smoother == "jacobi":
        p["omega"] = 2.0 / 3.0
    pa, pb = dict(p), dict(p, giveInfo=True)
    assert openmg_amd._device_setup_depth(A0, shape, dict(p, coarsestLevel=levels - 1, giveInfo=False)) == levels
    x_dev = openmg_amd.mgSolve(A0, b, pa)                          # device setup (giveInfo off)
    x_host, info = openmg_amd.mgSolve(A0, b, pb)                   # host lists
    assert pa["coarsestLevel"] == pb["coarsestLevel"] == len(info["R"]) == levels
    assert np.array_equal(x_dev, x_host), (name, int(np.sum(x_dev != x_host)))
    # the handle itself: which path each level takes, and the single-level entry points that build the row-kernel
    # format on first use from the closed-form orderings
    with _hip.Hierarchy.from_fine(A0, shape, levels, smoother=smoother, omega=p.get("omega", 1.0), dtype=dtype) as h, \
            _hip.Hierarchy(info["A"], info["R"], smoother=smoother, omega=p.get("omega", 1.0), dtype=dtype) as g:
        assert h.sizes == g.sizes
        for l in range(levels):
            assert h.level_flags(l)["plane"] == g.level_flags(l)["plane"] and h.level_flags(l)["stencil27"] == g.level_flags(l)["stencil27"]
        x = rng.standard_normal(A0.shape[0])
        if dtype == "float32":
            x = x.astype(np.float32).astype(np.float64)
        assert np.array_equal(h.spmv(0, x), g.spmv(0, x))
        r1, n1 = h.residual(0, b, x, want_norm=True)
        r2, n2 = g.residual(0, b, x, want_norm=True)
        assert np.array_equal(r1, r2) and n1 == n2
        h.resident_load(b)
        g.resident_load(b)
        assert h.resident_cycles(2, 1, 3) == g.resident_cycles(2, 1, 3)
        assert np.array_equal(h.resident_fetch(), g.resident_fetch())


def test_shapes_the_device_setup_leaves_to_the_host_route():
    A = sp.identity(12 ** 3, format="csr")
    assert openmg_amd._device_setup_depth(A, (12, 12, 12), {"coarsestLevel": 3, "minSize": 8}) == 0        # 3 is odd two levels down
    assert openmg_amd._device_setup_depth(A, (12, 12, 12), {"coarsestLevel": 3, "minSize": 8, "giveInfo": True}) == 0
    A = sp.identity(8 * 12 * 16, format="csr")
    assert openmg_amd._device_setup_depth(A, (8, 12, 16), {"coarsestLevel": 1, "minSize": 8}) == 0         # the reference's quirky offsets (Q6)
    assert openmg_amd._device_setup_depth(sp.identity(4096, format="csr"), (4096,), {"coarsestLevel": 2, "minSize": 8}) == 0
