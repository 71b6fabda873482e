"""omg_hierarchy_update_fine: new coefficients for a 27-point hierarchy that was set up on the device (BASELINE configs[4]:
"Galerkin RAP rebuilt on-device") — level 0 re-tiled, every Galerkin product re-formed by the closed-form streaming kernel
(csrc/stencil27.hip s27_rap_kernel), the coarsest operator factorised anew.  The reference would run
operators.coeffecientList again (openmg/operators.py:144-188); the updated hierarchy must be the hierarchy a fresh setup
with the new operator gives: the same iterate bit for bit, cycle after cycle — which holds only if every level's
coefficients, i.e. every Galerkin product of the chain, came out with the bits of the generic SciPy-order kernel."""
import ctypes

import numpy as np
import pytest

from openmg_amd import _hip, operators

pytestmark = pytest.mark.gpu


def run(h, b, x0, pre, post, cycles):
    h.resident_load(b, x0)
    return [h.resident_cycle(pre, post) for _ in range(cycles)], h.resident_fetch()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,grids", [((16, 16, 16), 3), ((32, 32, 32), 4), ((16, 24, 16), 3), ((8, 16, 8), 3), ((12, 8, 12), 3)])
def test_updated_hierarchy_is_the_freshly_built_one(shape, grids, dtype):
    A1 = operators.stencil27_variable(shape, seed=1)
    A2 = operators.stencil27_variable(shape, seed=2)
    assert np.array_equal(A1.indptr, A2.indptr) and np.array_equal(A1.indices, A2.indices)
    rng = np.random.default_rng(3)
    b = A2 @ rng.random(A2.shape[0])
    x0 = rng.standard_normal(A2.shape[0])
    if dtype == "float32":
        b, x0 = b.astype(np.float32).astype(np.float64), x0.astype(np.float32).astype(np.float64)
    with _hip.Hierarchy.from_fine(A2, shape, grids - 1, "colour", dtype=dtype) as fresh, \
            _hip.Hierarchy.from_fine(A1, shape, grids - 1, "colour", dtype=dtype) as h:
        assert all(h.level_flags(l)["stencil27"] for l in range(grids - 1))
        want = run(fresh, b, x0, 1, 1, 3)
        old = run(h, b, x0, 1, 1, 3)
        assert not np.array_equal(old[1], want[1])
        h.update_fine(A2.data)
        got = run(h, b, x0, 1, 1, 3)
        assert got[0] == want[0] and np.array_equal(got[1], want[1]), (shape, dtype)
        # the set-by-set schedule of the updated hierarchy is built from the NEW operator too
        h.use_plane(False)
        sets = run(h, b, x0, 1, 1, 3)
        assert np.array_equal(sets[1], want[1])
        h.use_plane(True)
        # and back again, from device-resident values
        hip = ctypes.CDLL("libamdhip64.so.7")
        hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipFree.argtypes = [ctypes.c_void_p]
        d = ctypes.c_void_p()
        data = np.ascontiguousarray(A1.data)
        assert hip.hipMalloc(ctypes.byref(d), data.nbytes) == 0
        try:
            assert hip.hipMemcpy(d, data.ctypes.data, data.nbytes, 1) == 0
            h.update_fine((d.value, data.size), on_device=True)
        finally:
            hip.hipFree(d)
        again = run(h, b, x0, 1, 1, 3)
        assert again[0] == old[0] and np.array_equal(again[1], old[1])


def test_update_is_refused_where_it_does_not_apply():
    shape = (16, 16, 16)
    A = operators.stencil27_variable(shape)
    with _hip.Hierarchy.from_fine(A, shape, 2, "colour") as h:
        with pytest.raises(_hip.HipError):
            h.update_fine(A.data[:-1])                           # not the pattern's number of entries
        bad = A.data.copy()
        interior = (5 * 16 + 5) * 16 + 5                         # cell (5, 5, 5): all 27 neighbours, the diagonal is entry 13
        bad[A.indptr[interior] + 13] = 0.0
        with pytest.raises(_hip.HipError):
            h.update_fine(bad)
        h.update_fine(A.data)
    A7 = operators.stencil_poisson(shape)
    with _hip.Hierarchy.from_fine(A7, shape, 2, "colour") as h:  # plane levels: nothing to re-tile
        with pytest.raises(_hip.HipError):
            h.update_fine(A7.data)
