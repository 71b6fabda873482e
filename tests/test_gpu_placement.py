"""Where the large arrays of a level lie in HBM is searched by timing at setup (hierarchy.hip place_finest_pool,
Stencil27Plan::place_tiles) — candidates from hipMalloc and from scattered pieces mapped in a shuffled
order (HIP's virtual-memory API).  Whatever the search keeps, the numbers must not move: the same bits in the iterate and
the same norms with every kind of candidate forced, for the plane passes (openmg/__init__.py:199-227 on the 7-point
hierarchy) and the 27-point kernels."""
import numpy as np
import pytest

from openmg_amd import _hip, operators

pytestmark = pytest.mark.gpu


def run(A0, shape, restrictions, dtype):
    rng = np.random.default_rng(3)
    b = A0 @ rng.random(A0.shape[0])
    with _hip.Hierarchy.from_fine(A0, shape, restrictions, "colour", dtype=dtype) as h:
        h.resident_load(b)
        norms = h.resident_cycles(1, 1, 3)
        return norms, h.resident_fetch()


@pytest.mark.parametrize("kind", ["plane", "stencil27"])
def test_every_kind_of_placement_gives_the_same_bits(monkeypatch, kind):
    shape = (256, 128, 256)                                              # 8.4 M unknowns: the search starts at 8 M
    A0 = operators.stencil_poisson(shape) if kind == "plane" else operators.stencil27_variable(shape)
    dtype = "float64" if kind == "plane" else "float32"
    monkeypatch.setenv("OMG_POOL_TRIALS", "1")
    monkeypatch.setenv("OMG_S27_TRIALS", "1")
    want = run(A0, shape, 3, dtype)                                      # no search: what hipMalloc gives
    for place in ("0", "2", "32"):
        monkeypatch.setenv("OMG_POOL_PLACE", place)
        monkeypatch.setenv("OMG_POOL_TRIALS", "3")
        monkeypatch.setenv("OMG_S27_TRIALS", "3")
        monkeypatch.setenv("OMG_PLACE_KEEP_LAST", "1")                   # the third candidate is what the cycles run on
        got = run(A0, shape, 3, dtype)
        assert got[0] == want[0], (kind, place)
        assert np.array_equal(got[1], want[1]), (kind, place)
    monkeypatch.delenv("OMG_POOL_PLACE")
    monkeypatch.delenv("OMG_PLACE_KEEP_LAST")


def test_scattered_candidates_are_given_back(monkeypatch):
    """Every candidate of the placement searches — hipMalloc or scattered pieces mapped with the virtual-memory API — is
    device memory again once the hierarchy is gone: free memory (hipMemGetInfo) after create / destroy with scattered
    candidates forced is what it was (round 5's ADVICE: one hipMemUnmap over many mappings need not undo any)."""
    shape = (256, 128, 256)
    A0 = operators.stencil_poisson(shape)
    monkeypatch.setenv("OMG_POOL_PLACE", "2")
    monkeypatch.setenv("OMG_POOL_TRIALS", "3")
    monkeypatch.setenv("OMG_PLACE_KEEP_LAST", "1")
    run(A0, shape, 3, "float64")                          # (code object, streams, caches: allocated once)
    free0 = _hip.device_mem_info()[0]
    for _ in range(3):
        run(A0, shape, 3, "float64")
    free1 = _hip.device_mem_info()[0]
    # three hierarchies x three candidates of 3 x 67 MB: a leak would be >= 600 MB
    assert free0 - free1 < 64 << 20, (free0, free1)
