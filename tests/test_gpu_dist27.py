"""BASELINE configs[4] on slabs (openmg_amd/csrc/dist27.hip): the 27-point octant-layout kernels on a rank's extended
slab — ghost aggregate planes, colours 0 .. 3 of the upper ghost plane relaxed redundantly, one exchange per sweep, the
Galerkin products per rank on the device — as LOOPBACK groups of 1, 2, 4 and 8 slabs on one GPU: the iterate bit for bit
the single-GPU hierarchy's (which runs the same kernels on the whole grid: no OMG_STENCIL27=0), the coarse operators
bit for bit the global products', and against the CPU oracle (openmg/__init__.py:151-236) at BASELINE's 1e-10."""
import numpy as np
import pytest
import scipy.sparse as sp

from openmg_amd import _hip, _hip_dist, dist, operators
from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu


def single_gpu(shape, grids, dtype):
    A0 = operators.stencil27_variable(shape)
    R = operators.restrictionList(shape, grids - 2, 1)
    assert len(R) == grids - 1
    A = operators.coeffecientList(A0, R)
    return A, R, _hip.Hierarchy(A, R, smoother="colour", dtype=dtype)


def slab_group(shape, world, n_levels, grids, dtype, p2p=0):
    plane = shape[1] * shape[2]
    per = shape[0] // world
    ranks = [_hip_dist.Slab27Rank(r, world, shape, dist.stencil27_variable_rows(shape, r * per * plane, (r + 1) * per * plane), n_levels,
                                  dtype=dtype) for r in range(world)]
    coarse = dist.assemble_coarse([r.coarse_rows() for r in ranks])
    tshape = tuple(s >> n_levels for s in shape)
    tails = [dist.make_tail(coarse, tshape, grids - n_levels, smoother="colour", dtype=dtype) for _ in ranks]
    for r, t in zip(ranks, tails):
        r.set_tail(t)
    return _hip_dist.Slab27Group(ranks, p2p=p2p), tails, coarse


def close(a, b, tol=1e-12):
    return all(abs(u - v) <= tol * abs(v) for u, v in zip(a, b))


CASES = [((16, 16, 16), 3, 1, 2), ((16, 16, 16), 3, 2, 2), ((16, 16, 16), 3, 4, 2), ((16, 16, 16), 3, 8, 1),
         ((32, 16, 32), 4, 4, 3), ((32, 16, 32), 4, 8, 2), ((32, 24, 32), 3, 2, 2)]


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,grids,world,n_levels", CASES)
def test_slab_groups_have_the_bits_of_the_single_gpu_hierarchy(shape, grids, world, n_levels, dtype):
    A, R, h = single_gpu(shape, grids, dtype)
    assert all(h.level_flags(l)["stencil27"] for l in range(grids - 1))
    g, tails, coarse = slab_group(shape, world, n_levels, grids, dtype)
    try:
        # the operator below the slabs, made from the ranks' own rows on the device: the bits of the global product
        want = sp.csr_matrix(A[n_levels])
        want.sort_indices()
        got = sp.csr_matrix(coarse)
        got.sort_indices()
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices) and np.array_equal(got.data, want.data)
        rng = np.random.default_rng(11)
        b = A[0] @ rng.random(A[0].shape[0])
        x0 = rng.standard_normal(A[0].shape[0])
        if dtype == "float32":
            b, x0 = b.astype(np.float32).astype(np.float64), x0.astype(np.float32).astype(np.float64)
        per = b.size // world
        for pre, post in ((1, 1), (1, 0), (2, 1), (0, 1), (0, 0), (2, 2)):
            h.resident_load(b, x0)
            want_norms = h.resident_cycles(pre, post, 3)
            want_x = h.resident_fetch()
            for r in g.ranks:
                r.load(b[r.rank * per:(r.rank + 1) * per], x0[r.rank * per:(r.rank + 1) * per])
            norms = g.cycles(pre, post, 3)
            x = np.concatenate([r.fetch() for r in g.ranks])
            assert np.array_equal(x, want_x), (shape, world, dtype, pre, post, int(np.sum(x != want_x)))
            assert close(norms, want_norms, 1e-12 if dtype == "float64" else 1e-6), (pre, post, norms, want_norms)
            if world > 1:
                assert g.ranks[1].info()["exchanges_last_call"] == 3 * ((pre + post) + (n_levels - 1) * (1 + pre + post))
    finally:
        g.close()
        for t in tails:
            t.close()
        h.close()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,grids,world,n_levels", [((16, 16, 16), 3, 2, 2), ((32, 16, 32), 4, 4, 3), ((32, 16, 32), 4, 8, 2), ((32, 24, 32), 3, 2, 2)])
def test_peer_store_exchanges_have_the_bits_of_the_single_gpu_hierarchy(shape, grids, world, n_levels, dtype):
    """Round 6 ("also give dist27 the peer-store exchange PlaneDist has"): the halo exchanges as stores into
    the neighbours' ghost planes ordered by flags — permission, push, landed — instead of copies / grouped send-recv; the
    same planes, the same bits, the same number of exchanges."""
    A, R, h = single_gpu(shape, grids, dtype)
    g, tails, _ = slab_group(shape, world, n_levels, grids, dtype, p2p=1)
    try:
        rng = np.random.default_rng(11)
        b = A[0] @ rng.random(A[0].shape[0])
        x0 = rng.standard_normal(A[0].shape[0])
        if dtype == "float32":
            b, x0 = b.astype(np.float32).astype(np.float64), x0.astype(np.float32).astype(np.float64)
        per = b.size // world
        for pre, post in ((1, 1), (1, 0), (2, 1), (0, 0)):
            h.resident_load(b, x0)
            want_norms = h.resident_cycles(pre, post, 3)
            want_x = h.resident_fetch()
            for r in g.ranks:
                r.load(b[r.rank * per:(r.rank + 1) * per], x0[r.rank * per:(r.rank + 1) * per])
            norms = g.cycles(pre, post, 3)
            x = np.concatenate([r.fetch() for r in g.ranks])
            assert np.array_equal(x, want_x), (shape, world, dtype, pre, post, int(np.sum(x != want_x)))
            assert close(norms, want_norms, 1e-12 if dtype == "float64" else 1e-6)
            assert g.ranks[1].info()["exchanges_last_call"] == 3 * ((pre + post) + (n_levels - 1) * (1 + pre + post))
    finally:
        g.close()
        for t in tails:
            t.close()
        h.close()


def test_a_peer_store_wait_gives_up_instead_of_hanging(monkeypatch):
    """A neighbour that never answers (its buffers mapped, peer mode on, but it does not run the cycle): the bounded wait
    (OMG_P2P_SPIN polls) ends, the call raises."""
    monkeypatch.setenv("OMG_P2P_SPIN", "2000")
    shape, world = (16, 16, 16), 2
    plane, per = 256, 8
    ranks = [_hip_dist.Slab27Rank(r, world, shape, dist.stencil27_variable_rows(shape, r * per * plane, (r + 1) * per * plane), 1) for r in range(world)]
    coarse = dist.assemble_coarse([r.coarse_rows() for r in ranks])
    tails = [dist.make_tail(coarse, (8, 8, 8), 2, smoother="colour") for _ in ranks]
    for r, t in zip(ranks, tails):
        r.set_tail(t)
    g = _hip_dist.Slab27Group(ranks, p2p=1)                 # (neighbour rows exchanged, buffers mapped)
    try:
        b = np.random.default_rng(2).random(shape[0] * plane)
        for r in ranks:
            r.load(b[r.rank * per * plane:(r.rank + 1) * per * plane])
        with pytest.raises((RuntimeError, _hip.HipError)):
            ranks[0].cycles(1, 1, 1)                        # rank 1 does not take part
    finally:
        g.close()
        for t in tails:
            t.close()


@pytest.mark.parametrize("pre,post", [(1, 1), (1, 0)])
def test_eight_slabs_against_the_oracle(pre, post):
    """16^3, 3 grids, fp64, 8 slabs of two planes: every cycle's norm within BASELINE's 1e-10 of the oracle's."""
    shape, grids, world, n_levels = (16, 16, 16), 3, 8, 1
    A0 = operators.stencil27_variable(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    Ro = orc.restriction_list(shape, grids - 2, 1)
    Ao = orc.coefficient_list(A0, Ro)
    sm = orc.make_smoother("colour", Ao)
    p = {"preIterations": pre, "postIterations": post, "coarsestLevel": len(Ro)}
    g, tails, _ = slab_group(shape, world, n_levels, grids, "float64")
    try:
        per = b.size // world
        for r in g.ranks:
            r.load(b[r.rank * per:(r.rank + 1) * per])
        norms = g.cycles(pre, post, 3)
        xo = None
        for k in range(3):
            xo, info = orc.mg_cycle(Ao, b, 0, Ro, p, initial=xo, smoother=sm)
            assert abs(norms[k] - info["norm"]) <= 1e-10 * info["norm"], (k, norms[k], info["norm"])
        x = np.concatenate([r.fetch() for r in g.ranks])
        np.testing.assert_allclose(x, xo, rtol=1e-9, atol=1e-12)
    finally:
        g.close()
        for t in tails:
            t.close()


def test_rows_that_are_not_the_stencil_are_refused():
    shape = (16, 16, 16)
    rows = dist.stencil27_variable_rows(shape, 0, 8 * 256).tolil()
    rows[300, 301] = 0.0
    bad = sp.csr_matrix(rows)
    bad.eliminate_zeros()
    with pytest.raises(_hip.HipError):
        _hip_dist.Slab27Rank(0, 2, shape, bad, 1)
    with pytest.raises(_hip.HipError):                           # six planes per rank, then three: an aggregate would straddle the ranks
        _hip_dist.Slab27Rank(0, 2, (12, 16, 16), dist.stencil27_variable_rows((12, 16, 16), 0, 6 * 256), 2)
