"""Distributed V-cycle on ONE GPU: every rank of a slab decomposition lives in this process
(loopback group, halos moved by device copies) and runs the same schedule as the RCCL path.
Checks that the iterate does not depend on the number of ranks, and that RCCL itself loads
and works for a single-rank communicator.  Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

from openmg_amd import _hip, _hip_dist, dist, operators

pytestmark = pytest.mark.gpu


def single_gpu(shape, grids, smoother, b, cycles, omega=0.8):
    R = [operators.restriction(tuple(s // 2 ** l for s in shape)) for l in range(grids - 1)]
    A = operators.coeffecientList(operators.stencil_poisson(shape), R)
    with _hip.Hierarchy(A, R, smoother=smoother, omega=omega) as h:
        h.resident_load(b)
        norms = [h.resident_cycle(1, 1) for _ in range(cycles)]
        return h.resident_fetch(), norms


def loopback(shape, grids, n_ranks, smoother, b, cycles, omega=0.8):
    part = dist.SlabPartition(shape, n_ranks, grids)
    levels, coarse, counts = dist.build_all_ranks(
        part, lambda q: dist.stencil_rows(shape, *part.rows(0, q)), smoother=smoother)
    ranks = [_hip_dist.DistRank(q, n_ranks, levels[q], coarse, counts, smoother=smoother, omega=omega)
             for q in range(n_ranks)]
    group = _hip_dist.DistGroup(ranks)
    try:
        for q, r in enumerate(ranks):
            r.load(b[slice(*part.rows(0, q))])
        norms = [group.cycle(1, 1) for _ in range(cycles)]
        x = np.concatenate([r.fetch() for r in ranks])
    finally:
        group.close()
    return x, norms


@pytest.mark.parametrize("shape,grids,n_ranks", [((32, 32, 32), 3, 2), ((32, 32, 32), 3, 4),
                                                 ((64, 32, 64), 4, 4), ((64, 64), 3, 2)])
def test_redblack_iterate_independent_of_rank_count(shape, grids, n_ranks):
    N = int(np.prod(shape))
    b = operators.stencil_poisson(shape) @ np.random.default_rng(12345).random(N)
    x1, n1 = single_gpu(shape, grids, "colour", b, 3)
    xd, nd = loopback(shape, grids, n_ranks, "colour", b, 3)
    assert np.array_equal(xd, x1)                      # bitwise: same arithmetic per row
    np.testing.assert_allclose(nd, n1, rtol=1e-13)     # norm: different summation grouping


@pytest.mark.parametrize("smoother", ["gs", "jacobi"])
def test_other_smoothers_loopback(smoother):
    shape, grids = (16, 16, 16), 3
    b = operators.stencil_poisson(shape) @ np.random.default_rng(5).random(4096)
    x1, n1 = single_gpu(shape, grids, smoother, b, 2)
    xd, nd = loopback(shape, grids, 2, smoother, b, 2)
    np.testing.assert_allclose(xd, x1, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(nd, n1, rtol=1e-12)


def test_initial_iterate_and_rccl_single_rank():
    """x0 != 0 needs a halo exchange before the first sweep; a 1-rank communicator exercises
    dlopen(librccl), ncclCommInitRank and the collective code path."""
    shape, grids = (32, 32, 32), 3
    N = 32 ** 3
    rng = np.random.default_rng(9)
    b = operators.stencil_poisson(shape) @ rng.random(N)
    part = dist.SlabPartition(shape, 1, grids)
    levels, coarse, counts = dist.build_all_ranks(part, lambda q: dist.stencil_rows(shape, 0, N), smoother="colour")
    r = _hip_dist.DistRank(0, 1, levels[0], coarse, counts, smoother="colour")
    try:
        r.connect(_hip_dist.rccl_unique_id())
        r.load(b)
        norms = [r.cycle(1, 1) for _ in range(2)]
        x = r.fetch()
    finally:
        r.close()
    x1, n1 = single_gpu(shape, grids, "colour", b, 2)
    assert np.array_equal(x, x1)
    np.testing.assert_allclose(norms, n1, rtol=1e-13)
    # non-zero initial iterate through the loopback group
    x0 = rng.random(N)
    part2 = dist.SlabPartition(shape, 2, grids)
    lv2, c2, k2 = dist.build_all_ranks(part2, lambda q: dist.stencil_rows(shape, *part2.rows(0, q)), smoother="colour")
    ranks = [_hip_dist.DistRank(q, 2, lv2[q], c2, k2, smoother="colour") for q in range(2)]
    g = _hip_dist.DistGroup(ranks)
    try:
        for q, rk in enumerate(ranks):
            sl = slice(*part2.rows(0, q))
            rk.load(b[sl], x0[sl])
        g.cycle(1, 1)
        xd = np.concatenate([rk.fetch() for rk in ranks])
    finally:
        g.close()
    Rl = [operators.restriction(tuple(s // 2 ** l for s in shape)) for l in range(grids - 1)]
    Al = operators.coeffecientList(operators.stencil_poisson(shape), Rl)
    with _hip.Hierarchy(Al, Rl, smoother="colour") as h:
        h.resident_load(b, x0)
        h.resident_cycle(1, 1)
        assert np.array_equal(xd, h.resident_fetch())


@pytest.mark.parametrize("n_ranks,n_dist", [(2, 3), (4, 2), (2, 2)])
def test_replicated_tail_matches_single_gpu(n_ranks, n_dist):
    """Levels below the last distributed one run replicated on every rank (dist.make_tail):
    still the same iterate as one GPU running all 5 grids."""
    shape, grids = (64, 64, 64), 5
    N = int(np.prod(shape))
    b = operators.stencil_poisson(shape) @ np.random.default_rng(77).random(N)
    x1, n1 = single_gpu(shape, grids, "colour", b, 3)
    part = dist.SlabPartition(shape, n_ranks, n_dist)
    levels, coarse, counts = dist.build_all_ranks(
        part, lambda q: dist.stencil_rows(shape, *part.rows(0, q)), smoother="colour")
    ranks = [_hip_dist.DistRank(q, n_ranks, levels[q], None, counts, smoother="colour",
                                tail=dist.make_tail(coarse, part.shapes[-1], grids - n_dist + 1, smoother="colour"))
             for q in range(n_ranks)]
    group = _hip_dist.DistGroup(ranks)
    try:
        for q, r in enumerate(ranks):
            r.load(b[slice(*part.rows(0, q))])
        nd = [group.cycle(1, 1) for _ in range(3)]
        xd = np.concatenate([r.fetch() for r in ranks])
    finally:
        group.close()
    assert np.array_equal(xd, x1)
    np.testing.assert_allclose(nd, n1, rtol=1e-13)


def test_boundary_first_sets_do_not_change_the_iterate(monkeypatch):
    """overlap=True orders every colour as (rows the neighbours need, interior rows) so that
    the exchange can run beside the interior launch; overlap=False keeps plain colour sets.
    Same iterate either way (and the same as one GPU).  (Small levels keep plain sets by
    default; the threshold is lowered here.)"""
    monkeypatch.setenv("OMG_OVERLAP_MIN_ROWS", "0")
    shape, grids, n_ranks = (32, 32, 32), 3, 4
    b = operators.stencil_poisson(shape) @ np.random.default_rng(3).random(32 ** 3)
    x1, n1 = single_gpu(shape, grids, "colour", b, 2)
    part = dist.SlabPartition(shape, n_ranks, grids)
    out = []
    for overlap in (True, False):
        levels, coarse, counts = dist.build_all_ranks(
            part, lambda q: dist.stencil_rows(shape, *part.rows(0, q)), smoother="colour", overlap=overlap)
        assert levels[1][0]["set_group"] == (2 if overlap else 1)
        assert levels[1][0]["n_sets"] == (4 if overlap else 2)
        ranks = [_hip_dist.DistRank(q, n_ranks, levels[q], coarse, counts, smoother="colour") for q in range(n_ranks)]
        group = _hip_dist.DistGroup(ranks)
        try:
            for q, r in enumerate(ranks):
                r.load(b[slice(*part.rows(0, q))])
            norms = [group.cycle(1, 1) for _ in range(2)]
            out.append((np.concatenate([r.fetch() for r in ranks]), norms))
        finally:
            group.close()
    assert np.array_equal(out[0][0], x1) and np.array_equal(out[1][0], x1)
    np.testing.assert_allclose(out[0][1], n1, rtol=1e-13)


def test_two_stream_overlap_schedule_on_one_gpu(monkeypatch):
    """OMG_FORCE_OVERLAP=1 makes the loopback group run the schedule the RCCL path uses on
    large levels: boundary rows relaxed / corrected first, their exchange on a second stream
    beside the interior launch, events between the streams.  Repeated to give a race a chance;
    the iterate must stay bit-identical to one GPU."""
    monkeypatch.setenv("OMG_FORCE_OVERLAP", "1")
    monkeypatch.setenv("OMG_OVERLAP_MIN_ROWS", "0")
    shape, grids, n_ranks = (64, 64, 64), 4, 4
    b = operators.stencil_poisson(shape) @ np.random.default_rng(8).random(64 ** 3)
    x1, n1 = single_gpu(shape, grids, "colour", b, 4)
    for _ in range(3):
        xd, nd = loopback(shape, grids, n_ranks, "colour", b, 4)
        assert np.array_equal(xd, x1)
        np.testing.assert_allclose(nd, n1, rtol=1e-13)


def test_default_thresholds_at_realistic_slab_size(monkeypatch):
    """128^3 over 2 ranks: 1M rows per rank at level 0 — above the default 2^19-row threshold,
    so boundary-first pairs and (forced here, as no RCCL peer exists) the two-stream schedule
    are what the real multi-GPU run takes at its two finest levels; deeper levels keep plain
    colour sets.  Replicated tail below level 2."""
    monkeypatch.setenv("OMG_FORCE_OVERLAP", "1")
    shape, grids, n_ranks, n_dist = (128, 128, 128), 5, 2, 3
    b = operators.stencil_poisson(shape) @ np.random.default_rng(11).random(128 ** 3)
    x1, n1 = single_gpu(shape, grids, "colour", b, 3)
    part = dist.SlabPartition(shape, n_ranks, n_dist)
    levels, coarse, counts = dist.build_all_ranks(
        part, lambda q: dist.stencil_rows(shape, *part.rows(0, q)), smoother="colour")
    assert [levels[0][l].get("set_group") for l in range(n_dist - 1)] == [2, 1]
    assert levels[0][0]["groups"] is not None and len(levels[0][0]["peers"]) == 2      # one peer, two colours
    ranks = [_hip_dist.DistRank(q, n_ranks, levels[q], None, counts, smoother="colour",
                                tail=dist.make_tail(coarse, part.shapes[-1], grids - n_dist + 1))
             for q in range(n_ranks)]
    group = _hip_dist.DistGroup(ranks)
    try:
        for q, r in enumerate(ranks):
            r.load(b[slice(*part.rows(0, q))])
        nd = [group.cycle(1, 1) for _ in range(3)]
        xd = np.concatenate([r.fetch() for r in ranks])
    finally:
        group.close()
    assert np.array_equal(xd, x1)
    np.testing.assert_allclose(nd, n1, rtol=1e-13)


# ------------------------------------------------------------- fp32 levels, 27-point operators --
def _single(A0, shape, grids, b, cycles, dtype):
    R = [operators.restriction(tuple(s // 2 ** l for s in shape)) for l in range(grids - 1)]
    A = operators.coeffecientList(A0, R)
    with _hip.Hierarchy(A, R, smoother="colour", dtype=dtype) as h:
        sets = h.level_sets(0)
        h.resident_load(b)
        norms = [h.resident_cycle(1, 1) for _ in range(cycles)]
        return h.resident_fetch(), norms, sets


def _loopback(rows_of, shape, grids, n_ranks, b, cycles, dtype, colouring="parity", n_dist=None):
    n_dist = grids if n_dist is None else n_dist
    part = dist.SlabPartition(shape, n_ranks, n_dist)
    levels, coarse, counts = dist.build_all_ranks(part, lambda q: rows_of(*part.rows(0, q)), smoother="colour",
                                                  colouring=colouring)
    tail = lambda: None if n_dist == grids else dist.make_tail(coarse, part.shapes[-1], grids - n_dist + 1, dtype=dtype)
    ranks = [_hip_dist.DistRank(q, n_ranks, levels[q], coarse if n_dist == grids else None, counts, smoother="colour",
                                tail=tail(), dtype=dtype) for q in range(n_ranks)]
    group = _hip_dist.DistGroup(ranks)
    try:
        for q, r in enumerate(ranks):
            r.load(b[slice(*part.rows(0, q))])
        norms = [group.cycle(1, 1) for _ in range(cycles)]
        x = np.concatenate([r.fetch() for r in ranks])
    finally:
        group.close()
    return x, norms


@pytest.mark.parametrize("n_ranks,n_dist", [(2, 3), (4, 2)])
def test_fp32_slabs_give_the_single_gpu_fp32_iterate(n_ranks, n_dist):
    """omg_dist_create_ex(OMG_DTYPE_F32): levels, halo messages and the coarse all-gather in
    float.  Rows of a colour are uncoupled, so the fp32 iterate is bit-identical to the
    single-GPU fp32 hierarchy whatever the number of slabs, with a direct coarse solve and with
    a replicated fp32 tail."""
    shape, grids = (32, 32, 32), 3
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(31).random(A0.shape[0])
    x1, n1, _ = _single(A0, shape, grids, b, 3, "float32")
    x64, _, _ = _single(A0, shape, grids, b, 3, "float64")
    xd, nd = _loopback(lambda lo, hi: dist.stencil_rows(shape, lo, hi), shape, grids, n_ranks, b, 3, "float32", n_dist=n_dist)
    assert np.array_equal(xd, x1)
    assert not np.array_equal(xd, x64)                       # it really ran in float
    np.testing.assert_allclose(nd, n1, rtol=1e-6)            # norm: fp32 residuals, double sums in another grouping


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_27_point_variable_coefficient_slabs(monkeypatch, dtype):
    """configs[4]'s operator over slabs: 8 colours by coordinate octant (what the single-GPU
    greedy colouring finds on a 27-point stencil, in the same order), one halo message per
    (neighbour, colour), Galerkin products from the rank's own rows.  Bit-identical to one GPU's set-by-set
    schedule on the operator as stored (OMG_STENCIL27=0: the 27-point kernels of stencil27.hip pad boundary rows
    to 27 entries, which changes how THEIR sums are associated)."""
    monkeypatch.setenv("OMG_STENCIL27", "0")
    shape, grids = (16, 16, 16), 3
    A0 = operators.stencil27_variable(shape)
    b = A0 @ np.random.default_rng(32).random(A0.shape[0])
    x1, n1, sets = _single(A0, shape, grids, b, 3, dtype)
    assert sets == 8
    for n_ranks in (2, 4):
        xd, nd = _loopback(lambda lo, hi: dist.stencil27_variable_rows(shape, lo, hi), shape, grids, n_ranks, b, 3, dtype,
                           colouring="octant")
        assert np.array_equal(xd, x1), n_ranks
        np.testing.assert_allclose(nd, n1, rtol=1e-13 if dtype == "float64" else 1e-6)
    with pytest.raises(ValueError):                          # red-black is not a colouring of a 27-point operator
        _loopback(lambda lo, hi: dist.stencil27_variable_rows(shape, lo, hi), shape, grids, 2, b, 1, dtype)


def test_split_scatter_prolongation_with_overlap(monkeypatch):
    """Long grid lines (R's blocks then hold a handful of row patterns): prolongation runs as a
    scatter over R's rows, split [first coarse planes | interior | last coarse planes] where the
    fine sets are (boundary, interior) pairs, with the exchange on the second stream beside the
    interior part.  Same bits as one GPU."""
    monkeypatch.setenv("OMG_OVERLAP_MIN_ROWS", "0")
    monkeypatch.setenv("OMG_FORCE_OVERLAP", "1")
    shape, grids, n_ranks = (256, 8, 256), 3, 2
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(41).random(A0.shape[0])
    x1, n1, _ = _single(A0, shape, grids, b, 2, "float64")
    part = dist.SlabPartition(shape, n_ranks, grids)
    levels, coarse, counts = dist.build_all_ranks(part, lambda q: dist.stencil_rows(shape, *part.rows(0, q)), smoother="colour")
    ranks = [_hip_dist.DistRank(q, n_ranks, levels[q], coarse, counts, smoother="colour") for q in range(n_ranks)]
    flags = ranks[0].level_flags(0)
    assert flags["paired_sets"] and flags["scatter_prolong"] and flags["split_scatter"], flags
    group = _hip_dist.DistGroup(ranks)
    try:
        for q, r in enumerate(ranks):
            r.load(b[slice(*part.rows(0, q))])
        nd = [group.cycle(1, 1) for _ in range(2)]
        xd = np.concatenate([r.fetch() for r in ranks])
    finally:
        group.close()
    assert np.array_equal(xd, x1)
    np.testing.assert_allclose(nd, n1, rtol=1e-13)


def test_27_point_slabs_with_paired_sets_overlap_and_tail(monkeypatch):
    """All the schedule options at once on the 8-colour operator: (boundary, interior) set pairs
    (16 sets), exchanges on the second stream, a replicated fp32 tail below the first level."""
    monkeypatch.setenv("OMG_OVERLAP_MIN_ROWS", "0")
    monkeypatch.setenv("OMG_FORCE_OVERLAP", "1")
    monkeypatch.setenv("OMG_STENCIL27", "0")                 # (rows summed as stored, on one GPU and in the replicated tail)
    shape, grids = (16, 16, 16), 3
    A0 = operators.stencil27_variable(shape)
    b = A0 @ np.random.default_rng(33).random(A0.shape[0])
    for dtype in ("float64", "float32"):
        x1, n1, _ = _single(A0, shape, grids, b, 2, dtype)
        xd, nd = _loopback(lambda lo, hi: dist.stencil27_variable_rows(shape, lo, hi), shape, grids, 2, b, 2, dtype,
                           colouring="octant", n_dist=2)
        assert np.array_equal(xd, x1), dtype
        np.testing.assert_allclose(nd, n1, rtol=1e-13 if dtype == "float64" else 1e-6)


@pytest.mark.parametrize("smoother,n_ranks,n_dist", [("colour", 4, 3), ("colour", 2, 2), ("jacobi", 2, 3), ("gs", 2, 3)])
def test_batched_cycles_over_slabs_return_every_norm(monkeypatch, smoother, n_ranks, n_dist):
    """omg_dist_group_cycles / omg_dist_cycles: n cycles, n global norms; with two colours (also as
    (boundary, interior) pairs with the two-stream schedule) or Jacobi the norm of cycle k is finished
    by cycle k + 1's first launches.  Same iterate and, up to the grouping of the block sums, same
    norms as single cycles and as one GPU; bitwise equal between the batched and the single-cycle
    slab runs."""
    monkeypatch.setenv("OMG_FORCE_OVERLAP", "1")
    monkeypatch.setenv("OMG_OVERLAP_MIN_ROWS", "0")
    shape, grids = (32, 32, 32), 4
    N = int(np.prod(shape))
    b = operators.stencil_poisson(shape) @ np.random.default_rng(21).random(N)
    x1, n1 = single_gpu(shape, grids, smoother, b, 5)
    out = []
    for how in ("single", "batch"):
        part = dist.SlabPartition(shape, n_ranks, n_dist)
        levels, coarse, counts = dist.build_all_ranks(
            part, lambda q: dist.stencil_rows(shape, *part.rows(0, q)), smoother=smoother)
        ranks = [_hip_dist.DistRank(q, n_ranks, levels[q], None, counts, smoother=smoother, omega=0.8,
                                    tail=dist.make_tail(coarse, part.shapes[-1], grids - n_dist + 1, smoother=smoother, omega=0.8))
                 for q in range(n_ranks)]
        group = _hip_dist.DistGroup(ranks)
        try:
            for q, r in enumerate(ranks):
                r.load(b[slice(*part.rows(0, q))])
            norms = [group.cycle(1, 1) for _ in range(5)] if how == "single" else group.cycles(1, 1, 2) + group.cycles(1, 1, 3)
            out.append((norms, np.concatenate([r.fetch() for r in ranks])))
        finally:
            group.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])
    if smoother == "colour":
        assert np.array_equal(out[1][1], x1)
    else:
        np.testing.assert_allclose(out[1][1], x1, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(out[1][0], n1, rtol=1e-12)
    # one-rank RCCL communicator: the real collective calls of the batched path
    part = dist.SlabPartition(shape, 1, 3)
    levels, coarse, counts = dist.build_all_ranks(part, lambda q: dist.stencil_rows(shape, 0, N), smoother=smoother)
    r = _hip_dist.DistRank(0, 1, levels[0], coarse, counts, smoother=smoother, omega=0.8)
    try:
        r.connect(_hip_dist.rccl_unique_id())
        assert r.rccl_ranks() == 1
        r.load(b)
        norms = r.cycles(1, 1, 3)
        x = r.fetch()
    finally:
        r.close()
    xs, ns = single_gpu(shape, 3, smoother, b, 3)
    np.testing.assert_allclose(norms, ns, rtol=1e-12)
    np.testing.assert_allclose(x, xs, rtol=1e-12, atol=1e-14)
