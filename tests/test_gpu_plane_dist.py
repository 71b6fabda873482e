"""The plane-pipelined slab runner (omg_pdist_*, csrc/dist.hip): all ranks of a 1-D decomposition in one process
on one GPU — the launches and exchanges of the multi-GPU schedule with device copies in place of RCCL — against
the single-GPU hierarchy: the iterate bit for bit, for 1, 2, 4 and 8 slabs; and a one-rank RCCL communicator
through the real collective calls."""
import numpy as np
import pytest

from openmg_amd import _hip, _hip_dist, operators

pytestmark = pytest.mark.gpu


def problem(shape, grids):
    from test_gpu_plane import hierarchy                # plain 2x2x2 aggregation for any even shape
    A, R = hierarchy(shape, grids)
    rng = np.random.default_rng(31)
    return A, R, A[0] @ rng.random(A[0].shape[0]), rng.standard_normal(A[0].shape[0])


def slabs(A, R, shape, n_ranks, n_dist, b, x0):
    """n_dist distributed levels, the rest replicated; every rank loaded with its planes."""
    shapes = [tuple(s >> l for s in shape) for l in range(len(A))]
    coef = [_hip_dist.star_coefficients(A[l], shapes[l]) for l in range(n_dist)]
    w = float(R[0].data[0])
    ranks = []
    per = b.size // n_ranks
    for r in range(n_ranks):
        tail = _hip.Hierarchy(A[n_dist:], R[n_dist:], smoother="colour")
        d = _hip_dist.PlaneDistRank(r, n_ranks, shape, coef, w, tail)
        d.load(b[r * per:(r + 1) * per], None if x0 is None else x0[r * per:(r + 1) * per])
        ranks.append(d)
    return ranks


@pytest.mark.parametrize("shape,grids,n_dist", [((32, 32, 32), 4, 2), ((64, 32, 48), 4, 2), ((32, 16, 16), 3, 1), ((64, 64, 64), 5, 3)])
def test_plane_slabs_have_the_bits_of_the_single_gpu_cycle(shape, grids, n_dist):
    A, R, b, x0 = problem(shape, grids)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        assert all(h.level_flags(l)["plane"] for l in range(len(R)))
        h.resident_load(b, x0)
        want_norms = [h.resident_cycle(1, 1) for _ in range(3)]
        want = h.resident_fetch()
    for n_ranks in (1, 2, 4, 8):
        if (shape[0] >> (n_dist - 1)) // n_ranks < 2:
            continue                                   # every distributed level needs two planes per rank
        g = _hip_dist.PlaneDistGroup(slabs(A, R, shape, n_ranks, n_dist, b, x0))
        try:
            norms = g.cycles(2) + g.cycles(1)
            got = np.concatenate([r.fetch() for r in g.ranks])
        finally:
            g.close()
        assert np.array_equal(got, want), (shape, n_ranks, int(np.sum(got != want)))
        np.testing.assert_allclose(norms, want_norms, rtol=1e-13)


@pytest.mark.parametrize("pre,post", [(1, 0), (0, 1), (0, 0)])
def test_plane_slabs_run_the_reference_default_cycle(pre, post):
    """V(1, 0) — the reference's default, openmg/__init__.py:22-23 —, V(0, 1), V(0, 0) on slabs: the passes without their
    relaxation (plane.hip SWEEP = false) over the same exchanges; the bits of the single-GPU cycle.  More than one sweep
    a side is refused (the set-by-set slab runner has them)."""
    shape, grids, n_dist = (32, 32, 32), 4, 2
    A, R, b, x0 = problem(shape, grids)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        h.resident_load(b, x0)
        want_norms = [h.resident_cycle(pre, post) for _ in range(3)]
        want = h.resident_fetch()
    for n_ranks in (1, 2, 4, 8):
        g = _hip_dist.PlaneDistGroup(slabs(A, R, shape, n_ranks, n_dist, b, x0))
        try:
            norms = g.cycles(2, pre, post) + g.cycles(1, pre, post)
            got = np.concatenate([r.fetch() for r in g.ranks])
            if n_ranks == 2:
                with pytest.raises(_hip.HipError):
                    g.cycles(1, 2, 1)
        finally:
            g.close()
        assert np.array_equal(got, want), (pre, post, n_ranks, int(np.sum(got != want)))
        np.testing.assert_allclose(norms, want_norms, rtol=1e-13)


def test_gated_passes_have_the_bits_of_stream_ordered_exchanges_and_a_missing_exchange_raises(monkeypatch):
    """Round 5: the finest level's passes as ONE launch whose edge chunks wait on a device flag while the exchange of their
    ghost planes runs on the side stream beside the inner chunks (OMG_PDIST_GATE, default on).  Same bits as the
    exchanges in stream order and as the single-GPU cycle; and when the flag is never raised (OMG_PDIST_GATE_POISON=1:
    the exchange 'did not happen') the bounded wait gives up and the call RAISES instead of hanging the device."""
    shape, grids, n_dist = (192, 64, 64), 4, 2                   # (a pass gates on slabs of >= 96 planes: PlanePlan::can_gate)
    monkeypatch.setenv("OMG_PLANE_TILE", "64,16,16")           # (workgroups of four waves: slabs this small otherwise take two-wave tiles, which do not gate)
    A, R, b, x0 = problem(shape, grids)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        h.resident_load(b, x0)
        want_norms = [h.resident_cycle(1, 1) for _ in range(5)]
        want = h.resident_fetch()
    for gate in ("1", "0"):
        monkeypatch.setenv("OMG_PDIST_GATE", gate)
        for n_ranks in (2, 4):
            g = _hip_dist.PlaneDistGroup(slabs(A, R, shape, n_ranks, n_dist, b, x0))
            assert g.ranks[0].info()["gated"] == (int(gate) if n_ranks == 2 else 0), g.ranks[0].info()
            try:
                norms = g.cycles(3) + g.cycles(2)
                got = np.concatenate([r.fetch() for r in g.ranks])
            finally:
                g.close()
            assert np.array_equal(got, want), (gate, n_ranks, int(np.sum(got != want)))
            np.testing.assert_allclose(norms, want_norms, rtol=1e-13)
    monkeypatch.setenv("OMG_PDIST_GATE", "1")
    monkeypatch.setenv("OMG_P2P_SPIN", "2000")
    g = _hip_dist.PlaneDistGroup(slabs(A, R, shape, 2, n_dist, b, x0))
    try:
        assert g.cycles(1) == want_norms[:1]
        monkeypatch.setenv("OMG_PDIST_GATE_POISON", "1")
        g.cycles(1)                                             # (this cycle's own ghost planes were posted before the switch)
        with pytest.raises(RuntimeError, match="bounded wait"):
            g.cycles(1)
    finally:
        monkeypatch.delenv("OMG_PDIST_GATE_POISON")
        g.close()


@pytest.mark.parametrize("mode", ["1", "2"])
def test_split_passes_have_the_bits_of_whole_passes(monkeypatch, mode):
    """OMG_PDIST_SPLIT: every slab pass as two launches — the slab's first and last four planes on the (priority) side
    stream with the exchanges behind them, the planes between on the main stream (off by default: it costs a rank more
    device time than the exchanges it hides).  Same bits; 2: also for a single slab."""
    monkeypatch.setenv("OMG_PDIST_SPLIT", mode)
    shape, grids, n_dist = (64, 32, 48), 4, 2
    A, R, b, x0 = problem(shape, grids)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        h.resident_load(b, x0)
        want_norms = [h.resident_cycle(1, 1) for _ in range(4)]
        want = h.resident_fetch()
    for n_ranks in (1, 2, 4):
        g = _hip_dist.PlaneDistGroup(slabs(A, R, shape, n_ranks, n_dist, b, x0))
        try:
            norms = g.cycles(3) + g.cycles(1)
            got = np.concatenate([r.fetch() for r in g.ranks])
        finally:
            g.close()
        assert np.array_equal(got, want), (mode, n_ranks, int(np.sum(got != want)))
        np.testing.assert_allclose(norms, want_norms, rtol=1e-13)


def test_one_rank_through_the_rccl_calls():
    """A one-rank communicator: omg_pdist_cycles runs the schedule's RCCL path (all-gather of the level below the
    slabs, all-reduce of the norm are skipped for one rank; connect / load / cycles / fetch are the calls bench.py
    makes per rank)."""
    shape, grids, n_dist = (32, 32, 32), 4, 2
    A, R, b, x0 = problem(shape, grids)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        h.resident_load(b)
        want_norms = [h.resident_cycle(1, 1) for _ in range(3)]
        want = h.resident_fetch()
    (d,) = slabs(A, R, shape, 1, n_dist, b, None)
    try:
        d.connect(_hip_dist.rccl_unique_id(), _hip_dist.rccl_unique_id())
        assert d.rccl_ranks() == 1
        d.load(b)
        norms = d.cycles(3)
        assert np.array_equal(d.fetch(), want)
        np.testing.assert_allclose(norms, want_norms, rtol=1e-13)
    finally:
        d.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_slabs_at_the_bench_shapes_have_the_bits_of_one_slab(world):
    """bench.py --gpus 8: (512, 512, 512), three slab levels of 64 / 32 / 16 planes per rank above a replicated 64^3
    hierarchy (--gpus 2 and 4: dist_bench.SHAPES).  Too large for the oracle: the size-independent property is that
    the decomposition does not change a bit — the slabs with ghost planes against ONE slab holding every plane,
    same launches otherwise."""
    from openmg_amd import dist_bench
    shape, n_levels, tail_grids = dist_bench.SHAPES[world], 3, 4
    assert dist_bench.plane_levels(shape, world, 4) == n_levels
    coef = [[v / 16.0 ** l for v in (-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0)] for l in range(n_levels)]
    tshape = tuple(s >> n_levels for s in shape)
    At = operators.stencil_poisson(tshape) / 16.0 ** n_levels
    Rt = operators.restrictionList(tshape, tail_grids - 2, 8)
    At = operators.coeffecientList(At, Rt)
    b = np.random.default_rng(5).random(shape[0] * shape[1] * shape[2])
    out = {}
    for n_ranks in (1, world):
        per = b.size // n_ranks
        ranks = []
        for r in range(n_ranks):
            d = _hip_dist.PlaneDistRank(r, n_ranks, shape, coef, 0.125, _hip.Hierarchy(At, Rt, smoother="colour"))
            d.load(b[r * per:(r + 1) * per])
            ranks.append(d)
        g = _hip_dist.PlaneDistGroup(ranks)
        try:
            norms = g.cycles(2)
            out[n_ranks] = (norms, np.concatenate([r.fetch() for r in g.ranks]))
        finally:
            g.close()
    assert np.array_equal(out[1][1], out[world][1])
    np.testing.assert_allclose(out[1][0], out[world][0], rtol=1e-13)     # per-slab partial sums, added in rank order
    assert out[1][0][1] < out[1][0][0]                  # and the norm goes down


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("shape,grids,n_dist", [((32, 32, 32), 4, 1), ((64, 32, 48), 4, 2), ((64, 64, 64), 5, 3)])
def test_peer_store_slabs_have_the_bits_of_the_single_gpu_cycle(shape, grids, n_dist, mode):
    """Peer mode (omg_pdist_p2p_*): the passes store their boundary planes into the neighbours' ghost planes and wait
    for each other's flags; no exchange launches.  All ranks in one process, so the peers are plain pointers."""
    A, R, b, x0 = problem(shape, grids)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        h.resident_load(b, x0)
        want_norms = [h.resident_cycle(1, 1) for _ in range(4)]
        want = h.resident_fetch()
    for n_ranks in (1, 2, 4, 8):
        if (shape[0] >> (n_dist - 1)) // n_ranks < 4:
            continue                                   # peer mode: four planes per rank on every distributed level
        g = _hip_dist.PlaneDistGroup(slabs(A, R, shape, n_ranks, n_dist, b, x0), p2p=mode)
        try:
            norms = g.cycles(3) + g.cycles(1)          # (two batches: the ghost planes are handed over again)
            got = np.concatenate([r.fetch() for r in g.ranks])
        finally:
            g.close()
        assert np.array_equal(got, want), (shape, n_ranks, int(np.sum(got != want)))
        np.testing.assert_allclose(norms, want_norms, rtol=1e-13)


def test_peer_store_wait_gives_up_instead_of_hanging():
    """A rank whose neighbour never writes: the wait is bounded, the status word says so, the device is not hung."""
    import os
    shape, grids, n_dist = (32, 32, 32), 4, 1
    A, R, b, x0 = problem(shape, grids)
    os.environ["OMG_P2P_SPIN"] = "2000"
    try:
        ranks = slabs(A, R, shape, 2, n_dist, b, x0)
    finally:
        del os.environ["OMG_P2P_SPIN"]
    try:
        ranks[0].p2p_local(ranks[1])
        ranks[0].p2p_enable(2)
        with pytest.raises(RuntimeError, match="gave up"):
            ranks[0].cycles(1, reduce=lambda v: v)     # rank 1 never runs: its READY flag never comes
        assert ranks[0].p2p_status() == 0              # (cleared by the read above)
    finally:
        for r in ranks:
            r.close()


@pytest.mark.parametrize("world,shape,grids,n_dist", [(2, (32, 32, 32), 4, 2), (4, (64, 32, 48), 4, 2)])
def test_peer_stores_between_processes(tmp_path, world, shape, grids, n_dist):
    """Real concurrency: one PROCESS per rank (all on this GPU), hipIpc mappings of each other's vectors and flags,
    gloo only as the control plane.  The iterate must have the bits of the single-GPU cycle."""
    import os
    import socket
    import subprocess
    import sys
    A, R, b, x0 = problem(shape, grids)
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        h.resident_load(b, x0)
        want_norms = [h.resident_cycle(1, 1) for _ in range(3)]
        want = h.resident_fetch()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "p2p_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(tmp_path), "x".join(map(str, shape)), str(grids), str(n_dist)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for r, p in enumerate(procs):
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, (r, e[-3000:])
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))["x"] for r in range(world)])
    assert np.array_equal(got, want), int(np.sum(got != want))
    for r in range(world):
        np.testing.assert_allclose(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))["norms"], want_norms, rtol=1e-13)


def test_bench_gpus_2_rehearsal_on_one_gpu():
    """`bench.py --gpus 2` end to end with both rank processes on this GPU (OMG_DIST_SHARED_GPU=1): the launcher, the
    gloo rendezvous, the preflight cycle, hipIpc mappings between the ranks, peer-store exchanges, the JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMG_DIST_SHARED_GPU="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu", "--size", "64", "--steps", "4",
                        "--warmup", "1", "--repeats", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 4
    c = d["config"]
    assert c["ranks_share_one_gpu"] is True and c["runner"].startswith("plane slabs")
    tail = c["norms_last_region_tail"]
    assert all(np.isfinite(tail)) and tail[-1] < tail[0]
