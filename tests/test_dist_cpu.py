"""Host logic of the multi-GPU path on CPU: slab partition, local operators, Galerkin from
local rows, smoother set keys, halo plans — executed by two REAL processes over
torch.distributed/gloo (world_size 2) with NumPy kernels, and compared with the
single-process oracle.  CPU only."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp

from openmg_amd import dist
from oracle import mg_oracle as orc


def scipy_spgemm(X, Y):
    return sp.csr_matrix(sp.csr_matrix(X) @ sp.csr_matrix(Y))


def test_partition_and_generators():
    part = dist.SlabPartition((16, 16, 16), 2, 3)
    assert part.shapes == [(16, 16, 16), (8, 8, 8), (4, 4, 4)]
    assert part.rows(0, 1) == (2048, 4096) and part.rows(2, 0) == (0, 32)
    assert part.bounds(1).tolist() == [0, 256, 512]
    with pytest.raises(ValueError):
        dist.SlabPartition((16, 16, 16), 2, 5)        # 1 plane per rank at the coarsest level: fine; 2 at level 3 -> odd above
    with pytest.raises(ValueError):
        dist.SlabPartition((16, 8, 32), 2, 2)         # reference's restriction offsets need shape[0] == shape[2]
    A = orc.stencil_poisson((6, 4, 6))
    rows = dist.stencil_rows((6, 4, 6), 24, 96)
    assert abs(rows - A[24:96]).max() == 0
    R = orc.restriction((8, 4, 8))
    assert abs(dist.restriction_rows((8, 4, 8), 8, 24) - R[8:24]).max() == 0
    assert abs(dist.restriction_rows((16,), 2, 7) - orc.restriction((16,))[2:7]).max() == 0
    assert abs(dist.restriction_rows((8, 8), 4, 12) - orc.restriction((8, 8))[4:12]).max() == 0
    keys, n = dist.set_keys((4, 4, 4), np.arange(64), "colour")
    assert n == 2 and np.array_equal(keys, orc.parity_colouring((4, 4, 4)))
    keys, n = dist.set_keys((4, 4, 4), np.arange(64), "gs")
    assert n == 10 and keys.max() == 9


@pytest.mark.parametrize("shape,n_ranks,grids", [((16, 16, 16), 2, 3), ((16, 8, 16), 4, 2), ((32, 32), 2, 3), ((64,), 4, 3)])
def test_local_hierarchy_equals_global(shape, n_ranks, grids):
    """Every rank's rows of every level (Galerkin from local rows only) == rows of the global
    hierarchy; halo plans are symmetric."""
    part = dist.SlabPartition(shape, n_ranks, grids)
    A0 = orc.stencil_poisson(shape)
    R = [orc.restriction(part.shapes[l]) for l in range(grids - 1)]
    A = orc.coefficient_list(A0, R)
    levels, coarse, counts = dist.build_all_ranks(part, lambda q: A0[slice(*part.rows(0, q))],
                                                  smoother="colour", spgemm=scipy_spgemm)
    assert abs(coarse - sp.csr_matrix(A[-1])).max() < 1e-15 and sum(counts) == A[-1].shape[0]
    for q in range(n_ranks):
        for l in range(grids):
            lo, hi = part.rows(l, q)
            lv = levels[q][l]
            n_loc = hi - lo
            want = sp.csr_matrix(A[l])[lo:hi]
            owned = lv["A"][:, :n_loc]
            assert abs(owned - want[:, lo:hi]).max() < 1e-15
            assert lv["A"].nnz == want.nnz                      # nothing lost into / out of the halo
            assert lv["n_halo"] == lv["recv_off"][-1] if len(lv["peers"]) else lv["n_halo"] == 0
            if l + 1 < grids:
                clo, chi = part.rows(l + 1, q)
                assert abs(lv["R"] - R[l][clo:chi, lo:hi]).max() == 0
            def tag(level, e):
                return -1 if level.get("groups") is None else int(level["groups"][e])

            for k, p in enumerate(lv["peers"]):                    # one entry per (peer, colour group)
                other = levels[int(p)][l]
                j = [e for e in range(len(other["peers"])) if other["peers"][e] == q and tag(other, e) == tag(lv, k)]
                assert len(j) == 1
                j = j[0]
                assert (lv["recv_off"][k + 1] - lv["recv_off"][k]) == (other["send_off"][j + 1] - other["send_off"][j])
            if l + 1 < grids and len(lv["peers"]):
                assert lv["groups"] is not None and set(lv["groups"]) <= {0, 1}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, shape, grids, smoother, out_dir, stencil="7pt"):
    from tests.dist_cpu_worker import run_rank
    run_rank(rank, world, port, shape, grids, smoother, out_dir, stencil)


def _oracle_cycles(A0, shape, grids, smoother, cycles=3):
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = orc.restriction_list(shape, grids - 2, 1)
    A = orc.coefficient_list(A0, R)
    assert len(A) == grids
    sm = orc.make_smoother(smoother, A, omega=0.8)
    p = {"preIterations": 1, "postIterations": 1, "coarsestLevel": len(R)}
    x, norms = None, []
    for _ in range(cycles):
        x, info = orc.mg_cycle(A, b, 0, R, p, initial=x, smoother=sm)
        norms.append(info["norm"])
    return x, norms


def _check_ranks(out_dir, world, x, norms):
    for rank in range(world):
        d = np.load(os.path.join(str(out_dir), "rank%d.npz" % rank))
        np.testing.assert_allclose(d["x"], x[int(d["lo"]):int(d["hi"])], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(d["norms"], norms, rtol=1e-11)


@pytest.mark.parametrize("smoother", ["colour", "gs", "jacobi"])
def test_two_process_gloo_cycle_matches_single_process_oracle(tmp_path, smoother):
    import torch.multiprocessing as mp
    shape, grids, world = (16, 16, 16), 3, 2
    mp.spawn(_worker, args=(world, _free_port(), shape, grids, smoother, str(tmp_path)), nprocs=world, join=True)
    x, norms = _oracle_cycles(orc.stencil_poisson(shape), shape, grids, smoother)
    _check_ranks(tmp_path, world, x, norms)


def test_two_process_gloo_cycle_27_point_eight_colours(tmp_path):
    """The 8-colour schedule (one message per neighbour and colour) over real message passing."""
    import torch.multiprocessing as mp
    from openmg_amd import operators
    shape, grids, world = (8, 8, 8), 2, 2
    mp.spawn(_worker, args=(world, _free_port(), shape, grids, "colour", str(tmp_path), "27var"), nprocs=world, join=True)
    A0 = operators.stencil27_variable(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = orc.restriction_list(shape, grids - 2, 1)
    A = orc.coefficient_list(A0, R)
    sm = orc.make_smoother("colour", A)
    p = {"preIterations": 1, "postIterations": 1, "coarsestLevel": len(R)}
    x, norms = None, []
    for _ in range(3):
        x, info = orc.mg_cycle(A, b, 0, R, p, initial=x, smoother=sm)
        norms.append(info["norm"])
    for rank in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        np.testing.assert_allclose(d["x"], x[int(d["lo"]):int(d["hi"])], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(d["norms"], norms, rtol=1e-11)


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_27_point_variable_coefficient_slabs_on_the_host(n_ranks):
    """configs[4]'s operator: rows of a slab == rows of the global generator; octant keys are a
    valid 8-colouring across the slab boundary (parity is rejected); the Galerkin operators a
    rank forms from its own rows are BIT-identical to the global (R A) R^T — the products run
    in global column order — and the plans carry one entry per (neighbour, colour)."""
    from openmg_amd import operators
    shape, grids = (16, 16, 16), 3
    part = dist.SlabPartition(shape, n_ranks, grids)
    A0 = operators.stencil27_variable(shape)
    for q in range(n_ranks):
        lo, hi = part.rows(0, q)
        assert abs(dist.stencil27_variable_rows(shape, lo, hi) - A0[lo:hi]).max() == 0
    R = [orc.restriction(part.shapes[l]) for l in range(grids - 1)]
    A = orc.coefficient_list(A0, R)
    rows_of = lambda q: dist.stencil27_variable_rows(shape, *part.rows(0, q))
    with pytest.raises(ValueError):
        dist.build_all_ranks(part, rows_of, smoother="colour", spgemm=scipy_spgemm)          # red-black: not a colouring here
    levels, coarse, counts = dist.build_all_ranks(part, rows_of, smoother="colour", spgemm=scipy_spgemm, colouring="octant")
    G, W = sp.csr_matrix(coarse), sp.csr_matrix(A[-1])
    G.sort_indices(); W.sort_indices()
    assert np.array_equal(G.indices, W.indices) and np.array_equal(G.data, W.data)
    for q in range(n_ranks):
        for l in range(grids):
            lo, hi = part.rows(l, q)
            lv = levels[q][l]
            want = sp.csr_matrix(A[l])[lo:hi]
            assert lv["A"].nnz == want.nnz
            own, ref = sp.csr_matrix(lv["A"][:, :hi - lo]), sp.csr_matrix(want[:, lo:hi])
            own.sort_indices(); ref.sort_indices()
            assert np.array_equal(own.data, ref.data)                                         # to the bit
            if l + 1 < grids:
                assert lv["n_sets"] == 8 and set(np.unique(lv["keys"])) == set(range(8))
                if len(lv["peers"]):
                    assert sorted(set(lv["groups"])) == list(range(8))
                    assert len(lv["peers"]) == 8 * len(set(lv["peers"]))


# ------------------------------------------------- the bench launcher (openmg_amd/launch.py) --
def _launch(world, out_dir, extra=(), timeout_s=240.0):
    import io
    import sys
    from openmg_amd import launch
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_cpu_worker.py")
    out, err = io.StringIO(), io.StringIO()
    code = launch.spawn_ranks(world, [sys.executable, worker, "--out", str(out_dir)] + list(extra),
                              timeout_s=timeout_s, out=out, err=err)
    return code, out.getvalue(), err.getvalue()


@pytest.mark.parametrize("world,shape,grids", [(2, (16, 16, 16), 3), (8, (32, 16, 32), 3)])
def test_launcher_runs_every_rank_and_relays_rank0_json(tmp_path, world, shape, grids):
    """`python bench.py --gpus N` without torch.distributed.run goes through launch.spawn_ranks:
    N rank processes with RANK / WORLD_SIZE / MASTER_* set, rank 0's JSON as the last stdout
    line, exit code 0.  Driven here with the gloo executor of the distributed schedule at world
    2 and at world 8 (8 slabs of 4 planes: the decomposition of BASELINE configs[3]/[4]), and
    every rank's iterate is checked against the single-process oracle."""
    import json
    code, out, err = _launch(world, tmp_path, ["--shape", ",".join(map(str, shape)), "--grids", str(grids)])
    assert code == 0, err
    last = json.loads(out.strip().splitlines()[-1])
    assert last["world"] == world
    x, norms = _oracle_cycles(orc.stencil_poisson(shape), shape, grids, "colour")
    np.testing.assert_allclose(last["norms"], norms, rtol=1e-11)
    _check_ranks(tmp_path, world, x, norms)


def test_launcher_deadline_and_failure_propagation(tmp_path):
    import time
    t0 = time.monotonic()
    code, out, err = _launch(2, tmp_path, ["--mode", "hang"], timeout_s=3.0)
    assert code == 124 and "deadline" in err and time.monotonic() - t0 < 30
    t0 = time.monotonic()
    code, out, err = _launch(3, tmp_path, ["--mode", "fail"], timeout_s=120.0)
    assert code == 7 and "rank 1 gives up" in err          # worst child code; the peers were ended, not waited for
    assert time.monotonic() - t0 < 60


def test_bench_parent_without_gpu_fails_fast_and_prints_no_result():
    """bench.py --gpus 2 started bare on a box without GPUs: the parent launches two rank
    processes, each fails loudly in require_gpu(), the parent returns non-zero within seconds
    (the round-1 behaviour was a gloo rendezvous hang) and prints no JSON line."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    from openmg_amd import _hip
    if _hip.device_count() > 0:
        pytest.skip("a GPU is visible: the no-GPU path cannot be exercised here")
    assert p.returncode != 0 and time.monotonic() - t0 < 240
    assert "no MI355X" in p.stderr or "no HIP device" in p.stderr
    assert not any(line.startswith("{") for line in p.stdout.splitlines())


def test_bench_overlap_autotune_candidates():
    """dist_bench times the cycle with the halo exchanges of the k largest levels on the second
    stream, for every k, and keeps the fastest: the candidate thresholds for bench.py's N = 8
    hierarchy (512^3 over 8 slabs, three distributed smoothed levels)."""
    from openmg_amd import dist_bench
    part = dist.SlabPartition((512, 512, 512), 8, 4)
    rows = [part.rows(l, 3)[1] - part.rows(l, 3)[0] for l in range(3)]
    assert rows == [16777216, 2097152, 262144]
    c = dist_bench.overlap_candidates(rows)
    assert [t for _, t in c] == [0, 16777216, 2097152, 1 << 62]
    assert dist_bench.overlap_candidates([4096]) == [("all levels", 0), ("none", 1 << 62)]


def _preflight_ranks(world, fault, fault_rank, timeout_s):
    import subprocess
    import sys
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "preflight_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), fault, str(fault_rank), str(timeout_s)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    out = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:                     # pragma: no cover - the failure this test exists for
            p.kill()
            o, e = p.communicate()
        out.append((p.returncode, o, e))
    return out


def test_preflight_healthy_ranks_agree_and_return_the_norm():
    res = _preflight_ranks(2, "none", -1, 60.0)
    for rank, (code, o, e) in enumerate(res):
        assert code == 0, e
        assert "rank %d norm 3" % rank in o


def test_preflight_stalled_rank_ends_every_rank_with_a_diagnosis():
    """one rank never reaches the exchange.  Every rank must END (non-zero) within the deadline
    and say where it was: the stalled one inside its cycle, the other inside the collective waiting for it."""
    from openmg_amd import preflight
    res = _preflight_ranks(2, "stall", 1, 6.0)
    for rank, (code, o, e) in enumerate(res):
        assert code == preflight.EXIT_TIMEOUT, (rank, code, e)
        assert "preflight: rank %d of 2 did not finish one cycle within 6 s: inside the cycle" % rank in e, e
        assert "device progress: level 0" in e
    assert "in the all-reduce of the norm" in res[0][2]       # rank 0 got as far as the collective
    assert "exchanging the ghost planes" in res[1][2]         # rank 1 never did


def test_preflight_norm_mismatch_between_ranks_is_fatal():
    from openmg_amd import preflight
    res = _preflight_ranks(2, "mismatch", 1, 60.0)
    for rank, (code, o, e) in enumerate(res):
        assert code == preflight.EXIT_MISMATCH, (rank, code, e)
        assert "differ between ranks after one cycle" in e and "[1]" in e


def test_preflight_cycle_that_raises_on_one_rank_reaches_every_rank():
    """a cycle that raises on ONE rank (a peer-store wait that gave up) used to skip the gather the other
    ranks were in, and the collectives paired up out of step.  Now every rank makes the same collectives, learns of
    the failure, raises the same error — and the agreed fallback that follows runs on all of them."""
    res = _preflight_ranks(2, "raise", 1, 60.0)
    for rank, (code, o, e) in enumerate(res):
        assert code == 0, (rank, code, e)
        assert "rank %d fallback agreed by 2 ranks after: preflight: the checked cycle raised on rank(s) 1 (RuntimeError: a peer-store wait gave up (injected))" % rank in o, o


def test_plane_levels_of_the_bench_shapes():
    """Which levels bench.py --gpus N hands to the plane-slab runner: every one needs an even number (>= 2) of planes
    per rank and even extents; the three bench shapes give three slab levels above a replicated 64^3 (N = 8) tail."""
    from openmg_amd import dist_bench
    for world, shape in dist_bench.SHAPES.items():
        if world == 1:
            continue
        assert dist_bench.plane_levels(shape, world, 4) == 3, (world, shape)
    assert dist_bench.plane_levels((512, 512, 512), 8, 7) == 6          # 64, 32, 16, 8, 4, 2 planes per rank: all n_dist - 1 levels
    assert dist_bench.plane_levels((512, 512, 512), 8, 8) == 6          # ... and ONE plane per rank is not a slab
    assert dist_bench.plane_levels((24, 16, 16), 4, 4) == 1             # 6 planes per rank, then 3: odd
    assert dist_bench.plane_levels((20, 16, 16), 8, 4) == 0             # 20 planes do not divide over 8 ranks
    assert dist_bench.plane_levels((32, 30, 32), 2, 4) == 1             # 30 lines, then 15: odd extents stop it
