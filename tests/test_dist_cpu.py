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
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch
        from tests.dist_cpu_executor import CpuRank

        class Comm:
            def sendrecv(self, sends, recvs):
                bufs = [torch.empty(n, dtype=torch.float64) for _, n in recvs]
                ops = [td.P2POp(td.isend, torch.from_numpy(a), p) for p, a in sends]
                ops += [td.P2POp(td.irecv, t, p) for (p, _), t in zip(recvs, bufs)]
                if ops:
                    for r in td.batch_isend_irecv(ops):
                        r.wait()
                return [t.numpy() for t in bufs]

            def allgather(self, a):
                out = [None] * world
                td.all_gather_object(out, np.asarray(a))
                return out

            def allreduce_sum(self, v):
                t = torch.tensor([v], dtype=torch.float64)
                td.all_reduce(t)
                return float(t[0])

        def all_gather(obj):
            out = [None] * world
            td.all_gather_object(out, obj)
            return out

        part = dist.SlabPartition(shape, world, grids)
        lo, hi = part.rows(0, rank)
        A_rows = dist.stencil_rows(shape, lo, hi) if stencil == "7pt" else dist.stencil27_variable_rows(shape, lo, hi)
        levels, coarse, counts = dist.build_this_rank(part, rank, A_rows, all_gather, smoother=smoother,
                                                      spgemm=scipy_spgemm,
                                                      colouring="parity" if stencil == "7pt" else "octant")
        u = np.random.default_rng(12345).random(part.n_rows(0))
        b_loc = A_rows @ u
        x_loc, norms = CpuRank(rank, levels, coarse, counts, smoother, Comm(), omega=0.8).run(b_loc, 3, 1, 1)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x_loc, norms=np.array(norms), lo=lo, hi=hi)
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("smoother", ["colour", "gs", "jacobi"])
def test_two_process_gloo_cycle_matches_single_process_oracle(tmp_path, smoother):
    import torch.multiprocessing as mp
    shape, grids, world = (16, 16, 16), 3, 2
    mp.spawn(_worker, args=(world, _free_port(), shape, grids, smoother, str(tmp_path)), nprocs=world, join=True)
    A0 = orc.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = orc.restriction_list(shape, grids - 2, 1)
    A = orc.coefficient_list(A0, R)
    assert len(A) == grids
    sm = orc.make_smoother({"colour": "colour", "gs": "gs", "jacobi": "jacobi"}[smoother], A, omega=0.8)
    p = {"preIterations": 1, "postIterations": 1, "coarsestLevel": len(R)}
    x = None
    norms = []
    for _ in range(3):
        x, info = orc.mg_cycle(A, b, 0, R, p, initial=x, smoother=sm)
        norms.append(info["norm"])
    for rank in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        np.testing.assert_allclose(d["x"], x[int(d["lo"]):int(d["hi"])], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(d["norms"], norms, rtol=1e-11)


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
