"""The 27-point slab schedule of openmg_amd/csrc/dist27.hip on CPU (tests/slab27_cpu_executor.py): ghost AGGREGATE
planes, colours 0 .. 3 of the upper ghost plane relaxed redundantly, one exchange per sweep — against the single-process
oracle cycle (openmg/__init__.py:151-236, colour-ordered sweep) for 1, 2, 4 and 8 ranks in-process and for two REAL
processes over torch.distributed / gloo.  CPU only."""
import os
import socket

import numpy as np
import pytest

from openmg_amd import dist_bench, operators
from oracle import mg_oracle as orc
from tests.slab27_cpu_executor import InProcessComm, Slab27Cpu, tail_of


def problem(shape, grids):
    A0 = operators.stencil27_variable(shape)
    R = orc.restriction_list(shape, grids - 2, 1)
    A = orc.coefficient_list(A0, R)
    assert len(A) == grids
    rng = np.random.default_rng(12345)
    b = A0 @ rng.random(A0.shape[0])
    x0 = rng.standard_normal(A0.shape[0])
    return A, R, b, x0


def oracle_cycles(A, R, b, x0, pre, post, cycles):
    sm = orc.make_smoother("colour", A)
    p = {"preIterations": pre, "postIterations": post, "coarsestLevel": len(R)}
    x, norms = x0.copy(), []
    for _ in range(cycles):
        x, info = orc.mg_cycle(A, b, 0, R, p, initial=x, smoother=sm)
        norms.append(info["norm"])
    return x, norms


@pytest.mark.parametrize("shape,grids,world,n_levels", [((16, 16, 16), 3, 1, 1), ((16, 16, 16), 3, 2, 2), ((16, 16, 16), 3, 4, 2),
                                                        ((16, 16, 16), 3, 8, 1), ((32, 16, 32), 4, 4, 3)])
@pytest.mark.parametrize("pre,post", [(1, 1), (1, 0), (2, 1), (0, 1)])
def test_slab_schedule_reproduces_the_single_process_cycle(shape, grids, world, n_levels, pre, post):
    if shape[0] == 32 and (pre, post) != (1, 1):
        pytest.skip("the larger grid once")
    A, R, b, x0 = problem(shape, grids)
    want_x, want_norms = oracle_cycles(A, R, b, x0, pre, post, 2)
    ex = Slab27Cpu(shape, world, n_levels, A, R, range(world), InProcessComm(), tail_of(A[n_levels:], R[n_levels:]))
    per = b.size // world
    ex.load(lambda r: b[r * per:(r + 1) * per], lambda r: x0[r * per:(r + 1) * per])
    xs, norms = ex.run(pre, post, 2)
    np.testing.assert_allclose(norms, want_norms, rtol=1e-11)
    for r in range(world):
        np.testing.assert_allclose(xs[r], want_x[r * per:(r + 1) * per], rtol=1e-10, atol=1e-13)
    # exchanges per cycle: p + q on the finest level, 1 + p + q on every other distributed level
    if world > 1:
        assert ex.exchanges == (pre + post) + (n_levels - 1) * (1 + pre + post)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gloo_rank(rank, world, port, shape, grids, n_levels, out_dir):
    import torch
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        class Comm:
            def exchange(self, sends, recvs):
                bufs = [torch.empty(n, dtype=torch.float64) for _, _, n in recvs]
                ops = [td.P2POp(td.isend, torch.from_numpy(np.ascontiguousarray(a)), dst) for _, dst, a in sends]
                ops += [td.P2POp(td.irecv, t, src) for (_, src, _), t in zip(recvs, bufs)]
                if ops:
                    for r in td.batch_isend_irecv(ops):
                        r.wait()
                return [t.numpy() for t in bufs]

            def allgather(self, own):
                out = [None] * world
                td.all_gather_object(out, np.asarray(own[0][1]))
                return out

            def allreduce_sum(self, v):
                t = torch.tensor([v], dtype=torch.float64)
                td.all_reduce(t)
                return float(t[0])

        A, R, b, x0 = problem(shape, grids)
        ex = Slab27Cpu(shape, world, n_levels, A, R, [rank], Comm(), tail_of(A[n_levels:], R[n_levels:]))
        per = b.size // world
        ex.load(lambda r: b[r * per:(r + 1) * per], lambda r: x0[r * per:(r + 1) * per])
        xs, norms = ex.run(1, 1, 2)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=xs[rank], norms=np.array(norms))
        td.barrier()
    finally:
        td.destroy_process_group()


def test_two_process_gloo_slab_schedule(tmp_path):
    import torch.multiprocessing as mp
    shape, grids, world, n_levels = (16, 16, 16), 3, 2, 2
    mp.spawn(_gloo_rank, args=(world, _free_port(), shape, grids, n_levels, str(tmp_path)), nprocs=world, join=True)
    A, R, b, x0 = problem(shape, grids)
    want_x, want_norms = oracle_cycles(A, R, b, x0, 1, 1, 2)
    per = b.size // world
    for rank in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        np.testing.assert_allclose(d["norms"], want_norms, rtol=1e-11)
        np.testing.assert_allclose(d["x"], want_x[rank * per:(rank + 1) * per], rtol=1e-10, atol=1e-13)


def test_slab27_levels_of_the_bench_shapes():
    """How many levels bench.py --gpus N --stencil 27var distributes: an even number (>= 2) of planes per rank and even
    extents on every one, a grid line within one wave (nx <= 512 in fp32, 256 in fp64)."""
    assert dist_bench.slab27_levels((512, 512, 512), 8, 4, "f32") == 3
    assert dist_bench.slab27_levels((512, 512, 512), 8, 4, "f64") == 0           # a 512-cell line does not fit a wave in fp64
    assert dist_bench.slab27_levels((256, 512, 256), 2, 4, "f64") == 3
    assert dist_bench.slab27_levels((256, 256, 256), 1, 3, "f32") == 2
    assert dist_bench.slab27_levels((24, 16, 16), 4, 4, "f32") == 1              # 6 planes per rank, then 3: odd
    assert dist_bench.slab27_levels((20, 16, 16), 8, 4, "f32") == 0
