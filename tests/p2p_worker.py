"""One rank of the peer-mode test (tests/test_gpu_plane_dist.py): WORLD processes share cuda:0, every rank opens the
others' IPC handles and the cycle's exchanges are stores into the other PROCESSES' memory, ordered by flags."""
import datetime
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    shape, grids, n_dist = tuple(int(v) for v in sys.argv[5].split("x")), int(sys.argv[6]), int(sys.argv[7])
    import torch
    import torch.distributed as td
    from openmg_amd import _hip, _hip_dist
    from test_gpu_plane_dist import problem
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                          timeout=datetime.timedelta(seconds=120))
    A, R, b, x0 = problem(shape, grids)
    shapes = [tuple(s >> l for s in shape) for l in range(len(A))]
    coef = [_hip_dist.star_coefficients(A[l], shapes[l]) for l in range(n_dist)]
    tail = _hip.Hierarchy(A[n_dist:], R[n_dist:], smoother="colour")
    d = _hip_dist.PlaneDistRank(rank, world, shape, coef, float(R[0].data[0]), tail)
    per = b.size // world
    handles = [None] * world
    td.all_gather_object(handles, d.p2p_handles())
    for r in range(world):
        if r != rank:
            d.p2p_open(r, handles[r])
    d.p2p_enable(2)                                        # the ranks share a GPU: wait launches, not waiting passes
    d.load(b[rank * per:(rank + 1) * per], x0[rank * per:(rank + 1) * per])

    def reduce(squares):
        t = torch.tensor(squares, dtype=torch.float64)
        td.all_reduce(t)
        return [float(v) for v in t]

    norms = d.cycles(2, reduce) + d.cycles(1, reduce)
    x = d.fetch()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x, norms=np.array(norms))
    td.barrier()                                           # nobody unmaps what a neighbour may still be writing
    d.close()
    tail.close()
    os._exit(0)


if __name__ == "__main__":
    main()
