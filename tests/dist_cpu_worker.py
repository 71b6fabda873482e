"""One rank of the distributed V-cycle schedule on CPU (test infrastructure): NumPy/SciPy + the
oracle's C sweeps for the local work, torch.distributed/gloo for the messages.  Importable
(tests/test_dist_cpu.py spawns `run_rank`) and runnable as a rank process under
openmg_amd.launch.spawn_ranks / torch.distributed.run (RANK, WORLD_SIZE, MASTER_* in the
environment), in which case rank 0 prints one JSON line as its last stdout line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def scipy_spgemm(X, Y):
    return sp.csr_matrix(sp.csr_matrix(X) @ sp.csr_matrix(Y))


def run_rank(rank, world, port, shape, grids, smoother, out_dir, stencil="7pt", cycles=3):
    import torch
    import torch.distributed as td
    from openmg_amd import dist
    from tests.dist_cpu_executor import CpuRank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
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
        x_loc, norms = CpuRank(rank, levels, coarse, counts, smoother, Comm(), omega=0.8).run(b_loc, cycles, 1, 1)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), x=x_loc, norms=np.array(norms), lo=lo, hi=hi)
        td.barrier()
        return norms
    finally:
        td.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="16,16,16")
    ap.add_argument("--grids", type=int, default=3)
    ap.add_argument("--smoother", default="colour")
    ap.add_argument("--out", required=True)
    ap.add_argument("--mode", default="run", choices=["run", "hang", "fail"],
                    help="hang: never finish (deadline test); fail: rank 1 exits with code 7 before the rendezvous")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if args.mode == "hang":
        print("rank %d waiting" % rank, flush=True)
        time.sleep(3600)
    if args.mode == "fail" and rank == 1:
        sys.stderr.write("rank 1 gives up\n")
        return 7
    shape = tuple(int(s) for s in args.shape.split(","))
    norms = run_rank(rank, world, int(os.environ["MASTER_PORT"]), shape, args.grids, args.smoother, args.out)
    if rank == 0:
        print("some banner line before the result")
        print(json.dumps({"world": world, "norms": norms, "shape": shape}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
