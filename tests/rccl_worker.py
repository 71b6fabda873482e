"""One rank of tests/test_gpu_rccl_shim.py: WORLD processes share cuda:0 and run the product's RCCL call sites
(csrc/dist.hip, csrc/dist27.hip: communicator set-up, grouped send / recv, all-gather, all-reduce) through the
test-only stand-in tests/fake_rccl (OMG_RCCL_LIB) — RCCL itself refuses two ranks on one device.

argv: mode out_dir shape grids n_dist dtype extra        (RANK / WORLD_SIZE / MASTER_PORT: launch.child_env)
modes: plane  — plane-pipelined slabs (omg_pdist_*), 7-point red-black fp64;  extra = "gate" switches the gated passes on
       sets   — the set-by-set runner (omg_dist_*);                           extra = smoother
       slab27 — 27-point slabs (omg_sdist_*);                                 extra = unused
Writes rank<r>.npz: the rank's part of the iterate, every norm, the communicator's size, the shim's status."""
import ctypes
import datetime
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    mode, out_dir = sys.argv[1], sys.argv[2]
    shape, grids, n_dist, dtype = tuple(int(v) for v in sys.argv[3].split("x")), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    extra = sys.argv[7] if len(sys.argv) > 7 else ""
    rank, world, port = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["MASTER_PORT"])
    shim = os.environ["OMG_RCCL_LIB"]
    import torch.distributed as td
    from openmg_amd import _hip, _hip_dist, dist, operators
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                          timeout=datetime.timedelta(seconds=300))

    def all_gather(obj):
        out = [None] * world
        td.all_gather_object(out, obj)
        return out

    def ids(n):
        box = [[_hip_dist.rccl_unique_id() for _ in range(n)] if rank == 0 else None]
        td.broadcast_object_list(box, src=0)
        return box[0]

    _hip_dist.set_device(0)
    n_glob = int(np.prod(shape))
    per = n_glob // world
    result = {}
    tails = []
    if mode == "plane":
        # the problem of tests/test_gpu_plane_dist.py: level l of the hierarchy is lap3(shape / 2^l) / 16^l
        coef = [[v / 16.0 ** l for v in (-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0)] for l in range(n_dist)]
        from test_gpu_plane import hierarchy
        At, Rt = hierarchy(tuple(s >> n_dist for s in shape), grids - n_dist, scale=1.0 / 16.0 ** n_dist)
        tail = _hip.Hierarchy(At, Rt, smoother="colour")
        tails.append(tail)
        b = np.random.default_rng([31, rank]).random(per)                    # (the parent makes the same pieces)
        x0 = np.random.default_rng([32, rank]).standard_normal(per)
        r = _hip_dist.PlaneDistRank(rank, world, shape, coef, 0.125, tail)
        r.connect(*ids(2))
        r.load(b, x0)
        if extra == "gate":
            r.set_gate(True)
            result["gated"] = r.info()["gated"]
        norms = r.cycles(2) + r.cycles(1)
        result["x"] = r.fetch()
        # the reference's default cycle V(1, 0) from the same start
        r.load(b, x0)
        norms += r.cycles(2, pre=1, post=0)
        result["x10"] = r.fetch()
    elif mode == "sets":
        smoother = extra or "colour"
        part = dist.SlabPartition(shape, world, n_dist)
        lo, hi = part.rows(0, rank)
        A_rows = dist.stencil_rows(shape, lo, hi)
        b = operators.stencil_poisson(shape) @ np.random.default_rng(12345).random(n_glob)
        levels, coarse, counts = dist.build_this_rank(part, rank, A_rows, all_gather, smoother=smoother)
        tail = dist.make_tail(coarse, part.shapes[-1], grids - n_dist + 1, smoother=smoother, omega=0.8, dtype=dtype)
        r = _hip_dist.DistRank(rank, world, levels, None, counts, smoother=smoother, omega=0.8, tail=tail, dtype=dtype)
        r.connect(ids(1)[0])
        r.load(b[lo:hi])
        norms = [r.cycle(1, 1) for _ in range(2)] + r.cycles(1, 1, 2)
        result["x"] = r.fetch()
    elif mode == "slab27":
        A_rows = dist.stencil27_variable_rows(shape, rank * per, (rank + 1) * per)
        r = _hip_dist.Slab27Rank(rank, world, shape, A_rows, n_dist, 0.125, dtype)
        coarse = dist.assemble_coarse(all_gather(r.coarse_rows()))
        tail = dist.make_tail(coarse, tuple(s >> n_dist for s in shape), grids - n_dist, smoother="colour", dtype=dtype)
        tails.append(tail)
        r.set_tail(tail)
        r.connect(ids(1)[0])                              # (collective: also the neighbours' coefficient rows)
        if extra == "p2p":
            # the halo exchanges as peer stores between the rank PROCESSES (hipIpc mappings of the neighbours' vectors and
            # flags); the gather below the slabs and the norm's reduction stay on the communicator
            handles = all_gather(r.p2p_handles())
            for nb in (rank - 1, rank + 1):
                if 0 <= nb < world:
                    r.p2p_open(nb, handles[nb])
            r.p2p_enable(1)
            td.barrier()
        u = np.random.default_rng(11).random(n_glob)
        b = A_rows @ u
        x0 = np.random.default_rng(12).standard_normal(n_glob)[rank * per:(rank + 1) * per]
        if dtype == "float32":
            b, x0 = b.astype(np.float32).astype(np.float64), x0.astype(np.float32).astype(np.float64)
        norms = []
        for k, (pre, post) in enumerate(((1, 1), (1, 0), (2, 1))):
            r.load(b, x0)
            norms += r.cycles(pre, post, 3)
            result["x%d%d" % (pre, post)] = r.fetch()
        result["exchanges"] = r.info()["exchanges_last_call"]
    elif mode == "stall":
        # a schedule that would DEADLOCK RCCL — rank 1 never posts the receives rank 0's sends wait for — must end as an
        # error after the stand-in's bounded wait (FRCCL_TIMEOUT_S), not as a hung GPU: rank 0 runs one cycle alone
        from test_gpu_plane import hierarchy
        coef = [[v / 16.0 ** l for v in (-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0)] for l in range(n_dist)]
        At, Rt = hierarchy(tuple(s >> n_dist for s in shape), grids - n_dist, scale=1.0 / 16.0 ** n_dist)
        tail = _hip.Hierarchy(At, Rt, smoother="colour")
        tails.append(tail)
        r = _hip_dist.PlaneDistRank(rank, world, shape, coef, 0.125, tail)
        r.connect(*ids(2))
        r.load(np.random.default_rng([31, rank]).random(per))
        norms = r.cycles(1)                                 # (both ranks: a cycle that completes)
        lib = ctypes.CDLL(shim)
        assert int(lib.frccl_status()) == 0
        td.barrier()
        outcome = "completed"
        if rank == 0:
            try:
                r.cycles(1)                                 # rank 1 does not take part
                r.sync()
            except Exception as e:                          # noqa: BLE001 - an error is one of the two acceptable endings
                outcome = "raised: %s" % e
            result["status_after"] = int(lib.frccl_status())
        else:
            result["status_after"] = 0
        result["outcome"] = outcome
        result["x"] = np.zeros(1)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **result)
        td.barrier()
        os._exit(0)                                         # (no orderly tear-down of a communicator whose peer has gone)
    else:
        raise SystemExit("unknown mode " + mode)
    r.sync()
    result["norms"] = np.array(norms)
    result["rccl_ranks"] = r.rccl_ranks()
    lib = ctypes.CDLL(shim)                                # the handle libopenmg_hip.so dlopen'ed
    lib.frccl_identity.restype = ctypes.c_char_p
    result["shim"] = lib.frccl_identity().decode()
    result["shim_status"] = int(lib.frccl_status())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **result)
    td.barrier()                                           # nobody tears a mailbox down a neighbour may still be writing
    r.close()
    for t in tails:
        t.close()
    td.barrier()
    os._exit(0)


if __name__ == "__main__":
    main()
