"""One rank of the preflight test (tests/test_dist_cpu.py): a gloo group of WORLD ranks, a "cycle" that is a real
all-reduce, and a fault injected by name on one rank."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, port, fault, fault_rank, timeout_s = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]),
                                                      sys.argv[4], int(sys.argv[5]), float(sys.argv[6]))
    import datetime
    import torch
    import torch.distributed as td
    from openmg_amd import preflight
    td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                          timeout=datetime.timedelta(seconds=120))
    progress = {"at": "level 0, exchanging the ghost planes of x"}

    def one_cycle():
        if fault == "stall" and rank == fault_rank:
            time.sleep(3600)                                  # a rank that never reaches the exchange
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        progress["at"] = "level 0, in the all-reduce of the norm"
        td.all_reduce(t)                                      # the others wait HERE for the stalled one
        v = float(t.item())
        if fault == "mismatch" and rank == fault_rank:
            v *= 1.0 + 1e-6
        if fault == "raise" and rank == fault_rank:
            raise RuntimeError("a peer-store wait gave up (injected)")     # AFTER the cycle's own collectives, as on the device
        return v

    def gather(v):
        out = [None] * world
        td.all_gather_object(out, v)
        return out

    try:
        norm = preflight.run(rank, world, one_cycle, gather, timeout_s, where=lambda: progress["at"])
    except RuntimeError as e:
        # what dist_bench does with a peer-mode cycle that raised: every rank agrees on the fallback
        agreed = gather(False)
        print("rank %d fallback agreed by %d ranks after: %s" % (rank, len(agreed), e))
        sys.stdout.flush()
        os._exit(0)
    print("rank %d norm %.17g" % (rank, norm))
    sys.stdout.flush()
    os._exit(0)                                               # no destroy: nothing left to wait for


if __name__ == "__main__":
    main()
