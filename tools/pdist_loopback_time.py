"""Per-rank device time of the plane-slab cycle at the --gpus N bench shape, measured on ONE GPU: all N slabs in a
loopback group on one stream (exchanges are device copies), time / N.  What is left to add for N GPUs is the
exchange latency (tools/exchange_probe.py), not device work."""
import os
import sys
import time

import numpy as np

from openmg_amd import _hip, _hip_dist, dist_bench, operators


def main():
    for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
        shape, n_levels, tail_grids = dist_bench.SHAPES[world], 3, 4
        coef = [[v / 16.0 ** l for v in (-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0)] for l in range(n_levels)]
        tshape = tuple(s >> n_levels for s in shape)
        Rt = operators.restrictionList(tshape, tail_grids - 2, 8)
        At = operators.coeffecientList(operators.stencil_poisson(tshape) / 16.0 ** n_levels, Rt)
        b = np.random.default_rng(5).random(shape[0] * shape[1] * shape[2])
        per = b.size // world
        ranks = []
        for r in range(world):
            d = _hip_dist.PlaneDistRank(r, world, shape, coef, 0.125, _hip.Hierarchy(At, Rt, smoother="colour"))
            d.load(b[r * per:(r + 1) * per])
            ranks.append(d)
        g = _hip_dist.PlaneDistGroup(ranks, p2p=int(os.environ.get("OMG_LOOPBACK_P2P", "0")))
        g.cycles(3)
        t = time.perf_counter()
        n = 20
        g.cycles(n)
        dt = (time.perf_counter() - t) / n
        g.close()
        print("world %d shape %s: %.1f us per cycle for all slabs, %.1f us per rank" % (world, shape, dt * 1e6, dt * 1e6 / world), flush=True)


if __name__ == "__main__":
    main()
