#!/usr/bin/env python3
"""What the two-launch (edge / inner) schedule of the slab passes costs ONE rank in device time: a single slab of the
--gpus N bench shape (no neighbours: the exchanges are no-ops) through the plane-slab runner, whole passes
(OMG_PDIST_SPLIT=0) against split passes forced on (OMG_PDIST_SPLIT=2).  Run once per setting:

    OMG_PDIST_SPLIT=0 python tools/pdist_split_cost.py ; OMG_PDIST_SPLIT=2 python tools/pdist_split_cost.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, _hip_dist, dist_bench, operators  # noqa: E402


def main():
    for world in (2, 4, 8):
        g = dist_bench.SHAPES[world]
        shape, n_levels = (g[0] // world, g[1], g[2]), 3
        coef = [[v / 16.0 ** l for v in (-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0)] for l in range(n_levels)]
        tshape = tuple(s >> n_levels for s in shape)
        At = operators.stencil_poisson(tshape) / 16.0 ** n_levels
        tail = _hip.Hierarchy([At], [], smoother="colour")                 # (the slab's own coarse grid, solved directly: same in both settings)
        b = np.random.default_rng(5).random(shape[0] * shape[1] * shape[2])
        d = _hip_dist.PlaneDistRank(0, 1, shape, coef, 0.125, tail)
        d.load(b)
        grp = _hip_dist.PlaneDistGroup([d], p2p=0)
        grp.cycles(3)
        t = time.perf_counter()
        n = 30
        norms = grp.cycles(n)
        dt = (time.perf_counter() - t) / n
        grp.close()
        print("OMG_PDIST_SPLIT=%s  one slab of the N = %d run, %s: %.1f us per cycle   (norm %.6e)"
              % (os.environ.get("OMG_PDIST_SPLIT", "1"), world, shape, dt * 1e6, norms[-1]), flush=True)


if __name__ == "__main__":
    main()
