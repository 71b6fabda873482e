#!/usr/bin/env python3
"""Schedule overhead of the distributed runner WITHOUT network cost: the same problem on one
GPU as (a) one ordinary hierarchy and (b) a 2- or 4-rank loopback group (pack kernels,
per-set exchanges as device copies, boundary-first sets, replicated tail)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, _hip_dist, dist, operators  # noqa: E402


def main():
    shape, grids, n_dist = (256, 256, 256), 5, 4
    N = int(np.prod(shape))
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(1).random(N)
    R = [operators.restriction(tuple(s // 2 ** l for s in shape)) for l in range(grids - 1)]
    A = operators.coeffecientList(A0, R)
    h = _hip.Hierarchy(A, R, smoother="colour")
    h.resident_load(b)
    for _ in range(3):
        h.resident_cycle(1, 1, want_norm=False)
    h.sync()
    t = time.perf_counter()
    for _ in range(20):
        h.resident_cycle(1, 1, want_norm=False)
    h.sync()
    base = (time.perf_counter() - t) / 20
    print("single hierarchy      %.3f ms/cycle" % (base * 1e3))
    h.close()
    del A, R
    for n_ranks in (2, 4):
        part = dist.SlabPartition(shape, n_ranks, n_dist)
        levels, coarse, counts = dist.build_all_ranks(part, lambda q: dist.stencil_rows(shape, *part.rows(0, q)),
                                                      smoother="colour")
        ranks = [_hip_dist.DistRank(q, n_ranks, levels[q], None, counts, smoother="colour",
                                    tail=dist.make_tail(coarse, part.shapes[-1], grids - n_dist + 1))
                 for q in range(n_ranks)]
        g = _hip_dist.DistGroup(ranks)
        for q, r in enumerate(ranks):
            r.load(b[slice(*part.rows(0, q))])
        for _ in range(3):
            g.cycle(1, 1, want_norm=False)
        ranks[0].sync()
        t = time.perf_counter()
        for _ in range(20):
            g.cycle(1, 1, want_norm=False)
        ranks[0].sync()
        dt = (time.perf_counter() - t) / 20
        print("%d-rank loopback group %.3f ms/cycle (+%.1f %%; includes %d redundant tails)"
              % (n_ranks, dt * 1e3, 100 * (dt / base - 1), n_ranks))
        g.close()


if __name__ == "__main__":
    main()
