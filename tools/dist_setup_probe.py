#!/usr/bin/env python3
"""Time one rank's share of the 8-GPU setup (512^3, rank 3) on a single GPU: catches memory
or run-time surprises of the host-side slab construction that the 1-GPU sandbox cannot
exercise end to end."""
import os
import sys
import time
import resource

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import dist  # noqa: E402


def main():
    shape, world, rank, n_dist = (512, 512, 512), 8, 3, 4
    t = time.perf_counter()
    part = dist.SlabPartition(shape, world, n_dist)
    lo, hi = part.rows(0, rank)
    A_rows = dist.stencil_rows(shape, lo, hi)
    print("stencil_rows %.1f s, nnz %d" % (time.perf_counter() - t, A_rows.nnz)); t = time.perf_counter()
    u = np.random.default_rng(12345).random(part.n_rows(0))
    b = A_rows @ u
    del u
    print("rhs %.1f s" % (time.perf_counter() - t)); t = time.perf_counter()
    s = dist.RankSetup(part, rank, A_rows, smoother="colour")
    for l in range(n_dist):
        halo = s.begin_level(l)
        print("begin_level(%d) %.1f s, halo %d, A_loc %s" % (l, time.perf_counter() - t, halo.size, s.levels[l]["A"].shape))
        t = time.perf_counter()
        # neighbours' halo lists by symmetry: what rank q needs from its neighbours mirrors ours
        bounds = part.bounds(l)
        plane = part.plane(l)
        halos = []
        for q in range(world):
            qlo, qhi = part.rows(l, q)
            h = []
            if q > 0:
                h.append(np.arange(qlo - plane, qlo))
            if q < world - 1:
                h.append(np.arange(qhi, qhi + plane))
            halos.append(np.concatenate(h))
        assert np.array_equal(halos[rank], halo)
        s.finish_level(l, halos)
        print("finish_level(%d) %.1f s" % (l, time.perf_counter() - t)); t = time.perf_counter()
    print("peak RSS %.1f GB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6))


if __name__ == "__main__":
    main()
