#!/usr/bin/env python3
"""configs[4]'s per-GPU workload (256^3 fp32 27-point) as ONE slab of the 27-point slab runner (omg_sdist, ghost aggregate
planes, 3 slab levels + replicated tail) against the single-GPU hierarchy, alternating regions inside one process."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, _hip_dist, dist, operators  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    n_levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    grids, steps = 5, 20
    shape = (size,) * 3
    A0 = operators.stencil27_variable(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    h = _hip.Hierarchy.from_fine(A0, shape, grids - 1, smoother="colour", dtype="float32")
    h.resident_load(b)
    r = _hip_dist.Slab27Rank(0, 1, shape, A0, n_levels, dtype="float32")
    del A0
    tail = dist.make_tail(r.coarse_rows(), tuple(s >> n_levels for s in shape), grids - n_levels, smoother="colour", dtype="float32")
    r.set_tail(tail)
    r.load(b)
    for _ in range(2):
        h.resident_cycles(1, 1, steps)
        r.cycles(1, 1, steps)
    out = {"single": [], "slab": []}
    for _ in range(5):
        h.sync()
        t0 = time.perf_counter()
        n1 = h.resident_cycles(1, 1, steps)
        out["single"].append((time.perf_counter() - t0) / steps * 1e3)
        t0 = time.perf_counter()
        n2 = r.cycles(1, 1, steps)
        out["slab"].append((time.perf_counter() - t0) / steps * 1e3)
    print("ms per cycle, single-GPU hierarchy:", ["%.4f" % v for v in out["single"]])
    print("ms per cycle, one slab (%d slab levels):" % n_levels, ["%.4f" % v for v in out["slab"]])
    print("ratio of medians: %.4f" % (sorted(out["slab"])[2] / sorted(out["single"])[2]))
    print("last norms:", n1[-1], n2[-1])


if __name__ == "__main__":
    main()
