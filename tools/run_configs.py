#!/usr/bin/env python3
"""Time the single-GPU BASELINE.json configs (and the reference-exact 'gs' smoother on the
3-D problem) — informational; bench.py reports only configs[2]."""
import sys
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from openmg_amd import _hip, operators  # noqa: E402

CASES = [
    ("configs[0] 1-D N=4096 (4,-1), 3 grids, lexicographic GS", (4096,), 3, "gs", "ref1d"),
    ("configs[1] 2-D 5-point 1024^2, 4 grids, weighted Jacobi (omega 2/3)", (1024, 1024), 4, "jacobi", "stencil"),
    ("configs[1] shape with red-black GS", (1024, 1024), 4, "colour", "stencil"),
    ("configs[2] 3-D 7-point 256^3, 5 grids, red-black GS", (256, 256, 256), 5, "colour", "stencil"),
    ("configs[2] shape with the reference's lexicographic GS (one wavefront launch per sweep; OMG_MARCH=0: 766 level sets)", (256, 256, 256), 5, "gs", "stencil"),
    ("3-D 7-point 128^3, 4 grids, lexicographic GS", (128, 128, 128), 4, "gs", "stencil"),
]


def main():
    graph = "--graph" in sys.argv
    only = [int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()]
    for k, (name, shape, grids, smoother, kind) in enumerate(CASES):
        if only and k not in only:
            continue
        A0 = operators.poisson(shape[0], sparse=True) if kind == "ref1d" else operators.stencil_poisson(shape)
        b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
        t0 = time.perf_counter()
        R = operators.restrictionList(shape, grids - 2, 8)
        A = operators.coeffecientList(A0, R)
        h = _hip.Hierarchy(A, R, smoother=smoother, omega=2.0 / 3.0)
        setup = time.perf_counter() - t0
        h.resident_load(b)
        if graph:
            h.use_graph(True)
        for _ in range(3):
            h.resident_cycle(1, 1, want_norm=False)
        h.sync()
        steps = 20
        t0 = time.perf_counter()
        for _ in range(steps):
            h.resident_cycle(1, 1, want_norm=False)
        h.sync()
        dt = (time.perf_counter() - t0) / steps
        norm = h.resident_cycle(1, 1)
        print("%s%-78s grids %d sets/level0 %4d  %9.3f ms/cycle  %9.1f cycles/s  setup %.2f s  norm %.3e"
              % ("[hipGraph] " if graph else "", name, len(A), h.level_sets(0), dt * 1e3, 1.0 / dt, setup, norm))
        sys.stdout.flush()
        h.close()


if __name__ == "__main__":
    main()
