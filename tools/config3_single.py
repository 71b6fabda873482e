#!/usr/bin/env python3
"""BASELINE configs[3]'s problem (3-D 7-point Poisson 512^3, 6 grids, V(1,1), red-black Gauss-Seidel,
fp64) on ONE MI355X: informational (the config is quoted on 8 GPUs, where each rank holds a 512x512x64
slab).  Checks the device's residual norm of the last cycle against a host recomputation with SciPy."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmg_amd import _hip, operators  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    grids = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    shape = (size,) * 3
    t0 = time.perf_counter()
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    R = operators.restrictionList(shape, grids - 2, 8)
    A = operators.coeffecientList(A0, R)
    h = _hip.Hierarchy(A, R, smoother="colour")
    t_setup = time.perf_counter() - t0
    h.resident_load(b)
    h.resident_cycles(1, 1, 3)
    steps = 20
    h.sync()
    t0 = time.perf_counter()
    norms = h.resident_cycles(1, 1, steps)
    h.sync()
    dt = (time.perf_counter() - t0) / steps
    x = h.resident_fetch()
    host = float(np.linalg.norm(b - A0 @ x))
    rel = abs(host - norms[-1]) / host
    print("3-D 7-point %d^3 (%d unknowns, %d stored entries), %d grids, red-black GS, fp64, one GPU:" % (size, A0.shape[0], A0.nnz, len(A)))
    print("  %.3f ms/cycle  %.1f V-cycles/s   generate %.1f s  setup %.1f s" % (dt * 1e3, 1.0 / dt, t_gen, t_setup))
    print("  residual norm after %d cycles: device %.12e  host (SciPy) %.12e  rel diff %.2e" % (steps + 3, norms[-1], host, rel))
    print("  contraction of the last cycle: %.4f" % (norms[-1] / norms[-2]))
    assert rel < 1e-10, rel
    h.close()


if __name__ == "__main__":
    main()
