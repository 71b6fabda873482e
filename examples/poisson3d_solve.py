#!/usr/bin/env python3
"""Solve a 3-D Poisson problem on one MI355X through the reference's entry point
(`mgSolve(A, b, parameters)`, openmg/__init__.py:28) and print, for each smoother the device
offers, the convergence history and the cost per V-cycle.

    python examples/poisson3d_solve.py [extent] [grids]        (default 128, 4)

'gs' is the reference's own iterate (lexicographic Gauss-Seidel, level-scheduled on the device),
'colour' is red-black Gauss-Seidel, 'jacobi' weighted Jacobi with omega = 2/3.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openmg  # noqa: E402  (the drop-in alias of openmg_amd)
from openmg_amd import _hip  # noqa: E402


def main():
    extent = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    grids = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    shape = (extent,) * 3
    A = openmg.operators.stencil_poisson(shape)
    zz, yy, xx = np.meshgrid(*(np.linspace(0.0, 1.0, extent + 2)[1:-1],) * 3, indexing="ij")
    u_exact = (np.sin(np.pi * xx) * np.sin(2 * np.pi * yy) * np.sin(3 * np.pi * zz)).ravel()
    b = A @ u_exact
    print("%d^3 unknowns, %d nonzeros, %d grids" % (extent, A.nnz, grids))
    print("%-8s %7s %14s %14s %10s" % ("smoother", "cycles", "||b - A u||", "max |u - u*|", "ms/cycle"))
    for smoother in ("gs", "colour", "jacobi"):
        params = {"problemShape": shape, "gridLevels": grids - 1, "preIterations": 1, "postIterations": 1,
                  "threshold": 1e-6 * np.linalg.norm(b), "cycles": 60, "giveInfo": True, "smoother": smoother}
        t0 = time.perf_counter()
        u, info = openmg.mgSolve(A, b, params)
        elapsed = time.perf_counter() - t0
        # the hierarchy mgSolve built comes back in info; reuse it to time resident cycles alone
        h = _hip.Hierarchy(info["A"], info["R"], smoother=smoother, omega=2.0 / 3.0)
        h.resident_load(b)
        h.resident_cycle(1, 1)
        t1 = time.perf_counter()
        for _ in range(10):
            h.resident_cycle(1, 1, want_norm=False)
        h.sync()
        per_cycle = (time.perf_counter() - t1) / 10
        h.close()
        print("%-8s %7d %14.6e %14.6e %10.3f   (solve incl. setup: %.2f s)"
              % (smoother, info["cycle"], info["norm"], np.abs(u - u_exact).max(), 1e3 * per_cycle, elapsed))


if __name__ == "__main__":
    main()
