"""The coarse solve beyond the old limits at full size: 27-point variable-coefficient 64^3 as a
coarsest operator — 262 144 unknowns, half-bandwidth 4161 — standalone (residual of the direct solve) and as the
coarsest level of mgSolve with the reference's gridLevels = 1 on 128^3.  Prints times and norms."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openmg_amd
from openmg_amd import _hip, operators

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
A = operators.stencil27_variable((size,) * 3)
n = A.shape[0]
b = np.random.default_rng(3).standard_normal(n)
t0 = time.perf_counter()
x = _hip.direct_solve(A, b)
t1 = time.perf_counter()
r = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
print("direct solve of the %d^3 27-point operator (n = %d): %.2f s including the factorisation, relative residual %.2e" % (size, n, t1 - t0, r))
fine = (2 * size,) * 3
A0 = operators.stencil27_variable(fine)
b0 = A0 @ np.random.default_rng(12345).random(A0.shape[0])
p = {"problemShape": fine, "gridLevels": 1, "preIterations": 1, "postIterations": 1, "cycles": 1, "threshold": 0,
     "giveInfo": True, "smoother": "colour", "minSize": 8}
t0 = time.perf_counter()
x1, info = openmg_amd.mgSolve(A0, b0, dict(p))
t1 = time.perf_counter()
print("mgSolve %d^3, gridLevels = 1 (coarsest %d^3), 1 cycle: %.2f s, norm %.6e (|b| = %.6e)" % (2 * size, size, t1 - t0, info["norm"], np.linalg.norm(b0)))
R, Al = info["R"], info["A"]
with _hip.Hierarchy(Al, R, smoother="colour") as h:
    print("coarse solver:", h.coarse_info())
    h.resident_load(b0)
    h.resident_cycles(1, 1, 2)
    h.sync()
    t0 = time.perf_counter()
    norms = h.resident_cycles(1, 1, 5)
    h.sync()
    print("5 resident cycles: %.2f ms per cycle, norms %s" % (1e3 * (time.perf_counter() - t0) / 5, ["%.4e" % v for v in norms]))
    bc = np.random.default_rng(1).random(n)
    h.coarse_solve(bc)
    t0 = time.perf_counter()
    xc = h.coarse_solve(bc)
    print("coarse solve through the hierarchy incl. PCIe: %.2f ms, relative residual %.2e" % (1e3 * (time.perf_counter() - t0), np.linalg.norm(bc - Al[1] @ xc) / np.linalg.norm(bc)))
