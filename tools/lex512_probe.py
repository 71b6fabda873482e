"""Lexicographic Gauss-Seidel V-cycles on 512^3 / 6 grids (one GPU): timing and a host check of the norm."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmg_amd import _hip, operators
shape, grids = (512, 512, 512), 6
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(5).random(A0.shape[0])
R = operators.restrictionList(shape, grids - 2, 8)
A = operators.coeffecientList(A0, R)
t0 = time.perf_counter()
h = _hip.Hierarchy(A, R, smoother="gs")
print("setup %.1f s" % (time.perf_counter() - t0), [h.level_flags(l)["march"] for l in range(grids - 1)])
h.resident_load(b)
t0 = time.perf_counter()
norms = [h.resident_cycle(1, 1) for _ in range(3)]
h.sync()
print("ms/cycle %.2f" % ((time.perf_counter() - t0) / 3 * 1e3), norms)
x = h.resident_fetch()
host = float(np.linalg.norm(b - A0 @ x))
print("host norm", host, "rel", abs(host - norms[-1]) / host)
