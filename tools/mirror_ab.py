#!/usr/bin/env python3
"""A/B inside one process, after the clocks have settled: the up pass marching down (OMG_PLANE_MIRROR=1) or up (0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openmg_amd import _hip, operators

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shape = (size,) * 3
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
h = _hip.Hierarchy.from_fine(A0, shape, 4, "colour")
h.resident_load(b)
h.resident_cycles(1, 1, 200)
res = {"0": [], "1": []}
for rep in range(12):
    for m in ("1", "0"):
        os.environ["OMG_PLANE_MIRROR"] = m
        h.resident_cycles(1, 1, 10)
        t0 = time.perf_counter()
        h.resident_cycles(1, 1, 40)
        res[m].append((time.perf_counter() - t0) / 40)
for m in ("1", "0"):
    v = sorted(res[m])
    print("mirror %s: ms per cycle median %.4f  min %.4f  max %.4f" % (m, 1e3 * v[len(v) // 2], 1e3 * v[0], 1e3 * v[-1]))
