#!/usr/bin/env python3
"""ms per cycle of the headline over several seconds of uninterrupted load (does the chip's clock keep rising?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openmg_amd import _hip, operators
shape = (256,) * 3
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
h = _hip.Hierarchy.from_fine(A0, shape, 4, "colour")
h.resident_load(b)
t_start = time.perf_counter()
out = []
for k in range(40):
    t0 = time.perf_counter()
    h.resident_cycles(1, 1, 500)
    t1 = time.perf_counter()
    out.append("%.2fs:%.4f" % (t1 - t_start, 1e3 * (t1 - t0) / 500))
print(" ".join(out))
