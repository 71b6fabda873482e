#!/usr/bin/env python3
"""Run lexicographic sweeps on a few grid shapes (for rocprofv3 --kernel-trace; tools/march_trace.py
prints the durations): single tiles give the time per step, tile chains the cost per hop."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openmg_amd import _hip, operators  # noqa: E402

shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(8, 8, 2048)]
for shape in shapes:
    A = operators.stencil_poisson(shape)
    n = A.shape[0]
    rng = np.random.default_rng(0)
    b, x = rng.random(n), rng.random(n)
    _hip.gauss_seidel(A, b, x, smoother="gs", iterations=6)
    print(shape, "done")
