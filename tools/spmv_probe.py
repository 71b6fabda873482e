#!/usr/bin/env python3
"""Matrix-free fine-grid SpMV of a plane level (plane.hip) at size^3: average launch time and fraction of the HBM peak on
the 2 w n bytes it has to move.  OMG_PLANE_SPMV=0: one thread per output pair; OMG_PLANE_SPMV_LZ: planes per chunk."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, operators  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = sys.argv[2] if len(sys.argv) > 2 else "float64"
shape = (size,) * 3
A0 = operators.stencil_poisson(shape)
h = _hip.Hierarchy.from_fine(A0, shape, max(1, size.bit_length() - 5), smoother="colour", dtype=dtype)
x = np.random.default_rng(1).random(A0.shape[0])
h.resident_load(x, x)
w = 8 if dtype == "float64" else 4
for _ in range(3):
    ms = h.spmv_time(100)
print("size %d %s SPMV=%s LZ=%s: %.2f us per launch, %.1f GB/s on 2 w n, %.3f of 8 TB/s" % (
    size, dtype, os.environ.get("OMG_PLANE_SPMV", "1"), os.environ.get("OMG_PLANE_SPMV_LZ", "auto"), 1e3 * ms, 2 * w * A0.shape[0] / ms / 1e6,
    2 * w * A0.shape[0] / ms / 1e6 / 8000.0))
y = h.spmv(0, x)
print("   max |y - A x| / max |A x| = %.2e" % (np.abs(y - A0 @ x).max() / np.abs(A0 @ x).max()))
