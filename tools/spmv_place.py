#!/usr/bin/env python3
"""Where the matrix-free SpMV writes, inside ONE process: the level's scratch vector (default), the right-hand side's vector,
allocations of its own at several staggers — is the fast / slow split between processes a matter of where x and y lie?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, operators  # noqa: E402

size = 256
shape = (size,) * 3
A0 = operators.stencil_poisson(shape)
h = _hip.Hierarchy.from_fine(A0, shape, 4, smoother="colour")
x = np.random.default_rng(1).random(A0.shape[0])
h.resident_load(x, x)
out = []
for y in ["tmp", "b", "o0", "o4", "o8", "o16", "o64", "o1024", "tmp", "own (default)"]:
    if y.startswith("own"):
        os.environ.pop("OMG_SPMV_Y", None)
    else:
        os.environ["OMG_SPMV_Y"] = y
    h.spmv_time(50)
    out.append("%s %.1f" % (y, 1e3 * h.spmv_time(100)))
print("us per launch by destination:", "  ".join(out))
