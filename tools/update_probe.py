#!/usr/bin/env python3
"""omg_hierarchy_update_fine at configs[4]'s per-GPU size (256^3, 27-point, fp32 levels): wall time of an update from
device-resident values, of update + one cycle; OMG_SETUP_TIMING=1 prints the phases."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, operators  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
grids = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shape = (size,) * 3
A = operators.stencil27_variable(shape)
b = A @ np.random.default_rng(12345).random(A.shape[0])
h = _hip.Hierarchy.from_fine(A, shape, grids - 1, "colour", dtype="float32")
h.resident_load(b)
print("norm after one cycle:", h.resident_cycle(1, 1))
hip = ctypes.CDLL("libamdhip64.so.7")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
d = ctypes.c_void_p()
data = np.ascontiguousarray(A.data * 1.5)
assert hip.hipMalloc(ctypes.byref(d), data.nbytes) == 0 and hip.hipMemcpy(d, data.ctypes.data, data.nbytes, 1) == 0
for k in range(3):
    t0 = time.perf_counter()
    h.update_fine((d.value, data.size), on_device=True)
    t1 = time.perf_counter()
    n = h.resident_cycle(1, 1)
    t2 = time.perf_counter()
    print("update %.2f ms, + one cycle %.2f ms (norm %.4e)" % (1e3 * (t1 - t0), 1e3 * (t2 - t0), n))
