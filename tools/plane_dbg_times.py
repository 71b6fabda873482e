"""Timing experiments with the -DOMG_PLANE_STAMPS build (OMG_LIB_PATH=openmg_amd/lib/libopenmg_stamps.so): how the
finest level's passes respond when part of their memory traffic is left out (OMG_PLANE_DBG bits: 1 no x stores,
2 no coarse stores, 4 no loads of x, 8 no loads of b; results are wrong, only the durations mean something)."""
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
for _ in range(100):
    h.resident_cycle(1, 1, want_norm=False)
h.profile_enable(True)
for _ in range(20):
    h.resident_cycle(1, 1, want_norm=False)
prof = h.profile_read()
print("OMG_PLANE_DBG=%s" % os.environ.get("OMG_PLANE_DBG", "0"), {k: round(1e3 * ms / c, 1) for k, (c, ms) in prof.items() if c and "plane" in k}, flush=True)
