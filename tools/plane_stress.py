"""Stress of the plane passes' intra-workgroup synchronisation (a wave waits for its neighbour waves, no barrier):
hundreds of cycles at 256^3 / 128^3 / 96^3, the iterate against the set-by-set schedule bit for bit, three hierarchies each."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from openmg_amd import _hip, operators
for size, grids, cycles in ((256, 5, 150), (128, 4, 300), (96, 4, 300)):
    shape = (size,) * 3
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(1).random(A0.shape[0])
    R = operators.restrictionList(shape, grids - 2, 8)
    A = operators.coeffecientList(A0, R)
    os.environ["OMG_PLANE"] = "0"
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        h.resident_load(b); h.resident_cycles(1, 1, cycles); ref = h.resident_fetch()
    os.environ["OMG_PLANE"] = "1"
    for rep in range(3):
        with _hip.Hierarchy(A, R, smoother="colour") as h:
            h.resident_load(b); h.resident_cycles(1, 1, cycles); x = h.resident_fetch()
        print(size, "rep", rep, "bitwise equal after", cycles, "cycles:", bool(np.array_equal(x, ref)), flush=True)
