"""Timing experiments on the var7 passes (OMG_VAR7_DBG bits: 1 no coefficient loads, 2 no arithmetic, 4 no plane loads, 8 no write-back)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmg_amd import _hip, operators
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shape = (size,) * 3
A0 = operators.stencil7_variable(shape)
b = A0 @ np.random.default_rng(1).random(A0.shape[0])
R = operators.restrictionList(shape, 3, 8)
A = operators.coeffecientList(A0, R)
for dbg in [int(v) for v in os.environ.get("DBGS", "0,1,4,15").split(",")]:
    os.environ["OMG_VAR7_DBG"] = str(dbg)
    for lz in os.environ.get("LZS", "0").split(","):
        h = _hip.Hierarchy(A, R, smoother="colour")
        h.resident_load(b)
        h.resident_cycles(1, 1, 5)
        h.profile_enable(True)
        h.resident_cycles(1, 1, 20)
        prof = h.profile_read()
        h.profile_enable(False)
        print("dbg %2d: " % dbg + "  ".join("%s %.1f us" % (k, 1e3 * v[1] / max(v[0], 1)) for k, v in prof.items() if v[0]))
        h.close()
