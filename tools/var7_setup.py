"""Where the setup of a variable-coefficient 7-point hierarchy goes (OMG_SETUP_TIMING=1), and mgSolve end to end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmg_amd
from openmg_amd import _hip, operators
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shape = (size,) * 3
A0 = operators.stencil7_variable(shape)
b = A0 @ np.random.default_rng(1).random(A0.shape[0])
_hip.spmv(operators.stencil_poisson((8, 8, 8)), np.ones(512))
for smoother in ("colour", "gs"):
    p = {"problemShape": shape, "gridLevels": 4, "preIterations": 1, "postIterations": 1, "cycles": 20, "threshold": 0, "smoother": smoother}
    t0 = time.perf_counter()
    x = openmg_amd.mgSolve(A0, b, dict(p))
    t1 = time.perf_counter()
    print("mgSolve %s, 20 cycles, %d^3 variable-coefficient 7-point: %.2f s" % (smoother, size, t1 - t0), flush=True)
