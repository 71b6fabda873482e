"""mgSolve with the reference's OWN parameters (no 'smoother' key: lexicographic Gauss-Seidel, V(1,0), threshold stop) end to
end, on the constant- and the variable-coefficient 7-point operator."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmg_amd
from openmg_amd import _hip, operators
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shape = (size,) * 3
_hip.spmv(operators.stencil_poisson((8, 8, 8)), np.ones(512))
for name, A0 in (("constant", operators.stencil_poisson(shape)), ("per-row", operators.stencil7_variable(shape))):
    b = A0 @ np.random.default_rng(1).random(A0.shape[0])
    for extra in ({}, {"smoother": "colour"}):
        p = dict({"problemShape": shape, "gridLevels": 4, "cycles": 10, "threshold": 0}, **extra)
        t0 = time.perf_counter()
        x = openmg_amd.mgSolve(A0, b, dict(p))
        print("%s coefficients %d^3, mgSolve %s, 10 cycles V(1,0): %.2f s" % (name, size, extra or "{default: 'gs'}", time.perf_counter() - t0), flush=True)
