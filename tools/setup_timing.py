#!/usr/bin/env python3
"""Where mgSolve's setup time goes at 256^3 (5 grids, red-black)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, operators  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    smoother = sys.argv[2] if len(sys.argv) > 2 else "colour"
    shape = (n, n, n)
    t = time.perf_counter()
    A0 = operators.stencil_poisson(shape)
    print("stencil_poisson      %.3f s" % (time.perf_counter() - t)); t = time.perf_counter()
    _hip.spmv(operators.stencil_poisson((8, 8, 8)), np.ones(512))   # device warm-up
    print("device warm-up       %.3f s" % (time.perf_counter() - t)); t = time.perf_counter()
    R = operators.restrictionList(shape, 3, 8)
    print("restrictionList      %.3f s" % (time.perf_counter() - t)); t = time.perf_counter()
    A = operators.coeffecientList(A0, R)
    print("coeffecientList      %.3f s" % (time.perf_counter() - t)); t = time.perf_counter()
    h = _hip.Hierarchy(A, R, smoother=smoother)
    print("Hierarchy (%s)   %.3f s" % (smoother, time.perf_counter() - t)); t = time.perf_counter()
    b = A0 @ np.ones(A0.shape[0])
    h.resident_load(b)
    for _ in range(10):
        norm = h.resident_cycle(1, 1)
    x = h.resident_fetch()
    print("load + 10 cycles + fetch %.3f s (norm %.3e)" % (time.perf_counter() - t, norm))
    h.close()


if __name__ == "__main__":
    main()
