#!/usr/bin/env python3
"""The reference's usage demo (openmg_usage_demo.py: simpleDemo + explainedDemo) run against
the MI355X implementation through the drop-in alias package.  Prints the same table:

    N    method     norm            seconds cycles
"""
import os
import sys
from time import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openmg  # noqa: E402


def simple_demo(verbose=True):
    """u = sin(x/10) on 100 points, 1-D (4,-1) operator; the reference documents residual norms
    0.805398, 0.107866, 0.018650, 0.003405 for its two-restriction hierarchy."""
    N = 100
    u_true = np.array([np.sin(x / 10.0) for x in np.linspace(0, 20, N)])
    A = openmg.operators.poisson(N, sparse=True)
    b = openmg.tools.flexibleMmult(A, u_true)
    params = {"problemShape": (N,), "gridLevels": 2, "cycles": 10, "iterations": 2, "verbose": verbose,
              "dense": True, "threshold": 1e-2, "giveInfo": True}
    u_mg, info = openmg.mgSolve(A, b, params)
    print("info: norm %.16g after %d cycles" % (info["norm"], info["cycle"]))
    return u_mg


def explained_demo(N, dense=False):
    threshold = 1e-14
    u_true = np.array([np.sin(x / 10.0) for x in range(N)])
    A = openmg.operators.poisson(N, sparse=True)
    b = openmg.tools.flexibleMmult(A, u_true)
    start = time()
    soln = openmg.solvers.coarseSolve(A, b)
    print(N, "direct", np.linalg.norm(openmg.tools.getresidual(b, A, soln, N)), time() - start)
    if N <= 200:
        start = time()
        soln = openmg.smoothToThreshold(A, b, np.zeros((N, 1)), threshold)
        print(N, "Gauss-Seidel", np.linalg.norm(openmg.tools.getresidual(b, A, soln, N)), time() - start)
    params = {"problemShape": (N,), "gridLevels": 3, "cycles": 0, "iterations": 2, "verbose": False, "dense": dense,
              "threshold": threshold, "giveInfo": True, "minSize": 30}
    start = time()
    soln, info = openmg.mgSolve(A, b, params)
    print(N, "%i-grid" % params["gridLevels"], np.linalg.norm(openmg.tools.getresidual(b, A, soln, N)),
          time() - start, info["cycle"])


if __name__ == "__main__":
    simple_demo()
    print()
    print("N    method     norm            seconds cycles")
    for n in [int(a) for a in sys.argv[1:]] or [200, 1000]:
        explained_demo(n)
        print()
