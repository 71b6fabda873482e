"""The reference's own smoother (lexicographic Gauss-Seidel, openmg/solvers.py:56-68) on operators the wavefront kernel does
not take — per-row coefficients: the level schedule, 3 n - 2 launches per sweep — eager and replayed from a hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmg_amd import _hip, operators
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
grids = int(sys.argv[2]) if len(sys.argv) > 2 else 4
shape = (size,) * 3
for name, A0 in (("7-point, per-row coefficients", operators.stencil7_variable(shape)),):
    b = A0 @ np.random.default_rng(1).random(A0.shape[0])
    R = operators.restrictionList(shape, grids - 2, 8)
    A = operators.coeffecientList(A0, R)
    h = _hip.Hierarchy(A, R, smoother="gs")
    print(name, "%d^3, %d grids; level flags" % (size, len(A)), h.level_flags(0), "sets on the finest level", h.level_sets(0))
    for graph in (False, True):
        h.use_graph(graph)
        h.resident_load(b)
        norms = [h.resident_cycle(1, 1) for _ in range(3)]
        h.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            h.resident_cycle(1, 1, want_norm=False)
        h.sync()
        dt = (time.perf_counter() - t0) / 5
        print("  hipGraph %s: %.2f ms per cycle, %.1f cycles/s; norms %s" % (graph, 1e3 * dt, 1 / dt, ["%.6e" % v for v in norms]))
    h.close()
