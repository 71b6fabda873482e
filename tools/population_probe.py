#!/usr/bin/env python3
"""The two populations of processes: is the state a property of the process, of its allocations, or of the
box?  One process: three hierarchies in turn (each freed before the next), each timed twice.  Run several in a row."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openmg_amd import _hip, operators
shape = tuple(int(v) for v in os.environ.get("PROBE_SHAPE", "256,256,256").split(","))
nr = int(os.environ.get("PROBE_RESTRICTIONS", "4"))
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
out = []
lists = len(sys.argv) > 1 and sys.argv[1] == "lists"       # the host-list route (bench.py's headline) instead of the device setup
if lists:
    R = operators.restrictionList(shape, nr - 1, 1)
    A = operators.coeffecientList(A0, R)
for k in range(3):
    h = _hip.Hierarchy(A, R, smoother="colour") if lists else _hip.Hierarchy.from_fine(A0, shape, nr, "colour")
    h.resident_load(b)
    h.resident_cycles(1, 1, 300)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); h.resident_cycles(1, 1, 40); t.append((time.perf_counter() - t0) / 40)
    out.append("%.4f" % (1e3 * sorted(t)[2]))
    h.close()
print("pid %d (%s): ms per cycle of three hierarchies in turn: %s" % (os.getpid(), "lists" if lists else "from_fine", " ".join(out)), flush=True)
