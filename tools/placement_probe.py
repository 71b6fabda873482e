#!/usr/bin/env python3
"""Does the time of the fine-grid passes depend on WHERE the level's vectors were allocated?  The same hierarchy is
created several times in one process (the earlier ones kept alive, or freed, so that hipMalloc hands out other ranges);
every instance times its own cycles."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openmg_amd import _hip, operators

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
keep = (sys.argv[2] == "keep") if len(sys.argv) > 2 else False
shape = (size,) * 3
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
held = []
for trial in range(8):
    h = _hip.Hierarchy.from_fine(A0, shape, 4, "colour")
    h.resident_load(b)
    h.resident_cycles(1, 1, 5)
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        h.resident_cycles(1, 1, 20)
        t.append((time.perf_counter() - t0) / 20)
    print("trial %d (%s): ms per cycle %s" % (trial, "earlier instances kept" if keep else "earlier instances freed", " ".join("%.4f" % (1e3 * v) for v in t)), flush=True)
    if keep:
        held.append(h)
    else:
        h.close()
