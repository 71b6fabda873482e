#!/usr/bin/env python3
"""One process = one line: steady-state ms per cycle (after 300 cycles) of the headline hierarchy and where its level-0
vectors were allocated (OMG_SETUP_TIMING=1 prints the pointers).  Run several times in a row: the pass time differs
between processes on one box by several per cent."""
import os, sys, time
os.environ["OMG_SETUP_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openmg_amd import _hip, operators
shape = (256,) * 3
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
h = _hip.Hierarchy.from_fine(A0, shape, 4, "colour")
h.resident_load(b)
h.resident_cycles(1, 1, 300)
t = []
for _ in range(5):
    t0 = time.perf_counter()
    h.resident_cycles(1, 1, 40)
    t.append((time.perf_counter() - t0) / 40)
h.profile_enable(True)
for _ in range(20):
    h.resident_cycle(1, 1, want_norm=False)
prof = h.profile_read()
h.profile_enable(False)
xcc = ""
if os.path.exists("/tmp/xcc_probe.so"):
    import ctypes
    lib = ctypes.CDLL("/tmp/xcc_probe.so")
    buf = (ctypes.c_uint * 480)()
    if lib.xcc_map(240, 512, 110 * 1024, buf) == 0:
        ids = [buf[2 * i] & 15 for i in range(240)]
        xcc = " xcc of workgroups 0..15: %s; b %% 8 holds for %d of 240" % ("".join(str(v) for v in ids[:16]), sum(1 for i, v in enumerate(ids) if v == (i + ids[0]) % 8))
print("steady ms per cycle %.4f  %s" % (1e3 * sorted(t)[2], {k: round(1e3 * ms / c, 1) for k, (c, ms) in prof.items() if c and "plane" in k}) + xcc, flush=True)
