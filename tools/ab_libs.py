#!/usr/bin/env python3
"""A/B of two builds of the library on the headline cycle: alternating processes (OMG_LIB_PATH), each 300 cycles of warm-up
then the median of five regions of 40 cycles; prints every run and the medians.
    python tools/ab_libs.py openmg_amd/lib/libopenmg_hip.so openmg_amd/lib/libopenmg_old.so 6"""
import os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from openmg_amd import _hip, operators
shape = (256,) * 3
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
h = _hip.Hierarchy.from_fine(A0, shape, 4, "colour", dtype=os.environ.get("AB_DTYPE", "float64"))
h.resident_load(b)
h.resident_cycles(1, 1, 300)
t = []
for _ in range(5):
    t0 = time.perf_counter()
    h.resident_cycles(1, 1, 40)
    t.append((time.perf_counter() - t0) / 40)
print("MS %%.5f" %% (1e3 * sorted(t)[2]))
''' % ROOT
# an argument NAME=VALUE instead of a library path: the default library with that environment variable set
libs = sys.argv[1:3]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
res = {l: [] for l in libs}
for i in range(n):
    for l in libs:
        if "=" in l and not l.endswith(".so"):
            env = dict(os.environ)
            for kv in l.split("+"):                 # (several variables: A=1+B=2)
                env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
        else:
            env = dict(os.environ, OMG_LIB_PATH=os.path.join(ROOT, l))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout
        ms = [float(x.split()[1]) for x in out.splitlines() if x.startswith("MS ")]
        if ms:
            res[l].append(ms[0])
for l in libs:
    v = res[l]
    print("%-40s median %.4f ms  min %.4f  max %.4f  (%s)" % (l, statistics.median(v), min(v), max(v), " ".join("%.4f" % x for x in v)))
