#!/usr/bin/env python3
"""The two populations of hierarchies against what rocm-smi says while their cycles run: clocks
(sclk, mclk, fclk, socclk), power, temperatures."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openmg_amd import _hip, operators


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showtemp", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        c = json.loads(out).get("card0", {})
        pick = {}
        for k, v in c.items():
            kl = k.lower()
            if "speed" in kl or "power" in kl or "temperature" in kl:
                pick[k.replace(" clock speed:", "").replace("Temperature (Sensor ", "T(").replace("Current Socket Graphics Package Power (W)", "W")] = v
        return pick
    except Exception as e:                                     # noqa: BLE001
        return {"error": str(e)}


shape = (256,) * 3
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
for k in range(3):
    h = _hip.Hierarchy.from_fine(A0, shape, 4, "colour")
    h.resident_load(b)
    h.resident_cycles(1, 1, 300)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); h.resident_cycles(1, 1, 40); t.append((time.perf_counter() - t0) / 40)
    # keep the GPU busy from a thread-free loop while rocm-smi samples: batches of 2000 cycles are ~0.55 s of queue
    h.resident_cycles(1, 1, 2000)
    s = smi()
    print("pid %d hierarchy %d: %.4f ms per cycle  %s" % (os.getpid(), k, 1e3 * sorted(t)[2], s), flush=True)
    h.close()
