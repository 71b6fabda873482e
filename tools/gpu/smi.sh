#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
{ rocm-smi --showtemp --showclocks --showpower --json 2>&1 | head -c 3000; echo; rocm-smi --showmemuse --showuse 2>&1 | tail -8; } > $o/smi.txt 2>&1
python - <<'PY' >> $o/smi.txt 2>&1
import subprocess, json, time, sys, os
sys.path.insert(0, os.getcwd())
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showtemp", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        c = d.get("card0", {})
        return {k: v for k, v in c.items() if any(s in k.lower() for s in ("temp", "sclk", "mclk", "power", "fclk"))}
    except Exception as e:
        return {"error": str(e)}
print("idle", smi())
import numpy as np
from openmg_amd import _hip, operators
shape = (256,) * 3
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
h = _hip.Hierarchy.from_fine(A0, shape, 4, "colour")
h.resident_load(b)
for rep in range(12):
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); h.resident_cycles(1, 1, 40); t.append((time.perf_counter() - t0) / 40)
    n = 0
    t1 = time.perf_counter()
    while time.perf_counter() - t1 < 8.0:
        h.resident_cycles(1, 1, 200); n += 200
    print("after %3d s of cycles: %.4f ms per cycle" % (int(8 * (rep + 1)), 1e3 * sorted(t)[2]), smi(), flush=True)
PY
