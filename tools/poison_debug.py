"""OMG_POISON=1 python tools/poison_debug.py — which path turns poisoned allocations into NaNs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from openmg_amd import _hip
import poison_worker as pw

def probe(name, A, R, dtype, smoother="colour"):
    rng = np.random.default_rng(7)
    b = A[0] @ rng.random(A[0].shape[0])
    x0 = rng.standard_normal(A[0].shape[0])
    with _hip.Hierarchy(A, R, smoother=smoother, dtype=dtype) as h:
        for fused in (True, False):
            h.use_plane(fused)
            for pre, post in ((1, 1), (2, 2)):
                h.resident_load(b, x0)
                norms = [h.resident_cycle(pre, post) for _ in range(2)]
                x = h.resident_fetch()
                print(name, dtype, "fused" if fused else "sets ", (pre, post), "norms", norms, "x finite", bool(np.all(np.isfinite(x))), flush=True)

for dtype in ("float64",):
    A, R = pw.hierarchy7((32, 32, 32), 3)
    probe("plane 32^3", A, R, dtype)
