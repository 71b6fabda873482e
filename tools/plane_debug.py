"""Debug: plane passes on / off in fp32 and fp64 on a small grid (norm traces, first differences)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openmg_amd import _hip, operators
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_plane import hierarchy, run

shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,16,16").split(","))
grids = int(sys.argv[2]) if len(sys.argv) > 2 else 2
A, R = hierarchy(shape, grids)
rng = np.random.default_rng(5)
b = (A[0] @ rng.random(A[0].shape[0])).astype(np.float32).astype(np.float64)
for dtype in ("float64", "float32"):
    with _hip.Hierarchy(A, R, smoother="colour", dtype=dtype) as h:
        print(dtype, "plane flags", [h.level_flags(l)["plane"] for l in range(grids - 1)], h.plane_info(0))
        got = run(h, b, 1, 1, 3)
        h.use_plane(False)
        ref = run(h, b, 1, 1, 3)
        print(" plane norms", got[0])
        print(" sets  norms", ref[0])
        d = np.flatnonzero(got[1] != ref[1])
        print(" differing", d.size, "of", got[1].size, "first", d[:10], "max abs", np.max(np.abs(got[1] - ref[1])))
