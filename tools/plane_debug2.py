"""Debug: impulse response of one cycle, plane on / off (fp32)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from openmg_amd import _hip
from test_gpu_plane import hierarchy, run
shape = (8, 8, 8)
A, R = hierarchy(shape, 2)
n = A[0].shape[0]
dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
for cell in ((3, 3, 2), (3, 3, 3)):
    x0 = np.zeros(n); x0[np.ravel_multi_index(cell, shape)] = 1.0
    b = np.zeros(n)
    with _hip.Hierarchy(A, R, smoother="colour", dtype=dtype) as h:
        got = run(h, b, 1, 1, 1, x0)
        h.use_plane(False)
        ref = run(h, b, 1, 1, 1, x0)
    print("impulse at", cell, "norm", got[0], ref[0])
    d = np.flatnonzero(np.abs(got[1] - ref[1]) > 1e-6)
    print(" differing", d.size)
    for i in d[:24]:
        print("   ", np.unravel_index(i, shape), "plane %.6f sets %.6f" % (got[1][i], ref[1][i]))
