#!/usr/bin/env python3
"""What does a 7-point operator with per-row coefficients (finite volumes, random cell-wise kappa, Dirichlet) cost today?
256^3, 5 grids, red-black, fp64, Galerkin products on the device; which kernels run it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
from openmg_amd import _hip, operators

def var7(shape, seed=7):
    nx, ny, nz = shape
    rng = np.random.default_rng(seed)
    kap = np.exp(rng.standard_normal((nz + 2, ny + 2, nx + 2)))         # cell-wise coefficient with a halo
    def face(a, b): return 2.0 * a * b / (a + b)
    c = kap[1:-1, 1:-1, 1:-1]
    w = {"xm": face(c, kap[1:-1, 1:-1, :-2]), "xp": face(c, kap[1:-1, 1:-1, 2:]), "ym": face(c, kap[1:-1, :-2, 1:-1]),
         "yp": face(c, kap[1:-1, 2:, 1:-1]), "zm": face(c, kap[:-2, 1:-1, 1:-1]), "zp": face(c, kap[2:, 1:-1, 1:-1])}
    diag = sum(w.values()).ravel()
    n = nx * ny * nz
    idx = np.arange(n).reshape(nz, ny, nx)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [diag]
    for key, (dz, dy, dx) in {"xm": (0, 0, -1), "xp": (0, 0, 1), "ym": (0, -1, 0), "yp": (0, 1, 0), "zm": (-1, 0, 0), "zp": (1, 0, 0)}.items():
        src = idx[max(0, -dz):nz - max(0, dz), max(0, -dy):ny - max(0, dy), max(0, -dx):nx - max(0, dx)]
        dst = idx[max(0, dz):nz + min(0, dz) or None, max(0, dy):ny + min(0, dy) or None, max(0, dx):nx + min(0, dx) or None]
        ww = w[key][max(0, -dz):nz - max(0, dz), max(0, -dy):ny - max(0, dy), max(0, -dx):nx - max(0, dx)]
        rows.append(src.ravel()); cols.append(dst.ravel()); vals.append(-ww.ravel())
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    A.sort_indices()
    return A

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shape = (size,) * 3
A0 = var7(shape)
b = A0 @ np.random.default_rng(1).random(A0.shape[0])
t = time.perf_counter()
R = operators.restrictionList(shape, 3, 8)
A = operators.coeffecientList(A0, R)
h = _hip.Hierarchy(A, R, smoother="colour")
print("setup %.2f s; level flags %s" % (time.perf_counter() - t, h.level_flags(0)))
h.resident_load(b)
norms = h.resident_cycles(1, 1, 10)
t0 = time.perf_counter()
h.resident_cycles(1, 1, int(os.environ.get("PROBE_CYCLES", "40")))
dt = (time.perf_counter() - t0) / 40
print("variable-coefficient 7-point %d^3: %.3f ms per cycle, %.1f V-cycles/s; norms %.3e -> %.3e; sets %d" % (size, 1e3 * dt, 1 / dt, norms[0], norms[-1], h.level_sets(0)))
x_fused = h.resident_fetch()
h.use_plane(False)
h.resident_load(b)
h.resident_cycles(1, 1, 10)
t0 = time.perf_counter()
h.resident_cycles(1, 1, int(os.environ.get("PROBE_CYCLES", "40")))
dt = (time.perf_counter() - t0) / 40
print("... the same hierarchy set by set (omg_hierarchy_use_plane(0)): %.3f ms per cycle, %.1f V-cycles/s; same bits: %s" % (1e3 * dt, 1 / dt, np.array_equal(h.resident_fetch(), x_fused)))
print(h.format_info(0, "A", 0))
