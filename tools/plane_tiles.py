"""A/B of plane-pipelined tilings at full size: OMG_PLANE_TILE candidates, per-kernel hipEvent times.
usage: plane_tiles.py [size] [grids] tile tile ...   (tile = TX,TY,LZ; 'auto' = the library's choice)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openmg_amd import _hip, operators

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
grids = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tiles = sys.argv[3:] or ["auto"]
dtype = os.environ.get("PLANE_DTYPE", "float64")
shape = (size,) * 3
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
R = operators.restrictionList(shape, grids - 2, 8)
A = operators.coeffecientList(A0, R)
ref = None
for tile in tiles:
    if tile == "auto":
        os.environ.pop("OMG_PLANE_TILE", None)
    else:
        os.environ["OMG_PLANE_TILE"] = tile
    with _hip.Hierarchy(A, R, smoother="colour", dtype=dtype) as h:
        h.resident_load(b)
        norms = h.resident_cycles(1, 1, 5)
        h.sync()
        t0 = time.perf_counter()
        K = 40
        h.resident_cycles(1, 1, K)
        h.sync()
        dt = (time.perf_counter() - t0) / K
        h.profile_enable(True)
        for _ in range(10):
            h.resident_cycle(1, 1, want_norm=False)
        prof = h.profile_read()
        h.profile_enable(False)
        x = h.resident_fetch()
        if ref is None:
            ref = x
        same = bool(np.array_equal(x, ref))
        info = h.plane_info(0)
        d, u = prof["plane_down"], prof["plane_up"]
        print("%-12s tile %s wg %d thr %d | cycle %.1f us (%.0f/s) | down %.1f us  up %.1f us | same bits %s"
              % (tile, (info["tile_x"], info["tile_y"], info["tile_z"]), info["workgroups"], info["threads"], dt * 1e6, 1 / dt,
                 1e3 * d[1] / max(d[0], 1), 1e3 * u[1] / max(u[0], 1), same), flush=True)
