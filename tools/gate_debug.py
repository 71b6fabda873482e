#!/usr/bin/env python3
"""The gated slab passes' missing-exchange path, step by step (debugging aid)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["OMG_PDIST_GATE"] = "1"
os.environ["OMG_PLANE_TILE"] = "64,16,16"
os.environ["OMG_P2P_SPIN"] = "2000"
from openmg_amd import _hip, _hip_dist  # noqa: E402
from test_gpu_plane_dist import problem, slabs  # noqa: E402

shape, grids, n_dist = (192, 64, 64), 4, 2
A, R, b, x0 = problem(shape, grids)
g = _hip_dist.PlaneDistGroup(slabs(A, R, shape, 2, n_dist, b, x0))
print(g.ranks[0].info())
print("cycle 1", g.cycles(1), [r.p2p_status() for r in g.ranks])
os.environ["OMG_PDIST_GATE_POISON"] = "1"
for k in range(3):
    try:
        print("cycle", k + 2, g.cycles(1))
    except RuntimeError as e:
        print("cycle", k + 2, "raised:", e)
    print("   status", [r.p2p_status() for r in g.ranks])
g.close()
