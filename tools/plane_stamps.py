"""Diagnostic: where a step of the plane-pipelined passes spends its time (needs a library built with
-DOMG_PLANE_STAMPS: make -C openmg_amd/csrc BUILD=build_stamps OUT=../lib/libopenmg_stamps.so EXTRA=-DOMG_PLANE_STAMPS,
run with OMG_LIB_PATH=openmg_amd/lib/libopenmg_stamps.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openmg_amd import _hip, operators
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shape = (size,) * 3
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
R = operators.restrictionList(shape, 3, 8)
A = operators.coeffecientList(A0, R)
with _hip.Hierarchy(A, R, smoother="colour") as h:
    h.resident_load(b)
    for _ in range(3):
        h.resident_cycle(1, 1)
