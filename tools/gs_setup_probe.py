import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openmg_amd
from openmg_amd import _hip, operators
shape = (256,)*3
_hip.spmv(operators.stencil_poisson((8, 8, 8)), np.ones(512))
A0 = operators.stencil_poisson(shape)
b = A0 @ np.random.default_rng(1).random(A0.shape[0])
t0=time.perf_counter(); R = operators.restrictionList(shape, 3, 8); t1=time.perf_counter(); A = operators.coeffecientList(A0, R); t2=time.perf_counter()
print("restrictionList %.3f coeffecientList %.3f" % (t1-t0, t2-t1), flush=True)
os.environ["OMG_SETUP_TIMING"]="1"
t0=time.perf_counter(); h = _hip.Hierarchy(A, R, smoother="gs"); print("Hierarchy gs %.3f" % (time.perf_counter()-t0), flush=True)
