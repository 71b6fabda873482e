#!/usr/bin/env python3
"""Where restrictionList / coeffecientList / mgCycle spend their host time at 256^3 (Python-level timers)."""
import ctypes
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import openmg_amd  # noqa: E402
from openmg_amd import _hip, operators  # noqa: E402


def T(label, t0):
    print("%-44s %8.1f ms" % (label, 1e3 * (time.perf_counter() - t0)))
    return time.perf_counter()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    shape = (n, n, n)
    A0 = operators.stencil_poisson(shape)
    _hip.spmv(operators.stencil_poisson((8, 8, 8)), np.ones(512))
    t = time.perf_counter()
    R0 = _hip.restriction(shape)
    t = T("_hip.restriction(256^3) (device build + fetch)", t)
    R0.sort_indices()
    t = T("R.sort_indices()", t)
    R = operators.restrictionList(shape, 3, 8)
    t = T("restrictionList (all levels)", t)
    X, Y = _hip.as_csr(R[0]), _hip.as_csr(A0)
    t = T("as_csr(R0), as_csr(A0)", t)
    vx, vy = _hip.csr_view(X), _hip.csr_view(Y)
    res = ctypes.c_void_p()
    nr, nc, nnz = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    _hip.check(_hip.lib().omg_rap(ctypes.byref(vx), ctypes.byref(vy), ctypes.byref(res), ctypes.byref(nr), ctypes.byref(nc), ctypes.byref(nnz)))
    t = T("omg_rap level 0 (upload + kernels)", t)
    indptr = np.empty(nr.value + 1, dtype=np.int32)
    indices = np.empty(nnz.value, dtype=np.int32)
    data = np.empty(nnz.value, dtype=np.float64)
    t = T("np.empty x3", t)
    _hip.check(_hip.lib().omg_csr_result_fetch(res, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data))
    t = T("omg_csr_result_fetch", t)
    A1 = sp.csr_matrix((data, indices, indptr), shape=(nr.value, nc.value))
    t = T("sp.csr_matrix((data, indices, indptr))", t)
    A = operators.coeffecientList(A0, R)
    t = T("coeffecientList (all levels)", t)
    h = _hip.Hierarchy(A, R, smoother="colour")
    t = T("Hierarchy", t)
    h.close()
    b = A0 @ np.ones(A0.shape[0])
    prm = {"coarsestLevel": len(R), "preIterations": 1, "postIterations": 1, "smoother": "colour"}
    t = time.perf_counter()
    x, info = openmg_amd.mgCycle(A, b, 0, R, prm)
    t = T("mgCycle first call", t)
    for _ in range(3):
        x, info = openmg_amd.mgCycle(A, b, 0, R, prm, initial=x)
        t = T("mgCycle call (initial given: Q2 path)", t)
    fp = openmg_amd._fingerprint(A, R, len(A), 1, 1.0, 0)
    t = T("  _fingerprint alone", t)
    y = np.empty(b.size)
    t = T("  np.empty(n)", t)
    hh = openmg_amd._hierarchy_for(A, R, len(A), *openmg_amd._smoother_of(prm))
    t = T("  _hierarchy_for (fingerprint + lookup)", t)
    nrm = hh.vcycle_ex(b, x, y, x, 1, 1)
    t = T("  vcycle_ex (b, x in; x out, x_pre out)", t)
    nrm = hh.vcycle_ex(b, x, y, None, 1, 1)
    t = T("  vcycle_ex without x_pre", t)
    openmg_amd.clear_cache()


if __name__ == "__main__":
    main()
