"""The bodies of tests/test_gpu_devarray.py, each run in a process of its own: PyTorch-ROCm brings its own copy of the HIP
runtime, and a caller with device tensors has initialised it BEFORE this package's library touches the GPU — the order a
fresh process gives; late in a long pytest process (hundreds of hierarchies made and destroyed through the other runtime
copy) PyTorch's initialisation found no device.  argv: case name, then its parameters."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

torch.cuda.init()

import openmg_amd
from openmg_amd import _hip, operators


def lists(shape, grid_levels):
    A0 = operators.stencil_poisson(shape)
    R = operators.restrictionList(shape, grid_levels - 1, 8)
    return operators.coeffecientList(A0, R), R


def chained(smoother, pre, post):
    pre, post = int(pre), int(post)
    shape = (32, 32, 32)
    A, R = lists(shape, 3)
    b = A[0] @ np.random.default_rng(3).random(A[0].shape[0])
    x0 = np.random.default_rng(4).standard_normal(b.size)
    p = {"coarsestLevel": len(R), "preIterations": pre, "postIterations": post, "smoother": smoother}
    # host arrays: the reference's contract, Q2 included
    xh = x0.copy()
    host_norms, host_initials = [], []
    for _ in range(3):
        start = xh
        xh, info = openmg_amd.mgCycle(A, b, 0, R, p, initial=start)
        host_norms.append(info["norm"])
        host_initials.append(start.copy())               # (what the call left in `initial`)
    # device arrays
    bd = torch.from_numpy(b).cuda()
    xd = torch.from_numpy(x0.copy()).cuda()
    for k in range(3):
        start = xd
        xd, info = openmg_amd.mgCycle(A, bd, 0, R, dict(p, trustOperators=True), initial=start)
        assert isinstance(xd, torch.Tensor) and xd.is_cuda and xd.dtype == torch.float64 and xd.data_ptr() != start.data_ptr()
        assert info["norm"] == host_norms[k]
        assert np.array_equal(start.cpu().numpy(), host_initials[k])      # Q2: pre-smoothed in place (untouched when pre = 0)
    assert np.array_equal(xd.cpu().numpy(), xh)
    # no initial: zeros (openmg/__init__.py:191-192)
    x1, i1 = openmg_amd.mgCycle(A, bd, 0, R, p)
    x2, i2 = openmg_amd.mgCycle(A, b, 0, R, p)
    assert np.array_equal(x1.cpu().numpy(), x2) and i1["norm"] == i2["norm"]
    openmg_amd.clear_cache()


def trusted():
    shape = (16, 16, 16)
    A, R = lists(shape, 2)
    b = A[0] @ np.ones(A[0].shape[0])
    p = {"coarsestLevel": len(R), "preIterations": 1, "postIterations": 1, "smoother": "colour", "trustOperators": True}
    x, info = openmg_amd.mgCycle(A, b, 0, R, p)
    calls = []
    real = _hip.host_checksum
    _hip.host_checksum = lambda a: calls.append(1) or real(a)
    try:
        x2, info2 = openmg_amd.mgCycle(A, b, 0, R, p)
        assert not calls and np.array_equal(x, x2)                        # same objects: taken on trust
        A2 = [M.copy() for M in A]
        A2[0].data[:] *= 2.0
        x3, info3 = openmg_amd.mgCycle(A2, b, 0, R, p)                   # other objects: checksummed, a new hierarchy
        assert calls and not np.array_equal(x3, x)
    finally:
        _hip.host_checksum = real
    bd = torch.from_numpy(b).cuda()
    for bad in (lambda: openmg_amd.mgCycle(A, b, 0, R, p, initial=bd), lambda: openmg_amd.mgCycle(A, bd.float(), 0, R, p)):
        try:
            bad()
        except TypeError:
            continue
        raise AssertionError("expected a TypeError")
    openmg_amd.clear_cache()


def mgsolve(give_info):
    give_info = give_info == "1"
    shape = (32, 32, 32)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(8).random(A0.shape[0])
    p = {"problemShape": shape, "gridLevels": 2, "cycles": 4, "threshold": 0, "preIterations": 1, "postIterations": 1,
         "smoother": "colour", "giveInfo": give_info}
    want = openmg_amd.mgSolve(A0, b, dict(p))
    got = openmg_amd.mgSolve(A0, torch.from_numpy(b).cuda().reshape(-1, 1), dict(p))      # ((n, 1) as the reference's tests pass b)
    if give_info:
        assert got[1]["norm"] == want[1]["norm"] and got[1]["cycle"] == 4
        got, want = got[0], want[0]
    assert isinstance(got, torch.Tensor) and got.is_cuda
    assert np.array_equal(got.cpu().numpy(), want)


if __name__ == "__main__":
    {"chained": chained, "trusted": trusted, "mgsolve": mgsolve}[sys.argv[1]](*sys.argv[2:])
    print("ok")
