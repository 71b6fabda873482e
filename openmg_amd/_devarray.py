"""Device arrays at the Python boundary: mgCycle / mgSolve accept `b` and `initial` that already live in HBM — any object
with `__cuda_array_interface__` (PyTorch-ROCm tensors have it) holding contiguous float64 — and return the same kind, so
that chained calls (openmg/__init__.py:132-138: uOut of one cycle is `initial` of the next) move nothing over PCIe.
PyTorch is used for what it is here for — device memory — and only when the caller hands device arrays in."""
import numpy as np

from . import _hip


def is_device_array(a):
    return a is not None and not isinstance(a, np.ndarray) and hasattr(a, "__cuda_array_interface__")


def address(a, n, what):
    """Device address of a contiguous float64 array of n elements (1-D or (n, 1), as the reference takes b)."""
    iface = a.__cuda_array_interface__
    size = 1
    for s in iface["shape"]:
        size *= int(s)
    if iface["typestr"] not in ("<f8", "=f8", "|f8"):
        raise TypeError("%s: device arrays must be float64 (got %s)" % (what, iface["typestr"]))
    if size != n:
        raise ValueError("%s has %d elements, the level has %d" % (what, size, n))
    if iface.get("strides") is not None:
        expect, ok = 8, True
        for s, st in zip(reversed(iface["shape"]), reversed(iface["strides"])):
            ok = ok and (int(s) == 1 or int(st) == expect)
            expect *= int(s)
        if not ok:
            raise ValueError("%s: device arrays must be contiguous" % what)
    return int(iface["data"][0])


def empty_like(a, n):
    """A new float64 device array of n elements of the caller's kind (PyTorch tensors; anything else with the
    interface gets a PyTorch tensor on the current device)."""
    import torch
    device = a.device if isinstance(a, torch.Tensor) else torch.device("cuda", torch.cuda.current_device())
    return torch.empty(n, dtype=torch.float64, device=device)


def synchronize():
    """What the caller's library enqueued (on whatever stream) is complete before the hierarchy's own stream reads it."""
    _hip.check(_hip.lib().omg_device_synchronize())
