"""Smoother and coarse solve with the reference's names (openmg/solvers.py), on the GPU."""
import numpy as np

from . import _hip

# Module-wide smoother choice for the standalone calls below; mgSolve/mgCycle take theirs
# from the parameters dict.  'gs' reproduces the reference's lexicographic sweep.
default_smoother = "gs"
default_omega = 1.0


def _inplace(x):
    """(work, writeback): a contiguous float64 1-D view of x, or a copy plus a flag."""
    arr = x if isinstance(x, np.ndarray) else None
    if arr is not None and arr.dtype == np.float64:
        flat = arr.reshape(-1)
        if flat.flags.c_contiguous and np.shares_memory(flat, arr):
            return flat, False
    return np.array(np.asarray(x, dtype=np.float64).reshape(-1), order="C"), True


def gaussSeidel(A, b, x, iterations=None, threshold=None, verbose=False, smoother=None, omega=None):
    """Gauss-Seidel on the device with the reference's stop rules (openmg/solvers.py:34-75):
    stop after `iterations` sweeps and/or once ||b - A x||_2 < threshold (absolute), the norm
    being tested before the first sweep too.  x is updated IN PLACE and returned (Q2)."""
    work, writeback = _inplace(x)
    sweeps = _hip.gauss_seidel(A, np.asarray(b).reshape(-1), work,
                               smoother=default_smoother if smoother is None else smoother,
                               omega=default_omega if omega is None else omega,
                               iterations=iterations, threshold=threshold)
    if verbose:
        print("gaussSeidel: %d sweep(s)" % sweeps)
    if writeback:
        if isinstance(x, np.ndarray):
            x[...] = work.reshape(x.shape)
        else:
            return work
    return x


def smooth(A, b, x, iterations, verbose=False):
    """openmg/solvers.py:28-29."""
    return gaussSeidel(A, b, x, iterations=iterations, verbose=verbose)


def smoothToThreshold(A, b, x, threshold, verbose=False):
    """openmg/solvers.py:31-32."""
    return gaussSeidel(A, b, x, threshold=threshold, verbose=verbose)


def coarseSolve(A, b):
    """Direct solve of A x = b (openmg/solvers.py:16-26); flat result like np.ravel."""
    return _hip.direct_solve(A, np.asarray(b).reshape(-1))
