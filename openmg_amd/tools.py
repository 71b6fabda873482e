"""Helpers with the reference's names (openmg/tools.py), backed by the HIP library.

Every product in here that involves a sparse operand runs on the GPU through
libopenmg_hip.so; there is no NumPy/SciPy arithmetic fallback.
"""
import numpy as np
import scipy.sparse as sp

from . import _hip


def _is_vector(y):
    return (not sp.issparse(y)) and (np.ndim(y) == 1 or (np.ndim(y) == 2 and np.shape(y)[1] == 1))


def flexibleMmult(x, y):
    """Product of two 2-D operands, either of which may be sparse (openmg/tools.py:18-26).

    matrix @ vector -> device CSR SpMV (the shape of y — (N,) or (N,1) — is preserved, as
    SciPy's `*` does); matrix @ matrix -> device SpGEMM, returned as CSR when BOTH operands
    are sparse and as ndarray otherwise (what SciPy's `*` / np.dot return).
    """
    if _is_vector(y):
        out = _hip.spmv(x, np.asarray(y))
        return out.reshape(np.shape(y)) if np.ndim(y) == 2 else out
    if (not sp.issparse(x)) and np.ndim(x) == 1:
        # row vector times matrix
        out = _hip.spmv(sp.csr_matrix(y).T.tocsr(), np.asarray(x))
        return out
    # SciPy's `sparse * dense-2-D` (and dense * sparse) returns a dense array, only sparse *
    # sparse stays sparse; np.dot of two dense operands is dense too
    prod = _hip.spgemm(x, y)
    return prod if (sp.issparse(x) and sp.issparse(y)) else prod.toarray()


def getresidual(b, A, x, N):
    """b - A x as an (N, 1) column (openmg/tools.py:12-15), computed on the device."""
    r = _hip.residual(A, np.asarray(b).reshape(N), np.asarray(x).reshape(N))
    return r.reshape((N, 1))


def dictUpdateNoClobber(updateDict, targetDict):
    """Copy the entries of updateDict that targetDict lacks (openmg/tools.py:29-40).

    >>> adict = {'a': 'A'}
    >>> out = dictUpdateNoClobber({'b': 'B', 'a': 'Z'}, adict)
    >>> adict == {'a': 'A', 'b': 'B'} and out is adict
    True
    """
    for key in updateDict:
        dictAddNoClobber(targetDict, key, updateDict[key])
    return targetDict


def dictAddNoClobber(dictionary, key, value):
    """Insert key only when absent (openmg/tools.py:43-53).

    >>> dictAddNoClobber({"hello": 42}, "hello", 0)
    {'hello': 42}
    """
    dictionary.setdefault(key, value)
    return dictionary


def product(iterableThing):
    """Integer product of a shape tuple (openmg/tools.py:56-60)."""
    total = 1
    for extent in iterableThing:
        total *= extent
    return total
