"""
openmg_amd — geometric multigrid V-cycle on AMD MI355X (gfx950), presenting the public
names of tsbertalan/openmg: mgSolve, mgCycle (alias mg_cycle), defaults, smooth,
smoothToThreshold, coarseSolve, tools, operators, solvers.

All numerical work runs in hand-written HIP kernels behind libopenmg_hip.so
(include/openmg_hip.h); this package is the NumPy/SciPy-facing host side.  There is no
CPU fallback: without the library or without a GPU the calls raise.

Extra keys understood in the `parameters` dict (ignored by the reference):
    'smoother'  'gs' (default: the reference's lexicographic Gauss-Seidel iterate),
                'colour' (multi-colour GS; red-black on 5/7-point stencils) or 'jacobi'
    'omega'     relaxation weight for 'jacobi' (default 2/3)
    'dtype'     'float64' (default, the reference's precision) or 'float32': the precision the
                levels are stored and computed in on the device (inputs / outputs stay float64)
    'trustOperators'  mgCycle only, default False: when the caller passes the SAME list members (object identity) it
                passed on an earlier call — e.g. infoDict['A'] / infoDict['R'] handed back unchanged — the per-call
                checksum of every byte of the lists (what recognises an operator edited in place) is skipped

`b` and `initial` may be DEVICE arrays (objects with `__cuda_array_interface__`, e.g. PyTorch-ROCm tensors: contiguous
float64): mgCycle / mgSolve then return a device array of the same kind and only the norm crosses PCIe.
"""
import weakref

import numpy as np
import scipy.sparse as sp

from . import _devarray, _hip
from . import operators, solvers, tools
from .solvers import coarseSolve, smooth, smoothToThreshold

__all__ = ["mgSolve", "mgCycle", "mg_cycle", "defaults", "smooth", "smoothToThreshold",
           "coarseSolve", "tools", "operators", "solvers", "clear_cache"]

# Same keys and values as the reference (openmg/__init__.py:16-27).  Like there, this dict
# is module-global and mgSolve writes 'coarsestLevel' into it on every call.
defaults = {
    "problemShape": (200,),
    "gridLevels": 2,
    "verbose": False,
    "threshold": 0.1,
    "cycles": 0,
    "preIterations": 1,
    "postIterations": 0,
    "dense": False,
    "giveInfo": False,
    "minSize": 8,
}

_DEFAULT_OMEGA = 2.0 / 3.0


def _smoother_of(parameters):
    kind = parameters.get("smoother", "gs")
    code = _hip.smoother_code(kind)
    omega = float(parameters.get("omega", _DEFAULT_OMEGA if code == _hip.SMOOTH_JACOBI else 1.0))
    return code, omega


def _dtype_of(parameters):
    return _hip.dtype_code(parameters.get("dtype", "float64"))


# ---- device hierarchy cache for repeated mgCycle calls ---------------------------------------
# mgCycle receives the A and R lists on every call (openmg/__init__.py:151); uploading
# them each time would dominate.  A cache entry holds the device hierarchy TOGETHER WITH strong
# references to the list members it was built from, so that neither their ids nor their buffer
# addresses can be recycled for other matrices while the entry lives, and the key carries a
# checksum of the stored arrays (every byte), so that a matrix edited in place (or rebuilt into
# the same buffers) misses.  Members that are not CSR are converted once per object (cached by id
# while the object lives).
_cache = {}
_CACHE_SLOTS = 2

def _array_checksum(a):
    """Checksum of the WHOLE array — an in-place edit of a few entries of a large operator must not hit
    the stale device copy.  omg_host_checksum hashes 4 MiB chunks on all host threads (one Python thread
    with xxhash: ~5 GB/s under the interpreter lock, 0.3 s per mgCycle call for a 256^3 operator)."""
    return _hip.host_checksum(np.asarray(a))


_csr_of = {}          # id(non-CSR sparse member) -> (weak reference, checksum of its own arrays, its CSR form)


def _as_csr_cached(M):
    """CSR form of a sparse member, converted once while the member lives and keeps its content."""
    if sp.isspmatrix_csr(M):
        return M
    arrays = [getattr(M, name) for name in ("data", "indices", "indptr", "row", "col", "offsets")
              if isinstance(getattr(M, name, None), np.ndarray)]
    if not arrays or any(a.dtype.kind == "O" for a in arrays):
        # LIL (object arrays of Python lists), DOK: no flat buffers to checksum — converted on every call
        return sp.csr_matrix(M)
    src = tuple(_array_checksum(a) for a in arrays)
    entry = _csr_of.get(id(M))
    if entry is not None and entry[0]() is M and entry[1] == src:
        return entry[2]
    C = sp.csr_matrix(M)
    try:
        ref = weakref.ref(M, lambda _r, k=id(M): _csr_of.pop(k, None))
    except TypeError:                           # not weakly referenceable: convert every time
        return C
    _csr_of[id(M)] = (ref, src, C)
    return C


def _fingerprint(A, R, n_levels, code, omega, dtype):
    def one(M):
        if sp.issparse(M):
            M = _as_csr_cached(M)
            return (M.shape, M.nnz, _array_checksum(M.indptr), _array_checksum(M.indices), _array_checksum(M.data))
        M = np.asarray(M)
        return (M.shape, _array_checksum(M))
    return (tuple(one(M) for M in A[:n_levels]), tuple(one(M) for M in R[:n_levels - 1]), code, omega, dtype)


def _hierarchy_for(A, R, n_levels, code, omega, dtype=_hip.DTYPE_F64, trust=False):
    if trust:
        # parameters['trustOperators']: the caller vouches that list members it has passed before are unchanged; an entry
        # built from the very same objects (identity, not equality) and the same smoother / precision is taken as it is
        for h, members, how in _cache.values():
            if (how == (code, omega, dtype) and len(members[0]) == n_levels
                    and all(a is m for a, m in zip(A[:n_levels], members[0]))
                    and all(r is m for r, m in zip(R[:n_levels - 1], members[1]))):
                return h
    key = _fingerprint(A, R, n_levels, code, omega, dtype)
    entry = _cache.get(key)
    if entry is None:
        while len(_cache) >= _CACHE_SLOTS:
            _cache.pop(next(iter(_cache)))[0].close()
        members = (list(A[:n_levels]), list(R[:n_levels - 1]))
        h = _hip.Hierarchy(members[0], members[1], smoother=code, omega=omega, dtype=dtype)
        entry = _cache[key] = (h, members, (code, omega, dtype))
    return entry[0]


def clear_cache():
    """Free the device hierarchies kept for repeated mgCycle calls."""
    while _cache:
        _cache.popitem()[1][0].close()


# ---- the two public drivers ---------------------------------------------------------------
def mgSolve(A_in, b, parameters):
    """Solve A_in u = b by V-cycles on the GPU; same contract as openmg.mgSolve
    (openmg/__init__.py:28-148).

    parameters must hold 'problemShape' and 'gridLevels'; the optional keys and their
    defaults are those of `openmg_amd.defaults`.  gridLevels = g builds g restriction
    operators, i.e. g+1 grids, unless 'minSize' stops the coarsening earlier.  At least one
    cycle is run; cycling stops when 'cycles' (> 0) is reached or the absolute residual
    2-norm falls below 'threshold' (> 0); if both are <= 0 a ValueError is raised (after the
    first cycle, as in the reference).  The caller's dict is completed with the defaults
    and its 'coarsestLevel' is overwritten with the depth actually built.

    Returns u, or (u, infoDict) when parameters['giveInfo'] is true; infoDict holds
    'cycle', 'norm', and the hierarchies 'R' and 'A' as SciPy CSR lists.
    """
    problemShape = parameters["problemShape"]
    gridLevels = parameters["gridLevels"]
    defaults["coarsestLevel"] = gridLevels - 1
    tools.dictUpdateNoClobber(defaults, parameters)
    verbose = parameters["verbose"]
    dense = parameters["dense"]
    code, omega = _smoother_of(parameters)

    pre, post = parameters["preIterations"], parameters["postIterations"]
    # (the device route qualifies levels for the FUSED paths, which exist for the colour orderings and 2-D weighted Jacobi; with
    # the reference's own lexicographic smoother it would build the Galerkin chain in HBM only to fetch it again for the host's
    # orderings and codings: 1.14 s of setup at 256^3 where the lists route below takes 0.7)
    fused_smoother = code == _hip.SMOOTH_GS_COLOUR or (code == _hip.SMOOTH_JACOBI and len(tuple(problemShape)) == 2)
    n_fused = _device_setup_depth(A_in, problemShape, parameters) if fused_smoother else 0
    if n_fused:
        # nobody asked for the operator lists (giveInfo off): restrictions, Galerkin products and the levels' qualification
        # for the fused paths stay in HBM (omg_hierarchy_create_from_fine); same hierarchy, same results
        R = A = None
        parameters["coarsestLevel"] = n_fused
        hierarchy = _hip.Hierarchy.from_fine(A_in, problemShape, n_fused, smoother=code, omega=omega, dtype=_dtype_of(parameters))
    else:
        R = operators.restrictionList(problemShape, parameters["coarsestLevel"], parameters["minSize"],
                                      dense=dense, verbose=verbose)
        parameters["coarsestLevel"] = len(R)
        A = operators.coeffecientList(A_in, R, dense=dense, verbose=verbose)
        hierarchy = _hip.Hierarchy(A, R, smoother=code, omega=omega, dtype=_dtype_of(parameters))
    depth = parameters["coarsestLevel"]
    b_on_device = _devarray.is_device_array(b)
    try:
        if b_on_device:
            _devarray.synchronize()
            hierarchy.resident_load_dev(_devarray.address(b, hierarchy.sizes[0], "b"))
        else:
            hierarchy.resident_load(np.asarray(b, dtype=np.float64).reshape(-1))
        if verbose:
            _announce_descent(depth)
        norm = hierarchy.resident_cycle(pre, post)
        cycle = 1
        if verbose:
            print("Residual norm from cycle %d is %f." % (cycle, norm))
        if parameters["threshold"] <= 0 and parameters["cycles"] <= 0:
            raise ValueError("Either parameters['threshold'] or parameters['cycles'] must be > 0.")

        def finished():
            by_count = parameters.get("cycles", 0) > 0 and cycle >= parameters["cycles"]
            by_norm = "threshold" in parameters and parameters["threshold"] > 0 and norm < parameters["threshold"]
            return by_count or by_norm

        if (not verbose and parameters.get("cycles", 0) > cycle and not parameters.get("threshold", 0) > 0):
            # stop rule = cycle count only: the remaining cycles go to the device in one call (every
            # cycle's norm is still computed; only the last one is observable here)
            norm = hierarchy.resident_cycles(pre, post, parameters["cycles"] - cycle)[-1]
            cycle = parameters["cycles"]
        while not finished():
            if verbose:
                print("cycle %i < cycles %i" % (cycle, parameters["cycles"]))
                _announce_descent(depth)
            cycle += 1
            norm = hierarchy.resident_cycle(pre, post)
            if verbose:
                print("Residual norm from cycle %d is %f." % (cycle, norm))
        if b_on_device:
            result = _devarray.empty_like(b, hierarchy.sizes[0])
            hierarchy.resident_fetch_dev(_devarray.address(result, hierarchy.sizes[0], "result"))
        else:
            result = hierarchy.resident_fetch()
    finally:
        hierarchy.close()

    infoDict = {"norm": norm, "cycle": cycle, "R": R, "A": A}
    if verbose:
        print("Returning mgSolve after %i cycle(s) with norm %f" % (cycle, norm))
    if parameters["giveInfo"]:
        return result, infoDict
    return result


def _device_setup_depth(A_in, problemShape, parameters):
    """How many restrictions operators.restrictionList would build (openmg/operators.py:128-141) when mgSolve's whole
    setup can stay on the device — 0 when it cannot: the caller wants the lists (giveInfo), the dense path, progress
    messages, or the shape is not one whose restriction is the plain aggregation at every level (2-D / 3-D, first and
    last extent equal, extents even on every restricted level) — then the ordinary route runs (and raises what the
    reference raises)."""
    if parameters.get("giveInfo") or parameters.get("dense") or parameters.get("verbose") or not sp.issparse(A_in):
        return 0
    try:
        shape = tuple(int(s) for s in problemShape)
    except TypeError:
        return 0
    dim = len(shape)
    if dim not in (2, 3) or shape[0] != shape[-1] or min(shape) < 2:
        return 0
    if int(np.prod(shape)) != A_in.shape[0] or A_in.shape[0] != A_in.shape[1]:
        return 0

    def rows_of(level):                 # rows of restriction(shape // 2**level), or None where the reference's quirks start
        ext = [s // 2 ** level for s in shape]
        if any(e < 2 or e % 2 for e in ext):
            return None
        return int(np.prod(ext)) // 2 ** dim

    first = rows_of(0)
    if first is None or first in (0, 1):
        return 0
    n = 1
    for level in range(1, parameters["coarsestLevel"] + 1):
        rows = rows_of(level)
        if rows is None or rows in (0, 1):
            return 0                    # (the reference raises or truncates here: let the ordinary route do that)
        if rows <= parameters["minSize"]:
            break
        n += 1
    return n


def _announce_descent(depth):
    for level in range(depth):
        print(level * " " + "calling mgCycle at level %i" % level)
    print(depth * " " + "direct solving at level %i" % depth)


def mgCycle(A, b, level, R, parameters, initial=None):
    """One V-cycle entered at `level`; same contract as openmg.mgCycle
    (openmg/__init__.py:151-236): pre-smooth, restrict the residual, recurse, prolong and
    correct, post-smooth; direct solve at parameters['coarsestLevel'].

    A, R are the lists made by operators.coeffecientList / restrictionList (any CSR lists
    of matching shapes work).  Returns (uOut, {'norm': ||b - A[level] uOut||_2}); the norm
    is 0 when `level` is the coarsest.  The device copy of the hierarchy is cached between
    calls (see clear_cache)."""
    coarsest = parameters["coarsestLevel"]
    if coarsest >= len(A) or coarsest > len(R):
        raise IndexError("parameters['coarsestLevel'] = %d but only %d operators / %d restrictions given"
                         % (coarsest, len(A), len(R)))
    code, omega = _smoother_of(parameters)
    on_device = _devarray.is_device_array(b)
    if _devarray.is_device_array(initial) and not on_device:
        raise TypeError("mgCycle: `initial` is a device array but `b` is not; pass both on the device or both on the host")
    if not on_device:
        b = np.asarray(b, dtype=np.float64).reshape(-1)
    if level >= coarsest:
        # the reference's `else` branch (:229-234): direct solve with A[level], norm 0
        if on_device:
            raise ValueError("mgCycle at the coarsest level with device arrays: use solvers.coarseSolve on host arrays")
        return solvers.coarseSolve(A[level], b), {"norm": 0}
    hierarchy = _hierarchy_for(A, R, coarsest + 1, code, omega, _dtype_of(parameters), trust=bool(parameters.get("trustOperators", False)))
    if parameters.get("verbose", False):
        for l in range(level, coarsest):
            print(l * " " + "calling mgCycle at level %i" % l)
        print(coarsest * " " + "direct solving at level %i" % coarsest)
    pre = parameters["preIterations"]
    if on_device:
        # b, initial, uOut in HBM: only the norm crosses PCIe (omg_vcycle_dev).  Q2 as on the host: with pre-smoothing the
        # caller's `initial` holds the pre-smoothed iterate afterwards; uOut is a new array (:220 / :224).
        n = hierarchy.sizes[level]
        _devarray.synchronize()
        x_in = None if initial is None else _devarray.address(initial, n, "initial")
        out = _devarray.empty_like(b, n)
        norm = hierarchy.vcycle_dev(_devarray.address(b, n, "b"), x_in, _devarray.address(out, n, "uOut"),
                                    x_in if (x_in is not None and pre > 0) else None, pre, parameters["postIterations"], level=level)
        return out, {"norm": norm}
    x_in = None if initial is None else np.ascontiguousarray(np.asarray(initial, dtype=np.float64).reshape(-1))
    x = np.empty(b.size)                    # uOut is a new array (openmg/__init__.py:220 / :224)
    in_place = (pre > 0 and isinstance(initial, np.ndarray) and initial.dtype == np.float64 and initial.flags.writeable
                and initial.size == b.size and np.shares_memory(initial, initial.reshape(-1)))
    if in_place and np.shares_memory(x_in, initial):
        # Q2: the reference's pre-smoother works IN PLACE on the caller's `initial`
        # (openmg/__init__.py:201 -> solvers.py:68,75): after the call it holds the pre-smoothed
        # iterate.  The device hands that iterate back from the same cycle (omg_vcycle_ex): x_in is read
        # before x_pre is written, so the two may be the caller's one buffer.
        norm = hierarchy.vcycle_ex(b, x_in, x, x_in, pre, parameters["postIterations"], level=level)
    elif in_place:
        smoothed = np.empty(b.size)
        norm = hierarchy.vcycle_ex(b, x_in, x, smoothed, pre, parameters["postIterations"], level=level)
        initial.reshape(-1)[:] = smoothed
    else:
        norm = hierarchy.vcycle_ex(b, x_in, x, None, pre, parameters["postIterations"], level=level)
    return x, {"norm": norm}


mg_cycle = mgCycle   # BASELINE.json's spelling
