"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (NumPy/SciPy + a small C helper, oracle/gs_oracle.c) of the
multigrid hot path of tsbertalan/openmg.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; ``openmg_amd`` never does.  Every function names the reference lines it
restates (paths relative to the reference root, /root/reference in the build
container).

Pinning: this restatement is checked against golden vectors recorded from the real
reference (tests/golden/*.npz, made by tests/golden/make_golden.py) in
tests/test_oracle_golden.py.  Pieces that have NO reference counterpart are marked
"parity unpinned by the reference": weighted Jacobi, the 2-D/3-D sparse stencil
generators, and anything multi-GPU.  Colour-ordered Gauss-Seidel IS pinned: it is the
reference's lexicographic sweep applied to a symmetrically permuted system (fixture g4).

SciPy (SpMV / SpGEMM / SuperLU) is the reference's own third-party arithmetic
(README.md:4 "Requires Numpy and Scipy", unpinned version); the fixtures were made with
SciPy 1.15.3 / NumPy 2.2.6.
"""
import ctypes
import os

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _clib():
    """Load oracle/libmg_oracle.so (built by `make -C oracle` or __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmg_oracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(path)
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        lib.oracle_gs_lex.argtypes = [ctypes.c_int64, i32p, i32p, f64p, f64p, f64p, ctypes.c_int]
        lib.oracle_gs_lex.restype = ctypes.c_int
        lib.oracle_gs_ordered.argtypes = [ctypes.c_int64, i32p, i32p, f64p, f64p, f64p, i32p, ctypes.c_int]
        lib.oracle_gs_ordered.restype = ctypes.c_int
        lib.oracle_residual.argtypes = [ctypes.c_int64, i32p, i32p, f64p, f64p, f64p, f64p]
        lib.oracle_residual.restype = None
        lib.oracle_spmv.argtypes = [ctypes.c_int64, i32p, i32p, f64p, f64p, f64p]
        lib.oracle_spmv.restype = None
        lib.oracle_jacobi.argtypes = [ctypes.c_int64, i32p, i32p, f64p, f64p, f64p, f64p, ctypes.c_double]
        lib.oracle_jacobi.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _csr(A):
    """CSR view with int32 index arrays and float64 data, stored column order untouched."""
    A = A if sp.isspmatrix_csr(A) else sp.csr_matrix(A)
    return (np.ascontiguousarray(A.indptr, dtype=np.int32),
            np.ascontiguousarray(A.indices, dtype=np.int32),
            np.ascontiguousarray(A.data, dtype=np.float64))


# ----------------------------------------------------------------------------
# defaults  (openmg/__init__.py:16-27).  The reference keeps ONE module-global dict
# and writes into it (Q1); the oracle mirrors that with its own global.
# ----------------------------------------------------------------------------
defaults = {
    "problemShape": (200,), "gridLevels": 2, "verbose": False, "threshold": 0.1,
    "cycles": 0, "preIterations": 1, "postIterations": 0, "dense": False,
    "giveInfo": False, "minSize": 8,
}


# ----------------------------------------------------------------------------
# tools.py
# ----------------------------------------------------------------------------
def flexible_mmult(x, y):
    """openmg/tools.py:18-26 — np.dot for two dense operands, overloaded `*` otherwise."""
    if not sp.issparse(x) and not sp.issparse(y):
        return np.dot(x, y)
    return x @ y if sp.issparse(x) and sp.issparse(y) else x * y


def get_residual(b, A, x, N):
    """openmg/tools.py:12-15 — b - A x as an (N, 1) column."""
    return np.asarray(b).reshape((N, 1)) - flexible_mmult(A, np.asarray(x).reshape((N, 1)))


def dict_update_no_clobber(update, target):
    """openmg/tools.py:29-53 — copy keys that the target does not have yet."""
    for k, v in update.items():
        if k not in target:
            target[k] = v
    return target


def product(seq):
    """openmg/tools.py:56-60."""
    out = 1
    for s in seq:
        out *= s
    return out


# ----------------------------------------------------------------------------
# operators.py
# ----------------------------------------------------------------------------
def restriction(shape, dense=False):
    """openmg/operators.py:15-89 — 2^alpha-cell aggregation, weight 1/2^alpha.

    Quirks kept on purpose: the second axis offset is shape[0] and the
    third is shape[0]*shape[1] (operators.py:46,78-81) while the coarse cell's first
    column comes from a C-order reshape (operators.py:64-68); rows pair up with coarse
    columns through zip(), which truncates (operators.py:74); LIL assignment *sets*
    an entry, so a duplicate (r, c) is stored once; an out-of-range column raises
    IndexError like LIL does.
    """
    shape = tuple(int(s) for s in shape)
    alpha = len(shape)
    N = product(shape)
    n = N // (2 ** alpha)
    if n in (0, 1):                                        # operators.py:53-56
        raise ValueError("New restriction matrix would have shape %s. Coarse set would have %d point(s)! "
                         "Try a larger problem or fewer gridLevels." % (str((n, N)), n))
    if alpha > 3:                                          # operators.py:69-71
        raise ValueError("restriction(): Greater than 3 dimensions is not implemented. (shape was %s .)"
                         % str(shape))
    grid = np.arange(N).reshape(shape)
    first = grid[(slice(None, None, 2),) * alpha].ravel()   # operators.py:64-68
    NX = shape[0]
    offs = [0, 1]
    if alpha >= 2:
        offs += [NX, NX + 1]
    if alpha == 3:
        NY = shape[1]
        offs += [NX * NY, NX * NY + 1, NX * NY + NX, NX * NY + NX + 1]
    m = min(n, first.size)                                 # zip truncation, operators.py:74
    rows = np.repeat(np.arange(m), len(offs))
    cols = (first[:m, None] + np.array(offs)[None, :]).ravel()
    if cols.size and cols.max() >= N:
        raise IndexError("column index (%d) out of range" % int(cols.max()))
    key = np.unique(rows.astype(np.int64) * N + cols)      # set-semantics for duplicates
    rows, cols = key // N, key % N
    R = sp.csr_matrix((np.full(rows.size, 1.0 / (2 ** alpha)), (rows, cols)), shape=(n, N))
    R.sort_indices()
    return R.toarray() if dense else R


def restriction_list(problemShape, coarsestLevel, minSize, dense=False, verbose=False):
    """openmg/operators.py:92-141 — one R per level transition.  The first R is always
    built; further ones until `coarsestLevel` transitions exist or the next coarse size
    would be <= minSize (operators.py:133-140)."""
    levels = coarsestLevel + 1
    shape = np.array(problemShape)
    R = [restriction(tuple(shape // 1), dense=dense)]
    level = 0
    while level < levels - 1:
        level += 1
        nxt = restriction(tuple(shape // (2 ** level)), dense=dense)
        if nxt.shape[0] <= minSize:
            break
        R.append(nxt)
    return R


def coefficient_list(A_in, R, dense=False, verbose=False):
    """openmg/operators.py:144-188 — Galerkin A[l] = (R[l-1] A[l-1]) R[l-1]^T via two
    SciPy products, exactly as the reference spells it (result columns are unsorted)."""
    A = [None] * (len(R) + 1)
    if dense:
        A[0] = A_in.todense() if sp.issparse(A_in) else A_in
    else:
        A[0] = sp.csr_matrix(A_in)
    for l in range(1, len(A)):
        A[l] = flexible_mmult(flexible_mmult(R[l - 1], A[l - 1]), R[l - 1].T)
    return A


def poisson1d_sparse(N):
    """openmg/operators.py:191-203 — diagonal 4, off-diagonals -1 (Q3)."""
    return sp.csr_matrix(sp.diags([-np.ones(N - 1), 4.0 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1]))


def poisson1d(shape, sparse=False):
    """openmg/operators.py:206-218 — dense variant is (2, -1)."""
    N = shape[0]
    if sparse:
        return poisson1d_sparse(N)
    return np.diag(-np.ones(N - 1), -1) + np.diag(2.0 * np.ones(N)) + np.diag(-np.ones(N - 1), 1)


def poisson2d(shape, sparse=False):
    """openmg/operators.py:221-243 — -4 on the diagonal, +1 on the first off-diagonals
    (`oneup`, not zeroed at grid-row ends) and +1 at offset 1+NX (`twoup` is an identity
    of size N-1-NX padded with 1+NX zero columns on the left), both mirrored (Q3).
    Checked against fixture gen_p2dense_*."""
    if sparse:
        raise NotImplementedError("Sparse poisson for alpha>1 is not yet implemented.")
    NX, NY = shape
    N = NX * NY
    A = -4.0 * np.eye(N)
    A += np.eye(N, k=1) + np.eye(N, k=-1)
    A += np.eye(N, k=1 + NX) + np.eye(N, k=-(1 + NX))
    return A


def poisson3d(shape, sparse=False):
    """openmg/operators.py:245-257 — -6 / +1 at i+1, i+NX, i+NX*NY, guarded only by < N."""
    if sparse:
        raise NotImplementedError("Sparse poisson for alpha>1 is not yet implemented.")
    NX, NY, NZ = shape
    N = NX * NY * NZ
    A = np.zeros((N, N))
    i = np.arange(N)
    A[i, i] = -6.0
    for off in (1, NX, NX * NY):
        j = i + off
        ok = j < N
        A[i[ok], j[ok]] = 1.0
    A += A.T
    return A


def poisson(shape, sparse=False):
    """openmg/operators.py:260-279."""
    if isinstance(shape, int):
        shape = (shape,)
    if len(shape) == 1:
        out = poisson1d(shape, sparse)
    elif len(shape) == 2:
        out = poisson2d(shape, sparse)
    elif len(shape) == 3:
        out = poisson3d(shape, sparse)
    else:
        raise ValueError("Only 1, 2 or 3 dimensions are allowed.")
    if sparse:
        out = sp.csr_matrix(out)
    return out


# ----------------------------------------------------------------------------
# solvers.py
# ----------------------------------------------------------------------------
def coarse_solve(A, b):
    """openmg/solvers.py:16-26 — spsolve (SuperLU) for sparse A, np.linalg.solve otherwise."""
    if sp.issparse(A):
        out = spla.spsolve(sp.csc_matrix(A), np.asarray(b))
    else:
        out = np.linalg.solve(A, b)
    return np.ravel(out)


def gauss_seidel(A, b, x, iterations=None, threshold=None, verbose=False):
    """openmg/solvers.py:34-75 — in-place lexicographic Gauss-Seidel; stops when
    `iterations` sweeps are done and/or ||b - A x||_2 < threshold (absolute), the check
    running BEFORE the first sweep as well (solvers.py:52-54).  x is mutated and
    returned (Q2).  Dense A goes through the same row loop (solvers.py:69-71)."""
    if iterations is None and threshold is None:
        iterations = 1
    N = x.size
    ip, ix, dv = _csr(A)
    bb = np.ascontiguousarray(np.asarray(b, dtype=np.float64).ravel())
    xv = x.reshape(-1)
    direct = xv.flags["C_CONTIGUOUS"] and xv.dtype == np.float64 and np.shares_memory(xv, x)
    work = xv if direct else np.ascontiguousarray(xv, dtype=np.float64)

    def stop(it):
        by_iter = iterations is not None and it >= iterations
        by_norm = False
        if threshold is not None:
            r = np.empty(N)
            _clib().oracle_residual(N, ip, ix, dv, bb, work, r)
            by_norm = np.linalg.norm(r) < threshold
        return by_iter or by_norm

    it = 0
    while not stop(it):
        rc = _clib().oracle_gs_lex(N, ip, ix, dv, bb, work, 1)
        if rc:
            raise ZeroDivisionError("row %d has no diagonal entry" % (rc - 1))
        it += 1
    if not direct:
        xv[...] = work
    return x


def smooth(A, b, x, iterations, verbose=False):
    """openmg/solvers.py:28-29."""
    return gauss_seidel(A, b, x, iterations=iterations, verbose=verbose)


def smooth_to_threshold(A, b, x, threshold, verbose=False):
    """openmg/solvers.py:31-32."""
    return gauss_seidel(A, b, x, threshold=threshold, verbose=verbose)


# ----------------------------------------------------------------------------
# __init__.py
# ----------------------------------------------------------------------------
def mg_cycle(A, b, level, R, parameters, initial=None, smoother=None):
    """openmg/__init__.py:151-236 — one recursive V-cycle.

    `smoother(A_l, b, x, iterations, level)` may replace the reference's `smooth`
    (used by the colour-ordered / Jacobi variants below); None = the reference's own.
    The dead product R[l]*b (:205-206, used only for its length) is skipped; the residual
    norm is evaluated at every level like the reference does (:227)."""
    b = np.asarray(b, dtype=np.float64)
    N = b.size
    if initial is None:
        initial = np.zeros((N,))                                            # :191-192
    if level < parameters["coarsestLevel"]:                                 # :199
        if smoother is None:
            u = smooth(A[level], b.ravel(), initial, parameters["preIterations"])   # :201
        else:
            u = smoother(A[level], b.ravel(), initial, parameters["preIterations"], level)
        NH = R[level].shape[0]
        residual = get_residual(b, A[level], u, N)                          # :209
        coarse_residual = np.asarray(flexible_mmult(R[level], residual.reshape((N, 1)))).reshape((NH,))  # :210
        coarse_corr = mg_cycle(A, coarse_residual, level + 1, R, parameters, smoother=smoother)[0]     # :213
        corr = np.asarray(flexible_mmult(R[level].transpose(), coarse_corr.reshape((NH, 1)))).reshape((N,))  # :214
        u_out = np.asarray(u).reshape((N,)) + corr                          # :220/:224
        if parameters["postIterations"] > 0:                                # :216-222
            if smoother is None:
                u_out = smooth(A[level], b.ravel(), u_out, parameters["postIterations"])
            else:
                u_out = smoother(A[level], b.ravel(), u_out, parameters["postIterations"], level)
        norm = float(np.linalg.norm(get_residual(b, A[level], u_out, N)))   # :227
    else:
        norm = 0                                                            # :232
        u_out = coarse_solve(A[level], b.reshape((N, 1)))                   # :234
    return u_out, {"norm": norm}


def mg_solve(A_in, b, parameters, smoother=None):
    """openmg/__init__.py:28-148 — setup, at least one cycle, then cycles until
    `cycles` is reached or the ABSOLUTE residual norm drops below `threshold`.
    Mutates `parameters` and this module's `defaults` like the reference (Q1); the
    both-stop-rules-off ValueError comes AFTER the first cycle (:118-119)."""
    shape = parameters["problemShape"]
    grid_levels = parameters["gridLevels"]
    defaults["coarsestLevel"] = grid_levels - 1                             # :95
    dict_update_no_clobber(defaults, parameters)                            # :96
    dense = parameters["dense"]
    R = restriction_list(shape, parameters["coarsestLevel"], parameters["minSize"], dense=dense)  # :103
    parameters["coarsestLevel"] = len(R)                                    # :106
    A = coefficient_list(A_in, R, dense=dense)                              # :109
    result, info = mg_cycle(A, b, 0, R, parameters, smoother=smoother)      # :112
    norm = info["norm"]
    cycle = 1
    if parameters["threshold"] <= 0 and parameters["cycles"] <= 0:          # :118-119
        raise ValueError("Either parameters['threshold'] or parameters['cycles'] must be > 0.")

    def stop(cycle, norm):                                                  # :120-129
        by_cycles = parameters.get("cycles", 0) > 0 and cycle >= parameters["cycles"]
        by_thresh = ("threshold" in parameters and parameters["threshold"] > 0
                     and norm < parameters["threshold"])
        return by_cycles or by_thresh

    while not stop(cycle, norm):                                            # :132-138
        cycle += 1
        result, info = mg_cycle(A, b, 0, R, parameters, initial=result, smoother=smoother)
        norm = info["norm"]
    info["cycle"], info["norm"], info["R"], info["A"] = cycle, norm, R, A   # :140-143
    if parameters["giveInfo"]:
        return result, info
    return result


# ----------------------------------------------------------------------------
# Extensions required by BASELINE.json's configs.  Anything marked UNPINNED has no
# counterpart in the reference.
# ----------------------------------------------------------------------------
def stencil_poisson(shape):
    """UNPINNED input generator: Dirichlet (2*dim, -1) 3/5/7-point Laplacian,
    C-order numbering, sorted CSR.  The reference has no sparse 2-D/3-D generator
    (operators.py:224,247 raise NotImplementedError)."""
    total = None
    for d, s in enumerate(shape):
        T = sp.diags([-np.ones(s - 1), 2.0 * np.ones(s), -np.ones(s - 1)], [-1, 0, 1], format="csr")
        term = None
        for e, se in enumerate(shape):
            f = T if e == d else sp.identity(se, format="csr")
            term = f if term is None else sp.kron(term, f, format="csr")
        total = term if total is None else total + term
    out = sp.csr_matrix(total)
    out.sort_indices()
    return out


def greedy_colouring(A):
    """Smallest-free-colour greedy colouring of the graph of A + A^T in natural row order.
    On a 5/7-point grid this is the red-black parity colouring, on a 27-point grid the
    2x2x2 eight-colouring.  Pure NumPy-free loop: use only at test sizes."""
    S = sp.csr_matrix(A)
    S = sp.csr_matrix(S + S.T)
    n = S.shape[0]
    colour = np.full(n, -1, dtype=np.int32)
    for i in range(n):
        nb = S.indices[S.indptr[i]:S.indptr[i + 1]]
        used = set(int(colour[j]) for j in nb if j != i and colour[j] >= 0)
        c = 0
        while c in used:
            c += 1
        colour[i] = c
    return colour


def parity_colouring(shape):
    """Red-black colouring of a grid in C-order numbering: (sum of coordinates) mod 2.
    Equal to greedy_colouring() on 3/5/7-point stencils (checked in tests) but vectorised,
    for sizes where the Python loop above is too slow."""
    return (np.indices(shape).reshape(len(shape), -1).sum(axis=0) % 2).astype(np.int32)


def colour_order(colour):
    """Rows of colour 0 first, then colour 1, ... (stable inside a colour)."""
    return np.argsort(colour, kind="stable").astype(np.int32)


def gs_ordered(A, b, x, order, iterations=1):
    """Gauss-Seidel visiting rows in `order`; equals the reference's gaussSeidel
    (solvers.py:56-68) on the system permuted by `order` — pinned by fixture g4."""
    ip, ix, dv = _csr(A)
    bb = np.ascontiguousarray(np.asarray(b, dtype=np.float64).ravel())
    work = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel())
    rc = _clib().oracle_gs_ordered(work.size, ip, ix, dv, bb, work,
                                   np.ascontiguousarray(order, dtype=np.int32), int(iterations))
    if rc:
        raise ZeroDivisionError("row %d has no diagonal entry" % (rc - 1))
    x.reshape(-1)[...] = work
    return x


def jacobi(A, b, x, iterations=1, omega=2.0 / 3.0):
    """UNPINNED (the reference has no Jacobi): x <- x + omega D^-1 (b - A x)."""
    ip, ix, dv = _csr(A)
    bb = np.ascontiguousarray(np.asarray(b, dtype=np.float64).ravel())
    cur = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel()).copy()
    nxt = np.empty_like(cur)
    for _ in range(int(iterations)):
        rc = _clib().oracle_jacobi(cur.size, ip, ix, dv, bb, cur, nxt, float(omega))
        if rc:
            raise ZeroDivisionError("row %d has no diagonal entry" % (rc - 1))
        cur, nxt = nxt, cur
    x.reshape(-1)[...] = cur
    return x


def make_smoother(kind, A_list, omega=2.0 / 3.0):
    """Smoother factory for mg_cycle/mg_solve's `smoother=` hook.
    'gs' = the reference's lexicographic sweep; 'colour' = greedy multi-colour GS
    (red-black on 5/7-point); 'jacobi' = weighted Jacobi (UNPINNED)."""
    if kind == "gs":
        return None
    if kind == "colour":
        orders = [colour_order(greedy_colouring(M)) for M in A_list]
        return lambda A, b, x, its, level: gs_ordered(A, b, x, orders[level], its)
    if kind == "jacobi":
        return lambda A, b, x, its, level: jacobi(A, b, x, its, omega)
    raise ValueError("unknown smoother %r" % (kind,))
