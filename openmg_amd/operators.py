"""Hierarchy setup with the reference's names (openmg/operators.py).

restriction() and the Galerkin product run on the GPU (omg_restriction / omg_rap); the
Poisson generators are test-input builders and stay NumPy like the reference's.
"""
import numpy as np
import scipy.sparse as sp

from . import _hip, tools


def restriction(shape, dense=False):
    """R of shape (N / 2**alpha, N): every coarse cell averages its 2**alpha fine cells
    (openmg/operators.py:15-89).  Index conventions follow the reference exactly, including
    the second-axis offset shape[0] and third-axis offset shape[0]*shape[1] (Q6)."""
    shape = tuple(int(s) for s in shape)
    alpha = len(shape)
    N = tools.product(shape)
    n = N // (2 ** alpha)
    if n in (0, 1):
        raise ValueError("New restriction matrix would have shape %s. Coarse set would have %d point(s)! "
                         "Try a larger problem or fewer gridLevels." % (str((n, N)), n))
    if alpha > 3:
        raise ValueError("restriction(): Greater than 3 dimensions is not implemented. "
                         "(shape was %s .)" % str(shape))
    R = _hip.restriction(shape)
    if any(s % 2 for s in shape) or len(set(shape)) > 1:
        # unequal / odd extents: the reference's LIL assignment stores a repeated (r, c)
        # once and zip() leaves trailing rows empty; canonicalise the same way.
        R = R.tocoo()
        key = np.unique(R.row.astype(np.int64) * N + R.col)
        R = sp.csr_matrix((np.full(key.size, 1.0 / 2 ** alpha), (key // N, key % N)), shape=(n, N))
    R.sort_indices()
    return R.toarray() if dense else R


def restrictionList(problemShape, coarsestLevel, minSize, dense=False, verbose=False):
    """One restriction per level transition (openmg/operators.py:92-141): always the first,
    then more until `coarsestLevel` exist or the next coarse size would be <= minSize."""
    if verbose:
        print("Generating restriction matrices; dense=%s" % dense)
    extents = np.array(problemShape)
    out = [restriction(tuple(extents // 1), dense=dense)]
    for level in range(1, coarsestLevel + 1):
        candidate = restriction(tuple(extents // (2 ** level)), dense=dense)
        if candidate.shape[0] <= minSize:
            break
        out.append(candidate)
    return out


def coeffecientList(A_in, R, dense=False, verbose=False):
    """[A_0, R_0 A_0 R_0^T, ...] (openmg/operators.py:144-188), each Galerkin product done on
    the device.  (Spelling as in the reference.)"""
    if verbose:
        print("Generating coefficient matrices; dense=%s ..." % dense, end=" ")
    levels = [sp.csr_matrix(A_in)]
    for Rl in R:
        levels.append(_hip.rap(sp.csr_matrix(Rl), levels[-1]))
    if dense:
        levels = [M.toarray() for M in levels]
    if verbose:
        print("made %i A matrices" % len(levels))
    return levels


# ---- generators (test inputs; quirks of the reference kept) ----------------------
def poisson1Dsparse(N):
    """Sparse 1-D operator with 4 on the diagonal and -1 beside it (openmg/operators.py:191-203)."""
    return sp.csr_matrix(sp.diags([-np.ones(N - 1), 4.0 * np.ones(N), -np.ones(N - 1)], [-1, 0, 1]))


def poisson1D(shape, sparse=False):
    """Dense 1-D operator is (2, -1) — not the sparse one's (4, -1) (openmg/operators.py:206-218)."""
    N = shape[0]
    if sparse:
        return poisson1Dsparse(N)
    return 2.0 * np.eye(N) - np.eye(N, k=1) - np.eye(N, k=-1)


def poisson2D(shape, sparse=False):
    """Dense only: -4 / +1 at offsets 1 and NX+1, bands not cut at grid-row ends
    (openmg/operators.py:221-243)."""
    if sparse:
        raise NotImplementedError("Sparse poisson for alpha>1 is not yet implemented.")
    NX, NY = shape
    N = NX * NY
    return (-4.0 * np.eye(N) + np.eye(N, k=1) + np.eye(N, k=-1)
            + np.eye(N, k=NX + 1) + np.eye(N, k=-(NX + 1)))


def poisson3D(shape, sparse=False):
    """Dense only: +1 at i+1, i+NX, i+NX*NY wherever that index is < N, mirrored; the
    reference sets -6 on the diagonal BEFORE `A += A.T`, so the stored diagonal is -12
    (openmg/operators.py:245-257; fixture gen_p3dense_*)."""
    if sparse:
        raise NotImplementedError("Sparse poisson for alpha>1 is not yet implemented.")
    NX, NY, NZ = shape
    N = NX * NY * NZ
    upper = np.eye(N, k=1) + np.eye(N, k=NX) + np.eye(N, k=NX * NY)
    upper = np.minimum(upper, 1.0)          # NX == 1 would stack two bands
    return -12.0 * np.eye(N) + upper + upper.T


def poissonnd(shape, sparse=False):
    """Dispatch on len(shape) (openmg/operators.py:260-279)."""
    if isinstance(shape, int):
        shape = (shape,)
    if len(shape) == 1:
        out = poisson1D(shape, sparse)
    elif len(shape) == 2:
        out = poisson2D(shape, sparse)
    elif len(shape) == 3:
        out = poisson3D(shape, sparse)
    else:
        raise ValueError("Only 1, 2 or 3 dimensions are allowed.")
    return sp.csr_matrix(out) if sparse else out


poisson = poissonnd


def stencil_poisson(shape):
    """Dirichlet 3/5/7-point Laplacian (2*dim on the diagonal, -1 off it), C-order numbering,
    sorted CSR with int32 indices, assembled directly (no Kronecker products, so 256^3 takes
    seconds).  NOT in the reference (its sparse 2-D/3-D generators raise NotImplementedError);
    these are BASELINE.json's synthetic inputs."""
    shape = tuple(int(s) for s in shape)
    dim = len(shape)
    N = int(np.prod(shape))
    strides = [int(np.prod(shape[d + 1:])) for d in range(dim)]
    coords = np.unravel_index(np.arange(N, dtype=np.int64), shape)
    # neighbours in ascending column order: -stride_0, ..., -stride_{dim-1}, 0, +stride_{dim-1}, ..., +stride_0
    offsets, valid = [], []
    for d in range(dim):
        offsets.append(-strides[d]); valid.append(coords[d] > 0)
    offsets.append(0); valid.append(np.ones(N, dtype=bool))
    for d in reversed(range(dim)):
        offsets.append(strides[d]); valid.append(coords[d] < shape[d] - 1)
    counts = np.zeros(N, dtype=np.int64)
    for v in valid:
        counts += v
    indptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    nnz = int(indptr[-1])
    indices = np.empty(nnz, dtype=np.int32)
    data = np.empty(nnz, dtype=np.float64)
    cursor = indptr[:-1].copy()
    rows = np.arange(N, dtype=np.int64)
    for off, v in zip(offsets, valid):
        pos = cursor[v]
        indices[pos] = (rows[v] + off).astype(np.int32)
        data[pos] = 2.0 * dim if off == 0 else -1.0
        cursor += v
    out = sp.csr_matrix((data, indices, indptr.astype(np.int32)), shape=(N, N))
    out.has_sorted_indices = True
    return out


def _q1_element_stiffness():
    """8 x 8 stiffness matrix of the trilinear (Q1) element on the unit cube for -div(grad u),
    local nodes numbered (dz, dy, dx) in C order; 2-point Gauss quadrature is exact here."""
    g = np.array([0.5 - 0.5 / np.sqrt(3.0), 0.5 + 0.5 / np.sqrt(3.0)])
    corners = [(dz, dy, dx) for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)]
    K = np.zeros((8, 8))
    for z in g:
        for y in g:
            for x in g:
                grads = []
                for (cz, cy, cx) in corners:
                    fz, fy, fx = (z if cz else 1 - z), (y if cy else 1 - y), (x if cx else 1 - x)
                    sz, sy, sx = (1.0 if cz else -1.0), (1.0 if cy else -1.0), (1.0 if cx else -1.0)
                    grads.append((sz * fy * fx, fz * sy * fx, fz * fy * sx))
                G = np.array(grads)
                K += G @ G.T / 8.0
    return K, corners


def stencil7_variable(shape, seed=2024):
    """7-point variable-coefficient operator (NOT in the reference; the ordinary real input of mgSolve): cell-centred
    finite volumes of -div(kappa grad u) on a box of `shape` cells, homogeneous Dirichlet boundary, kappa =
    exp(U(-1, 1) ln 10) per cell (two decades, as BASELINE configs[4] draws it) from default_rng(seed); the coupling of two
    cells is the harmonic mean of their coefficients.  Symmetric (bit for bit) positive definite.  C-order numbering,
    sorted CSR with int32 indices."""
    shape = tuple(int(s) for s in shape)
    if len(shape) != 3:
        raise ValueError("stencil7_variable needs a 3-D shape")
    nz, ny, nx = shape
    rng = np.random.default_rng(seed)
    kap = np.exp(rng.uniform(-1.0, 1.0, size=(nz + 2, ny + 2, nx + 2)) * np.log(10.0))     # cells incl. one boundary layer

    def face(a, b):
        return 2.0 * a * b / (a + b)
    c = kap[1:-1, 1:-1, 1:-1]
    w = {(0, 0, -1): face(c, kap[1:-1, 1:-1, :-2]), (0, 0, 1): face(c, kap[1:-1, 1:-1, 2:]),
         (0, -1, 0): face(c, kap[1:-1, :-2, 1:-1]), (0, 1, 0): face(c, kap[1:-1, 2:, 1:-1]),
         (-1, 0, 0): face(c, kap[:-2, 1:-1, 1:-1]), (1, 0, 0): face(c, kap[2:, 1:-1, 1:-1])}
    diag = ((((w[(-1, 0, 0)] + w[(0, -1, 0)]) + w[(0, 0, -1)]) + w[(0, 0, 1)]) + w[(0, 1, 0)]) + w[(1, 0, 0)]
    n = nx * ny * nz
    idx = np.arange(n, dtype=np.int64).reshape(nz, ny, nx)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [diag.ravel()]
    for (dz, dy, dx), ww in w.items():
        sel = (slice(max(0, -dz), nz - max(0, dz)), slice(max(0, -dy), ny - max(0, dy)), slice(max(0, -dx), nx - max(0, dx)))
        rows.append(idx[sel].ravel())
        cols.append((idx[sel] + (dz * ny + dy) * nx + dx).ravel())
        vals.append(-ww[sel].ravel())
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    A.sort_indices()
    A.indices = A.indices.astype(np.int32)
    A.indptr = A.indptr.astype(np.int32)
    return A


def stencil27_variable(shape, seed=2024, rows=None):
    """27-point variable-coefficient operator of BASELINE.json configs[4] as SURVEY.md 8(d)
    specifies it (NOT in the reference): Q1 finite-element stiffness of -div(kappa grad u) on a
    box of unit cells whose unknowns are the interior nodes `shape` (homogeneous Dirichlet
    boundary), kappa = exp(U(-1, 1) ln 10) per cell from default_rng(seed).  Symmetric positive
    definite.  C-order numbering, sorted CSR with int32 indices, all 27 couplings of a node
    stored (the six along the axes cancel to ~0 only where kappa is locally constant),
    assembled directly in CSR (the 256^3 operator has 450 M entries).  rows=(lo, hi): only those
    rows, as a (hi - lo) x N matrix with global columns (one slab of a distributed run)."""
    shape = tuple(int(s) for s in shape)
    if len(shape) != 3:
        raise ValueError("stencil27_variable needs a 3-D shape")
    N = int(np.prod(shape))
    strides = (shape[1] * shape[2], shape[2], 1)
    K, corners = _q1_element_stiffness()
    rng = np.random.default_rng(seed)
    # one cell per (node, node + 1) interval including the two boundary layers: (n + 1)^3 cells;
    # cell c spans nodes c - 1 .. c along each axis (node -1 and node n are the Dirichlet boundary)
    kappa = np.exp(rng.uniform(-1.0, 1.0, size=tuple(s + 1 for s in shape)) * np.log(10.0))
    row_lo, row_hi = (0, N) if rows is None else (int(rows[0]), int(rows[1]))
    M = row_hi - row_lo
    cz, cy, cx = np.unravel_index(np.arange(row_lo, row_hi, dtype=np.int64), shape)
    coords = (cz, cy, cx)
    dirs = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]   # ascending column offset

    def valid_of(d):
        v = np.ones(M, dtype=bool)
        for ax in range(3):
            if d[ax] < 0:
                v &= coords[ax] > 0
            elif d[ax] > 0:
                v &= coords[ax] < shape[ax] - 1
        return v
    valid = [valid_of(d) for d in dirs]
    counts = np.zeros(M, dtype=np.int64)
    for v in valid:
        counts += v
    indptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    nnz = int(indptr[-1])
    indices = np.empty(nnz, dtype=np.int32)
    data = np.empty(nnz, dtype=np.float64)
    cursor = indptr[:-1].copy()
    gid = np.arange(row_lo, row_hi, dtype=np.int64)
    for d, v in zip(dirs, valid):
        off = d[0] * strides[0] + d[1] * strides[1] + d[2] * strides[2]
        # A[i, i + d] = sum over the cells that contain both nodes: local node a of the cell is i,
        # local node a + d is i + d; the cell's origin node is i - a, i.e. cell index i - a + 1
        val = np.zeros(int(v.sum()))
        pz, py, px = cz[v], cy[v], cx[v]
        for ia, a in enumerate(corners):
            b = (a[0] + d[0], a[1] + d[1], a[2] + d[2])
            if min(b) < 0 or max(b) > 1:
                continue
            ib = corners.index(b)
            val += kappa[pz - a[0] + 1, py - a[1] + 1, px - a[2] + 1] * K[ia, ib]
        pos = cursor[v]
        indices[pos] = (gid[v] + off).astype(np.int32)
        data[pos] = val
        cursor += v
    out = sp.csr_matrix((data, indices, indptr.astype(np.int32)), shape=(M, N))
    out.has_sorted_indices = True
    return out
