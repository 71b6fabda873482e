"""Host side of the multi-GPU V-cycle: 1-D slab decomposition of the level hierarchy.

Every rank (one process per GPU) owns a contiguous block of grid planes along axis 0 — the
slowest axis of the C-order numbering the reference uses (openmg/operators.py:64-68) — at
every level.  This module builds, for ONE rank, what `omg_dist_create` needs:

  * the owned rows of A_l with columns renumbered to [owned | halo];
  * the owned block of the restriction R_l (local because slabs are cut on aggregate
    boundaries: planes pair up as (2k, 2k+1), openmg/operators.py:73-84);
  * the Galerkin operator of the next level, A_{l+1} = R_l A_l R_l^T (openmg/operators.py:
    184-186), computed from the rank's own rows only;
  * smoother set keys that agree across ranks (red-black by global coordinate parity, or the
    hyperplane index of the lexicographic sweep);
  * the halo plan: which owned unknowns go to which neighbour, and where received ones land.

Index work only; the sparse products run wherever `spgemm` runs (the GPU by default).
Nothing here has a counterpart in the reference, which is single-process.
"""
import os

import numpy as np
import scipy.sparse as sp


# ------------------------------------------------------------------------------ partition --
class SlabPartition:
    """Row ranges of every level for `n_ranks` slabs along axis 0."""

    def __init__(self, problemShape, n_ranks, n_grids):
        self.shape0 = tuple(int(s) for s in problemShape)
        self.n_ranks = int(n_ranks)
        self.n_grids = int(n_grids)
        self.shapes = [tuple(s // 2 ** l for s in self.shape0) for l in range(self.n_grids)]
        for l, shp in enumerate(self.shapes):
            if min(shp) < 1:
                raise ValueError("level %d has an empty extent: %s" % (l, shp))
            if shp[0] % self.n_ranks:
                raise ValueError("level %d: %d planes do not split evenly over %d ranks" % (l, shp[0], self.n_ranks))
            if l + 1 < self.n_grids:
                if any(s % 2 for s in shp):
                    raise ValueError("level %d: odd extent in %s" % (l, shp))
                if (shp[0] // self.n_ranks) % 2:
                    raise ValueError("level %d: %d planes per rank is odd — aggregates would straddle ranks"
                                     % (l, shp[0] // self.n_ranks))
            if len(shp) == 3 and shp[0] != shp[2]:
                # the reference's restriction uses shape[0] as the second-axis offset (Q6)
                raise ValueError("3-D shapes need shape[0] == shape[2] for the reference's restriction offsets")
            if len(shp) == 2 and shp[0] != shp[1]:
                raise ValueError("2-D shapes need equal extents for the reference's restriction offsets")

    def plane(self, level):
        return int(np.prod(self.shapes[level][1:], dtype=np.int64))

    def planes(self, level, rank):
        per = self.shapes[level][0] // self.n_ranks
        return rank * per, (rank + 1) * per

    def rows(self, level, rank):
        lo, hi = self.planes(level, rank)
        return lo * self.plane(level), hi * self.plane(level)

    def n_rows(self, level):
        return int(np.prod(self.shapes[level], dtype=np.int64))

    def bounds(self, level):
        """Row offsets of all ranks (length n_ranks + 1)."""
        return np.array([self.rows(level, q)[0] for q in range(self.n_ranks)] + [self.n_rows(level)], dtype=np.int64)


# ----------------------------------------------------------------------------- generators --
def stencil_rows(shape, row_lo, row_hi):
    """Rows [row_lo, row_hi) of the Dirichlet 3/5/7-point Laplacian (operators.stencil_poisson)
    with GLOBAL column indices, assembled directly."""
    shape = tuple(int(s) for s in shape)
    dim = len(shape)
    N = int(np.prod(shape, dtype=np.int64))
    rows = np.arange(row_lo, row_hi, dtype=np.int64)
    coords = np.unravel_index(rows, shape)
    strides = [int(np.prod(shape[d + 1:], dtype=np.int64)) for d in range(dim)]
    offsets, valid = [], []
    for d in range(dim):
        offsets.append(-strides[d]); valid.append(coords[d] > 0)
    offsets.append(0); valid.append(np.ones(rows.size, dtype=bool))
    for d in reversed(range(dim)):
        offsets.append(strides[d]); valid.append(coords[d] < shape[d] - 1)
    counts = np.zeros(rows.size, dtype=np.int64)
    for v in valid:
        counts += v
    indptr = np.zeros(rows.size + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    indices = np.empty(int(indptr[-1]), dtype=np.int64)
    data = np.empty(int(indptr[-1]), dtype=np.float64)
    cursor = indptr[:-1].copy()
    for off, v in zip(offsets, valid):
        pos = cursor[v]
        indices[pos] = rows[v] + off
        data[pos] = 2.0 * dim if off == 0 else -1.0
        cursor += v
    return sp.csr_matrix((data, indices, indptr), shape=(rows.size, N))


def stencil27_variable_rows(shape, row_lo, row_hi, seed=2024):
    """Rows [row_lo, row_hi) of operators.stencil27_variable(shape, seed) with GLOBAL column
    indices.  Every rank draws the whole kappa field from the same seed (it is one double per
    cell) and assembles only its rows."""
    from . import operators
    return operators.stencil27_variable(shape, seed, rows=(int(row_lo), int(row_hi)))


def restriction_rows(shape, row_lo, row_hi):
    """Rows [row_lo, row_hi) of operators.restriction(shape) (openmg/operators.py:15-89) with
    GLOBAL column indices; even extents only (SlabPartition checks that)."""
    shape = tuple(int(s) for s in shape)
    alpha = len(shape)
    N = int(np.prod(shape, dtype=np.int64))
    cshape = tuple(s // 2 for s in shape)
    r = np.arange(row_lo, row_hi, dtype=np.int64)
    cc = np.unravel_index(r, cshape)
    first = np.ravel_multi_index(tuple(2 * c for c in cc), shape)
    NX = shape[0]
    offs = [0, 1]
    if alpha >= 2:
        offs += [NX, NX + 1]
    if alpha == 3:
        NXY = shape[0] * shape[1]
        offs += [NXY, NXY + 1, NXY + NX, NXY + NX + 1]
    offs = np.sort(np.array(offs, dtype=np.int64))
    cols = (first[:, None] + offs[None, :]).ravel()
    indptr = np.arange(0, (r.size + 1) * offs.size, offs.size, dtype=np.int64)
    data = np.full(cols.size, 1.0 / 2 ** alpha)
    return sp.csr_matrix((data, cols, indptr), shape=(r.size, N))


def set_keys(shape, rows, smoother, colouring="parity"):
    """Smoother set of each global row of a grid-stencil operator, the same on every rank.
    'colour': with colouring 'parity' the parity of the coordinate sum (red-black; what the
    single-GPU greedy colouring gives on 3/5/7-point stencils), with 'octant' the bits
    (c0 % 2, c1 % 2, ...) most significant first — 2^dim colours, what the greedy colouring
    gives on 9/27-point stencils.  'gs': hyperplane index i0+i1+... — the level schedule of the
    lexicographic sweep on 3/5/7-point stencils.  'jacobi': one set."""
    if smoother in ("jacobi",):
        return None, 0
    coords = np.unravel_index(np.asarray(rows, dtype=np.int64), shape)
    total = np.zeros(np.size(rows), dtype=np.int64)
    for c in coords:
        total += c
    if smoother == "colour" and colouring == "octant":
        key = np.zeros(np.size(rows), dtype=np.int64)
        for c in coords:
            key = 2 * key + c % 2
        return key.astype(np.int32), 2 ** len(shape)
    if smoother == "colour":
        if colouring != "parity":
            raise ValueError("unknown colouring %r" % (colouring,))
        return (total % 2).astype(np.int32), 2
    if smoother == "gs":
        return total.astype(np.int32), int(sum(shape) - len(shape) + 1)
    raise ValueError("unknown smoother %r" % (smoother,))


# ------------------------------------------------------------------------------- one rank --
def compress_columns(M, own_lo, own_hi, bounds=None, group_of=None):
    """Renumber the GLOBAL columns of M: owned -> 0..n_loc-1, the others (the halo) -> n_loc +
    position in the halo list, which is ordered by (owning rank, message group, global id).
    `group_of(gids)` gives the message group of remote unknowns (their smoother colour, so that
    after relaxing colour c only colour-c values travel); None = one group.
    Returns (M_local, halo_gids, halo_groups)."""
    M = sp.csr_matrix(M)
    cols = M.indices.astype(np.int64)
    owned = (cols >= own_lo) & (cols < own_hi)
    halo = np.unique(cols[~owned])
    groups = np.zeros(halo.size, dtype=np.int32) if group_of is None else np.asarray(group_of(halo), dtype=np.int32)
    owner = np.zeros(halo.size, dtype=np.int64) if bounds is None else np.searchsorted(bounds, halo, side="right") - 1
    order = np.lexsort((halo, groups, owner))
    rank_of_sorted = np.empty(halo.size, dtype=np.int64)
    rank_of_sorted[order] = np.arange(halo.size)
    new = np.empty_like(cols)
    new[owned] = cols[owned] - own_lo
    new[~owned] = (own_hi - own_lo) + rank_of_sorted[np.searchsorted(halo, cols[~owned])]
    out = sp.csr_matrix((M.data.copy(), new.astype(np.int32), M.indptr.astype(np.int32)),
                        shape=(M.shape[0], (own_hi - own_lo) + halo.size))
    return out, halo[order], groups[order]


def make_halo_plans(halo_lists, bounds, halo_groups=None):
    """halo_lists[q] = global ids rank q needs, ordered by (owner, group, id) as
    compress_columns leaves them; halo_groups[q] their message groups (None: one group);
    bounds = row offsets of the ranks.  One plan ENTRY per (peer, group): returns per rank
    dict(peers, groups, recv_off, send_off, send_idx) with peers/groups of length n_entries and
    send_idx in the sender's local (natural) numbering, in the receiver's halo order."""
    n = len(halo_lists)
    if halo_groups is None:
        halo_groups = [np.zeros(len(h), dtype=np.int32) for h in halo_lists]
    owner = [np.searchsorted(bounds, h, side="right") - 1 for h in halo_lists]
    n_groups = 1 + max([int(g.max()) for g in halo_groups if len(g)] + [0])

    def want(q, p, g):                       # ids q wants from p in group g (q's halo order)
        sel = (owner[q] == p) & (halo_groups[q] == g)
        return halo_lists[q][sel]

    plans = []
    for q in range(n):
        peers, groups, recv_off, send_off, send_idx = [], [], [0], [0], []
        for p in range(n):
            if p == q:
                continue
            for g in range(n_groups):
                mine, theirs = want(q, p, g), want(p, q, g)
                if mine.size == 0 and theirs.size == 0:
                    continue
                peers.append(p)
                groups.append(g)
                recv_off.append(recv_off[-1] + mine.size)
                idx = theirs - bounds[q]
                send_idx.append(idx)
                send_off.append(send_off[-1] + idx.size)
        if recv_off[-1] != len(halo_lists[q]):
            raise ValueError("rank %d: halo columns outside every rank's range" % q)
        # the entries must tile the halo region in order
        chk = np.concatenate([want(q, p, g) for p, g in zip(peers, groups)]) if peers else np.zeros(0, dtype=np.int64)
        if not np.array_equal(chk, halo_lists[q]):
            raise ValueError("rank %d: halo list is not ordered by (owner, group, id)" % q)
        plans.append({"peers": np.array(peers, dtype=np.int32),
                      "groups": np.array(groups, dtype=np.int32),
                      "recv_off": np.array(recv_off, dtype=np.int64),
                      "send_off": np.array(send_off, dtype=np.int64),
                      "send_idx": (np.concatenate(send_idx) if send_idx else np.zeros(0)).astype(np.int32)})
    return plans


class RankSetup:
    """Builds one rank's levels.  Drive it level by level:

        for l in range(n_grids):
            halo = setup.begin_level(l)                 # local
            all_halos = <all-gather of halo over ranks>  # collective (or a loop, in-process)
            setup.finish_level(l, all_halos)            # local
        coarse = <all-gather of setup.coarse_rows()>    # collective

    `spgemm(X, Y)` does the two Galerkin products (default: the device SpGEMM)."""

    def __init__(self, part, rank, A0_rows, smoother="colour", spgemm=None, overlap=True, colouring="parity"):
        """colouring: 'parity' (3/5/7-point stencils) or 'octant' (9/27-point), see set_keys; it
        must be a valid colouring of every level's operator on every rank (checked)."""
        self.part, self.rank, self.smoother, self.overlap = part, int(rank), smoother, bool(overlap)
        self.colouring = colouring
        if spgemm is None:
            from . import _hip
            spgemm = _hip.spgemm
        self.spgemm = spgemm
        lo, hi = part.rows(0, rank)
        A0_rows = sp.csr_matrix(A0_rows)
        if A0_rows.shape != (hi - lo, part.n_rows(0)):
            raise ValueError("A0_rows must hold this rank's rows with global columns: expected %s, got %s"
                             % ((hi - lo, part.n_rows(0)), A0_rows.shape))
        self._A_glob = A0_rows           # owned rows of the current level, global columns
        self.levels = []
        self._halo = None

    def begin_level(self, l):
        part, q = self.part, self.rank
        lo, hi = part.rows(l, q)
        last = l + 1 == part.n_grids
        # message groups = smoother colours: after relaxing colour c only colour-c boundary
        # values change, so only they are sent (half the bytes for red-black)
        group_of = None
        if not last and self.smoother == "colour":
            group_of = lambda gids: set_keys(part.shapes[l], gids, "colour", self.colouring)[0]
        A_loc, halo, halo_groups = compress_columns(self._A_glob, lo, hi, part.bounds(l), group_of)
        lv = {"A": A_loc, "R": None, "n_halo": int(halo.size), "keys": None, "n_sets": 0}
        if not last:
            keys, n_sets = set_keys(part.shapes[l], np.arange(lo, hi), self.smoother, self.colouring)
            if keys is not None:
                self._check_keys(A_loc, keys, set_keys(part.shapes[l], halo, self.smoother, self.colouring)[0], hi - lo)
            lv["set_group"] = 1
            min_rows = int(os.environ.get("OMG_OVERLAP_MIN_ROWS", 1 << 19))   # same rule as csrc/dist.hip
            if keys is not None and self.smoother == "colour" and self.overlap and hi - lo >= min_rows:
                # rows that touch the halo (for a symmetric pattern: exactly the rows the
                # neighbours need) go FIRST inside their colour, so that their exchange can run
                # while the interior rows of the colour are still being relaxed
                interior = np.ones(hi - lo, dtype=np.int32)
                touches = np.unique(np.repeat(np.arange(hi - lo), np.diff(A_loc.indptr))[A_loc.indices >= hi - lo])
                interior[touches] = 0
                keys = (2 * keys + interior).astype(np.int32)
                n_sets *= 2
                lv["set_group"] = 2
            lv["keys"], lv["n_sets"] = keys, n_sets
            clo, chi = part.rows(l + 1, q)
            R_glob = restriction_rows(part.shapes[l], clo, chi)
            if R_glob.indices.min() < lo or R_glob.indices.max() >= hi:
                raise ValueError("level %d: an aggregate straddles the slab boundary" % l)
            R_loc = sp.csr_matrix((R_glob.data, (R_glob.indices - lo).astype(np.int32),
                                   R_glob.indptr.astype(np.int32)), shape=(chi - clo, hi - lo))
            lv["R"] = R_loc
            # Galerkin: (R_loc A_loc) Z, Z = rows of R^T for [owned | halo] fine unknowns
            ext = np.concatenate([np.arange(lo, hi, dtype=np.int64), halo])
            cplane, fplane = part.plane(l + 1), part.plane(l)
            c_lo = max(0, clo - cplane)                      # one coarse plane of margin each side
            c_hi = min(part.n_rows(l + 1), chi + cplane)
            f_lo, f_hi = 2 * (c_lo // cplane) * fplane, 2 * (c_hi // cplane) * fplane
            if halo.size and (halo.min() < f_lo or halo.max() >= f_hi):
                raise ValueError("level %d: halo reaches beyond the neighbouring plane pair" % l)
            R_ext = restriction_rows(part.shapes[l], c_lo, c_hi)
            R_ext = sp.csr_matrix((R_ext.data, R_ext.indices - f_lo, R_ext.indptr), shape=(c_hi - c_lo, f_hi - f_lo))
            Rt_ext = sp.csr_matrix(R_ext.T)                  # (fine rows f_lo..f_hi) x (coarse rows c_lo..c_hi)
            # The products run with the fine unknowns numbered in GLOBAL order (lower halo, owned,
            # upper halo), not [owned | halo]: SpGEMM adds the terms of an output entry in the
            # column order of its left factor's row, and only this numbering makes that order —
            # and with it every rounding of a variable-coefficient coarse operator — the one of
            # the global product R A R^T, whatever the number of ranks.
            ext_sorted = np.sort(ext)
            to_sorted = np.searchsorted(ext_sorted, ext)
            A_sorted = sp.csr_matrix((A_loc.data, to_sorted[A_loc.indices].astype(np.int32), A_loc.indptr),
                                     shape=(hi - lo, ext.size))
            Z = sp.csr_matrix(Rt_ext[ext_sorted - f_lo])
            if (np.diff(Z.indptr) == 0).any():
                raise ValueError("level %d: a halo unknown's aggregate lies outside the neighbouring planes" % l)
            RA = self.spgemm(R_loc, A_sorted)
            An = sp.csr_matrix(self.spgemm(RA, Z))
            self._A_next = sp.csr_matrix((An.data, An.indices.astype(np.int64) + c_lo, An.indptr),
                                         shape=(chi - clo, part.n_rows(l + 1)))
        self.levels.append(lv)
        self._halo = halo
        return halo, halo_groups

    def finish_level(self, l, all_halos):
        """all_halos: every rank's begin_level(l) result, in rank order."""
        lists = [np.asarray(h[0], dtype=np.int64) for h in all_halos]
        groups = [np.asarray(h[1], dtype=np.int32) for h in all_halos]
        plan = make_halo_plans(lists, self.part.bounds(l), groups)[self.rank]
        if self.smoother != "colour" or l + 1 == self.part.n_grids:
            plan["groups"] = None                 # one message per peer, sent after every set
        self.levels[l].update(plan)
        if l + 1 < self.part.n_grids:
            self._A_glob = self._A_next
        return self.levels[l]

    def coarse_rows(self):
        """Owned rows of the coarsest operator with global columns (all-gather and stack them)."""
        return self._A_glob

    @staticmethod
    def _check_keys(A_loc, keys, halo_keys, n_loc):
        """Rows of one smoother set must be mutually uncoupled, across the slab boundary too."""
        row = np.repeat(np.arange(A_loc.shape[0]), np.diff(A_loc.indptr))
        col = A_loc.indices
        all_keys = np.concatenate([keys, halo_keys if halo_keys is not None else np.zeros(0, dtype=np.int32)])
        off = col != row
        if (all_keys[col[off]] == keys[row[off]]).any():
            raise ValueError("the coordinate-based smoother sets are not independent for this operator "
                             "(colouring='parity' fits 3/5/7-point grid stencils, 'octant' 9/27-point ones)")


def make_tail(coarse_global, shape, n_grids, smoother="colour", omega=1.0, dtype="float64"):
    """Replicated tail: an ordinary single-GPU hierarchy over the levels BELOW the last
    distributed one.  `coarse_global` is the whole operator of that level (shape `shape`),
    `n_grids` how many grids the tail has (1 = direct solve only).  Every rank builds the same
    object; DistRank(..., coarse_global=None, tail=...) uses it as its coarse solver, so the
    small levels — pure exchange latency in a slab decomposition — need no communication."""
    from . import _hip, operators
    shapes = [tuple(s // 2 ** l for s in shape) for l in range(n_grids)]
    R = [operators.restriction(shapes[l]) for l in range(n_grids - 1)]
    A = operators.coeffecientList(sp.csr_matrix(coarse_global), R)
    return _hip.Hierarchy(A, R, smoother=smoother, omega=omega, dtype=dtype)


def assemble_coarse(rows_per_rank):
    G = sp.vstack([sp.csr_matrix(r) for r in rows_per_rank], format="csr")
    G.sum_duplicates()
    return G


def build_all_ranks(part, A0_rows_of, smoother="colour", spgemm=None, overlap=True, colouring="parity"):
    """In-process construction of EVERY rank (loopback groups, tests).  A0_rows_of(rank) gives
    that rank's fine rows.  Returns (levels_per_rank, coarse_global, coarse_counts)."""
    setups = [RankSetup(part, q, A0_rows_of(q), smoother=smoother, spgemm=spgemm, overlap=overlap, colouring=colouring)
              for q in range(part.n_ranks)]
    for l in range(part.n_grids):
        halos = [s.begin_level(l) for s in setups]
        for s in setups:
            s.finish_level(l, halos)
    coarse = assemble_coarse([s.coarse_rows() for s in setups])
    last = part.n_grids - 1
    counts = [part.rows(last, q)[1] - part.rows(last, q)[0] for q in range(part.n_ranks)]
    return [s.levels for s in setups], coarse, counts


def build_this_rank(part, rank, A0_rows, all_gather, smoother="colour", spgemm=None, overlap=True,
                    colouring="parity"):
    """SPMD construction: `all_gather(obj)` returns the list of every rank's obj (e.g.
    torch.distributed.all_gather_object).  Returns (levels, coarse_global, coarse_counts)."""
    s = RankSetup(part, rank, A0_rows, smoother=smoother, spgemm=spgemm, overlap=overlap, colouring=colouring)
    for l in range(part.n_grids):
        halo = s.begin_level(l)
        s.finish_level(l, all_gather(halo))
    coarse = assemble_coarse(all_gather(s.coarse_rows()))
    last = part.n_grids - 1
    counts = [part.rows(last, q)[1] - part.rows(last, q)[0] for q in range(part.n_ranks)]
    return s.levels, coarse, counts
