"""CPU executor of the 27-point slab schedule of openmg_amd/csrc/dist27.hip (test infrastructure).

The schedule in NumPy/SciPy, rank by rank, with caller-supplied message passing: vectors on the rank's EXTENDED slab (one
ghost aggregate plane = two grid planes on either side), the 8-colour sweep with colours 0 .. 3 of the upper ghost plane
relaxed redundantly, ONE exchange per sweep (colours 4 .. 7 of the boundary aggregate planes both ways), the coarse
right-hand side's colours 0 .. 3 of the upper ghost plane exchanged after each restriction, the prolongation applied to
the ghost planes from the coarse level's ghost cells, the level below the slabs gathered and cycled replicated.
What it proves on CPU: that this exchange pattern reproduces the single-process cycle (openmg/__init__.py:151-236 with
the colour-ordered sweep of openmg/solvers.py:56-68) for any number of ranks."""
import numpy as np
import scipy.sparse as sp

from oracle import mg_oracle as orc


def colour_of(shape_ext):
    """Octant colour (i & 1) + 2 (j & 1) + 4 (k & 1) of every cell of a C-ordered grid (planes k, lines j, cells i)."""
    k, j, i = np.unravel_index(np.arange(int(np.prod(shape_ext))), shape_ext)
    return (i & 1) + 2 * (j & 1) + 4 * (k & 1)


class Slab27Level:
    def __init__(self, A_global, shape, rank, world):
        nz, ny, nx = shape
        self.nzo = nz // world
        assert self.nzo % 2 == 0 and self.nzo >= 2
        self.shape_ext = (self.nzo + 4, ny, nx)
        self.plane = ny * nx
        self.n_ext = self.plane * (self.nzo + 4)
        k0 = rank * self.nzo
        # global row / column of every cell of the extended slab (-1: outside the grid)
        kk = np.arange(k0 - 2, k0 + self.nzo + 2)
        inside = (kk >= 0) & (kk < nz)
        gid = np.where(inside[:, None], kk[:, None] * self.plane + np.arange(self.plane)[None, :], -1).ravel()
        self.gid = gid
        # rows of the owned planes and of the upper ghost aggregate plane's EVEN plane (the redundantly relaxed rows),
        # columns in the extended numbering (every coupling of such a row lies inside the extended slab)
        want = np.zeros(self.n_ext, dtype=bool)
        want[2 * self.plane:(self.nzo + 2) * self.plane] = True
        if rank + 1 < world:
            want[(self.nzo + 2) * self.plane:(self.nzo + 3) * self.plane] = True
        rows = np.flatnonzero(want)
        sub = sp.csr_matrix(A_global)[gid[rows]].tocoo()
        lo = (k0 - 2) * self.plane
        col_ext = sub.col - lo
        assert col_ext.min() >= 0 and col_ext.max() < self.n_ext
        self.A = sp.csr_matrix((sub.data, (rows[sub.row], col_ext)), shape=(self.n_ext, self.n_ext))
        self.diag = self.A.diagonal()
        col = colour_of(self.shape_ext)
        plane_of = np.arange(self.n_ext) // self.plane
        owned = (plane_of >= 2) & (plane_of < self.nzo + 2)
        self.owned = owned
        upper_even = (plane_of == self.nzo + 2) & (rank + 1 < world)
        # rows a sweep relaxes, colour by colour
        self.rows_of_colour = [np.flatnonzero((col == c) & (owned | (upper_even if c < 4 else False))) for c in range(8)]
        self.colour = col
        self.plane_of = plane_of
        self.x = np.zeros(self.n_ext)
        self.b = np.zeros(self.n_ext)

    def sel(self, agg_plane, colours):
        """cells of aggregate plane `agg_plane` of the extended slab with one of `colours`"""
        return np.flatnonzero((self.plane_of // 2 == agg_plane) & np.isin(self.colour, list(colours)))


class Slab27Cpu:
    """Drives the ranks in `local` (all of them in-process, or one per process over gloo)."""

    def __init__(self, shape, world, n_levels, A_levels, R_levels, local, comm, tail_cycle):
        self.world, self.n_levels, self.comm, self.tail_cycle = world, n_levels, comm, tail_cycle
        self.local = list(local)
        self.shapes = [tuple(s >> l for s in shape) for l in range(n_levels + 1)]
        self.lv = {r: [Slab27Level(A_levels[l], self.shapes[l], r, world) for l in range(n_levels)] for r in self.local}
        self.exchanges = 0

    # what: 0 x after a sweep, 1 x after a load, 2 the right-hand side
    def halo(self, l, what):
        up = [] if what == 2 else [4, 5, 6, 7]
        down = {0: [4, 5, 6, 7], 1: list(range(8)), 2: [0, 1, 2, 3]}[what]
        sends, recvs = [], []
        for r in self.local:
            L = self.lv[r][l]
            v = L.b if what == 2 else L.x
            hz = (L.nzo + 4) // 2
            if r + 1 < self.world:
                if up:
                    sends.append((r, r + 1, v[L.sel(hz - 2, up)].copy()))
                recvs.append((r, r + 1, L.sel(hz - 1, down)))
            if r > 0:
                sends.append((r, r - 1, v[L.sel(1, down)].copy()))
                if up:
                    recvs.append((r, r - 1, L.sel(0, up)))
        if sends or recvs:
            self.exchanges += 1
        got = self.comm.exchange(sends, [(dst, src, idx.size) for dst, src, idx in recvs])
        for (dst, src, idx), buf in zip(recvs, got):
            L = self.lv[dst][l]
            (L.b if what == 2 else L.x)[idx] = buf

    def sweep(self, l):
        for c in range(8):                                   # openmg/solvers.py:56-68 on the colour-ordered rows
            for r in self.local:
                L = self.lv[r][l]
                rows = L.rows_of_colour[c]
                L.x[rows] += (L.b[rows] - L.A[rows] @ L.x) / L.diag[rows]
        self.halo(l, 0)

    def cycle(self, l, pre, post, x_zero):
        last = l + 1 == self.n_levels
        if x_zero:
            for r in self.local:
                self.lv[r][l].x[:] = 0.0
        for _ in range(pre):
            self.sweep(l)
        for r in self.local:
            L = self.lv[r][l]
            res = np.where(L.owned, L.b - L.A @ L.x, 0.0)
            # restriction: the aggregate of the extended slab (K, J, I) = cell (K + 1, J, I) of the next extended slab
            nzc, nyc, nxc = (L.nzo + 4) // 2, L.shape_ext[1] // 2, L.shape_ext[2] // 2
            agg = res.reshape(nzc, 2, nyc, 2, nxc, 2)
            bc = np.zeros((nzc, nyc, nxc))
            for dk in range(2):                              # R's column order (openmg/operators.py:73-84)
                for dj in range(2):
                    for di in range(2):
                        bc += 0.125 * agg[:, dk, :, dj, :, di]
            L.bc_ext = bc                                    # planes K = 0 .. nzc - 1
        if last:
            own = [(r, self.lv[r][l].bc_ext[1:-1].ravel()) for r in self.local]
            full = self.comm.allgather(own)
            e_full = self.tail_cycle(np.concatenate(full), pre, post).reshape(self.shapes[l + 1])
            for r in self.local:
                L = self.lv[r][l]
                nzc = (L.nzo + 4) // 2
                k0 = r * (nzc - 2)
                e = np.zeros((nzc,) + e_full.shape[1:])
                for K in range(nzc):
                    if 0 <= k0 - 1 + K < e_full.shape[0]:
                        e[K] = e_full[k0 - 1 + K]
                L.e_ext = e
        else:
            for r in self.local:
                L, C = self.lv[r][l], self.lv[r][l + 1]
                bc = np.zeros(C.shape_ext)
                bc[1:1 + L.bc_ext.shape[0]] = L.bc_ext
                owned_c = C.owned.reshape(C.shape_ext)
                C.b[:] = np.where(owned_c, bc, 0.0).ravel()
            self.halo(l + 1, 2)
            self.cycle(l + 1, pre, post, True)
            for r in self.local:
                L, C = self.lv[r][l], self.lv[r][l + 1]
                nzc = (L.nzo + 4) // 2
                L.e_ext = C.x.reshape(C.shape_ext)[1:1 + nzc]
        for r in self.local:                                 # x += R^T e on every plane of the extended slab
            L = self.lv[r][l]
            e = L.e_ext
            L.x += 0.125 * np.repeat(np.repeat(np.repeat(e, 2, axis=0), 2, axis=1), 2, axis=2).ravel()
        for _ in range(post):
            self.sweep(l)

    def load(self, b_of_rank, x0_of_rank=None):
        for r in self.local:
            L = self.lv[r][0]
            L.b[:] = 0.0
            L.x[:] = 0.0
            L.b[L.owned] = b_of_rank(r)
            if x0_of_rank is not None:
                L.x[L.owned] = x0_of_rank(r)
        self.halo(0, 2)
        self.halo(0, 1)

    def run(self, pre, post, n_cycles):
        norms = []
        for _ in range(n_cycles):
            self.exchanges = 0
            self.cycle(0, pre, post, False)
            sq = 0.0
            for r in self.local:
                L = self.lv[r][0]
                res = (L.b - L.A @ L.x)[L.owned]
                sq += float(res @ res)
            norms.append(float(np.sqrt(self.comm.allreduce_sum(sq))))
        return {r: self.lv[r][0].x[self.lv[r][0].owned].copy() for r in self.local}, norms


class InProcessComm:
    """Every rank lives in this process."""

    def exchange(self, sends, recvs):
        box = {}
        for src, dst, buf in sends:
            box[(src, dst)] = buf
        return [box[(src, dst)] for dst, src, _ in recvs]

    def allgather(self, own):
        return [a for _, a in sorted(own, key=lambda t: t[0])]

    def allreduce_sum(self, v):
        return v


def tail_of(A_tail_levels, R_tail_levels):
    """Zero-start V-cycle over the levels below the slabs (the replicated tail) by the oracle."""
    sm = orc.make_smoother("colour", A_tail_levels) if len(A_tail_levels) > 1 else None

    def run(b, pre, post):
        if len(A_tail_levels) == 1:
            return orc.coarse_solve(A_tail_levels[0], b)
        p = {"preIterations": pre, "postIterations": post, "coarsestLevel": len(R_tail_levels)}
        x, _ = orc.mg_cycle(A_tail_levels, b, 0, R_tail_levels, p, initial=None, smoother=sm)
        return x
    return run
