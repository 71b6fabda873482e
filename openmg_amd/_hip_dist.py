"""ctypes wrappers of the multi-GPU entry points (omg_dist_* in include/openmg_hip.h)."""
import ctypes

import numpy as np
import scipy.sparse as sp

from ._hip import DistLevelView, as_csr, check, csr_view, dtype_code, lib, smoother_code, vec


def peer_access(device, peer_device):
    """Can `device` address `peer_device`'s memory?  (asked before peer mode maps anything)"""
    can = ctypes.c_int(0)
    check(lib().omg_peer_access(int(device), int(peer_device), ctypes.byref(can)))
    return bool(can.value)


def set_device(device):
    check(lib().omg_set_device(int(device)))


def self_exchange_us(n_bytes, reps=200):
    """Microseconds per grouped ncclSend + ncclRecv of n_bytes to this rank itself (one-rank communicator):
    the floor of one halo exchange (omg_rccl_self_exchange_time)."""
    us = ctypes.c_double(0.0)
    check(lib().omg_rccl_self_exchange_time(int(n_bytes), int(reps), ctypes.byref(us)))
    return us.value


def rccl_unique_id():
    """128-byte RCCL bootstrap id (make it on rank 0, broadcast it, connect everywhere)."""
    buf = ctypes.create_string_buffer(128)
    check(lib().omg_rccl_unique_id(buf))
    return buf.raw


class DistRank:
    """One rank's slab of a row-partitioned hierarchy (omg_dist).  `levels` is the list made
    by openmg_amd.dist: dicts with A, R (CSR, local column numbering), n_halo, keys, n_sets,
    peers, send_off, send_idx, recv_off."""

    def __init__(self, rank, n_ranks, levels, coarse_global, coarse_counts, smoother="colour", omega=1.0,
                 tail=None, dtype="float64"):
        """coarse_global: the whole operator of the last distributed level (direct solve), or None
        when `tail` — a _hip.Hierarchy over the levels below it — does that job.  dtype: precision
        of the levels on the device (and of the halo messages); host vectors are float64."""
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        self._tail = tail
        keep = []
        views = (DistLevelView * len(levels))()
        empty = sp.csr_matrix((0, 0))
        for l, lv in enumerate(levels):
            A = as_csr(lv["A"])
            R = as_csr(lv["R"]) if lv.get("R") is not None else as_csr(empty)
            keys = None if lv.get("keys") is None else np.ascontiguousarray(lv["keys"], dtype=np.int32)
            peers = np.ascontiguousarray(lv["peers"], dtype=np.int32)
            send_off = np.ascontiguousarray(lv["send_off"], dtype=np.int64)
            send_idx = np.ascontiguousarray(lv["send_idx"], dtype=np.int32)
            recv_off = np.ascontiguousarray(lv["recv_off"], dtype=np.int64)
            keep += [A, R, keys, peers, send_off, send_idx, recv_off]
            v = views[l]
            v.A, v.R = csr_view(A), csr_view(R)
            v.n_halo = int(lv["n_halo"])
            v.keys = None if keys is None else keys.ctypes.data
            v.n_sets = int(lv.get("n_sets", 0))
            v.n_peers = len(peers)
            v.peers = peers.ctypes.data
            v.send_off = send_off.ctypes.data
            v.send_idx = send_idx.ctypes.data
            v.recv_off = recv_off.ctypes.data
            v.set_group = int(lv.get("set_group", 1))
            groups = lv.get("groups")
            if groups is not None:
                groups = np.ascontiguousarray(groups, dtype=np.int32)
                keep.append(groups)
                v.entry_group = groups.ctypes.data
            else:
                v.entry_group = None
        G = None if coarse_global is None else as_csr(coarse_global)
        gv = None if G is None else csr_view(G)
        counts = (ctypes.c_int64 * self.n_ranks)(*[int(c) for c in coarse_counts])
        self.n_local = levels[0]["A"].shape[0]
        self.n_halo = int(levels[0]["n_halo"])
        self.nnz_local = int(levels[0]["A"].nnz)
        h = ctypes.c_void_p()
        check(lib().omg_dist_create_ex(self.rank, self.n_ranks, len(levels), views,
                                       None if gv is None else ctypes.byref(gv), counts,
                                       smoother_code(smoother), float(omega), dtype_code(dtype), ctypes.byref(h)))
        self._h = h
        if tail is not None:
            check(lib().omg_dist_set_tail(self._h, tail._h))
        del keep

    def close(self):
        if getattr(self, "_h", None):
            lib().omg_dist_destroy(self._h)
            self._h = None
        if getattr(self, "_tail", None) is not None:
            self._tail.close()
            self._tail = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def connect(self, unique_id):
        check(lib().omg_dist_connect(self._h, ctypes.c_char_p(unique_id)))

    def rccl_ranks(self):
        """Ranks of the RCCL communicator this rank joined (0 before connect)."""
        n = ctypes.c_int(0)
        check(lib().omg_dist_rccl_ranks(self._h, ctypes.byref(n)))
        return n.value

    def set_stream(self, hip_stream):
        check(lib().omg_dist_set_stream(self._h, ctypes.c_void_p(hip_stream or 0)))

    def sync(self):
        check(lib().omg_dist_sync(self._h))

    def load(self, b_local, x0_local=None):
        b = vec(b_local, self.n_local)
        x0 = None if x0_local is None else vec(x0_local, self.n_local)
        check(lib().omg_dist_load(self._h, b.ctypes.data, None if x0 is None else x0.ctypes.data))

    def fetch(self):
        x = np.empty(self.n_local)
        check(lib().omg_dist_fetch(self._h, x.ctypes.data))
        return x

    def cycle(self, pre, post, want_norm=True):
        if want_norm:
            norm = ctypes.c_double(0.0)
            check(lib().omg_dist_cycle(self._h, int(pre), int(post), ctypes.byref(norm)))
            return norm.value
        check(lib().omg_dist_cycle(self._h, int(pre), int(post), None))
        return None


    def cycles(self, pre, post, n_cycles):
        """n_cycles V-cycles back to back (collective); every cycle's global norm (omg_dist_cycles)."""
        norms = (ctypes.c_double * max(int(n_cycles), 1))()
        check(lib().omg_dist_cycles(self._h, int(pre), int(post), int(n_cycles), norms))
        return [float(norms[k]) for k in range(int(n_cycles))]

    def spmv_time(self, reps=20):
        """Average milliseconds of y = A_0 x over this rank's rows (omg_dist_spmv_time)."""
        ms = ctypes.c_double(0.0)
        check(lib().omg_dist_spmv_time(self._h, int(reps), ctypes.byref(ms)))
        return ms.value

    def level_flags(self, level):
        f = ctypes.c_int(0)
        check(lib().omg_dist_level_flags(self._h, int(level), ctypes.byref(f)))
        return {"scatter_prolong": bool(f.value & 2), "paired_sets": bool(f.value & 4), "split_scatter": bool(f.value & 8)}

    def format_info(self, level, op="A", set=-1):
        from ._hip import Hierarchy
        out = (ctypes.c_int64 * len(Hierarchy.FORMAT_FIELDS))()
        check(lib().omg_dist_format_info(self._h, int(level), {"A": 0, "R": 1, "P": 2}[op], int(set), out))
        return dict(zip(Hierarchy.FORMAT_FIELDS, [int(v) for v in out]))


class DistGroup:
    """Loopback group: every rank of a decomposition inside this process, on one GPU."""

    def __init__(self, ranks):
        self.ranks = list(ranks)
        arr = (ctypes.c_void_p * len(self.ranks))(*[r._h for r in self.ranks])
        g = ctypes.c_void_p()
        check(lib().omg_dist_group_create(len(self.ranks), arr, ctypes.byref(g)))
        self._g = g

    def cycle(self, pre, post, want_norm=True):
        norm = ctypes.c_double(0.0)
        check(lib().omg_dist_group_cycle(self._g, int(pre), int(post), ctypes.byref(norm) if want_norm else None))
        return norm.value if want_norm else None

    def cycles(self, pre, post, n_cycles):
        norms = (ctypes.c_double * max(int(n_cycles), 1))()
        check(lib().omg_dist_group_cycles(self._g, int(pre), int(post), int(n_cycles), norms))
        return [float(norms[k]) for k in range(int(n_cycles))]

    def close(self):
        if getattr(self, "_g", None):
            lib().omg_dist_group_destroy(self._g)
            self._g = None
        for r in self.ranks:
            r.close()


# ---- plane-pipelined slabs (omg_pdist_*) ------------------------------------------------------
def star_coefficients(A, shape):
    """The seven coefficients (-K, -J, -I, diagonal, +I, +J, +K) of a constant-coefficient star stencil on a
    C-ordered grid of `shape` (planes, lines, cells), read off the row of cell (1, 1, 1)."""
    A = as_csr(A)
    nz, ny, nx = (int(s) for s in shape)
    r = (1 * ny + 1) * nx + 1
    lo, hi = A.indptr[r], A.indptr[r + 1]
    cols, vals = A.indices[lo:hi], A.data[lo:hi]
    want = [r - nx * ny, r - nx, r - 1, r, r + 1, r + nx, r + nx * ny]
    if list(cols) != want:
        raise ValueError("not a seven-point star stencil with ascending columns")
    return [float(v) for v in vals]


class PlaneDistRank:
    """One rank's slab of a constant-coefficient 7-point hierarchy run as plane-pipelined passes (omg_pdist).
    shape: the GLOBAL finest grid (planes, lines, cells); coefficients: seven per distributed level; tail: a
    _hip.Hierarchy over the levels below the slabs (kept alive here)."""

    def __init__(self, rank, n_ranks, shape, coefficients, weight, tail):
        nz, ny, nx = (int(s) for s in shape)
        c = np.ascontiguousarray(np.asarray(coefficients, dtype=np.float64).reshape(-1, 7))
        h = ctypes.c_void_p()
        check(lib().omg_pdist_create(int(rank), int(n_ranks), nx, ny, nz, c.shape[0], c.ctypes.data, float(weight), ctypes.byref(h)))
        self._h = h
        self._tail = tail
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        self.n_local = nx * ny * (nz // int(n_ranks))
        check(lib().omg_pdist_set_tail(self._h, tail._h))

    def connect(self, unique_id, unique_id_side):
        """Two RCCL bootstrap ids (rccl_unique_id() twice on rank 0, broadcast): the cycle's communicator and the
        one of the exchanges that run on the second stream."""
        a = ctypes.create_string_buffer(bytes(unique_id), 128)
        b = ctypes.create_string_buffer(bytes(unique_id_side), 128)
        check(lib().omg_pdist_connect(self._h, a, b))

    def rccl_ranks(self):
        n = ctypes.c_int(0)
        check(lib().omg_pdist_rccl_ranks(self._h, ctypes.byref(n)))
        return n.value

    def load(self, b_local, x0_local=None):
        b = vec(b_local, self.n_local)
        x0 = None if x0_local is None else vec(x0_local, self.n_local)
        check(lib().omg_pdist_load(self._h, b.ctypes.data, None if x0 is None else x0.ctypes.data))

    def fetch(self):
        x = np.empty(self.n_local, dtype=np.float64)
        check(lib().omg_pdist_fetch(self._h, x.ctypes.data))
        return x

    def sync(self):
        check(lib().omg_pdist_sync(self._h))

    def set_gate(self, enable=True):
        """Gated passes (the finest level's passes as one launch whose edge chunks wait for exchanges that run beside them)."""
        check(lib().omg_pdist_set_gate(self._h, 1 if enable else 0))

    def info(self):
        out = (ctypes.c_int64 * 8)()
        check(lib().omg_pdist_info(self._h, out))
        keys = ("levels", "gated", "tile_x", "tile_y", "tile_z", "workgroups", "threads", "gate_tile_z")
        return dict(zip(keys, [int(v) for v in out]))

    PHASES = {0: "not started", 1: "halo exchange of x", 2: "halo exchange of the right-hand side", 3: "down pass",
              4: "halo exchange of x for the up pass", 5: "gather + replicated tail", 6: "halo exchange of the correction",
              7: "up pass"}

    def trace(self, enable=True):
        check(lib().omg_pdist_trace(self._h, 1 if enable else 0))

    def progress(self):
        """(cycle, level, last phase the DEVICE has completed) — read without synchronising."""
        w = ctypes.c_uint(0)
        check(lib().omg_pdist_progress(self._h, ctypes.byref(w)))
        return w.value >> 16, (w.value >> 8) & 0xFF, self.PHASES.get(w.value & 0xFF, "phase %d" % (w.value & 0xFF))

    def cycles(self, n_cycles, reduce=None, pre=1, post=1):
        """n cycles -> every cycle's global residual norm.  reduce (peer mode without a communicator): a callable
        taking this rank's list of squared-residual sums and returning the sums over all ranks.  pre, post in {0, 1}."""
        norms = (ctypes.c_double * max(int(n_cycles), 1))()
        if reduce is None:
            check(lib().omg_pdist_cycles_ex(self._h, int(pre), int(post), int(n_cycles), norms))
            out = [float(norms[k]) for k in range(int(n_cycles))]
        else:
            if (int(pre), int(post)) != (1, 1):
                raise ValueError("cycles(reduce=...) runs V(1,1) only (omg_pdist_cycles_squares has no sweep counts)")
            check(lib().omg_pdist_cycles_squares(self._h, int(n_cycles), norms))
            out = [float(v) ** 0.5 for v in reduce([float(norms[k]) for k in range(int(n_cycles))])]
        status = self.p2p_status()
        if status:
            raise RuntimeError("rank %d: a bounded wait inside a pass gave up (status %d: bit 0 a neighbour's flag — the "
                               "peer-store exchanges did not complete —, bit 1 a neighbouring wave of a workgroup); the "
                               "results of this batch are not valid" % (self.rank, status))
        return out

    # ---- peer mode (include/openmg_hip.h: omg_pdist_p2p_*) ----
    p2p_mode = 0

    def p2p_handles(self):
        """bytes: this rank's IPC handles, for the other ranks' p2p_open."""
        n = ctypes.c_int(0)
        check(lib().omg_pdist_p2p_handle_count(self._h, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(64 * n.value)
        check(lib().omg_pdist_p2p_handles(self._h, buf, n.value))
        return bytes(buf.raw)

    def p2p_open(self, peer_rank, handles):
        buf = ctypes.create_string_buffer(bytes(handles), len(handles))
        check(lib().omg_pdist_p2p_open(self._h, int(peer_rank), buf, len(handles) // 64))

    def p2p_local(self, other):
        check(lib().omg_pdist_p2p_local(self._h, other._h))

    def p2p_enable(self, mode):
        check(lib().omg_pdist_p2p_enable(self._h, int(mode)))
        self.p2p_mode = int(mode)

    def p2p_status(self):
        v = ctypes.c_uint(0)
        check(lib().omg_pdist_p2p_status(self._h, ctypes.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "_h", None):
            lib().omg_pdist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PlaneDistGroup:
    """All ranks of a plane-slab decomposition in one process on one GPU (device copies in place of RCCL)."""

    def __init__(self, ranks, p2p=0):
        """p2p 1 / 2: the ranks store into each other's vectors (peer mode; 2: with wait launches) instead of copies."""
        self.ranks = list(ranks)
        if p2p:
            for a in self.ranks:
                for b in self.ranks:
                    if a is not b:
                        a.p2p_local(b)
            for a in self.ranks:
                a.p2p_enable(p2p)
        arr = (ctypes.c_void_p * len(self.ranks))(*[r._h for r in self.ranks])
        g = ctypes.c_void_p()
        check(lib().omg_pdist_group_create(len(self.ranks), arr, ctypes.byref(g)))
        self._g = g

    def cycles(self, n_cycles, pre=1, post=1):
        norms = (ctypes.c_double * max(int(n_cycles), 1))()
        check(lib().omg_pdist_group_cycles_ex(self._g, int(pre), int(post), int(n_cycles), norms))
        for r in self.ranks:
            if r.p2p_status():
                raise RuntimeError("rank %d: a bounded wait inside a pass gave up" % r.rank)
        return [float(norms[k]) for k in range(int(n_cycles))]

    def close(self):
        if getattr(self, "_g", None):
            lib().omg_pdist_group_destroy(self._g)
            self._g = None
        for r in self.ranks:
            r.close()


# ---- 27-point slabs (omg_sdist_*) ----------------------------------------------------------------
class Slab27Rank:
    """One rank's slab of a 27-point hierarchy with per-row coefficients on the octant-layout kernels (omg_sdist).
    shape: the GLOBAL finest grid (planes, lines, cells); A_rows: this rank's rows of the finest operator with global
    columns (dist.stencil27_variable_rows); n_levels distributed levels; the Galerkin products are made on the device.
    After construction: coarse_rows() -> gather over the ranks -> a tail hierarchy -> set_tail()."""

    def __init__(self, rank, n_ranks, shape, A_rows, n_levels, weight=0.125, dtype="float64"):
        nz, ny, nx = (int(s) for s in shape)
        A = as_csr(A_rows)
        h = ctypes.c_void_p()
        view = csr_view(A)
        check(lib().omg_sdist_create(int(rank), int(n_ranks), nx, ny, nz, int(n_levels), ctypes.byref(view), float(weight),
                                     dtype_code(dtype), ctypes.byref(h)))
        self._h = h
        self._tail = None
        self.rank, self.n_ranks, self.n_levels = int(rank), int(n_ranks), int(n_levels)
        self.n_local = nx * ny * (nz // int(n_ranks))

    def coarse_rows(self):
        """This rank's rows of the operator below the slabs, global columns (scipy CSR)."""
        nr, nc, nnz = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib().omg_sdist_coarse_size(self._h, ctypes.byref(nr), ctypes.byref(nc), ctypes.byref(nnz)))
        indptr = np.empty(nr.value + 1, dtype=np.int32)
        indices = np.empty(max(nnz.value, 1), dtype=np.int32)
        data = np.empty(max(nnz.value, 1), dtype=np.float64)
        check(lib().omg_sdist_coarse_fetch(self._h, indptr.ctypes.data, indices.ctypes.data, data.ctypes.data))
        return sp.csr_matrix((data[:nnz.value], indices[:nnz.value], indptr), shape=(nr.value, nc.value))

    def set_tail(self, tail):
        check(lib().omg_sdist_set_tail(self._h, tail._h))
        self._tail = tail

    def connect(self, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(lib().omg_sdist_connect(self._h, buf))

    def rccl_ranks(self):
        n = ctypes.c_int(0)
        check(lib().omg_sdist_rccl_ranks(self._h, ctypes.byref(n)))
        return n.value

    def info(self, level=0):
        out = (ctypes.c_int64 * 8)()
        check(lib().omg_sdist_info(self._h, int(level), out))
        keys = ("nx", "ny", "owned_planes", "aggregates_per_lane", "workgroups", "waves_per_workgroup", "levels", "exchanges_last_call")
        return dict(zip(keys, [int(v) for v in out]))

    def load(self, b_local, x0_local=None):
        b = vec(b_local, self.n_local)
        x0 = None if x0_local is None else vec(x0_local, self.n_local)
        check(lib().omg_sdist_load(self._h, b.ctypes.data, None if x0 is None else x0.ctypes.data))

    def fetch(self):
        x = np.empty(self.n_local, dtype=np.float64)
        check(lib().omg_sdist_fetch(self._h, x.ctypes.data))
        return x

    def sync(self):
        check(lib().omg_sdist_sync(self._h))

    def cycles(self, pre, post, n_cycles):
        norms = (ctypes.c_double * max(int(n_cycles), 1))()
        check(lib().omg_sdist_cycles(self._h, int(pre), int(post), int(n_cycles), norms))
        out = [float(norms[k]) for k in range(int(n_cycles))]
        if self.p2p_mode and self.p2p_status():
            raise RuntimeError("rank %d: a bounded wait for a neighbour's flag gave up (the peer-store exchanges did not "
                               "complete); the results of this batch are not valid" % self.rank)
        return out

    # ---- peer mode for the halo exchanges (include/openmg_hip.h: omg_sdist_p2p_*) ----
    p2p_mode = 0

    def p2p_handles(self):
        """bytes: this rank's IPC handles, for its neighbours' p2p_open."""
        n = ctypes.c_int(0)
        check(lib().omg_sdist_p2p_handle_count(self._h, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(64 * n.value)
        check(lib().omg_sdist_p2p_handles(self._h, buf, n.value))
        return bytes(buf.raw)

    def p2p_open(self, peer_rank, handles):
        buf = ctypes.create_string_buffer(bytes(handles), len(handles))
        check(lib().omg_sdist_p2p_open(self._h, int(peer_rank), buf, len(handles) // 64))

    def p2p_local(self, other):
        check(lib().omg_sdist_p2p_local(self._h, other._h))

    def p2p_enable(self, mode):
        check(lib().omg_sdist_p2p_enable(self._h, int(mode)))
        self.p2p_mode = int(mode)

    def p2p_status(self):
        v = ctypes.c_uint(0)
        check(lib().omg_sdist_p2p_status(self._h, ctypes.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "_h", None):
            lib().omg_sdist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Slab27Group:
    """All ranks of a 27-point slab decomposition in one process on one GPU (device copies in place of RCCL)."""

    def __init__(self, ranks, p2p=0):
        """p2p 1: the ranks store into each other's ghost planes (peer mode) instead of the copies."""
        self.ranks = list(ranks)
        if p2p:
            for a in self.ranks:
                for b in self.ranks:
                    if abs(a.rank - b.rank) == 1:
                        a.p2p_local(b)
            for a in self.ranks:
                a.p2p_enable(p2p)
        arr = (ctypes.c_void_p * len(self.ranks))(*[r._h for r in self.ranks])
        g = ctypes.c_void_p()
        check(lib().omg_sdist_group_create(len(self.ranks), arr, ctypes.byref(g)))
        self._g = g

    def cycles(self, pre, post, n_cycles):
        norms = (ctypes.c_double * max(int(n_cycles), 1))()
        check(lib().omg_sdist_group_cycles(self._g, int(pre), int(post), int(n_cycles), norms))
        for r in self.ranks:
            if r.p2p_mode and r.p2p_status():
                raise RuntimeError("rank %d: a bounded wait for a neighbour's flag gave up" % r.rank)
        return [float(norms[k]) for k in range(int(n_cycles))]

    def close(self):
        if getattr(self, "_g", None):
            lib().omg_sdist_group_destroy(self._g)
            self._g = None
        for r in self.ranks:
            r.close()
