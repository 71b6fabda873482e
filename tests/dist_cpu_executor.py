"""CPU executor of the distributed V-cycle schedule (test infrastructure).

Runs ONE rank of the plan made by openmg_amd.dist with NumPy/SciPy + the oracle's C sweeps
for the local work and caller-supplied communication callbacks, in the same order as
csrc/dist.hip's Runner: exchange after every smoother set and after prolongation, all-gather
before the coarse solve, all-reduce for the norm."""
import numpy as np
import scipy.sparse.linalg as spla

from oracle import mg_oracle as orc


class CpuRank:
    def __init__(self, rank, levels, coarse_global, coarse_counts, smoother, comm, omega=2.0 / 3.0):
        self.rank, self.levels, self.G, self.counts = rank, levels, coarse_global, list(coarse_counts)
        self.smoother, self.comm, self.omega = smoother, comm, omega
        self.x = [np.zeros(lv["A"].shape[1]) for lv in levels]
        self.b = [np.zeros(lv["A"].shape[0]) for lv in levels]
        self.csr = [orc._csr(lv["A"]) for lv in levels]

    def exchange(self, l, group=-1):
        """group >= 0: only the plan entries that carry that colour (like csrc/dist.hip)."""
        lv = self.levels[l]
        n_loc = lv["A"].shape[0]
        sends, recvs = [], []
        tags = lv.get("groups")
        for k, p in enumerate(lv["peers"]):
            if group >= 0 and tags is not None and int(tags[k]) != group:
                continue
            idx = lv["send_idx"][lv["send_off"][k]:lv["send_off"][k + 1]]
            sends.append((int(p), np.ascontiguousarray(self.x[l][idx])))
            recvs.append((int(p), n_loc + int(lv["recv_off"][k]), n_loc + int(lv["recv_off"][k + 1])))
        got = self.comm.sendrecv(sends, [(p, hi - lo) for p, lo, hi in recvs])
        for (p, lo, hi), buf in zip(recvs, got):
            self.x[l][lo:hi] = buf

    def smooth(self, l, its):
        lv = self.levels[l]
        n_loc = lv["A"].shape[0]
        ip, ix, dv = self.csr[l]
        for _ in range(its):
            if lv["keys"] is None:
                new = np.empty(n_loc)
                rc = orc._clib().oracle_jacobi(n_loc, ip, ix, dv, self.b[l], self.x[l], new, self.omega)
                assert rc == 0
                self.x[l][:n_loc] = new
                self.exchange(l)
                continue
            for s in range(lv["n_sets"]):
                order = np.flatnonzero(lv["keys"] == s).astype(np.int32)
                if order.size:
                    rc = orc._clib().oracle_gs_ordered(order.size, ip, ix, dv, self.b[l], self.x[l], order, 1)
                    assert rc == 0
                self.exchange(l, group=s // int(lv.get("set_group", 1)))

    def cycle(self, l, pre, post):
        last = len(self.levels) - 1
        if l >= last:
            full = np.concatenate(self.comm.allgather(self.b[l]))
            lo = sum(self.counts[:self.rank])
            sol = spla.spsolve(self.G.tocsc(), full)
            self.x[l][:self.b[l].size] = sol[lo:lo + self.b[l].size]
            return
        lv = self.levels[l]
        n_loc = lv["A"].shape[0]
        self.smooth(l, pre)
        r = self.b[l] - lv["A"] @ self.x[l]
        self.b[l + 1] = lv["R"] @ r
        self.x[l + 1][:] = 0.0
        self.cycle(l + 1, pre, post)
        nc = lv["R"].shape[0]
        self.x[l][:n_loc] += lv["R"].T @ self.x[l + 1][:nc]
        self.exchange(l)
        if post > 0:
            self.smooth(l, post)

    def run(self, b_loc, n_cycles, pre, post):
        self.b[0] = np.array(b_loc, dtype=np.float64)
        norms = []
        for _ in range(n_cycles):
            self.cycle(0, pre, post)
            lv = self.levels[0]
            r = self.b[0] - lv["A"] @ self.x[0]
            norms.append(float(np.sqrt(self.comm.allreduce_sum(float(r @ r)))))
        return self.x[0][:lv["A"].shape[0]].copy(), norms
