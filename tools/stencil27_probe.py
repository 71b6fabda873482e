#!/usr/bin/env python3
"""Kernel rates on a 27-point operator (rows of 27 entries: 75 rows per 2048-entry block)."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, operators  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    T = sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="csr")
    A0 = sp.csr_matrix(-sp.kron(sp.kron(T, T), T) + sp.diags(np.full(n ** 3, 28.0)))
    A0.sort_indices()
    shape = (n, n, n)
    R = operators.restrictionList(shape, 3, 8)
    A = operators.coeffecientList(A0, R)
    b = A0 @ np.random.default_rng(1).random(A0.shape[0])
    for smoother in ("colour", "jacobi"):
        h = _hip.Hierarchy(A, R, smoother=smoother, omega=0.6)
        h.resident_load(b)
        for _ in range(3):
            h.resident_cycle(1, 1, want_norm=False)
        h.sync()
        h.profile_enable(True)
        for _ in range(10):
            h.resident_cycle(1, 1, want_norm=False)
        prof = h.profile_read()
        nn, nnz = A0.shape[0], A0.nnz
        full = 12 * nnz + 4 * (nn + 1) + 24 * nn
        print(smoother, "sets", h.level_sets(0), "nnz/row %.1f" % (nnz / nn))
        for name, (cnt, ms) in prof.items():
            if cnt:
                us = 1e3 * ms / cnt
                per = {"smoother_set_sweep": full / h.level_sets(0), "residual": full if smoother == "jacobi" else full * (1 - 1 / h.level_sets(0)),
                       "residual_norm": full if smoother == "jacobi" else full * (1 - 1 / h.level_sets(0))}.get(name)
                print("   %-20s %8.1f us %s" % (name, us, ("%.0f GB/s" % (per / us / 1e3)) if per else ""))
        h.close()


if __name__ == "__main__":
    main()
