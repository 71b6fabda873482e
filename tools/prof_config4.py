#!/usr/bin/env python3
"""Torch-free driver for rocprofv3: BASELINE configs[4]'s per-GPU workload (27-point variable coefficient, fp32), K cycles.

    rocprofv3 --kernel-trace ... -- python3 tools/prof_config4.py --size 256 --steps 6 [--batched 1]
"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, operators  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--grids", type=int, default=5)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batched", type=int, default=1)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--cache", default="/tmp/cfg4")
    args = ap.parse_args()
    shape = (args.size,) * 3
    base = os.path.join(args.cache, "cfg4_%d" % args.size)
    if os.path.exists(base + "_data.npy"):
        A0 = sp.csr_matrix((np.load(base + "_data.npy"), np.load(base + "_indices.npy"), np.load(base + "_indptr.npy")), shape=(args.size ** 3,) * 2)
    else:
        A0 = operators.stencil27_variable(shape)
        os.makedirs(args.cache, exist_ok=True)
        np.save(base + "_data.npy", A0.data); np.save(base + "_indices.npy", A0.indices); np.save(base + "_indptr.npy", A0.indptr)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, args.grids - 2, 8)
    A = operators.coeffecientList(A0, R)
    h = _hip.Hierarchy(A, R, smoother="colour", dtype=args.dtype)
    h.resident_load(b)
    h.resident_cycles(1, 1, 2)
    h.sync()
    if args.batched:
        norms = h.resident_cycles(1, 1, args.steps)
    else:
        norms = [h.resident_cycle(1, 1) for _ in range(args.steps)]
    h.sync()
    print("norms", norms[-3:])
    h.close()


if __name__ == "__main__":
    main()
