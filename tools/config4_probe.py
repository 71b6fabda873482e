#!/usr/bin/env python3
"""BASELINE.json configs[4] on ONE GPU: 27-point variable-coefficient operator, fp32 levels,
Galerkin products on the device.  (The config itself is 512^3 on 8 GPUs = this workload per
GPU; the multi-GPU runner is fp64-only so far)

    python tools/config4_probe.py --size 256 --dtype float32
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, operators  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--grids", type=int, default=5)
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cache", default="", help="directory to keep the generated operator in between runs (A/B builds)")
    args = ap.parse_args()
    shape = (args.size,) * 3
    w = 4 if args.dtype == "float32" else 8
    t = time.perf_counter()
    A0 = None
    if args.cache:
        import scipy.sparse as sp
        base = os.path.join(args.cache, "cfg4_%d" % args.size)
        if os.path.exists(base + "_data.npy"):
            A0 = sp.csr_matrix((np.load(base + "_data.npy"), np.load(base + "_indices.npy"), np.load(base + "_indptr.npy")),
                               shape=(args.size ** 3,) * 2)
    if A0 is None:
        A0 = operators.stencil27_variable(shape)              # Q1 stiffness, kappa per cell, default_rng(2024)
        if args.cache:
            os.makedirs(args.cache, exist_ok=True)
            np.save(base + "_data.npy", A0.data); np.save(base + "_indices.npy", A0.indices); np.save(base + "_indptr.npy", A0.indptr)
    t_gen = time.perf_counter() - t
    b = A0 @ np.random.default_rng(2).random(A0.shape[0])
    t = time.perf_counter()
    R = operators.restrictionList(shape, args.grids - 2, 8)
    A = operators.coeffecientList(A0, R)                       # device SpGEMM (omg_rap), fp64, SciPy's summation order
    t_rap = time.perf_counter() - t
    t = time.perf_counter()
    h = _hip.Hierarchy(A, R, smoother="colour", dtype=args.dtype)
    h.resident_load(b)
    t_up = time.perf_counter() - t
    flags = h.level_flags(0)
    norms = [h.resident_cycle(1, 1) for _ in range(3)]
    h.sync()
    t = time.perf_counter()
    for _ in range(args.steps):
        h.resident_cycle(1, 1, want_norm=False)
    h.sync()
    dt_nonorm = (time.perf_counter() - t) / args.steps
    h.resident_cycles(1, 1, 3)
    t = time.perf_counter()
    h.resident_cycles(1, 1, args.steps)              # every cycle's norm computed (as bench.py times it)
    dt = (time.perf_counter() - t) / args.steps
    h.profile_enable(True)
    for _ in range(args.steps):
        h.resident_cycle(1, 1, want_norm=False)
    prof = h.profile_read()
    h.profile_enable(False)
    n, nnz = A0.shape[0], A0.nnz
    n_sets = h.level_sets(0)
    if flags["stencil27"]:
        fmt = {"format_bytes": 27 * w * n, "kernels": "stencil27.hip (octant layout, 27 coefficients per row)"}
    else:
        fmt = h.format_info(0, "A")
    out = {"workload": "27-point variable-coefficient, %d^3, %d grids, V(1,1) %d-colour Gauss-Seidel, %s" % (args.size, len(A), n_sets, args.dtype),
           "unknowns": n, "nnz": nnz, "level_rows": [M.shape[0] for M in A], "level_nnz": [M.nnz for M in A],
           "generator_s": round(t_gen, 2), "restriction_and_device_rap_s": round(t_rap, 2), "upload_s": round(t_up, 2),
           "stencil27_kernels": flags["stencil27"],
           "ms_per_cycle": round(1e3 * dt, 4), "vcycles_per_s": round(1 / dt, 2),
           "ms_per_cycle_without_norms": round(1e3 * dt_nonorm, 4), "first_norms": norms,
           "format": fmt, "spmv_GBps_csr_equivalent": None, "kernels": {}}
    if not flags["stencil27"]:
        spmv_ms = h.spmv_time(10)
        out["spmv_us"] = round(1e3 * spmv_ms, 2)
        out["spmv_GBps_csr_equivalent"] = round(((w + 4) * nnz + 4 * (n + 1) + 2 * w * n) / spmv_ms / 1e6, 1)
        out["spmv_GBps_format"] = round((fmt["format_bytes"] + 2 * w * n) / spmv_ms / 1e6, 1)
    for name, (cnt, ms) in prof.items():
        if cnt:
            out["kernels"][name] = {"launches_per_cycle": cnt / args.steps, "avg_us": round(1e3 * ms / cnt, 2)}
    print(json.dumps(out))
    h.close()


if __name__ == "__main__":
    main()
