#!/usr/bin/env python3
"""Benchmark of the multigrid V-cycle hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one V-cycle (pre-smooth, residual, restrict, recurse, coarse solve, prolong +
correct, post-smooth, residual norm — openmg/__init__.py:151-236) on a synthetic 3-D 7-point
Poisson problem with b, x and the whole hierarchy already resident in HBM.  The workload at
N = 1 is BASELINE.json configs[2]: 256^3, 5 grids, red-black Gauss-Seidel, fp64, V(1,1).

Prints ONE JSON line (see README of the task): metric V-cycles/s, plus
  roofline      the fine-grid residual kernel r = b - A x (the SpMV-class kernel the metric
                names): algorithmic bytes / average launch time measured with hipEvents on
                the kernel's own stream inside the timed region, against 8 TB/s;
  cpu_baseline  the CPU oracle's V-cycle timed on this box's host (one core: the oracle's C
                sweeps and SciPy's CSR kernels are single-threaded) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def spmv_bytes(n, nnz, w=8):
    """nnz*(w+4) + 4*(n+1) + 2*w*n."""
    return nnz * (w + 4) + 4 * (n + 1) + 2 * w * n


def residual_bytes(n, nnz, w=8):
    return spmv_bytes(n, nnz, w) + w * n


def kernel_name(fmt, mode):
    """Which kernel walks an operator held like `fmt` (csrc/csr_kernels.hip launch_rows_range)."""
    if fmt["rows"] and fmt["pattern_rows"] == fmt["rows"] and os.environ.get("OMG_PATTERN_KERNEL", "1") != "0":
        return "rows_pattern_kernel<%s>" % mode
    return "rows_kernel<%s>" % mode


def build_problem(size, grids, smoother, dtype="float64"):
    import openmg_amd
    from openmg_amd import _hip, operators
    shape = (size, size, size)
    A0 = operators.stencil_poisson(shape)
    u_true = np.random.default_rng(12345).random(A0.shape[0])
    b = A0 @ u_true
    R = operators.restrictionList(shape, grids - 2, 8)          # gridLevels = grids - 1 -> coarsestLevel = grids - 2 (D5)
    A = operators.coeffecientList(A0, R)                        # Galerkin products on the device
    h = _hip.Hierarchy(A, R, smoother=smoother, dtype=dtype)
    meta = {"n": A0.shape[0], "nnz": A0.nnz, "grids": len(A),
            "level_rows": [M.shape[0] for M in A], "level_nnz": [M.nnz for M in A]}
    return h, b, meta


def cpu_baseline(size, grids, cycles):
    """The CPU oracle (oracle/, a 'port' of the reference's algorithm: the reference itself
    is Python 2 and cannot run here) on a bounded sample: `size`^3, same grids, V(1,1)
    red-black.  Returned in 256^3-equivalent V-cycles/s (work per cycle scales with n)."""
    from oracle import mg_oracle as orc
    shape = (size, size, size)
    from openmg_amd import operators
    A0 = operators.stencil_poisson(shape)                       # input generator only
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = orc.restriction_list(shape, grids - 2, 8)
    A = orc.coefficient_list(A0, R)
    orders = [orc.colour_order(orc.parity_colouring(tuple(s // 2 ** l for s in shape))) for l in range(len(A))]
    sm = lambda M, bb, x, its, level: orc.gs_ordered(M, bb, x, orders[level], its)
    p = {"preIterations": 1, "postIterations": 1, "coarsestLevel": len(R)}
    x = None
    x, _ = orc.mg_cycle(A, b, 0, R, p, initial=x, smoother=sm)  # warm-up
    t0 = time.perf_counter()
    for _ in range(cycles):
        x, info = orc.mg_cycle(A, b, 0, R, p, initial=x, smoother=sm)
    dt = time.perf_counter() - t0
    # fine-grid mat-vec on the CPU (SciPy csr_matvec, one core) with the same byte accounting
    xv = np.ones(A0.shape[0])
    A0 @ xv
    t1 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        A0 @ xv
    spmv_s = (time.perf_counter() - t1) / reps
    return cycles / dt, dt, spmv_bytes(A0.shape[0], A0.nnz) / spmv_s / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=256, help="grid extent per axis (default: BASELINE config 3)")
    ap.add_argument("--grids", type=int, default=5)
    ap.add_argument("--smoother", default="colour")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                    help="precision the levels are stored / computed in (f64 = BASELINE configs[2], the default)")
    ap.add_argument("--stencil", default="7pt", choices=["7pt", "27var"],
                    help="multi-GPU leg only: 27var = BASELINE configs[4]'s 27-point variable-coefficient operator")
    ap.add_argument("--graph", type=int, default=0, help="replay the cycle from a hipGraph")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-plain", action="store_true", help="skip the plain-CSR (OMG_COMPRESS=0) comparison leg")
    ap.add_argument("--dist", type=int, default=0, help="force the multi-GPU code path even with one rank (debug)")
    ap.add_argument("--watchdog", type=int, default=900, help="multi-GPU: abort the rank after this many seconds")
    ap.add_argument("--overlap", type=int, default=1,
                    help="multi-GPU: relax boundary rows first and exchange them while the interior rows run")
    ap.add_argument("--dist-grids", type=int, default=4,
                    help="multi-GPU: grids handled across ranks (the last of them and all below run replicated)")
    ap.add_argument("--cpu-size", type=int, default=256, help="CPU baseline leg: grid extent (default: the full workload)")
    ap.add_argument("--cpu-cycles", type=int, default=16)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or args.dist:
        from openmg_amd import dist_bench
        return dist_bench.main(args)

    import torch                                   # device sync / contract plumbing only
    from openmg_amd import _hip
    _hip.require_gpu()
    torch.cuda.set_device(0)

    t_setup = time.perf_counter()
    w = 8 if args.dtype == "f64" else 4
    h, b, meta = build_problem(args.size, args.grids, args.smoother, "float64" if w == 8 else "float32")
    h.resident_load(b)
    setup_s = time.perf_counter() - t_setup
    pre = post = 1
    if args.graph:
        h.use_graph(True)
    for _ in range(args.warmup):
        h.resident_cycle(pre, post, want_norm=False)
    h.sync()
    torch.cuda.synchronize()

    # Timed region.  Only the roofline kernel (fine-grid residual, one launch per cycle) is
    # bracketed by hipEvents here: every event pair opens a ~10 us gap in the stream, so
    # timing all eight level-0 launches would cost the cycle ~4 %.  (A hipGraph replay cannot
    # carry the events; with --graph 1 the kernel is timed in the second region below.)
    in_region = not args.graph
    if in_region:
        h.profile_enable(["residual"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h.resident_cycle(pre, post, want_norm=False)
    h.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed = h.profile_read() if in_region else None
    h.profile_enable(False)
    norm = h.resident_cycle(pre, post, want_norm=True)          # untimed: read the norm back once

    # Second, untimed region: every level-0 kernel class, for the per-kernel table.
    h.use_graph(False)
    h.profile_enable(True)
    for _ in range(args.steps):
        h.resident_cycle(pre, post, want_norm=False)
    prof = h.profile_read()
    h.profile_enable(False)
    if timed is None:
        timed = prof
    # The metric's "fine-grid SpMV GB/s": plain y = A x over the whole level-0 operator as it
    # sits in HBM for the cycle, 20 back-to-back launches in one hipEvent bracket (untimed region).
    spmv_ms = h.spmv_time(20)
    spmv_b = spmv_bytes(meta["n"], meta["nnz"], w)
    fmt_all = h.format_info(0, "A")
    spmv_fmt = fmt_all["format_bytes"] + 2 * w * meta["n"]          # operator in its device format + x read + y written
    fine_spmv = {"kernel": "y = A x, all rows of the fine grid (%s)" % kernel_name(fmt_all, "ROW_SPMV"),
                 "avg_launch_us": round(spmv_ms * 1e3, 2),
                 "algorithmic_bytes": spmv_b, "GBps": round(spmv_b / spmv_ms / 1e6, 1),
                 "frac_of_peak": round(spmv_b / spmv_ms / 1e6 / HBM_PEAK_GBS, 4),
                 "format_bytes": spmv_fmt, "format_GBps": round(spmv_fmt / spmv_ms / 1e6, 1)}

    n, nnz = meta["n"], meta["nnz"]
    launches, ms = timed["residual"]
    avg_s = (ms / launches) * 1e-3
    # Rows the fine-grid residual launch covers.  With a Gauss-Seidel ordering the last set's
    # residual comes out of the smoother launch that relaxed it (bit-identical),
    # so the launch visits the other sets only: the red half for red-black.
    n_sets = h.level_sets(0)
    covered = range(n_sets - 1) if h.level_fused(0) else range(n_sets)
    rows_c = sum(h.set_info(0, s)[0] for s in covered)
    nnz_c = sum(h.set_info(0, s)[1] for s in covered)
    # algorithmic bytes (w = value width): entries w+4 B, row pointers 4 B, b and r w B per
    # covered row, and the whole of x once (the covered rows together reference every unknown)
    res_bytes = (w + 4) * nnz_c + 4 * (rows_c + 1) + 2 * w * rows_c + w * n
    achieved = res_bytes / avg_s / 1e9
    # ... and the bytes the launch has to move with the operator in its DEVICE format (lossless
    # block-dictionary recoding of the CSR, DESIGN.md "Device format"): same vectors, fewer
    # operator bytes.  `achieved` above is the CSR-equivalent rate SURVEY 8(d) defines; the
    # rate at which HBM is actually driven is format_bytes (or the PMC traffic) over the time.
    fmt_sets = [h.format_info(0, "A", s) for s in covered]
    fmt_bytes = sum(f["format_bytes"] for f in fmt_sets) + 2 * w * rows_c + w * n
    fmt_cov = {k: sum(f[k] for f in fmt_sets) for k in ("rows", "nnz", "blocks", "pattern_rows", "coldict_nnz", "valdict_nnz")}
    # HBM traffic of that launch from the PMC counters cannot be collected from inside this
    # process; it is taken from the committed rocprofv3 pass of the SAME kernel and problem
    # (profiles/, method recorded there) and only when the byte accounting matches exactly.
    traffic, traffic_src = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_residual.json")))
        if int(pmc["algorithmic_bytes"]) == int(res_bytes) and int(pmc.get("format_bytes", -1)) == int(fmt_bytes) and w == 8:
            traffic, traffic_src = pmc["traffic_bytes"], pmc["source"]
    except (OSError, KeyError, ValueError):
        pass
    # Per-class table of the level-0 launches (second region): average time and algorithmic
    # bytes of ONE launch -> GB/s.  Smoother launches are per set; with the fused last set two
    # of the 2*n_sets launches per cycle also write the residual (8 B per covered row more).
    n_c = meta["level_rows"][1] if len(meta["level_rows"]) > 1 else 0
    set_rows = [h.set_info(0, s)[0] for s in range(n_sets)]
    set_nnz = [h.set_info(0, s)[1] for s in range(n_sets)]
    avg_set_bytes = sum((w + 4) * z + 4 * (r + 1) + 3 * w * r for r, z in zip(set_rows, set_nnz)) / max(n_sets, 1) + w * n / max(n_sets, 1)
    class_bytes = {
        "smoother_set_sweep": avg_set_bytes,
        "residual": res_bytes,
        "restrict": (w + 4) * n + 4 * (n_c + 1) + w * n + 2 * w * n_c,    # R entries, indptr, r read, b_c + cleared x_c written
        "prolong_add": (w + 4) * n + 4 * (n + 1) + w * n_c + 2 * w * n,   # P entries, indptr, e read, x read + written
        "residual_norm": res_bytes - w * rows_c,                          # as the residual launch, nothing stored
    }
    # the same launches with the operators in their device format
    fA = [h.format_info(0, "A", s) for s in range(n_sets)]
    fR, fP = (h.format_info(0, "R"), h.format_info(0, "P")) if n_c else ({"format_bytes": 0}, {"format_bytes": 0})
    scatter = bool(n_c) and h.level_flags(0)["scatter_prolong"]
    class_fmt = {
        "smoother_set_sweep": sum(f["format_bytes"] + 3 * w * f["rows"] for f in fA) / max(n_sets, 1) + w * n / max(n_sets, 1),
        "residual": fmt_bytes,
        "restrict": fR["format_bytes"] + w * n + 2 * w * n_c,
        # prolongation: a pass over P = R^T, or (aggregation R, row-pattern coded) a scatter over R's rows
        "prolong_add": (fR["format_bytes"] if scatter else fP["format_bytes"]) + w * n_c + 2 * w * n,
        "residual_norm": fmt_bytes - w * rows_c,
    }
    kernels = {}
    for name, (cnt, tot) in prof.items():
        if cnt:
            us = 1e3 * tot / cnt
            kernels[name] = {"launches_per_cycle": cnt / args.steps, "avg_us": round(us, 2),
                             "algorithmic_bytes": int(class_bytes[name]),
                             "GBps": round(class_bytes[name] / us / 1e3, 1),
                             "format_bytes": int(class_fmt[name]),
                             "format_GBps": round(class_fmt[name] / us / 1e3, 1)}
    roofline = {"bound": "hbm", "kernel": "fine grid r = b - A x (%s)" % kernel_name(fmt_cov, "ROW_RESIDUAL"),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes": res_bytes, "avg_launch_us": round(avg_s * 1e6, 2),
                "format_bytes": fmt_bytes, "format_GBps": round(fmt_bytes / avg_s / 1e9, 1),
                "format_frac": round(fmt_bytes / avg_s / 1e9 / HBM_PEAK_GBS, 4),
                "device_format": {"row_pattern_rows": fmt_cov["pattern_rows"], "rows": fmt_cov["rows"],
                                  "column_coded_nnz": fmt_cov["coldict_nnz"], "value_coded_nnz": fmt_cov["valdict_nnz"],
                                  "nnz": fmt_cov["nnz"], "OMG_COMPRESS": os.environ.get("OMG_COMPRESS", "15 (default)")},
                "rows_covered": rows_c, "nnz_covered": nnz_c, "fused_last_set": h.level_fused(0),
                "level0_kernels": kernels}

    # The same launch with the operator as PLAIN int32 CSR (OMG_COMPRESS=0): the figure the
    # north-star target (">= 50 % of the HBM roofline on the fine-grid SpMV") is about.  Second
    # hierarchy, untimed region, same hipEvent bracketing of the residual launch.
    plain = None
    if not args.no_plain and os.environ.get("OMG_COMPRESS", "15") != "0":
        h.close()
        keep = os.environ.get("OMG_COMPRESS")
        os.environ["OMG_COMPRESS"] = "0"
        try:
            h2, b2, _ = build_problem(args.size, args.grids, args.smoother, "float64" if w == 8 else "float32")
        finally:
            if keep is None:
                del os.environ["OMG_COMPRESS"]
            else:
                os.environ["OMG_COMPRESS"] = keep
        h2.resident_load(b2)
        for _ in range(args.warmup):
            h2.resident_cycle(pre, post, want_norm=False)
        h2.profile_enable(["residual"])
        t1 = time.perf_counter()
        for _ in range(args.steps):
            h2.resident_cycle(pre, post, want_norm=False)
        h2.sync()
        plain_elapsed = time.perf_counter() - t1
        pl, pms = h2.profile_read()["residual"]
        h2.profile_enable(False)
        p_spmv_ms = h2.spmv_time(20)
        p_us = 1e3 * pms / pl
        plain = {"what": "operator held as plain int32 CSR (OMG_COMPRESS=0), same problem, untimed region",
                 "residual_kernel": "rows_kernel<ROW_RESIDUAL>", "avg_launch_us": round(p_us, 2),
                 "achieved": round(res_bytes / p_us / 1e3, 1), "frac": round(res_bytes / p_us / 1e3 / HBM_PEAK_GBS, 4),
                 "fine_grid_spmv_us": round(p_spmv_ms * 1e3, 2), "fine_grid_spmv_GBps": round(spmv_b / p_spmv_ms / 1e6, 1),
                 "fine_grid_spmv_frac": round(spmv_b / p_spmv_ms / 1e6 / HBM_PEAK_GBS, 4),
                 "vcycles_per_s": round(args.steps / plain_elapsed, 1)}
        h2.close()
    roofline["plain_csr"] = plain
    if fmt_bytes < res_bytes:
        roofline["note"] = ("frac counts SURVEY 8(d)'s CSR bytes; the operator sits in HBM in a lossless row-pattern / "
                            "dictionary coding (DESIGN.md 'Device format'), so the launch moves format_bytes, HBM is "
                            "driven at format_frac of peak, and frac may exceed 1.  plain_csr is the same launch on "
                            "plain int32 CSR.")

    cpu = None
    if not args.no_cpu:
        rate, dt, cpu_spmv = cpu_baseline(args.cpu_size, args.grids, args.cpu_cycles)
        scale = (args.cpu_size / float(args.size)) ** 3
        cpu = {"value": round(rate * scale, 5), "unit": "V-cycles/s", "cores": 1, "kind": "port",
               "sample": "%d V(1,1) red-black cycles of the CPU oracle on %d^3 (%d grids) in %.1f s%s; "
                         "host has %d cores, the oracle uses 1"
                         % (args.cpu_cycles, args.cpu_size, args.grids, dt,
                            "" if args.cpu_size == args.size else
                            ", scaled by (%d/%d)^3 to %d^3" % (args.cpu_size, args.size, args.size),
                            os.cpu_count() or 0),
               "fine_grid_spmv_GBps": round(cpu_spmv, 2)}

    out = {
        "metric": "V-cycles/sec, 3-D 7-point Poisson %d^3" % args.size,
        "value": round(args.steps / elapsed, 3),
        "unit": "V-cycles/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": "3-D 7-point Poisson %d^3, %d-grid V(1,1) cycle, %s Gauss-Seidel, %s, "
                               "int32 CSR (BASELINE configs[2]%s)" % (args.size, meta["grids"],
                                                                      "red-black" if args.smoother == "colour" else args.smoother,
                                                                      "fp64" if w == 8 else "fp32",
                                                                      "" if w == 8 else " run in fp32: NOT the headline configuration"),
                   "unknowns": n, "nnz": nnz, "grids": meta["grids"], "pre": pre, "post": post,
                   "smoother": args.smoother, "hipgraph": bool(args.graph),
                   "final_residual_norm": norm, "setup_s": round(setup_s, 2)},
        "roofline": roofline,
        "fine_grid_spmv": fine_spmv,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    h.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
