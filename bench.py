#!/usr/bin/env python3
"""Benchmark of the multigrid V-cycle hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one V-cycle (pre-smooth, residual, restrict, recurse, coarse solve, prolong +
correct, post-smooth, residual norm — openmg/__init__.py:151-236) on a synthetic 3-D 7-point
Poisson problem with b, x and the whole hierarchy already resident in HBM.  The workload at
N = 1 is BASELINE.json configs[2]: 256^3, 5 grids, red-black Gauss-Seidel, fp64, V(1,1).

N > 1: started by torch.distributed.run (RANK / WORLD_SIZE in the environment) every process is
one rank; started bare (`python bench.py --gpus N`) the parent launches the N rank processes
itself (openmg_amd/launch.py) before it touches the GPU, relays rank 0's JSON line and returns
the worst child exit code.

Prints ONE JSON line: metric V-cycles/s (median of --repeats timed regions of K cycles), plus
  roofline      the fine-grid residual kernel r = b - A x (the SpMV-class kernel of the metric):
                bytes the launch has to move, with the operator in the format it has in HBM,
                / average launch time measured with hipEvents on the kernel's own stream inside
                the timed regions, against 8 TB/s (frac <= 1 by construction); the rate in
                SURVEY 8(d)'s plain-CSR bytes is reported beside it as csr_equiv_GBps;
  csr_path      the same problem with every operator held as plain int32 CSR (OMG_COMPRESS=0),
                timed with the same loop: V-cycles/s, residual and fine-grid SpMV launches against
                SURVEY 8(d)'s CSR byte counts;
  reference_smoother  the same problem with the reference's own lexicographic Gauss-Seidel as the
                smoother (the headline uses the red-black ordering BASELINE configs[2] names);
  cpu_baseline  the CPU oracle's V-cycle timed on this box's host (one core: the oracle's C
                sweeps and SciPy's CSR kernels are single-threaded) on a bounded sample.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def spmv_bytes(n, nnz, w=8):
    """nnz*(w+4) + 4*(n+1) + 2*w*n."""
    return nnz * (w + 4) + 4 * (n + 1) + 2 * w * n


def kernel_source_hash():
    """Identity of the kernel sources a PMC profile was taken with (profiles/*.json carry it)."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "openmg_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()[:16]


def git_head():
    """Commit of the sources this line was measured on: OMG_GIT_HEAD (the GPU boxes have no .git) or git itself."""
    head = os.environ.get("OMG_GIT_HEAD")
    if head:
        return head
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        return None


def kernel_name(fmt, mode, union=False):
    """Which kernel walks an operator held like `fmt` (csrc/csr_kernels.hip launch_rows_range)."""
    if union:
        return "rows_union_kernel<%s>" % mode
    if fmt["rows"] and fmt["pattern_rows"] == fmt["rows"] and os.environ.get("OMG_PATTERN_KERNEL", "1") != "0":
        return "rows_pattern_kernel<%s>" % mode
    return "rows_kernel<%s>" % mode


def build_problem(size, grids, smoother, dtype="float64"):
    import numpy as np
    from openmg_amd import _hip, operators
    shape = (size, size, size)
    t0 = time.perf_counter()
    A0 = operators.stencil_poisson(shape)                       # synthetic input (NumPy, host)
    u_true = np.random.default_rng(12345).random(A0.shape[0])
    b = A0 @ u_true
    _hip.device_count()                                         # loads the library
    t1 = time.perf_counter()
    # what mgSolve does before its first cycle (openmg/__init__.py:103-109): R list, Galerkin
    # products (on the device), then the device hierarchy (ordering, coding, upload, coarse factors)
    R = operators.restrictionList(shape, grids - 2, 8)          # gridLevels = grids - 1 -> coarsestLevel = grids - 2 (D5)
    A = operators.coeffecientList(A0, R)
    h = _hip.Hierarchy(A, R, smoother=smoother, dtype=dtype)
    t2 = time.perf_counter()
    meta = {"n": A0.shape[0], "nnz": A0.nnz, "grids": len(A),
            "level_rows": [M.shape[0] for M in A], "level_nnz": [M.nnz for M in A],
            "generate_s": t1 - t0, "setup_s": t2 - t1}
    return h, b, meta


def cpu_baseline(size, grids, cycles):
    """The CPU oracle (oracle/, a 'port' of the reference's algorithm: the reference itself
    is Python 2 and cannot run here) on a bounded sample: `size`^3, same grids, V(1,1)
    red-black.  Returned in 256^3-equivalent V-cycles/s (work per cycle scales with n)."""
    import numpy as np
    from oracle import mg_oracle as orc
    from openmg_amd import operators
    shape = (size, size, size)
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


def timed_regions(h, sync, steps, warmup, repeats, pre, post, classes=("residual",)):
    """`warmup` untimed cycles, then `repeats` timed regions of EXACTLY `steps` cycles each,
    every one bracketed by a device synchronisation on both sides.  Inside the regions the
    launches of `classes` are bracketed by hipEvents on the hierarchy's own stream (one pair per
    cycle for 'residual': ~1 % of the cycle).  Returns (elapsed seconds per region, profile, the
    last region's norms)."""
    h.resident_cycles(pre, post, warmup)
    sync()
    if classes:
        h.profile_enable(list(classes))
    times, norms = [], []
    for _ in range(repeats):
        sync()
        t0 = time.perf_counter()
        # K cycles enqueued back to back, every cycle's residual norm computed and returned at the end
        # of the region (omg_resident_cycles: mgSolve's loop with a cycle-count stop rule)
        norms = h.resident_cycles(pre, post, steps)
        sync()
        times.append(time.perf_counter() - t0)
    prof = h.profile_read() if classes else None
    h.profile_enable(False)
    return times, prof, norms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps cycles each; value = steps / median region time (BASELINE.md §3)")
    ap.add_argument("--size", type=int, default=256, help="grid extent per axis (default: BASELINE config 3)")
    ap.add_argument("--grids", type=int, default=5)
    ap.add_argument("--smoother", default="colour")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                    help="precision the levels are stored / computed in (f64 = BASELINE configs[2], the default)")
    ap.add_argument("--stencil", default="7pt", choices=["7pt", "27var"],
                    help="multi-GPU leg only: 27var = BASELINE configs[4]'s 27-point variable-coefficient operator")
    ap.add_argument("--graph", type=int, default=0, help="replay the cycle from a hipGraph")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-plain", action="store_true", help="skip the plain-CSR (OMG_COMPRESS=0) leg")
    ap.add_argument("--no-lex", action="store_true", help="skip the leg with the reference's lexicographic Gauss-Seidel")
    ap.add_argument("--dist", type=int, default=0, help="force the multi-GPU code path even with one rank (debug)")
    ap.add_argument("--watchdog", type=int, default=900, help="multi-GPU: abort after this many seconds")
    ap.add_argument("--overlap", type=int, default=1,
                    help="multi-GPU: relax boundary rows first and exchange them while the interior rows run")
    ap.add_argument("--dist-grids", type=int, default=4,
                    help="multi-GPU: grids handled across ranks (the last of them and all below run replicated)")
    ap.add_argument("--cpu-size", type=int, default=256, help="CPU baseline leg: grid extent (default: the full workload)")
    ap.add_argument("--cpu-cycles", type=int, default=16)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher.  Nothing GPU-related has been
        # imported yet (launch.py imports neither torch nor the HIP library).
        from openmg_amd import launch
        return launch.spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                  timeout_s=args.watchdog + 60)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or args.dist:
        from openmg_amd import dist_bench
        return dist_bench.main(args)

    import torch                                   # device sync / contract plumbing only
    from openmg_amd import _hip
    _hip.require_gpu()
    torch.cuda.set_device(0)
    repeats = max(1, args.repeats)

    w = 8 if args.dtype == "f64" else 4
    np_dtype = "float64" if w == 8 else "float32"
    h, b, meta = build_problem(args.size, args.grids, args.smoother, np_dtype)
    h.resident_load(b)
    setup_s, generate_s = meta["setup_s"], meta["generate_s"]
    pre = post = 1
    if args.graph:
        h.use_graph(True)

    def sync():
        h.sync()
        torch.cuda.synchronize()

    # Timed regions.  Only the roofline kernel (fine-grid residual, one launch per cycle) is
    # bracketed by hipEvents there: every event pair opens a gap in the stream, so timing all
    # eight level-0 launches would cost the cycle ~4 %.  (A hipGraph replay cannot carry the
    # events; with --graph 1 the kernel is timed in the second region below.)
    in_region = not args.graph
    times, timed, region_norms = timed_regions(h, sync, args.steps, args.warmup, repeats, pre, post,
                                               ("residual",) if in_region else ())
    elapsed = statistics.median(times)
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
    n, nnz = meta["n"], meta["nnz"]
    spmv_csr = spmv_bytes(n, nnz, w)
    fmt_all = h.format_info(0, "A")
    spmv_fmt = fmt_all["format_bytes"] + 2 * w * n              # operator in its device format + x read + y written
    fine_spmv = {"kernel": "y = A x, all rows of the fine grid (%s)" % kernel_name(fmt_all, "ROW_SPMV", h.level_flags(0)["union_walk"]),
                 "avg_launch_us": round(spmv_ms * 1e3, 2),
                 "bytes_per_launch": spmv_fmt, "achieved": round(spmv_fmt / spmv_ms / 1e6, 1),
                 "frac": round(spmv_fmt / spmv_ms / 1e6 / HBM_PEAK_GBS, 4),
                 "csr_equiv_bytes": spmv_csr, "csr_equiv_GBps": round(spmv_csr / spmv_ms / 1e6, 1)}

    launches, ms = timed["residual"]
    avg_s = (ms / launches) * 1e-3
    # Rows the fine-grid residual launch covers.  With a Gauss-Seidel ordering the last set's
    # residual comes out of the smoother launch that relaxed it (bit-identical),
    # so the launch visits the other sets only: the red half for red-black.
    n_sets = h.level_sets(0)
    covered = range(n_sets - 1) if h.level_fused(0) else range(n_sets)
    rows_c = sum(h.set_info(0, s)[0] for s in covered)
    nnz_c = sum(h.set_info(0, s)[1] for s in covered)
    # SURVEY 8(d)'s bytes of that launch on plain CSR (w = value width): entries w+4 B, row
    # pointers 4 B, b and r w B per covered row, the whole of x once
    res_csr = (w + 4) * nnz_c + 4 * (rows_c + 1) + 2 * w * rows_c + w * n
    # ... and the bytes the launch HAS TO MOVE with the operator in its device format (lossless
    # block recoding): same vectors, the operator as it is stored.  The roofline
    # fraction is taken on these: a launch cannot be credited with bytes it does not read.
    fmt_sets = [h.format_info(0, "A", s) for s in covered]
    res_fmt = sum(f["format_bytes"] for f in fmt_sets) + 2 * w * rows_c + w * n
    fmt_cov = {k: sum(f[k] for f in fmt_sets) for k in ("rows", "nnz", "blocks", "pattern_rows", "coldict_nnz", "valdict_nnz")}
    # HBM traffic from the PMC counters cannot be collected from inside this process.  It is taken
    # from a committed rocprofv3 pass of the same launch ONLY when that pass was made with the
    # kernel sources of this build (hash recorded in the profile) on the same problem.
    traffic, traffic_src = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_residual.json")))
        if pmc.get("kernel_src_sha") == kernel_source_hash() and int(pmc["bytes_per_launch"]) == int(res_fmt) and w == 8:
            traffic = pmc["traffic_bytes"]
            traffic_src = "NOT measured in this run: rocprofv3 --pmc passes of the same build, " + pmc["source"]
    except (OSError, KeyError, ValueError):
        pass
    # Per-class table of the level-0 launches (second region): average time, bytes one launch has
    # to move in the device format, and the CSR-equivalent rate.
    n_c = meta["level_rows"][1] if len(meta["level_rows"]) > 1 else 0
    set_rows = [h.set_info(0, s)[0] for s in range(n_sets)]
    set_nnz = [h.set_info(0, s)[1] for s in range(n_sets)]
    class_csr = {
        "smoother_set_sweep": sum((w + 4) * z + 4 * (r + 1) + 3 * w * r for r, z in zip(set_rows, set_nnz)) / max(n_sets, 1) + w * n / max(n_sets, 1),
        "residual": res_csr,
        "restrict": (w + 4) * n + 4 * (n_c + 1) + w * n + 2 * w * n_c,    # R entries, indptr, r read, b_c + cleared x_c written
        "prolong_add": (w + 4) * n + 4 * (n + 1) + w * n_c + 2 * w * n,   # P entries, indptr, e read, x read + written
        "residual_norm": res_csr - w * rows_c,                            # as the residual launch, nothing stored
    }
    fA = [h.format_info(0, "A", s) for s in range(n_sets)]
    fR, fP = (h.format_info(0, "R"), h.format_info(0, "P")) if n_c else ({"format_bytes": 0}, {"format_bytes": 0})
    scatter = bool(n_c) and h.level_flags(0)["scatter_prolong"]
    class_fmt = {
        "smoother_set_sweep": sum(f["format_bytes"] + 3 * w * f["rows"] for f in fA) / max(n_sets, 1) + w * n / max(n_sets, 1),
        "residual": res_fmt,
        "restrict": fR["format_bytes"] + w * n + 2 * w * n_c,
        # prolongation: a pass over P = R^T, or (aggregation R, row-pattern coded) a scatter over R's rows
        "prolong_add": (fR["format_bytes"] if scatter else fP["format_bytes"]) + w * n_c + 2 * w * n,
        "residual_norm": res_fmt - w * rows_c,
    }
    kernels = {}
    for name, (cnt, tot) in prof.items():
        if cnt:
            us = 1e3 * tot / cnt
            kernels[name] = {"launches_per_cycle": cnt / args.steps, "avg_us": round(us, 2),
                             "bytes_per_launch": int(class_fmt[name]),
                             "achieved": round(class_fmt[name] / us / 1e3, 1),
                             "frac": round(class_fmt[name] / us / 1e3 / HBM_PEAK_GBS, 4),
                             "csr_equiv_GBps": round(class_csr[name] / us / 1e3, 1)}
    achieved = res_fmt / avg_s / 1e9
    roofline = {"bound": "hbm", "kernel": "fine grid r = b - A x (%s)" % kernel_name(fmt_cov, "ROW_RESIDUAL", h.level_flags(0)["union_walk"]),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "bytes_per_launch": res_fmt,
                "bytes_definition": "operator bytes as stored in HBM (device format, DESIGN.md §4) + b and r of the covered rows + x once",
                "avg_launch_us": round(avg_s * 1e6, 2), "launches_timed": launches,
                "csr_equiv_bytes": res_csr, "csr_equiv_GBps": round(res_csr / avg_s / 1e9, 1),
                "device_format": {"row_pattern_rows": fmt_cov["pattern_rows"], "rows": fmt_cov["rows"],
                                  "column_coded_nnz": fmt_cov["coldict_nnz"], "value_coded_nnz": fmt_cov["valdict_nnz"],
                                  "nnz": fmt_cov["nnz"], "OMG_COMPRESS": os.environ.get("OMG_COMPRESS", "15 (default)")},
                "rows_covered": rows_c, "nnz_covered": nnz_c, "fused_last_set": h.level_fused(0),
                "level0_kernels": kernels}

    # The same problem with every operator as PLAIN int32 CSR (OMG_COMPRESS=0) — what the north
    # star's "CSR SpMV ... >= 50 % of the HBM roofline on the fine-grid SpMV" describes — timed
    # with the same loop (same warm-up, steps, repeats, event bracketing).
    csr_path = None
    if not args.no_plain and os.environ.get("OMG_COMPRESS", "15") != "0":
        h.close()
        keep = os.environ.get("OMG_COMPRESS")
        os.environ["OMG_COMPRESS"] = "0"
        try:
            h2, b2, _ = build_problem(args.size, args.grids, args.smoother, np_dtype)
        finally:
            if keep is None:
                del os.environ["OMG_COMPRESS"]
            else:
                os.environ["OMG_COMPRESS"] = keep
        h2.resident_load(b2)

        def sync2():
            h2.sync()
            torch.cuda.synchronize()

        p_times, p_prof, _ = timed_regions(h2, sync2, args.steps, args.warmup, repeats, pre, post, ("residual",))
        p_elapsed = statistics.median(p_times)
        pl, pms = p_prof["residual"]
        p_spmv_ms = h2.spmv_time(20)
        p_us = 1e3 * pms / pl
        csr_path = {"what": "same problem and timed loop, every operator held as plain int32 CSR (OMG_COMPRESS=0)",
                    "vcycles_per_s": round(args.steps / p_elapsed, 3), "ms_per_step": round(1e3 * p_elapsed / args.steps, 4),
                    "ms_per_step_all": [round(1e3 * t / args.steps, 4) for t in p_times],
                    "residual": {"kernel": "rows_kernel<ROW_RESIDUAL>", "avg_launch_us": round(p_us, 2),
                                 "bytes_per_launch": res_csr, "achieved": round(res_csr / p_us / 1e3, 1),
                                 "frac": round(res_csr / p_us / 1e3 / HBM_PEAK_GBS, 4)},
                    "fine_grid_spmv": {"kernel": "rows_kernel<ROW_SPMV>", "avg_launch_us": round(p_spmv_ms * 1e3, 2),
                                       "bytes_per_launch": spmv_csr, "achieved": round(spmv_csr / p_spmv_ms / 1e6, 1),
                                       "frac": round(spmv_csr / p_spmv_ms / 1e6 / HBM_PEAK_GBS, 4)}}
        h2.close()

    # The same problem with the reference's OWN smoother — in-place lexicographic Gauss-Seidel
    # (openmg/solvers.py:56-68) — instead of the red-black ordering: same loop, fewer repeats.
    lex_path = None
    if not args.no_lex and args.smoother == "colour":
        h3, b3, meta3 = build_problem(args.size, args.grids, "gs", np_dtype)
        h3.resident_load(b3)

        def sync3():
            h3.sync()
            torch.cuda.synchronize()

        l_times, _, l_norms = timed_regions(h3, sync3, args.steps, min(args.warmup, 2), min(repeats, 3), pre, post, ())
        l_elapsed = statistics.median(l_times)
        lex_path = {"what": "same problem and timed loop with the reference's lexicographic Gauss-Seidel (openmg/solvers.py:56-68) "
                            "as the smoother; grid star stencils run a sweep as one wavefront launch (march.hip, "
                            "OMG_MARCH=0: one launch per level set), bit-identical either way",
                    "vcycles_per_s": round(args.steps / l_elapsed, 3), "ms_per_step": round(1e3 * l_elapsed / args.steps, 4),
                    "ms_per_step_all": [round(1e3 * t / args.steps, 4) for t in l_times],
                    "wavefront_levels": [bool(h3.level_flags(l)["march"]) for l in range(meta3["grids"] - 1)],
                    "level_sets_fine_grid": h3.level_sets(0), "setup_s": round(meta3["setup_s"], 2),
                    "norms_last_region_tail": l_norms[-3:]}
        h3.close()

    h.close()
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
        "value_plain_csr": None if csr_path is None else csr_path["vcycles_per_s"],
        "config": {"workload": "3-D 7-point Poisson %d^3, %d-grid V(1,1) cycle, %s Gauss-Seidel, %s, "
                               "int32 CSR at the boundary (BASELINE configs[2]%s); `value`: operators in the lossless "
                               "device format (row patterns), `value_plain_csr`: the same cycle on plain CSR"
                               % (args.size, meta["grids"], "red-black" if args.smoother == "colour" else args.smoother,
                                  "fp64" if w == 8 else "fp32", "" if w == 8 else " run in fp32: NOT the headline configuration"),
                   "unknowns": n, "nnz": nnz, "grids": meta["grids"], "pre": pre, "post": post,
                   "smoother": args.smoother, "hipgraph": bool(args.graph),
                   "repeats": repeats, "ms_per_step_all": [round(1e3 * t / args.steps, 4) for t in times],
                   "final_residual_norm": norm,
                   "norms": "every cycle of a timed region computes its residual norm (all K are returned at the region's "
                            "end; the red rows' share of cycle k is formed by cycle k+1's first red launch, which computes "
                            "those residuals anyway - bit-identical, tests/test_gpu_parity.py; OMG_NO_PRENORM=1 gives it a launch of its own)",
                   "norms_last_region_tail": region_norms[-3:],
                   "setup_s": round(setup_s, 2),
                   "setup_what": "restrictionList + coeffecientList (device Galerkin products) + device hierarchy; "
                                 "generating the synthetic operator and right-hand side on the host took generate_s",
                   "generate_s": round(generate_s, 2),
                   "kernel_src_sha": kernel_source_hash(), "git_head": git_head()},
        "roofline": roofline,
        "csr_path": csr_path,
        "reference_smoother": lex_path,
        "fine_grid_spmv": fine_spmv,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
