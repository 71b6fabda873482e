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
  roofline      the dominant kernel — the plane-pipelined down pass of the fine grid (red-black
                sweep + residual + restriction in one launch; without plane passes: the fine-grid
                residual kernel r = b - A x): bytes the launch has to move / average launch time
                measured with hipEvents on the kernel's own stream inside the timed regions, against
                8 TB/s (frac <= 1 by construction); the rate in SURVEY 8(d)'s plain-CSR bytes of the
                launches it replaces is reported beside it as csr_equiv_GBps;
  set_schedule  the same problem with the set-by-set schedule (OMG_PLANE=0) the plane passes replace;
  csr_path      the same problem with every operator held as plain int32 CSR (OMG_COMPRESS=0,
                OMG_PLANE=0), timed with the same loop: V-cycles/s, residual and fine-grid SpMV
                launches against SURVEY 8(d)'s CSR byte counts;
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
    """Commit of the sources this line was measured on: OMG_GIT_HEAD — its PRESENCE means "resolved", even empty (the
    GPU boxes have no .git; no rank of a multi-GPU run forks git from a process that has initialised the GPU) — or
    git itself."""
    if "OMG_GIT_HEAD" in os.environ:
        return os.environ["OMG_GIT_HEAD"] or None
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


_PROBLEM = {}


def build_problem(size, grids, smoother, dtype="float64"):
    """Synthetic operator, right-hand side, R list and Galerkin products (made once per (size, grids) and
    shared by the legs), then the device hierarchy for this leg's smoother / environment."""
    import numpy as np
    from openmg_amd import _hip, operators
    shape = (size, size, size)
    key = (size, grids)
    if key not in _PROBLEM:
        t0 = time.perf_counter()
        A0 = operators.stencil_poisson(shape)                       # synthetic input (NumPy, host)
        u_true = np.random.default_rng(12345).random(A0.shape[0])
        b = A0 @ u_true
        # loads the library, creates the HIP context and loads the code object: once per PROCESS, not part of a
        # hierarchy's setup (reported as device_init_s)
        t_i = time.perf_counter()
        _hip.spmv(operators.stencil_poisson((8, 8, 8)), np.ones(512))
        _PROBLEM["device_init_s"] = time.perf_counter() - t_i
        t1 = time.perf_counter()
        # what mgSolve does before its first cycle (openmg/__init__.py:103-109): R list, Galerkin
        # products (on the device)
        R = operators.restrictionList(shape, grids - 2, 8)          # gridLevels = grids - 1 -> coarsestLevel = grids - 2 (D5)
        A = operators.coeffecientList(A0, R)
        t2 = time.perf_counter()
        init_s = _PROBLEM.get("device_init_s", 0.0)
        _PROBLEM.clear()
        _PROBLEM["device_init_s"] = init_s
        _PROBLEM[key] = (A0, b, R, A, t1 - t0 - init_s, t2 - t1)
    A0, b, R, A, gen_s, rap_s = _PROBLEM[key]
    t2 = time.perf_counter()
    h = _hip.Hierarchy(A, R, smoother=smoother, dtype=dtype)        # ordering, coding, upload, coarse factors
    t3 = time.perf_counter()
    meta = {"n": A0.shape[0], "nnz": A0.nnz, "grids": len(A),
            "level_rows": [M.shape[0] for M in A], "level_nnz": [M.nnz for M in A],
            "generate_s": gen_s, "setup_s": rap_s + (t3 - t2), "rap_s": rap_s, "hierarchy_s": t3 - t2,
            "device_init_s": _PROBLEM.get("device_init_s", 0.0)}
    return h, b, meta


class env_override:
    """Environment switches of one leg (read by the library when a hierarchy is created)."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.keep = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.keep.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


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


def timed_regions(h, sync, steps, warmup, repeats, pre, post, classes=("residual",), graph=False):
    """`warmup` untimed cycles, then `repeats` timed regions of EXACTLY `steps` cycles each,
    every one bracketed by a device synchronisation on both sides.  Inside the regions the
    launches of `classes` are bracketed by hipEvents on the hierarchy's own stream (one pair per
    cycle and class: ~1 % of the cycle).  graph: the cycles are replays of a captured hipGraph
    (omg_resident_cycle, one norm read back per cycle) instead of one batched call.  Returns (elapsed
    seconds per region, profile, the last region's norms)."""
    def cycles(k):
        if graph:
            return [h.resident_cycle(pre, post) for _ in range(k)]
        # K cycles enqueued back to back, every cycle's residual norm computed and returned at the end
        # of the region (omg_resident_cycles: mgSolve's loop with a cycle-count stop rule)
        return h.resident_cycles(pre, post, k)

    cycles(warmup)
    sync()
    if classes:
        h.profile_enable(list(classes))
    times, norms = [], []
    for _ in range(repeats):
        sync()
        t0 = time.perf_counter()
        norms = cycles(steps)
        sync()
        times.append(time.perf_counter() - t0)
    prof = h.profile_read() if classes else None
    h.profile_enable(False)
    return times, prof, norms


def residual_launch_bytes(h, meta, w):
    """The fine-grid residual launch r = b - A x of the set-by-set schedule: rows it covers, SURVEY 8(d)'s
    CSR bytes, and the bytes it has to move with the operator in its device format."""
    n = meta["n"]
    n_sets = h.level_sets(0)
    # With a Gauss-Seidel ordering the last set's residual comes out of the smoother launch that
    # relaxed it (bit-identical), so the launch visits the other sets only.
    covered = range(n_sets - 1) if h.level_fused(0) else range(n_sets)
    rows_c = sum(h.set_info(0, s)[0] for s in covered)
    nnz_c = sum(h.set_info(0, s)[1] for s in covered)
    res_csr = (w + 4) * nnz_c + 4 * (rows_c + 1) + 2 * w * rows_c + w * n
    fmt_sets = [h.format_info(0, "A", s) for s in covered]
    res_fmt = sum(f["format_bytes"] for f in fmt_sets) + 2 * w * rows_c + w * n
    fmt_cov = {k: sum(f[k] for f in fmt_sets) for k in ("rows", "nnz", "blocks", "pattern_rows", "coldict_nnz", "valdict_nnz")}
    return rows_c, nnz_c, res_csr, res_fmt, fmt_cov


def class_bytes(h, meta, w):
    """Per class of level-0 launch: bytes one launch has to move (device format) and SURVEY 8(d)'s CSR bytes."""
    n = meta["n"]
    n_c = meta["level_rows"][1] if len(meta["level_rows"]) > 1 else 0
    n_sets = h.level_sets(0)
    rows_c, nnz_c, res_csr, res_fmt, _ = residual_launch_bytes(h, meta, w)
    set_rows = [h.set_info(0, s)[0] for s in range(n_sets)]
    set_nnz = [h.set_info(0, s)[1] for s in range(n_sets)]
    nnz = meta["nnz"]
    # one plane-pipelined pass reads x and b once and writes x once; the down pass also writes the coarse
    # right-hand side (and the coarse initial iterate unless the coarse level's own down pass takes it as
    # zero), the up pass reads the coarse correction; 4 bytes per coarse cell for its slot in the coarse ordering
    coarse_plane = len(meta["level_rows"]) > 2 and h.level_flags(1)["plane"]
    plane_down = 3 * w * n + n_c * (w + 4 + (0 if coarse_plane else w))
    # (the up pass is booked WITHOUT the 4 bytes per coarse cell of the slot map it also reads: 419 MB at 256^3)
    plane_up = 3 * w * n + n_c * w
    csr = {
        "smoother_set_sweep": sum((w + 4) * z + 4 * (r + 1) + 3 * w * r for r, z in zip(set_rows, set_nnz)) / max(n_sets, 1) + w * n / max(n_sets, 1),
        "residual": res_csr,
        "restrict": (w + 4) * n + 4 * (n_c + 1) + w * n + 2 * w * n_c,    # R entries, indptr, r read, b_c + cleared x_c written
        "prolong_add": (w + 4) * n + 4 * (n + 1) + w * n_c + 2 * w * n,   # P entries, indptr, e read, x read + written
        "residual_norm": res_csr - w * rows_c,                            # as the residual launch, nothing stored
        # what the launches a pass replaces move on plain CSR: two set sweeps, the residual, the restriction /
        # the prolongation, two set sweeps, the norm (SURVEY 8(d) V-cycle model, level 0)
        "plane_down": 2 * ((w + 4) * nnz / 2 + 4 * (n / 2 + 1) + 3 * w * n / 2 + w * n / 2) + ((w + 4) * nnz + 4 * (n + 1) + 3 * w * n)
                      + (w + 4) * n + 4 * (n_c + 1) + w * n + 2 * w * n_c,
        "plane_up": (w + 4) * n + 4 * (n + 1) + w * n_c + 2 * w * n + 2 * ((w + 4) * nnz / 2 + 4 * (n / 2 + 1) + 3 * w * n / 2 + w * n / 2)
                    + ((w + 4) * nnz + 4 * (n + 1) + 2 * w * n),
    }
    fA = [h.format_info(0, "A", s) for s in range(n_sets)]
    fR, fP = (h.format_info(0, "R"), h.format_info(0, "P")) if n_c else ({"format_bytes": 0}, {"format_bytes": 0})
    scatter = bool(n_c) and h.level_flags(0)["scatter_prolong"]
    fmt = {
        "smoother_set_sweep": sum(f["format_bytes"] + 3 * w * f["rows"] for f in fA) / max(n_sets, 1) + w * n / max(n_sets, 1),
        "residual": res_fmt,
        "restrict": fR["format_bytes"] + w * n + 2 * w * n_c,
        # prolongation: a pass over P = R^T, or (aggregation R, row-pattern coded) a scatter over R's rows
        "prolong_add": (fR["format_bytes"] if scatter else fP["format_bytes"]) + w * n_c + 2 * w * n,
        "residual_norm": res_fmt - w * rows_c,
        "plane_down": plane_down,
        "plane_up": plane_up,
    }
    return fmt, csr


def kernel_table(prof, steps, fmt, csr):
    kernels = {}
    for name, (cnt, tot) in prof.items():
        if cnt:
            us = 1e3 * tot / cnt
            kernels[name] = {"launches_per_cycle": cnt / steps, "avg_us": round(us, 2),
                             "bytes_per_launch": int(fmt[name]),
                             "achieved": round(fmt[name] / us / 1e3, 1),
                             "frac": round(fmt[name] / us / 1e3 / HBM_PEAK_GBS, 4),
                             "csr_equiv_GBps": round(csr[name] / us / 1e3, 1)}
    return kernels


def pmc_traffic(name, bytes_per_launch, w):
    """HBM traffic from the PMC counters cannot be collected from inside this process.  It is taken from a
    committed rocprofv3 pass of the same launch ONLY when that pass was made with the kernel sources of this
    build (hash recorded in the profile) on the same problem.  Third value: the kernel's average duration in
    that rocprofv3 run (the profiler's clock, beside this run's hipEvents)."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
        if pmc.get("kernel_src_sha") == kernel_source_hash() and int(pmc["bytes_per_launch"]) == int(bytes_per_launch) and w == 8:
            return (pmc["traffic_bytes"], "NOT measured in this run: rocprofv3 --pmc passes of the same build, " + pmc["source"],
                    pmc.get("avg_duration_under_collection_us"))
    except (OSError, KeyError, ValueError):
        pass
    return None, None, None


def config4_leg(size, grids, steps, warmup, repeats, sync_of):
    """BASELINE configs[4]'s per-GPU workload on ONE GPU: 27-point variable-coefficient operator (Q1 stiffness of
    -div(kappa grad u), kappa = exp(U(-1,1) ln 10) per cell, default_rng(2024): SURVEY 8(d)), fp32 levels, Galerkin
    products rebuilt on the device, 8-colour Gauss-Seidel V(1,1) — the same timed loop as the headline: regions of
    `steps` batched cycles, every cycle's norm computed."""
    import numpy as np
    from openmg_amd import _hip, operators
    shape = (size,) * 3
    t0 = time.perf_counter()
    A0 = operators.stencil27_variable(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    t1 = time.perf_counter()
    # restrictions, Galerkin products (rebuilt on the device: configs[4]) and the levels' 27-point tiles, all in HBM
    # (omg_hierarchy_create_from_fine: what mgSolve runs when nobody asks for the operator lists)
    h = _hip.Hierarchy.from_fine(A0, shape, grids - 1, smoother="colour", dtype="float32")
    t3 = t2 = time.perf_counter()
    A = [None] * grids
    h.resident_load(b)
    n, nnz, w = A0.shape[0], A0.nnz, 4
    fused = bool(h.level_flags(0)["stencil27"])
    times, prof, norms = timed_regions(h, sync_of(h), steps, warmup, repeats, 1, 1, ("smoother_set_sweep",))
    e = statistics.median(times)
    cnt, ms = prof["smoother_set_sweep"]
    out = {"what": "BASELINE configs[4] per GPU: 3-D 27-point variable-coefficient Poisson %d^3, %d grids, fp32 levels, Galerkin products "
                   "on the device, 8-colour Gauss-Seidel V(1,1); same timed loop as `value` (batched cycles, every norm computed)" % (size, len(A)),
           "vcycles_per_s": round(steps / e, 3), "ms_per_step": round(1e3 * e / steps, 4),
           "ms_per_step_all": [round(1e3 * t / steps, 4) for t in times], "dtype": "f32",
           "unknowns": n, "nnz": nnz, "stencil27_kernels": fused, "level_flags": [h.level_flags(l) for l in range(len(A) - 1)],
           "generate_s": round(t1 - t0, 2), "device_setup_s": round(t3 - t1, 2),
           "norms_last_region_tail": norms[-3:]}
    if fused and cnt:
        # the dominant kernel: one smoothing sweep of the fine grid = four pair launches of stencil27.hip.  Bytes it has
        # to move: the 27 coefficients of every row once (27 w n), b read and x written once (2 w n), and the iterate
        # read once per launch (every launch needs every colour's neighbours: 4 w n)
        sweep_bytes = (27 + 2 + 4) * w * n
        us = 1e3 * ms / cnt
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_s27_sweep.json")))
            if pmc.get("kernel_src_sha") == kernel_source_hash() and size == 256:
                traffic = pmc["traffic_bytes"]
                traffic_src = "NOT measured in this run: rocprofv3 --pmc passes of the same build, " + pmc["source"]
        except (OSError, KeyError, ValueError):
            pass
        out["roofline"] = {"bound": "hbm", "kernel": "s27_sweep_kernel x 4 (one 8-colour Gauss-Seidel sweep of the fine grid, stencil27.hip)",
                           "achieved": round(sweep_bytes / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(sweep_bytes / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                           "bytes_per_sweep": sweep_bytes, "avg_sweep_us": round(us, 2), "sweeps_timed": cnt,
                           "bytes_definition": "27 coefficients per row (27 w n) + b read, x written (2 w n) + the iterate read once per launch (4 w n)",
                           "csr_equiv_bytes": (w + 4) * nnz + 4 * (n + 8) + 3 * w * n + 8 * w * n,
                           "csr_equiv_GBps": round(((w + 4) * nnz + 4 * (n + 8) + 3 * w * n + 8 * w * n) / us / 1e3, 1)}
    # new coefficients into the same hierarchy (omg_hierarchy_update_fine: level 0 re-tiled, every Galerkin product re-formed
    # on the device by the closed-form streaming kernel, the coarsest operator's inverse anew) — from host memory here
    try:
        t4 = time.perf_counter()
        h.update_fine(A0.data)
        out["update_fine_s"] = round(time.perf_counter() - t4, 3)
        out["update_fine_what"] = "omg_hierarchy_update_fine with the values in HOST memory (3.6 GB over PCIe included); device-resident values: profiles/r05_update_fine.txt"
    except Exception as ex:                                     # noqa: BLE001 - reported, not fatal for the bench line
        out["update_fine_s"] = None
        out["update_fine_what"] = "failed: %s" % ex
    # ... and with the values already in HBM (a tensor's storage), the second of two calls + one cycle
    try:
        import torch
        dev_vals = torch.from_numpy(np.ascontiguousarray(A0.data)).cuda()
        torch.cuda.synchronize()
        for _ in range(2):
            t5 = time.perf_counter()
            h.update_fine((dev_vals.data_ptr(), int(dev_vals.numel())), on_device=True)
            t6 = time.perf_counter()
            h.resident_cycle(1, 1, want_norm=True)
            t7 = time.perf_counter()
        out["update_fine_device_ms"] = round(1e3 * (t6 - t5), 2)
        out["update_fine_device_plus_cycle_ms"] = round(1e3 * (t7 - t5), 2)
        del dev_vals
    except Exception as ex:                                     # noqa: BLE001
        out["update_fine_device_ms"] = None
        out["update_fine_device_what"] = "failed: %s" % ex
    # the whole level-0 kernel table (untimed region)
    h.profile_enable(True)
    for _ in range(min(steps, 10)):
        h.resident_cycle(1, 1, want_norm=True)
    tab = h.profile_read()
    h.profile_enable(False)
    out["level0_kernels"] = {k: {"launches_per_cycle": c / min(steps, 10), "avg_us": round(1e3 * m / c, 2)} for k, (c, m) in tab.items() if c}
    h.close()
    return out


def config0_leg(steps, warmup, repeats, sync_of):
    """BASELINE configs[0] — the reference's own CPU-runnable case: 1-D Poisson N = 4096 (operators.poisson: 4, -1), 3 grids,
    the reference's lexicographic Gauss-Seidel (openmg/solvers.py:56-68), V(1,1), fp64.  A 1-D sweep is a first-order
    recurrence: one wave walks it (march.hip line_gs_kernel)."""
    import numpy as np
    from openmg_amd import _hip, operators
    shape = (4096,)
    A0 = operators.poisson(shape[0], sparse=True)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, 1, 8)
    A = operators.coeffecientList(A0, R)
    h = _hip.Hierarchy(A, R, smoother="gs")
    h.resident_load(b)
    times, _, norms = timed_regions(h, sync_of(h), steps, warmup, repeats, 1, 1, ())
    e = statistics.median(times)
    out = {"what": "BASELINE configs[0]: 1-D Poisson N = 4096, 3 grids, the reference's lexicographic Gauss-Seidel, V(1,1), fp64; same timed loop as `value`",
           "vcycles_per_s": round(steps / e, 1), "ms_per_step": round(1e3 * e / steps, 4), "grids": len(A),
           "row_updates_per_cycle": 2 * sum(int(M.shape[0]) for M in A[:-1]),
           "ns_per_row_update_incl_everything": round(1e9 * e / steps / (2 * sum(int(M.shape[0]) for M in A[:-1])), 1),
           "norms_last_region_tail": norms[-2:]}
    h.close()
    return out


def var7_leg(size, grids, steps, warmup, repeats, sync_of):
    """A 7-point operator with PER-ROW coefficients (openmg_amd.operators.stencil7_variable: finite volumes of
    -div(kappa grad u), kappa over two decades) — the ordinary variable-coefficient input of mgSolve — on the headline's
    grid: size^3, `grids` grids, red-black, V(1,1), fp64.  Large levels run each half of the cycle as one launch (var7.hip);
    the same hierarchy set by set (omg_hierarchy_use_plane(0)) in the same timed loop beside it, and that both give the same bits."""
    import numpy as np
    from openmg_amd import _hip, operators
    shape = (size,) * 3
    A0 = operators.stencil7_variable(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, grids - 2, 8)
    A = operators.coeffecientList(A0, R)
    t0 = time.perf_counter()
    h = _hip.Hierarchy(A, R, smoother="colour")
    setup = time.perf_counter() - t0
    flags = [bool(h.level_flags(l)["var7"]) for l in range(len(R))]
    h.resident_load(b)
    times, _, norms = timed_regions(h, sync_of(h), steps, warmup, repeats, 1, 1, ())
    e = statistics.median(times)
    # the finest level's two launches, hipEvents on the hierarchy's stream (an untimed region)
    h.profile_enable(["plane_down", "plane_up"])
    h.resident_cycles(1, 1, steps)
    prof = h.profile_read()
    h.profile_enable(False)
    x_fused = h.resident_fetch()
    h.use_plane(False)
    h.resident_load(b)
    h.resident_cycles(1, 1, warmup + steps * (repeats + 1))             # (the cycles the fused path has run)
    same = bool(np.array_equal(h.resident_fetch(), x_fused))
    h.resident_load(b)
    times_s, _, _ = timed_regions(h, sync_of(h), steps, warmup, min(repeats, 3), 1, 1, ())
    es = statistics.median(times_s)
    n = A0.shape[0]
    w = 8
    # what one pass of the finest level has to move: the row's seven coefficients (7 w), x read, b read, x written (3 w) per
    # unknown; + w (+ 4 for its slot) per coarse unknown
    pass_bytes = 10 * w * n + (w + 4) * (n // 8)
    out = {"what": "3-D 7-point operator with per-row coefficients (finite volumes, kappa over two decades) %d^3, %d grids, red-black V(1,1), "
                   "fp64: fused passes on the levels of 128^3 and more (var7.hip), same timed loop as `value`" % (size, len(A)),
           "vcycles_per_s": round(steps / e, 1), "ms_per_step": round(1e3 * e / steps, 4),
           "ms_per_step_all": [round(1e3 * t / steps, 4) for t in times],
           "set_schedule_vcycles_per_s": round(steps / es, 1), "same_bits_as_the_set_schedule": same,
           "fused_levels": flags, "hierarchy_s": round(setup, 2), "norms_last_region_tail": norms[-2:]}
    if prof and prof.get("plane_down") and prof["plane_down"][0] and prof.get("plane_up") and prof["plane_up"][0]:
        d_us = 1e3 * prof["plane_down"][1] / prof["plane_down"][0]
        u_us = 1e3 * prof["plane_up"][1] / prof["plane_up"][0]
        out["roofline"] = {"bound": "hbm", "kernel": "var7_pass_kernel<down> of the finest level: last pre-smoothing sweep + residual + restriction in one launch",
                           "bytes_per_launch": pass_bytes, "avg_launch_us": round(d_us, 1), "up_pass_avg_launch_us": round(u_us, 1),
                           "achieved": round(pass_bytes / d_us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(pass_bytes / d_us / 1e3 / HBM_PEAK_GBS, 4),
                           "bytes_definition": "per fine unknown: the row's seven coefficients (7 w), x read, b read, x written (3 w); per coarse "
                                               "unknown: right-hand side written (w) + its slot read (4)"}
    h.close()
    # the same operator with the reference's OWN smoother (lexicographic Gauss-Seidel, openmg/solvers.py:56-68; the default of
    # the drop-in): per-row coefficients have no pattern table — a third wave streams the rows' coefficients (march.hip, round 6;
    # as a level schedule of 3 n - 2 launches per sweep this cycle took 18 ms)
    t0 = time.perf_counter()
    hl = _hip.Hierarchy(A, R, smoother="gs")
    setup_l = time.perf_counter() - t0
    hl.resident_load(b)
    lsteps = max(4, steps // 2)
    times_l, _, norms_l = timed_regions(hl, sync_of(hl), lsteps, 2, min(repeats, 3), 1, 1, ())
    el = statistics.median(times_l)
    out["reference_smoother"] = {"vcycles_per_s": round(lsteps / el, 1), "ms_per_step": round(1e3 * el / lsteps, 4),
                                 "wavefront_levels": [bool(hl.level_flags(l)["march"]) for l in range(len(R))],
                                 "hierarchy_s": round(setup_l, 2), "norms_last_region_tail": norms_l[-2:]}
    hl.close()
    # what a caller of the drop-in pays on this input: mgSolve for 20 cycles including its whole setup — on the device when
    # nobody asks for the operator lists (round 6: the large levels' rows are checked and scattered into their coefficient
    # arrays in HBM, the small levels below them fetched and coded by the host)
    import openmg_amd
    p_s = {"problemShape": shape, "gridLevels": grids - 1, "cycles": 20, "threshold": 0, "preIterations": 1, "postIterations": 1, "smoother": "colour"}
    t0 = time.perf_counter()
    openmg_amd.mgSolve(A0, b, dict(p_s))
    out["mgsolve_20_cycles_s"] = round(time.perf_counter() - t0, 3)
    return out


def config1_leg(steps, warmup, repeats, sync_of):
    """BASELINE configs[1]: 2-D 5-point Poisson 1024^2, 4 grids, fp64: weighted Jacobi (the smoother the config names)
    and red-black Gauss-Seidel (fused tile passes), V(1,1), same timed loop."""
    import numpy as np
    from openmg_amd import _hip, operators
    shape = (1024, 1024)
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, 2, 8)
    A = operators.coeffecientList(A0, R)
    out = {"what": "BASELINE configs[1]: 2-D 5-point Poisson 1024^2, 4 grids, fp64, V(1,1); same timed loop as `value`"}
    for name, smoother, kw in (("weighted_jacobi", "jacobi", {"omega": 2.0 / 3.0}), ("red_black", "colour", {})):
        h = _hip.Hierarchy(A, R, smoother=smoother, **kw)
        h.resident_load(b)
        times, _, norms = timed_regions(h, sync_of(h), steps * 4, warmup, repeats, 1, 1, ())
        e = statistics.median(times)
        out[name] = {"vcycles_per_s": round(steps * 4 / e, 1), "ms_per_step": round(1e3 * e / (steps * 4), 4),
                     "plane_levels": [bool(h.level_flags(l)["plane"]) for l in range(len(A) - 1)],
                     "norms_last_region_tail": norms[-2:]}
        h.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=11,
                    help="timed regions of --steps cycles each; value = steps / median region time (BASELINE.md §3)")
    ap.add_argument("--size", type=int, default=256, help="grid extent per axis (default: BASELINE config 3)")
    ap.add_argument("--grids", type=int, default=5)
    ap.add_argument("--smoother", default="colour")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                    help="precision the levels are stored / computed in (f64 = BASELINE configs[2], the default)")
    ap.add_argument("--stencil", default="7pt", choices=["7pt", "27var"],
                    help="multi-GPU leg only: 27var = BASELINE configs[4]'s 27-point variable-coefficient operator")
    ap.add_argument("--graph", type=int, default=0, help="replay every cycle of the timed regions from a hipGraph")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-plain", action="store_true", help="skip the plain-CSR (OMG_COMPRESS=0) leg")
    ap.add_argument("--no-sets", action="store_true", help="skip the set-by-set schedule (OMG_PLANE=0) leg")
    ap.add_argument("--no-lex", action="store_true", help="skip the leg with the reference's lexicographic Gauss-Seidel")
    ap.add_argument("--no-dropin", action="store_true", help="skip the legs through the Python drop-in (mgCycle per call, mgSolve end to end)")
    ap.add_argument("--no-config4", action="store_true", help="skip the configs[4] leg (27-point variable-coefficient, fp32, 256^3)")
    ap.add_argument("--no-config1", action="store_true", help="skip the configs[1] leg (2-D 1024^2)")
    ap.add_argument("--config4-size", type=int, default=256)
    ap.add_argument("--dist", type=int, default=0, help="force the multi-GPU code path even with one rank (debug)")
    ap.add_argument("--watchdog", type=int, default=900, help="multi-GPU: abort after this many seconds")
    ap.add_argument("--overlap", type=int, default=1,
                    help="multi-GPU: relax boundary rows first and exchange them while the interior rows run")
    ap.add_argument("--dist-grids", type=int, default=4,
                    help="multi-GPU: grids handled across ranks (the last of them and all below run replicated)")
    ap.add_argument("--cpu-size", type=int, default=256, help="CPU baseline leg: grid extent (default: the full workload)")
    ap.add_argument("--cpu-cycles", type=int, default=4)
    args = ap.parse_args()

    # identity of the sources, resolved BEFORE anything initialises the GPU (no fork from a GPU process) and
    # handed to the children of a multi-GPU launch
    os.environ.setdefault("OMG_GIT_HEAD", git_head() or "")
    src_sha = kernel_source_hash()

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
    n, nnz = meta["n"], meta["nnz"]
    plane = meta["grids"] > 1 and h.level_flags(0)["plane"]
    meta["plane_levels"] = [h.level_flags(l)["plane"] for l in range(meta["grids"] - 1)]
    if args.graph:
        h.use_graph(True)

    def syncer(hh):
        def sync():
            hh.sync()
            torch.cuda.synchronize()
        return sync

    # Timed regions.  Only the roofline kernel (one launch per cycle) is bracketed by hipEvents there:
    # every event pair opens a gap in the stream.  (A hipGraph replay cannot carry the events; with
    # --graph 1 the kernel is timed in the second region below.)
    roof_class = "plane_down" if plane else "residual"
    times, timed, region_norms = timed_regions(h, syncer(h), args.steps, args.warmup, repeats, pre, post,
                                               () if args.graph else (roof_class,), graph=bool(args.graph))
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
    spmv_csr = spmv_bytes(n, nnz, w)
    if plane:
        # a plane level applies its operator matrix-free, like its cycle does: x read once, y written once
        spmv_fmt = 2 * w * n
        spmv_kernel = "y = A x, all rows of the fine grid (plane_spmv_kernel: matrix-free, the level's seven coefficients)"
    else:
        fmt_all = h.format_info(0, "A")
        spmv_fmt = fmt_all["format_bytes"] + 2 * w * n          # operator in its device format + x read + y written
        spmv_kernel = "y = A x, all rows of the fine grid (%s)" % kernel_name(fmt_all, "ROW_SPMV", h.level_flags(0)["union_walk"])
    fine_spmv = {"kernel": spmv_kernel,
                 "avg_launch_us": round(spmv_ms * 1e3, 2),
                 "bytes_per_launch": spmv_fmt, "achieved": round(spmv_fmt / spmv_ms / 1e6, 1),
                 "frac": round(spmv_fmt / spmv_ms / 1e6 / HBM_PEAK_GBS, 4),
                 "csr_equiv_bytes": spmv_csr, "csr_equiv_GBps": round(spmv_csr / spmv_ms / 1e6, 1)}

    fmt_b, csr_b = class_bytes(h, meta, w)
    kernels = kernel_table(prof, args.steps, fmt_b, csr_b)
    launches, ms = timed[roof_class]
    avg_s = (ms / launches) * 1e-3
    if plane:
        # The dominant kernel of the cycle: the plane-pipelined DOWN pass of the fine grid — red-black sweep,
        # residual and restriction of openmg/__init__.py:201-210 in one launch.  Algorithmic bytes per unit
        # (DESIGN.md §5): per fine unknown 3 w (x read, b read, x written), per coarse unknown w (its right-hand
        # side) + 4 (its slot in the coarse ordering).  The operator itself costs nothing: seven coefficients.
        bytes_roof = fmt_b["plane_down"]
        traffic, traffic_src, rocprof_us = pmc_traffic("r06_pmc_plane_down.json", bytes_roof, w)
        achieved = bytes_roof / avg_s / 1e9
        roofline = {"bound": "hbm", "kernel": "plane_kernel<down>: fine-grid red-black sweep + residual + restriction in one launch (plane.hip)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "bytes_per_launch": int(bytes_roof),
                    "bytes_definition": "per fine unknown: x read, b read, x written (3 w); per coarse unknown: right-hand side written (w) "
                                        "+ its slot in the coarse ordering read (4)",
                    "avg_launch_us": round(avg_s * 1e6, 2), "avg_launch_us_source": "hipEvents on the kernel's own stream inside the timed regions",
                    "avg_launch_us_rocprof": rocprof_us,
                    "avg_launch_us_rocprof_source": "NOT measured in this run: rocprofv3's average duration of the same kernel of the same build "
                                                    "(profiles/r06_pmc_plane_down.json; per grid size over a whole bench run: "
                                                    "profiles/r06_bench_kernel_stats_by_grid.txt)" if rocprof_us else None,
                    "launches_timed": launches,
                    "csr_equiv_bytes": int(csr_b["plane_down"]), "csr_equiv_GBps": round(csr_b["plane_down"] / avg_s / 1e9, 1),
                    "tiling": h.plane_info(0),
                    "level0_kernels": kernels}
    else:
        rows_c, nnz_c, res_csr, res_fmt, fmt_cov = residual_launch_bytes(h, meta, w)
        traffic, traffic_src, _ = pmc_traffic("r02_pmc_residual.json", res_fmt, w)
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
    # The reference's OWN parameter dict: preIterations 1, postIterations 0 (openmg/__init__.py:22-23), same hierarchy
    # and loop: the up pass then runs without its relaxation (prolongation + correction + the norm's squares).
    # the same setup kept on the device (omg_hierarchy_create_from_fine: what mgSolve runs when the caller does not ask
    # for the operator lists): upload of the fine operator, restrictions, Galerkin products, qualification, tilings
    setup_device_s = None
    if args.smoother == "colour":
        from openmg_amd import _hip as _h
        t_s = time.perf_counter()
        h_dev = _h.Hierarchy.from_fine(_PROBLEM[(args.size, args.grids)][0], (args.size,) * 3, args.grids - 1, smoother=args.smoother, dtype=np_dtype)
        setup_device_s = time.perf_counter() - t_s
        h_dev.close()
    default_cycle = None
    if plane and args.smoother == "colour":
        h.resident_load(b)
        t_d, _, n_d = timed_regions(h, syncer(h), args.steps, min(args.warmup, 3), min(repeats, 7), 1, 0, ())
        e_d = statistics.median(t_d)
        default_cycle = {"what": "same hierarchy and timed loop with the reference's default sweep counts V(1,0) "
                                 "(openmg/__init__.py:22-23), red-black ordering, plane passes",
                         "vcycles_per_s": round(args.steps / e_d, 3), "ms_per_step": round(1e3 * e_d / args.steps, 4),
                         "ms_per_step_all": [round(1e3 * t / args.steps, 4) for t in t_d],
                         "plane": bool(h.level_flags(0)["plane"]), "norms_last_region_tail": n_d[-3:]}
    h.close()

    # What a caller of the DROP-IN pays (north star: "openmg.mg_cycle(A, b, ...) is a drop-in"): mgCycle is handed the
    # A / R lists and host vectors on every call (openmg/__init__.py:151), mgSolve builds the hierarchy itself (:103-109).
    dropin = None
    if not args.no_dropin and args.smoother == "colour":
        import numpy as np
        import openmg_amd
        A0, b_h, R_l, A_l = _PROBLEM[(args.size, args.grids)][:4]
        prm = {"coarsestLevel": len(R_l), "preIterations": 1, "postIterations": 1, "smoother": "colour",
               "dtype": np_dtype}
        t0 = time.perf_counter()
        x_d, info_d = openmg_amd.mgCycle(A_l, b_h, 0, R_l, prm)                  # first call: uploads the hierarchy
        first_s = time.perf_counter() - t0
        calls = []
        for _ in range(4):
            t0 = time.perf_counter()
            x_d, info_d = openmg_amd.mgCycle(A_l, b_h, 0, R_l, prm, initial=x_d)
            calls.append(time.perf_counter() - t0)
        # the same chain with b, initial and uOut as DEVICE arrays and the list members taken on trust (they are the objects
        # of the calls above): nothing but the norm crosses PCIe (omg_vcycle_dev)
        b_t = torch.from_numpy(b_h).cuda()
        x_t, info_t = openmg_amd.mgCycle(A_l, b_t, 0, R_l, dict(prm, trustOperators=True))
        dev_calls = []
        for _ in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x_t, info_t = openmg_amd.mgCycle(A_l, b_t, 0, R_l, dict(prm, trustOperators=True), initial=x_t)
            torch.cuda.synchronize()
            dev_calls.append(time.perf_counter() - t0)
        x_chk = None
        for _ in range(9):
            x_chk, info_chk = openmg_amd.mgCycle(A_l, b_h, 0, R_l, prm, initial=x_chk)
        dev_same = bool(np.array_equal(x_t.cpu().numpy(), x_chk)) and info_t["norm"] == info_chk["norm"]
        del b_t, x_t
        openmg_amd.clear_cache()
        p_s = {"problemShape": (args.size,) * 3, "gridLevels": args.grids - 1, "cycles": 20, "threshold": 0,
               "preIterations": 1, "postIterations": 1, "smoother": "colour", "dtype": np_dtype}
        t0 = time.perf_counter()
        u_s = openmg_amd.mgSolve(A0, b_h, dict(p_s))                           # what a caller writes: setup stays on the device
        solve_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        u_i, info_s = openmg_amd.mgSolve(A0, b_h, dict(p_s, giveInfo=True))    # ... asking for the R / A lists: through the host
        solve_info_s = time.perf_counter() - t0
        same = bool(np.array_equal(u_s, u_i))
        dropin = {"what": "openmg_amd.mgCycle(A, b, 0, R, parameters, initial) with the A / R lists and host vectors per call "
                          "(cached device hierarchy, every byte of the lists checksummed per call, b and x over PCIe), and "
                          "openmg_amd.mgSolve(A_in, b, parameters) for 20 cycles including its whole setup: on the device "
                          "(omg_hierarchy_create_from_fine) when the caller does not ask for the operator lists, through host "
                          "lists with giveInfo",
                  "mgcycle_first_call_s": round(first_s, 3),
                  "mgcycle_call_ms": round(1e3 * statistics.median(calls), 2),
                  "mgcycle_call_ms_all": [round(1e3 * t, 2) for t in calls],
                  "mgcycle_norm": info_d["norm"],
                  "mgcycle_device_arrays_call_ms": round(1e3 * statistics.median(dev_calls), 3),
                  "mgcycle_device_arrays_call_ms_all": [round(1e3 * t, 3) for t in dev_calls],
                  "mgcycle_device_arrays_what": "b, initial, uOut as PyTorch-ROCm tensors, parameters['trustOperators'] = True (no checksum of the "
                                                "lists: the same objects as the calls before); nine chained calls give the bits of nine host-array calls: %s" % dev_same,
                  "mgsolve_20_cycles_s": round(solve_s, 3), "mgsolve_20_cycles_giveinfo_s": round(solve_info_s, 3),
                  "mgsolve_same_iterate_both_routes": same,
                  "mgsolve_norm": info_s["norm"], "mgsolve_cycles": info_s["cycle"]}

    def leg(env, smoother, what, reps, want_spmv):
        """The same problem and timed loop under other switches."""
        with env_override(**env):
            h2, b2, m2 = build_problem(args.size, args.grids, smoother, np_dtype)
        h2.resident_load(b2)
        cls = () if smoother != "colour" else ("residual",)
        t2, p2, n2 = timed_regions(h2, syncer(h2), args.steps, min(args.warmup, 3), reps, pre, post, cls)
        e2 = statistics.median(t2)
        out = {"what": what, "vcycles_per_s": round(args.steps / e2, 3), "ms_per_step": round(1e3 * e2 / args.steps, 4),
               "ms_per_step_all": [round(1e3 * t / args.steps, 4) for t in t2], "hierarchy_s": round(m2["hierarchy_s"], 2)}
        if cls:
            rows_c, nnz_c, res_csr, res_fmt, _ = residual_launch_bytes(h2, m2, w)
            pl, pms = p2["residual"]
            p_us = 1e3 * pms / pl
            plain = env.get("OMG_COMPRESS") == "0"
            nb = res_csr if plain else res_fmt
            out["residual"] = {"kernel": "rows_kernel<ROW_RESIDUAL>" if plain else "rows_union_kernel<ROW_RESIDUAL>",
                               "avg_launch_us": round(p_us, 2), "bytes_per_launch": nb, "achieved": round(nb / p_us / 1e3, 1),
                               "frac": round(nb / p_us / 1e3 / HBM_PEAK_GBS, 4)}
            if want_spmv:
                p_spmv_ms = h2.spmv_time(20)
                out["fine_grid_spmv"] = {"kernel": "rows_kernel<ROW_SPMV>", "avg_launch_us": round(p_spmv_ms * 1e3, 2),
                                         "bytes_per_launch": spmv_csr, "achieved": round(spmv_csr / p_spmv_ms / 1e6, 1),
                                         "frac": round(spmv_csr / p_spmv_ms / 1e6 / HBM_PEAK_GBS, 4)}
        else:
            out["wavefront_levels"] = [bool(h2.level_flags(l)["march"]) for l in range(m2["grids"] - 1)]
            out["line_scan_levels"] = [bool(h2.level_flags(l)["march_scan"]) for l in range(m2["grids"] - 1)]
            out["level_sets_fine_grid"] = h2.level_sets(0)
            out["norms_last_region_tail"] = n2[-3:]
        h2.close()
        return out

    # The set-by-set schedule the plane passes replace (round 2's headline path): eight level-0 launches per cycle.
    set_path = None
    if plane and not args.no_sets:
        set_path = leg({"OMG_PLANE": "0"}, args.smoother,
                       "same problem and timed loop, set-by-set schedule (OMG_PLANE=0): operators in the lossless device "
                       "format, eight level-0 launches per cycle; same iterate bit for bit", min(repeats, 3), False)
    # The same problem with every operator as PLAIN int32 CSR (OMG_COMPRESS=0) walked by the CSR row kernels
    # (OMG_PLANE=0) — what the north star's "CSR SpMV ... >= 50 % of the HBM roofline on the fine-grid SpMV"
    # describes — timed with the same loop.
    csr_path = None
    if not args.no_plain and os.environ.get("OMG_COMPRESS", "15") != "0":
        csr_path = leg({"OMG_COMPRESS": "0", "OMG_PLANE": "0"}, args.smoother,
                       "same problem and timed loop, every operator held as plain int32 CSR and walked by the CSR row kernels "
                       "(OMG_COMPRESS=0, OMG_PLANE=0)", repeats, True)
    # The same problem with the reference's OWN smoother — in-place lexicographic Gauss-Seidel
    # (openmg/solvers.py:56-68) — instead of the red-black ordering: same loop, fewer repeats.
    lex_path = None
    if not args.no_lex and args.smoother == "colour":
        lex_path = leg({}, "gs",
                       "same problem and timed loop with the reference's lexicographic Gauss-Seidel (openmg/solvers.py:56-68) "
                       "as the smoother; grid star stencils run a sweep as one wavefront launch (march.hip, "
                       "OMG_MARCH=0: one launch per level set), bit-identical either way", min(repeats, 3), False)
        # ... and with the opt-in line-scan sweep (OMG_MARCH_SCAN=1, round 6: a wave resolves a grid line's recurrence at once;
        # a few ulp per row away from the sequential loop's bits, inside the 1e-10 contract — tests/test_gpu_march.py)
        scan_path = leg({"OMG_MARCH_SCAN": "1"}, "gs",
                        "the reference's lexicographic Gauss-Seidel with OMG_MARCH_SCAN=1: the line-scan sweep (march.hip "
                        "scan_gs_kernel), not bit-identical to the sequential loop", min(repeats, 3), False)
        lex_path["line_scan"] = {k: scan_path[k] for k in ("vcycles_per_s", "ms_per_step", "ms_per_step_all", "line_scan_levels",
                                                           "norms_last_region_tail")}
        na, nb = lex_path["norms_last_region_tail"][-1], scan_path["norms_last_region_tail"][-1]
        lex_path["line_scan"]["norm_rel_diff_from_the_wavefront_kernel"] = abs(na - nb) / max(abs(na), 1e-300)

    config4 = None
    if not args.no_config4 and args.dtype == "f64" and args.smoother == "colour":
        _PROBLEM.clear()                               # (the 256^3 7-point operator: 1.9 GB of host memory)
        config4 = config4_leg(args.config4_size, args.grids, max(10, args.steps), 3, min(repeats, 7), syncer)
    var7 = None
    if not args.no_config1 and args.dtype == "f64" and args.smoother == "colour":
        var7 = var7_leg(args.size, args.grids, args.steps, 3, min(repeats, 5), syncer)
    config0 = None
    if not args.no_config1 and args.dtype == "f64":
        config0 = config0_leg(args.steps, 3, min(repeats, 5), syncer)
    config1 = None
    if not args.no_config1 and args.dtype == "f64" and args.smoother == "colour":
        config1 = config1_leg(args.steps, 3, min(repeats, 9), syncer)

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

    # What the driver's record keeps of a line is its scalars (nested tables under `config` / `roofline` and the legs' own
    # objects are dropped): the figures the other legs exist for are therefore ALSO flat scalars of `roofline` / `config`.
    #  * SURVEY 8(d) as it is worded — the fine-grid SpMV over plain int32 CSR (1 740 111 876 B at 256^3), the `csr_path` leg;
    #  * the cycle with the reference's own (lexicographic) smoother; BASELINE configs[4]'s cycle and its sweep's fraction;
    #  * traffic_ratio: HBM bytes the PMC passes counted for the roofline kernel over the bytes it has to move.
    if csr_path is not None and "fine_grid_spmv" in csr_path:
        roofline["csr_spmv_bytes"] = csr_path["fine_grid_spmv"]["bytes_per_launch"]
        roofline["csr_spmv_us"] = csr_path["fine_grid_spmv"]["avg_launch_us"]
        roofline["csr_spmv_frac"] = csr_path["fine_grid_spmv"]["frac"]
        roofline["csr_cycle_vcycles_per_s"] = csr_path["vcycles_per_s"]
    roofline["matrix_free_spmv_frac"] = fine_spmv["frac"]
    roofline["matrix_free_spmv_us"] = fine_spmv["avg_launch_us"]
    if dropin is not None:
        roofline["mgcycle_host_arrays_call_ms"] = dropin["mgcycle_call_ms"]
        roofline["mgcycle_device_arrays_call_ms"] = dropin["mgcycle_device_arrays_call_ms"]
    if lex_path is not None:
        roofline["reference_smoother_vcycles_per_s"] = lex_path["vcycles_per_s"]
        roofline["reference_smoother_line_scan_vcycles_per_s"] = lex_path["line_scan"]["vcycles_per_s"]
    if var7 is not None:
        roofline["var7_vcycles_per_s"] = var7["vcycles_per_s"]
        roofline["var7_set_schedule_vcycles_per_s"] = var7["set_schedule_vcycles_per_s"]
        roofline["var7_reference_smoother_vcycles_per_s"] = var7["reference_smoother"]["vcycles_per_s"]
        roofline["var7_mgsolve_20_cycles_s"] = var7["mgsolve_20_cycles_s"]
    if config4 is not None:
        roofline["config4_vcycles_per_s"] = config4.get("vcycles_per_s")
        roofline["config4_sweep_frac"] = (config4.get("roofline") or {}).get("frac")
    if roofline.get("traffic") and roofline.get("bytes_per_launch"):
        roofline["traffic_ratio"] = round(roofline["traffic"] / float(roofline["bytes_per_launch"]), 4)
    steady_ms = round(1e3 * min(times[len(times) // 2:]) / args.steps, 4)
    headline_shape = args.size == 256 and w == 8 and plane
    population = ("fast" if steady_ms <= 0.272 else "slow") if headline_shape else None

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
        "value_set_schedule": None if set_path is None else set_path["vcycles_per_s"],
        "value_plain_csr": None if csr_path is None else csr_path["vcycles_per_s"],
        "config": {"workload": "3-D 7-point Poisson %d^3, %d-grid V(1,1) cycle, %s Gauss-Seidel, %s, "
                               "int32 CSR at the boundary (BASELINE configs[2]%s); `value`: %s; `value_plain_csr`: the same "
                               "cycle on plain CSR"
                               % (args.size, meta["grids"], "red-black" if args.smoother == "colour" else args.smoother,
                                  "fp64" if w == 8 else "fp32", "" if w == 8 else " run in fp32: NOT the headline configuration",
                                  "each half of the cycle over a grid-stencil level is one plane-pipelined launch (plane.hip); "
                                  "`value_set_schedule`: the same cycle set by set on the lossless device format" if plane else
                                  "operators in the lossless device format (row patterns)"),
                   "unknowns": n, "nnz": nnz, "grids": meta["grids"], "pre": pre, "post": post,
                   "smoother": args.smoother, "hipgraph": bool(args.graph),
                   "plane_levels": [bool(f) for f in meta.get("plane_levels", [])],
                   "repeats": repeats, "ms_per_step_all": [round(1e3 * t / args.steps, 4) for t in times],
                   # the two kinds of allocation of the finest level's vectors (the level's passes ~7 % apart;
                   # hierarchy.hip place_finest_pool searches for the fast kind at setup): which one this hierarchy ended up with,
                   # by its steady regions (the first regions of any process are ~3 % slower)
                   "process_population": {"steady_ms_per_step": steady_ms,
                                          "first_region_ms_per_step": round(1e3 * times[0] / args.steps, 4),
                                          "which": population,
                                          "rule": "256^3 fp64 headline: steady cycle <= 0.272 ms = the finest level's vectors are the fast kind of allocation (0.258-0.270), above = the slow kind (0.273-0.290); profiles/r05_pool_placement.txt"},
                   # (the same, flat: the driver's record keeps scalars only)
                   "process_population_which": population, "process_population_steady_ms": steady_ms,
                   "final_residual_norm": norm,
                   "norms": "every cycle of a timed region computes its residual norm (all K are returned at the region's end): "
                            + ("the up pass of the fine grid leaves the squared residuals of its rows as one partial per workgroup, "
                               "one launch per 64 cycles adds them up" if plane else
                               "the red rows' share of cycle k is formed by cycle k+1's first red launch, which computes "
                               "those residuals anyway - bit-identical, tests/test_gpu_parity.py; OMG_NO_PRENORM=1 gives it a launch of its own"),
                   "norms_last_region_tail": region_norms[-3:],
                   "setup_s": round(setup_s, 2),
                   "setup_what": "restrictionList + coeffecientList (device Galerkin products: rap_s) + device hierarchy (hierarchy_s), i.e. what "
                                 "mgSolve does before its first cycle (openmg/__init__.py:103-109), in a process whose HIP context exists "
                                 "(creating it and loading the code object, once per process: device_init_s); generating the synthetic "
                                 "operator and right-hand side on the host took generate_s",
                   "device_init_s": round(meta.get("device_init_s", 0.0), 3),
                   "setup_device_s": None if setup_device_s is None else round(setup_device_s, 3),
                   "setup_device_what": "the same setup without host lists in between (omg_hierarchy_create_from_fine; mgSolve's route "
                                        "when giveInfo is off): fine operator up, restrictions + Galerkin products + the levels' "
                                        "qualification in HBM, coarsest operator down for its factorisation",
                   "rap_s": round(meta["rap_s"], 2), "hierarchy_s": round(meta["hierarchy_s"], 2),
                   "generate_s": round(generate_s, 2),
                   "kernel_src_sha": src_sha, "git_head": os.environ.get("OMG_GIT_HEAD") or None},
        "roofline": roofline,
        "default_cycle": default_cycle,
        "dropin": dropin,
        "config4": config4,
        "config0": config0,
        "config1": config1,
        "var7": var7,
        "set_schedule": set_path,
        "csr_path": csr_path,
        "reference_smoother": lex_path,
        "fine_grid_spmv": fine_spmv,
        "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
