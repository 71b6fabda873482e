"""bench.py's multi-GPU leg: one process per GPU (torch.distributed.run), weak scaling.

Per-GPU work is fixed at 256^3 unknowns: N = 2 -> 256x512x256, N = 4 -> 512x256x512,
N = 8 -> 512^3 (BASELINE configs[3]); shapes keep shape[0] == shape[2] because the
reference's restriction uses shape[0] as its second-axis offset (openmg/operators.py:78).
Six grids (configs[3]) for N >= 2.  `value` is in 256^3-equivalent V-cycles/s = cycles/s x
(global unknowns / 256^3), so that it aggregates over GPUs."""
import json
import os
import statistics
import sys
import time

import numpy as np

SHAPES = {1: (256, 256, 256), 2: (256, 512, 256), 4: (512, 256, 512), 8: (512, 512, 512)}


def overlap_candidates(level_rows):
    """(name, OMG_OVERLAP_MIN_ROWS) candidates for the exchange / compute overlap: every distributed
    level, the k largest levels for k = 1 .. n - 1, none.  level_rows: owned rows of the smoothed
    distributed levels, finest first."""
    sizes = sorted(set(int(n) for n in level_rows), reverse=True)
    return [("all levels", 0)] + [("levels >= %d rows" % n, n) for n in sizes[:-1]] + [("none", 1 << 62)]


def _bench_identity():
    """(hash of the kernel sources, commit) as bench.py reports them.  The commit is what bench.py resolved BEFORE
    anything initialised the GPU and exported as OMG_GIT_HEAD (empty: no .git): a rank never forks git."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_omg_bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.kernel_source_hash(), (os.environ.get("OMG_GIT_HEAD") or None)


def plane_levels(shape, world, n_dist):
    """How many levels the plane-pipelined slab runner can distribute (0: not applicable): the first n_dist - 1
    levels, as long as every one of them has even extents and at least two planes per rank."""
    n = 0
    nz, ny, nx = shape
    for _ in range(max(0, n_dist - 1)):
        if nz % world or (nz // world) % 2 or (nz // world) < 2 or ny % 2 or nx % 2:
            break
        n += 1
        nz, ny, nx = nz // 2, ny // 2, nx // 2
    return n


def slab27_levels(shape, world, n_dist, dtype):
    """How many levels the 27-point slab runner (csrc/dist27.hip) can distribute (0: not applicable): the first
    n_dist - 1 levels, as long as every one of them has even extents, an even number (>= 2) of planes per rank and a
    grid line that fits one wave (512 cells in fp32, 256 in fp64)."""
    n = 0
    nz, ny, nx = shape
    line = 512 if dtype == "f32" else 256
    for _ in range(max(0, n_dist - 1)):
        if nz % world or (nz // world) % 2 or (nz // world) < 2 or ny % 2 or nx % 2 or nx > line:
            break
        n += 1
        nz, ny, nx = nz // 2, ny // 2, nx // 2
    return n


def _median_ms(run, sync, steps, warmup, repeats=3):
    run(max(1, warmup))
    times = []
    for _ in range(repeats):
        sync()
        t0 = time.perf_counter()
        run(steps)
        sync()
        times.append((time.perf_counter() - t0) / steps)
    return 1e3 * statistics.median(times)


def one_gpu_baselines_plane(args, shape, grids, torch):
    """The two one-GPU numbers an N-GPU line of the 7-point workload is read against, measured by rank 0 ON ITS OWN GPU in
    the same run (the other ranks wait): (i) BASELINE configs[2] as `bench.py --gpus 1` runs it — (256 s)^3, `--grids`
    grids, the single-GPU hierarchy (`vs_n1_config2`: what a weak-scaling ratio over the driver's N = 1 line means, a ratio
    between two DIFFERENT hierarchies); (ii) the SAME global problem and hierarchy on one GPU — one slab holding every
    plane, same passes, no exchanges (`vs_one_gpu_same_problem`: the strong-scaling speed-up of this run).
    OMG_BENCH_BASELINES=0 skips both (keys stay, values None)."""
    from . import _hip, _hip_dist, operators
    out = {"n1_config2_ms_per_cycle": None, "one_gpu_same_problem_ms_per_cycle": None}
    if os.environ.get("OMG_BENCH_BASELINES", "1") == "0":
        return out
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import importlib.util
    spec = importlib.util.spec_from_file_location("_omg_bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def sync_of(obj):
        def sync():
            obj.sync()
            torch.cuda.synchronize()
        return sync

    h, b, _ = mod.build_problem(args.size, args.grids, args.smoother, "float64")
    try:
        h.resident_load(b)
        out["n1_config2_ms_per_cycle"] = round(_median_ms(lambda k: h.resident_cycles(1, 1, k), sync_of(h), args.steps, args.warmup), 4)
    finally:
        h.close()
        mod._PROBLEM.clear()
    del b
    n_levels = plane_levels(shape, 1, max(2, min(args.dist_grids, grids)))
    coef = [[v / 16.0 ** l for v in (-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0)] for l in range(n_levels)]
    tshape = tuple(s >> n_levels for s in shape)
    tgrids = grids - n_levels
    At = operators.stencil_poisson(tshape) / 16.0 ** n_levels
    Rt = operators.restrictionList(tshape, tgrids - 2, 8) if tgrids >= 2 else []
    tail = _hip.Hierarchy(operators.coeffecientList(At, Rt), Rt, smoother=args.smoother)
    d = _hip_dist.PlaneDistRank(0, 1, shape, coef, 0.125, tail)
    try:
        d.load(np.random.default_rng(12345).random(d.n_local))
        out["one_gpu_same_problem_ms_per_cycle"] = round(_median_ms(lambda k: d.cycles(k), sync_of(d), args.steps, args.warmup), 4)
    finally:
        d.close()
        tail.close()
    return out


def one_gpu_baselines_slab27(args, shape, grids, torch, np_dtype):
    """As one_gpu_baselines_plane, for the 27-point workload: (i) ONE slab of the N = 1 problem ((256 s)^3, `--grids` grids:
    what `bench.py --gpus 1 --dist 1 --stencil 27var` runs), (ii) one slab holding the whole global problem — measured only
    while that operator's CSR (27 nnz per row, 12 bytes each on the host) stays under ~12 GB, i.e. up to N = 2."""
    from . import _hip_dist, dist
    out = {"n1_config4_ms_per_cycle": None, "one_gpu_same_problem_ms_per_cycle": None, "note": None}
    if os.environ.get("OMG_BENCH_BASELINES", "1") == "0":
        return out

    def one_slab(shp, g):
        n_lv = slab27_levels(shp, 1, max(2, min(args.dist_grids, g)), args.dtype)
        n = shp[0] * shp[1] * shp[2]
        A_rows = dist.stencil27_variable_rows(shp, 0, n)
        b = A_rows @ np.random.default_rng(12345).random(n)
        d = _hip_dist.Slab27Rank(0, 1, shp, A_rows, n_lv, 0.125, np_dtype)
        del A_rows
        tail = dist.make_tail(dist.assemble_coarse([d.coarse_rows()]), tuple(v >> n_lv for v in shp), g - n_lv, smoother="colour", dtype=np_dtype)
        d.set_tail(tail)
        try:
            d.load(b)

            def sync():
                d.sync()
                torch.cuda.synchronize()
            return round(_median_ms(lambda k: d.cycles(1, 1, k), sync, args.steps, args.warmup), 4)
        finally:
            d.close()
            tail.close()

    s1 = int(args.size)
    out["n1_config4_ms_per_cycle"] = one_slab((s1, s1, s1), args.grids)
    if shape[0] * shape[1] * shape[2] <= 2 * 256 ** 3:
        out["one_gpu_same_problem_ms_per_cycle"] = one_slab(shape, grids)
    else:
        out["note"] = "the same problem on one GPU was not measured: its host CSR (%.0f GB) is beyond one rank's set-up budget" % (
            12e-9 * 27 * shape[0] * shape[1] * shape[2])
    return out


def main_slab27(args, rank, world, shape, grids, n_levels, all_gather, td, torch, watchdog):
    """BASELINE configs[4]: the 27-point variable-coefficient operator, 8-colour Gauss-Seidel, on slabs with ghost
    aggregate planes run by the octant-layout kernels (omg_sdist_*): per-rank Galerkin products on the device, one
    exchange per sweep."""
    import numpy as np
    from . import _hip, _hip_dist, dist, preflight
    np_dtype = "float64" if args.dtype == "f64" else "float32"
    w = 8 if args.dtype == "f64" else 4
    t_setup = time.perf_counter()
    plane = shape[1] * shape[2]
    lo = rank * (shape[0] // world) * plane
    hi = lo + (shape[0] // world) * plane
    A_rows = dist.stencil27_variable_rows(shape, lo, hi)
    n_glob = shape[0] * plane
    u = np.random.default_rng(12345).random(n_glob)
    b_loc = A_rows @ u
    nnz_loc, n_loc = A_rows.nnz, hi - lo
    del u
    t_gen = time.perf_counter()
    r = _hip_dist.Slab27Rank(rank, world, shape, A_rows, n_levels, 0.125, np_dtype)
    del A_rows
    # the operator below the slabs: every rank's rows gathered, the levels under it replicated as an ordinary hierarchy
    coarse = dist.assemble_coarse(all_gather(r.coarse_rows()))
    tshape = tuple(s >> n_levels for s in shape)
    tgrids = grids - n_levels
    tail = dist.make_tail(coarse, tshape, tgrids, smoother="colour", dtype=np_dtype)
    del coarse
    r.set_tail(tail)
    if world > 1:
        ident = [_hip_dist.rccl_unique_id() if rank == 0 else None]
        td.broadcast_object_list(ident, src=0)
        r.connect(ident[0])                                     # (collective: also the neighbours' coefficient rows)
    r.load(b_loc)
    setup_s = time.perf_counter() - t_setup
    pre = post = 1
    first_norm = preflight.run(rank, world, lambda: r.cycles(pre, post, 1)[0], all_gather, min(120.0, max(20.0, args.watchdog / 4.0)),
                               where=lambda: "27-point slab cycle")
    exchanges = r.info(0)["exchanges_last_call"]
    # Peer mode for the halo exchanges (omg_sdist_p2p_*, round 6): on request only (OMG_DIST_P2P=1) — tried after the RCCL
    # preflight, kept only if every rank mapped its neighbours and one checked cycle from the same start gives the RCCL
    # cycle's norm on every rank.  (There is no fallback inside the timed regions as the plane slabs have: a wait that gives
    # up there raises.)
    exchange = ("RCCL grouped send/recv of ghost aggregate planes (colours 4..7 both ways after every sweep, colours 0..3 of the "
                "coarse right-hand side after every restriction)")
    p2p_note = "not tried (OMG_DIST_P2P=1 asks for it)"
    if world > 1 and os.environ.get("OMG_DIST_P2P", "auto") == "1":
        devices = all_gather(int(torch.cuda.current_device()))
        nbs = [q for q in (rank - 1, rank + 1) if 0 <= q < world]
        try:
            reach = all(_hip_dist.peer_access(devices[rank], devices[q]) for q in nbs)
            mine = r.p2p_handles() if reach else None
            note = None if reach else "rank %d's GPU cannot address a neighbour's memory (hipDeviceCanAccessPeer)" % rank
        except Exception as e:                                  # noqa: BLE001 - any failure means "stay with RCCL"
            mine, note = None, "export failed on rank %d: %s" % (rank, e)
        handles = all_gather(mine)
        ok = all(h is not None for h in handles)
        if ok:
            try:
                for q in nbs:
                    r.p2p_open(q, handles[q])
                r.p2p_enable(1)
            except Exception as e:                              # noqa: BLE001
                ok, note = False, "mapping failed on rank %d: %s" % (rank, e)
        if all(all_gather(bool(ok))):
            r.load(b_loc)
            try:
                peer_norm = preflight.run(rank, world, lambda: r.cycles(pre, post, 1)[0], all_gather, min(120.0, max(20.0, args.watchdog / 4.0)),
                                          where=lambda: "27-point slab cycle, peer stores")
                ok = abs(peer_norm - first_norm) <= 1e-9 * abs(first_norm)
                note = None if ok else "norm after one cycle %.17g, RCCL cycle gave %.17g" % (peer_norm, first_norm)
            except RuntimeError as e:
                ok, note = False, str(e)
        else:
            ok = False
        if all(all_gather(bool(ok))):
            exchange = "peer stores into the neighbours' ghost aggregate planes (xGMI), flags; the gather below the slabs and the norm's reduction over RCCL"
            p2p_note = "checked against the RCCL cycle: same norm on every rank"
        else:
            if r.p2p_mode:
                r.p2p_enable(0)
            notes = [n for n in all_gather(note) if n]
            p2p_note = "rejected: " + (notes[0] if notes else "another rank failed")
    r.load(b_loc)
    trajectory = r.cycles(pre, post, 1)
    for _ in range(args.warmup):
        trajectory += r.cycles(pre, post, 1)
    times = []
    for _ in range(max(1, getattr(args, "repeats", 1))):
        r.sync()
        torch.cuda.synchronize()
        td.barrier()
        t0 = time.perf_counter()
        region = r.cycles(pre, post, args.steps)         # K cycles back to back, every cycle's global norm computed
        r.sync()
        torch.cuda.synchronize()
        td.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)             # the slowest rank's time
        times.append(float(t[0]))
        trajectory += region
    elapsed = statistics.median(times)
    norm = r.cycles(pre, post, 1)[0]
    rccl_ranks = r.rccl_ranks()
    r.sync()
    td.barrier()
    base = None
    if rank == 0 and world > 1:
        try:
            base = one_gpu_baselines_slab27(args, shape, grids, torch, np_dtype)
        except Exception as e:                                  # noqa: BLE001
            base = {"n1_config4_ms_per_cycle": None, "one_gpu_same_problem_ms_per_cycle": None, "note": None, "error": "%s: %s" % (type(e).__name__, e)}
    td.barrier()
    if rank == 0:
        equiv = n_glob / float(256 ** 3)
        # what a rank's fine-grid launches have to move per V(1,1) cycle: two sweeps of (27 + 2 + 4) w n
        # and the residual of six of eight colours + restriction, (6/8 27 + 2 + 1) w n + w n / 8
        sweep_bytes = (27 + 2 + 4) * w * n_loc
        cycle_bytes = 2 * sweep_bytes + int((0.75 * 27 + 3) * w * n_loc) + w * n_loc // 8
        src_sha, head = _bench_identity()
        info = r.info(0)
        out = {
            "metric": "V-cycles/sec (256^3-unknown equivalents), 3-D 27-point variable-coefficient Poisson, weak scaling",
            "value": round(args.steps / elapsed * equiv, 3),
            "unit": "256^3-equivalent V-cycles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "3-D 27-point variable-coefficient Poisson (Q1 stiffness) %s, %d-grid V(1,1) cycle, 8-colour Gauss-Seidel, %s, "
                                   "Galerkin products per rank on the device, 1-D slabs over %d GPUs: octant-layout kernels on slabs with "
                                   "ghost aggregate planes, one exchange per sweep"
                                   % ("x".join(map(str, shape)), grids, "fp64" if w == 8 else "fp32", world),
                       "unknowns": n_glob, "unknowns_per_gpu": n_loc, "nnz_per_gpu": nnz_loc, "grids": grids,
                       "distributed_grids": n_levels, "replicated_tail_grids": tgrids, "runner": "27-point slabs (omg_sdist)",
                       "exchange": exchange, "peer_mode": p2p_note,
                       "halo_exchanges_per_cycle": exchanges, "rccl_ranks": rccl_ranks, "repeats": len(times),
                       "ranks_share_one_gpu": os.environ.get("OMG_DIST_SHARED_GPU", "0") in ("1", "rccl") and world > 1,
                       "preflight_norm": first_norm, "kernel_src_sha": src_sha, "git_head": head,
                       "ms_per_step_all": [round(1e3 * t / args.steps, 4) for t in times],
                       "pre": pre, "post": post, "cycles_per_s": round(args.steps / elapsed, 3),
                       "final_residual_norm": norm, "norms_last_region_tail": trajectory[-3:],
                       "workgroups": info["workgroups"], "aggregates_per_lane": info["aggregates_per_lane"],
                       "generate_s": round(t_gen - t_setup, 2), "setup_s": round(setup_s, 2)},
            "vs_n1_config2": (round((args.steps / elapsed * equiv) / ((args.size / 256.0) ** 3 * 1e3 / base["n1_config4_ms_per_cycle"]), 4)
                              if base and base["n1_config4_ms_per_cycle"] else None),
            "vs_one_gpu_same_problem": (round(base["one_gpu_same_problem_ms_per_cycle"] / (1e3 * elapsed / args.steps), 4)
                                        if base and base["one_gpu_same_problem_ms_per_cycle"] else None),
            "one_gpu_baselines": base,
            # a LOWER bound of the fine-grid launches' rate: the bytes rank 0's fine-grid launches have to move per cycle over
            # the WHOLE cycle's time (exchanges, coarser levels and the replicated tail included)
            "roofline": {"bound": "hbm", "kernel": "rank 0's fine-grid launches per cycle: 2 sweeps (4 pair launches each) + residual of 6 colours + restriction "
                                                   "(bytes needed / whole cycle time: a lower bound)",
                         "achieved": round(cycle_bytes / (elapsed / args.steps) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(cycle_bytes / (elapsed / args.steps) / 1e9 / 8000.0, 4), "traffic": None,
                         "bytes_per_cycle": cycle_bytes, "bytes_per_sweep": sweep_bytes},
            "cpu_baseline": None,
        }
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out))
        sys.stdout.flush()
    r.close()
    tail.close()
    watchdog.cancel()
    td.barrier()
    td.destroy_process_group()
    return 0


def main_plane(args, rank, world, shape, grids, n_levels, all_gather, td, torch, watchdog):
    """The 7-point red-black cycle as plane-pipelined passes on slabs with ghost planes (omg_pdist_*)."""
    import numpy as np
    from . import _hip, _hip_dist, dist, operators, preflight
    t_setup = time.perf_counter()
    lo = rank * (shape[0] // world) * shape[1] * shape[2]
    hi = lo + (shape[0] // world) * shape[1] * shape[2]
    A_rows = dist.stencil_rows(shape, lo, hi)
    n_glob = shape[0] * shape[1] * shape[2]
    u = np.random.default_rng(12345).random(n_glob)
    b_loc = A_rows @ u
    nnz_loc, n_loc = A_rows.nnz, hi - lo
    del u, A_rows
    # level l of the hierarchy is lap3(shape / 2^l) / 16^l exactly (tests/test_gpu_parity.py); the levels below the
    # slabs run replicated as an ordinary hierarchy on the gathered right-hand side
    coef = [[v / 16.0 ** l for v in (-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0)] for l in range(n_levels)]
    tshape = tuple(s >> n_levels for s in shape)
    tgrids = grids - n_levels
    At = operators.stencil_poisson(tshape) / 16.0 ** n_levels
    Rt = operators.restrictionList(tshape, tgrids - 2, 8) if tgrids >= 2 else []
    tail = _hip.Hierarchy(operators.coeffecientList(At, Rt), Rt, smoother=args.smoother)
    r = _hip_dist.PlaneDistRank(rank, world, shape, coef, 0.125, tail)
    # OMG_DIST_SHARED_GPU=1: every rank on GPU 0 — a REHEARSAL of the multi-process path (launcher, rendezvous, preflight,
    # hipIpc peer mappings, flags between processes) where there is one GPU; RCCL cannot put two ranks on one device, so
    # the exchanges are peer stores with wait launches and the norms are added over gloo.  Not a scaling measurement.
    # OMG_DIST_SHARED_GPU=rccl (with OMG_RCCL_LIB naming tests/fake_rccl's library): every rank on GPU 0 too, but through the
    # RCCL call sites themselves — communicators, grouped send / recv, all-gather, all-reduce — as on N GPUs.
    shared = os.environ.get("OMG_DIST_SHARED_GPU", "0") == "1" and world > 1
    shared_rccl = os.environ.get("OMG_DIST_SHARED_GPU", "0") == "rccl" and world > 1

    def reduce(squares):
        t = torch.tensor(squares, dtype=torch.float64)
        td.all_reduce(t)
        return [float(v) for v in t]

    run_cycles = (lambda n: r.cycles(n, reduce)) if shared else r.cycles
    if shared:
        handles = all_gather(r.p2p_handles())
        for peer in range(world):
            if peer != rank:
                r.p2p_open(peer, handles[peer])
        r.p2p_enable(2)
    else:
        ident = [(_hip_dist.rccl_unique_id(), _hip_dist.rccl_unique_id()) if rank == 0 else None]
        td.broadcast_object_list(ident, src=0)
        r.connect(*ident[0])
    r.load(b_loc)
    setup_s = time.perf_counter() - t_setup
    # ONE checked cycle: every rank's norm under a deadline, compared with rank 0's; a rank that hangs says where
    r.trace(True)
    first_norm = preflight.run(rank, world, lambda: run_cycles(1)[0], all_gather, min(120.0, max(20.0, args.watchdog / 4.0)),
                               where=lambda: "cycle %d, level %d, last completed phase: %s" % r.progress())
    r.trace(False)
    # Peer mode (xGMI peer stores fused into the passes, no exchange launches): tried when every level has four planes
    # per rank, kept only if EVERY rank could map every other rank's buffers and one checked cycle from the same start
    # gives the norm the RCCL cycle gave; otherwise the RCCL exchanges stay.  OMG_DIST_P2P=0: not tried; =1: tried
    # with one rank too.
    exchange = "RCCL grouped send/recv of ghost planes"
    p2p_note = "not tried"
    want = os.environ.get("OMG_DIST_P2P", "auto")
    if shared:
        exchange = "peer stores between PROCESSES SHARING ONE GPU (hipIpc mappings, wait launches): a rehearsal, not a scaling measurement"
        p2p_note = "required (no RCCL between ranks on one device)"
    elif shared_rccl:
        exchange = "grouped send/recv of ghost planes through the RCCL call sites, ranks SHARING ONE GPU (test stand-in for librccl): a rehearsal, not a scaling measurement"
        p2p_note = "not tried (ranks share a GPU)"
    elif want != "0" and (world > 1 or want == "1") and all((shape[0] >> l) // world >= 4 for l in range(n_levels)):
        def agree(ok):
            return all(all_gather(bool(ok)))
        devices = all_gather(int(torch.cuda.current_device()))
        try:
            # (asked first: a store through a mapping the hardware cannot serve is a memory fault, not an exception)
            reach = all(_hip_dist.peer_access(devices[rank], devices[peer]) for peer in range(world) if peer != rank)
            mine = r.p2p_handles() if reach else None
            if not reach:
                p2p_note = "rank %d's GPU cannot address every other rank's memory (hipDeviceCanAccessPeer)" % rank
        except Exception as e:                                  # noqa: BLE001 - any failure means "stay with RCCL"
            mine, p2p_note = None, "export failed on rank %d: %s" % (rank, e)
        handles = all_gather(mine)
        ok = all(h is not None for h in handles)
        if ok:
            try:
                for peer in range(world):
                    if peer != rank:
                        r.p2p_open(peer, handles[peer])
                r.p2p_enable(1)
            except Exception as e:                              # noqa: BLE001
                ok, p2p_note = False, "mapping failed on rank %d: %s" % (rank, e)
        if agree(ok):
            r.load(b_loc)
            try:
                r.trace(True)
                peer_norm = preflight.run(rank, world, lambda: r.cycles(1)[0], all_gather, min(120.0, max(20.0, args.watchdog / 4.0)),
                                          where=lambda: "peer mode, cycle %d, level %d, last completed phase: %s" % r.progress())
                r.trace(False)
                ok = abs(peer_norm - first_norm) <= 1e-12 * abs(first_norm)
                if not ok:
                    p2p_note = "norm after one cycle %.17g, RCCL cycle gave %.17g" % (peer_norm, first_norm)
            except RuntimeError as e:
                ok, p2p_note = False, str(e)
            if agree(ok):
                exchange = "peer stores into the neighbours' ghost planes (xGMI), flags, no exchange launches"
                p2p_note = "checked against the RCCL cycle: same norm on every rank"
            else:
                r.p2p_enable(0)
                notes = [n for n in all_gather(p2p_note) if n != "not tried"]
                p2p_note = "rejected: " + (notes[0] if notes else "another rank failed")
        else:
            r.p2p_enable(0)
            notes = [n for n in all_gather(p2p_note) if n != "not tried"]
            p2p_note = "unavailable: " + (notes[0] if notes else "another rank failed")
    # Gated passes over RCCL (csrc/dist.hip PlaneDist::gate): the finest level's passes as ONE launch each whose edge chunks
    # wait on a device flag while the exchange of their ghost planes runs beside the inner chunks.  Tried only where the
    # slabs qualify (>= 96 planes per rank, <= 192 edge workgroups) and peer mode is not in use; kept only if one checked
    # cycle from the same start gives the stream-ordered cycle's norm on every rank.  OMG_DIST_GATE=0: not tried.
    gate_note = "not applicable (slabs of fewer than 96 planes, or more than 192 edge workgroups)"
    if not shared and not r.p2p_mode and world > 1 and os.environ.get("OMG_DIST_GATE", "1") != "0":
        r.set_gate(True)
        if all(all_gather(bool(r.info()["gated"]))):
            # two cycles each way from the same start: in the first the down pass's ghost exchange is still stream-ordered,
            # the exchange posted on the side stream behind the PREVIOUS cycle and signalled through the gate flag is first
            # used in the second — both norms are compared
            r.set_gate(False)
            r.load(b_loc)
            ordered = run_cycles(2)
            r.set_gate(True)
            r.load(b_loc)
            gated = {}

            def two_gated_cycles():
                gated["norms"] = run_cycles(2)
                return gated["norms"][1]
            try:
                preflight.run(rank, world, two_gated_cycles, all_gather, min(120.0, max(20.0, args.watchdog / 4.0)), where=lambda: "gated passes")
                ok = all(abs(g - o) <= 1e-12 * abs(o) for g, o in zip(gated["norms"], ordered))
                note = ("checked against two stream-ordered cycles: same norms" if ok
                        else "norms %r, stream-ordered cycles gave %r" % (gated["norms"], ordered))
            except RuntimeError as e:
                ok, note = False, str(e)
            if all(all_gather(bool(ok))):
                gate_note = "on: " + note
            else:
                r.set_gate(False)
                gate_note = "rejected: " + next(n for n in all_gather(note) if not n.startswith("checked"))
        else:
            r.set_gate(False)

    class CyclesFailed(RuntimeError):
        """Some rank's cycles raised inside a timed run; raised on EVERY rank at the same point."""

    def timed_run():
        """From the loaded right-hand side: one cycle, the warm-up, the timed regions -> (times, every norm in order).
        A rank whose cycles raise (a peer-store wait that gave up) keeps making the same collectives as the others and
        tells them through the reductions, so that all ranks leave together (CyclesFailed) instead of one crashing and
        the others waiting for the watchdog."""
        failed = [0.0]

        def cycles(n):
            if failed[0]:
                return [float("nan")] * n
            try:
                return run_cycles(n)
            except RuntimeError as e:
                sys.stderr.write("bench.py rank %d: cycles raised inside the timed run: %s\n" % (rank, e))
                failed[0] = 1.0
                return [float("nan")] * n

        def anyone_failed(extra=0.0):
            t = torch.tensor([extra, failed[0]], dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)         # the slowest rank's time; whether any rank failed
            return float(t[0]), bool(t[1])

        r.load(b_loc)
        trajectory = cycles(1)
        for _ in range(args.warmup):
            trajectory += cycles(1)
        if anyone_failed()[1]:
            raise CyclesFailed("warm-up")
        times_ = []
        for _ in range(max(1, getattr(args, "repeats", 1))):
            r.sync()
            torch.cuda.synchronize()
            td.barrier()
            t0 = time.perf_counter()
            region = cycles(args.steps)                  # K cycles back to back, every cycle's global norm computed
            r.sync()
            torch.cuda.synchronize()
            td.barrier()
            slowest, bad = anyone_failed(time.perf_counter() - t0)
            if bad:
                raise CyclesFailed("timed region")
            times_.append(slowest)
            trajectory += region
        return times_, trajectory

    try:
        times, trajectory = timed_run()
    except CyclesFailed as e:
        if r.p2p_mode and not shared:
            # peer mode gave up somewhere: every rank is here (the failure was agreed inside timed_run); the RCCL
            # exchanges are the checked fallback
            r.p2p_enable(0)
            exchange = "RCCL grouped send/recv of ghost planes"
            p2p_note = "rejected DURING the timed run (%s): the reported run is the RCCL one" % e
        elif gate_note.startswith("on"):
            # a gated pass's bounded wait gave up: back to the exchanges in stream order, on every rank
            r.set_gate(False)
            gate_note = "rejected DURING the timed run (%s): the reported run has the exchanges in stream order" % e
        else:
            raise
        times, trajectory = timed_run()
    if r.p2p_mode and not shared:
        # every norm of the timed peer-mode run against the same cycles over RCCL (untimed): a hand-over that went
        # wrong once in hundreds of cycles must not survive into the reported number
        r.p2p_enable(0)
        r.load(b_loc)
        check = []
        while len(check) < len(trajectory):
            check += r.cycles(min(args.steps, len(trajectory) - len(check)))
        same = all(abs(a - c) <= 1e-11 * abs(c) for a, c in zip(trajectory, check))
        if all(all_gather(bool(same))):
            p2p_note += "; all %d norms of the timed run equal the same cycles' over RCCL" % len(trajectory)
        else:
            bad = next(k for k, (a, c) in enumerate(zip(trajectory, check)) if not abs(a - c) <= 1e-11 * abs(c)) if not same else -1
            p2p_note = "rejected AFTER the timed run (rank %d: first differing cycle %d): the reported run is the RCCL one" % (rank, bad)
            p2p_note = next(n for n in all_gather(p2p_note) if n.startswith("rejected"))
            exchange = "RCCL grouped send/recv of ghost planes"
            times, trajectory = timed_run()
    region_norms = trajectory[-args.steps:]
    elapsed = statistics.median(times)
    norm = run_cycles(1)[0]
    rccl_ranks = r.rccl_ranks()
    r.sync()
    td.barrier()
    base = None
    if rank == 0 and world > 1:
        try:                                                    # (a baseline that fails must not cost the run its line)
            base = one_gpu_baselines_plane(args, shape, grids, torch)
        except Exception as e:                                  # noqa: BLE001
            base = {"n1_config2_ms_per_cycle": None, "one_gpu_same_problem_ms_per_cycle": None, "error": "%s: %s" % (type(e).__name__, e)}
    td.barrier()
    if rank == 0:
        equiv = n_glob / float(256 ** 3)
        # per cycle a rank's two fine-grid passes move 3 w per owned unknown + (w + 4) per coarse one, each
        n_c = n_loc // 8
        pass_bytes = 3 * 8 * n_loc + n_c * (8 + 4)
        src_sha, head = _bench_identity()
        out = {
            "metric": "V-cycles/sec (256^3-unknown equivalents), 3-D 7-point Poisson, weak scaling",
            "value": round(args.steps / elapsed * equiv, 3),
            "unit": "256^3-equivalent V-cycles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "3-D 7-point Poisson %s, %d-grid V(1,1) cycle, red-black Gauss-Seidel, fp64, 1-D slabs over %d GPUs: "
                                   "plane-pipelined passes on slabs with ghost planes, whole ghost planes exchanged"
                                   % ("x".join(map(str, shape)), grids, world),
                       "unknowns": n_glob, "unknowns_per_gpu": n_loc, "nnz_per_gpu": nnz_loc, "grids": grids,
                       "distributed_grids": n_levels, "replicated_tail_grids": tgrids, "runner": "plane slabs (omg_pdist)",
                       "exchange": exchange, "peer_mode": p2p_note, "gated_passes": gate_note, "ranks_share_one_gpu": bool(shared or shared_rccl),
                       "rccl_ranks": rccl_ranks, "repeats": len(times), "preflight_norm": first_norm,
                       "kernel_src_sha": src_sha, "git_head": head,
                       "ms_per_step_all": [round(1e3 * t / args.steps, 4) for t in times],
                       "pre": 1, "post": 1, "cycles_per_s": round(args.steps / elapsed, 3),
                       "final_residual_norm": norm, "norms_last_region_tail": region_norms[-3:],
                       "setup_s": round(setup_s, 2)},
            # a LOWER bound of the fine-grid passes' rate: the bytes both of rank 0's fine-grid passes have to move per
            # cycle over the WHOLE cycle's time (exchanges, coarser levels and the replicated tail included)
            # the two one-GPU baselines of this line, measured by rank 0 in this run (one_gpu_baselines_plane): `value` over the
            # N = 1 line's workload (configs[2]: another hierarchy, one grid fewer) in the same 256^3-equivalent unit, and this
            # run's cycle rate over the SAME problem's on one GPU
            "vs_n1_config2": (round((args.steps / elapsed * equiv) / ((args.size / 256.0) ** 3 * 1e3 / base["n1_config2_ms_per_cycle"]), 4)
                              if base and base["n1_config2_ms_per_cycle"] else None),
            "vs_one_gpu_same_problem": (round(base["one_gpu_same_problem_ms_per_cycle"] / (1e3 * elapsed / args.steps), 4)
                                        if base and base["one_gpu_same_problem_ms_per_cycle"] else None),
            "one_gpu_baselines": base,
            "roofline": {"bound": "hbm", "kernel": "rank 0's two fine-grid plane passes per cycle (bytes needed / whole cycle time: a lower bound)",
                         "achieved": round(2 * pass_bytes / (elapsed / args.steps) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(2 * pass_bytes / (elapsed / args.steps) / 1e9 / 8000.0, 4), "traffic": None,
                         "bytes_per_launch": pass_bytes},
            "cpu_baseline": None,
        }
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out))
        sys.stdout.flush()
    r.close()
    tail.close()
    watchdog.cancel()
    td.barrier()
    td.destroy_process_group()
    return 0


def main(args):
    import datetime
    import threading
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if world not in SHAPES:
        raise SystemExit("bench.py --gpus must be 1, 2, 4 or 8")

    # A rendezvous or collective that never completes would hang the whole node job: the
    # watchdog is armed BEFORE anything that can block (init_process_group included) and ends
    # this rank with a non-zero code.  Nothing is ever re-exec'ed.
    def _abort():
        sys.stderr.write("bench.py rank %d: watchdog expired after %d s, aborting\n" % (rank, args.watchdog))
        sys.stderr.flush()
        os._exit(3)

    watchdog = threading.Timer(args.watchdog, _abort)
    watchdog.daemon = True
    watchdog.start()

    import torch
    import torch.distributed as td
    from . import _hip, _hip_dist, dist

    _hip.require_gpu()
    if os.environ.get("OMG_DIST_SHARED_GPU", "0") in ("1", "rccl"):
        local = 0                                             # (rehearsal on one GPU: main_plane)
    torch.cuda.set_device(local)
    _hip_dist.set_device(local)
    # control plane (ids, barriers, timing max) over gloo; the data plane is RCCL inside
    # libopenmg_hip.so on the rank's HIP stream
    td.init_process_group("gloo", rank=rank, world_size=world,
                          timeout=datetime.timedelta(seconds=max(30, min(args.watchdog, 600))))

    def all_gather(obj):
        out = [None] * world
        td.all_gather_object(out, obj)
        return out

    scale = args.size / 256.0
    shape = tuple(int(s * scale) for s in SHAPES[world])
    grids = args.grids if world == 1 else args.grids + 1
    t_setup = time.perf_counter()
    # levels 0..n_dist-2 are smoothed across ranks; level n_dist-1 and everything below it run
    # replicated on every rank as an ordinary single-GPU hierarchy (dist.make_tail)
    n_dist = max(2, min(args.dist_grids, grids))
    # constant-coefficient 7-point stencil, red-black, fp64: the plane-pipelined slab runner (OMG_DIST_PLANE=0: the
    # set-by-set runner with one exchange per colour below)
    n_plane = plane_levels(shape, world, n_dist)
    if (args.stencil == "7pt" and args.smoother == "colour" and args.dtype == "f64" and n_plane >= 1
            and os.environ.get("OMG_DIST_PLANE", "1") != "0"):
        return main_plane(args, rank, world, shape, grids, n_plane, all_gather, td, torch, watchdog)
    # 27-point operator with per-row coefficients, 8 colours: the octant-layout slab runner (OMG_DIST_SLAB27=0: the set-by-set
    # runner below)
    n_s27 = slab27_levels(shape, world, n_dist, args.dtype)
    if args.stencil == "27var" and args.smoother == "colour" and n_s27 >= 1 and os.environ.get("OMG_DIST_SLAB27", "1") != "0":
        return main_slab27(args, rank, world, shape, grids, n_s27, all_gather, td, torch, watchdog)
    part = dist.SlabPartition(shape, world, n_dist)
    lo, hi = part.rows(0, rank)
    w = 8 if args.dtype == "f64" else 4
    np_dtype = "float64" if w == 8 else "float32"
    if args.stencil == "27var":                               # BASELINE configs[4]
        A_rows = dist.stencil27_variable_rows(shape, lo, hi)
        colouring = "octant"
    else:
        A_rows = dist.stencil_rows(shape, lo, hi)
        colouring = "parity"
    u = np.random.default_rng(12345).random(part.n_rows(0))
    b_loc = A_rows @ u
    del u
    # Boundary-first (boundary, interior) set pairs on EVERY distributed level: which of those
    # levels then really run their halo exchanges on the second stream beside the interior rows is
    # measured below (the threshold is read by the runner at every cycle), not guessed.
    # (OMG_BENCH_AUTOTUNE=0 keeps the default threshold; =force runs the probe loop with one rank too)
    mode = os.environ.get("OMG_BENCH_AUTOTUNE", "1")
    tune = bool(args.overlap) and (world > 1 or mode == "force") and "OMG_OVERLAP_MIN_ROWS" not in os.environ and mode != "0"
    if tune:
        os.environ["OMG_OVERLAP_MIN_ROWS"] = "0"
    levels, coarse, counts = dist.build_this_rank(part, rank, A_rows, all_gather, smoother=args.smoother,
                                                  overlap=bool(args.overlap), colouring=colouring)
    nnz_loc, n_loc = A_rows.nnz, hi - lo
    del A_rows
    tail = dist.make_tail(coarse, part.shapes[-1], grids - n_dist + 1, smoother=args.smoother, dtype=np_dtype)
    r = _hip_dist.DistRank(rank, world, levels, None, counts, smoother=args.smoother, tail=tail, dtype=np_dtype)
    ident = [_hip_dist.rccl_unique_id() if rank == 0 else None]
    td.broadcast_object_list(ident, src=0)
    r.connect(ident[0])
    r.load(b_loc)
    setup_s = time.perf_counter() - t_setup

    pre = post = 1
    autotune = None
    if tune:
        # Exchange / compute overlap threshold: levels with at least this many owned rows send their
        # boundary values on the second stream while the interior rows are relaxed (two events per
        # exchange); smaller levels exchange in line.  Every candidate is timed over a few cycles
        # (slowest rank's time), the fastest is kept for the timed regions.  Identical iterates.
        cands = overlap_candidates([part.rows(l, rank)[1] - part.rows(l, rank)[0] for l in range(n_dist - 1)])
        autotune = {}
        probe = max(10, args.steps)
        for _ in range(max(args.warmup, 5)):                      # clocks and caches warm before the first candidate
            r.cycle(pre, post, want_norm=False)
        for name, thr in cands:
            os.environ["OMG_OVERLAP_MIN_ROWS"] = str(thr)
            for _ in range(3):
                r.cycle(pre, post, want_norm=False)
            r.sync()
            td.barrier()
            t0 = time.perf_counter()
            r.cycles(pre, post, probe)
            r.sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            autotune[name] = (thr, float(t[0]) / probe)
        best = min(autotune, key=lambda k: autotune[k][1])        # the same on every rank: the times were all-reduced
        os.environ["OMG_OVERLAP_MIN_ROWS"] = str(autotune[best][0])
        autotune = {"chosen": best, "ms_per_cycle": {k: round(1e3 * v[1], 4) for k, v in autotune.items()}}
    for _ in range(args.warmup):
        r.cycle(pre, post, want_norm=False)
    times = []
    for _ in range(max(1, getattr(args, "repeats", 1))):
        r.sync()
        torch.cuda.synchronize()
        td.barrier()
        t0 = time.perf_counter()
        region_norms = r.cycles(pre, post, args.steps)  # K cycles back to back, every cycle's global norm computed
        r.sync()
        torch.cuda.synchronize()
        td.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)           # the slowest rank's time
        times.append(float(t[0]))
    elapsed = statistics.median(times)
    norm = r.cycle(pre, post, want_norm=True)
    # rank 0's fine-grid SpMV over its own rows (untimed region, hipEvents on the rank's stream)
    spmv_ms = r.spmv_time(20)
    fmt = r.format_info(0, "A")

    rccl_ranks = r.rccl_ranks()
    if rank == 0:
        fmt_bytes = fmt["format_bytes"] + 2 * w * n_loc
        csr_bytes = (w + 4) * nnz_loc + 4 * (n_loc + 1) + 2 * w * n_loc
        n_glob = part.n_rows(0)
        equiv = n_glob / float(256 ** 3)
        out = {
            "metric": "V-cycles/sec (256^3-unknown equivalents), 3-D %s, weak scaling"
                      % ("7-point Poisson" if args.stencil != "27var" else "27-point variable-coefficient Poisson"),
            "value": round(args.steps / elapsed * equiv, 3),
            "unit": "256^3-equivalent V-cycles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "3-D %s %s, %d-grid V(1,1) cycle, %s Gauss-Seidel, %s, int32 CSR, "
                                   "1-D slabs over %d GPUs, RCCL halo exchange"
                                   % ("7-point Poisson" if args.stencil != "27var" else "27-point variable-coefficient Poisson (Q1 stiffness)",
                                      "x".join(map(str, shape)), grids,
                                      ("red-black" if colouring == "parity" else "8-colour") if args.smoother == "colour" else args.smoother,
                                      "fp64" if w == 8 else "fp32", world),
                       "unknowns": n_glob, "unknowns_per_gpu": n_loc, "nnz_per_gpu": nnz_loc, "grids": grids,
                       "distributed_grids": n_dist - 1, "replicated_tail_grids": grids - n_dist + 1,
                       "rccl_ranks": rccl_ranks, "repeats": len(times), "overlap_autotune": autotune,
                       "ranks_share_one_gpu": os.environ.get("OMG_DIST_SHARED_GPU", "0") in ("1", "rccl") and world > 1,
                       "kernel_src_sha": _bench_identity()[0], "git_head": _bench_identity()[1],
                       "ms_per_step_all": [round(1e3 * t / args.steps, 4) for t in times],
                       "pre": pre, "post": post, "cycles_per_s": round(args.steps / elapsed, 3),
                       "final_residual_norm": norm, "norms_last_region_tail": region_norms[-3:],
                       "setup_s": round(setup_s, 2)},
            # (the set-by-set runner measures no one-GPU baselines: the keys are there, empty)
            "vs_n1_config2": None, "vs_one_gpu_same_problem": None, "one_gpu_baselines": None,
            # rank 0's y = A x over its own rows: bytes the launch has to move with the operator in
            # its device format over the launch time, frac <= 1 by construction;
            # the rate in SURVEY 8(d)'s plain-CSR bytes is csr_equiv_GBps
            "roofline": {"bound": "hbm", "kernel": "y = A_local x on rank 0 (all local rows; x incl. halo)",
                         "achieved": round(fmt_bytes / spmv_ms / 1e6, 1),
                         "peak": 8000.0, "unit": "GB/s",
                         "frac": round(fmt_bytes / spmv_ms / 1e6 / 8000.0, 4),
                         "traffic": None, "avg_launch_us": round(1e3 * spmv_ms, 2),
                         "bytes_per_launch": fmt_bytes,
                         "csr_equiv_bytes": csr_bytes, "csr_equiv_GBps": round(csr_bytes / spmv_ms / 1e6, 1),
                         "device_format": {k: fmt[k] for k in ("rows", "nnz", "pattern_rows", "coldict_nnz", "valdict_nnz")}},
            "cpu_baseline": None,
        }
        # RCCL (NCCL_DEBUG=VERSION) and gloo write banners through C stdio; push them out first
        # so that the JSON line is the LAST line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out))
        sys.stdout.flush()
    r.close()
    watchdog.cancel()
    td.barrier()
    td.destroy_process_group()
    return 0
