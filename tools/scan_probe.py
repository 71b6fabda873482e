"""The line-scan lexicographic sweep (OMG_MARCH_SCAN=1, march.hip scan_gs_kernel) against the bit-exact wavefront kernel:
largest relative difference after a few sweeps, and the time of a sweep, per grid shape.
    python tools/scan_probe.py [check|time] [size ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from openmg_amd import _hip, operators


def sweeps(A, b, x0, its, scan, dtype="float64"):
    os.environ["OMG_MARCH_SCAN"] = "1" if scan else "0"
    x = x0.copy()
    t0 = time.perf_counter()
    assert _hip.gauss_seidel(A, b, x, smoother="gs", iterations=its) == its
    return x, time.perf_counter() - t0


def check(shapes):
    rng = np.random.default_rng(5)
    for shape in shapes:
        A = operators.stencil_poisson(shape)
        f = rng.choice(np.array([1.0, 0.5, 3.0]), size=A.shape[0])
        A = sp.csr_matrix(sp.diags(f) @ A)
        A.sort_indices()
        n = A.shape[0]
        b, x0 = rng.standard_normal(n), rng.standard_normal(n)
        for its in (1, 3):
            got, _ = sweeps(A, b, x0, its, True)
            ref, _ = sweeps(A, b, x0, its, False)
            err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
            print("shape %-16s sweeps %d: max |scan - march| / max |march| = %.2e %s" % (shape, its, err, "" if err < 1e-13 else "  <-- BAD"), flush=True)


def timing(sizes):
    for size in sizes:
        shape = (size,) * 3
        A0 = operators.stencil_poisson(shape)
        b = A0 @ np.random.default_rng(1).random(A0.shape[0])
        grids = max(2, int(np.log2(size)) - 3)
        R = operators.restrictionList(shape, grids - 2, 8)
        A = operators.coeffecientList(A0, R)
        for scan in (False, True):
            os.environ["OMG_MARCH_SCAN"] = "1" if scan else "0"
            h = _hip.Hierarchy(A, R, smoother="gs")
            h.resident_load(b)
            norms = [h.resident_cycle(1, 1) for _ in range(3)]
            h.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                h.resident_cycle(1, 1, want_norm=False)
            h.sync()
            dt = (time.perf_counter() - t0) / 5
            print("%d^3 %d grids scan=%d: %.3f ms per cycle, %.1f cycles/s; norms %s" % (size, len(A), scan, 1e3 * dt, 1 / dt, ["%.10e" % v for v in norms]), flush=True)
            h.close()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what == "sweep":
        # python tools/scan_probe.py sweep 4x256x256 ...  (under rocprofv3 --kernel-trace + tools/scan_trace.py: time per launch)
        rng = np.random.default_rng(5)
        for spec in sys.argv[2:]:
            shape = tuple(int(v) for v in spec.split("x"))
            A = operators.stencil_poisson(shape)
            n = A.shape[0]
            b, x0 = rng.standard_normal(n), rng.standard_normal(n)
            sweeps(A, b, x0, 10, True)
    elif what == "check":
        check([(8, 8, 8), (12, 20, 30), (17, 9, 33), (5, 64, 16), (33, 5, 7), (48, 48, 48), (64, 64, 64), (40, 72, 64), (100, 30, 20), (10, 8, 300), (9, 12, 130), (6, 7, 100), (20, 20, 256)])
    else:
        timing([int(v) for v in sys.argv[2:]] or [64, 128])
