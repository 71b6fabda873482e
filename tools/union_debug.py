#!/usr/bin/env python3
"""Compare the union-walk kernels with the one-row-per-thread pattern kernel operation by operation."""
import os
import sys
import subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(shape, grids):
    from openmg_amd import _hip, operators
    A0 = operators.stencil_poisson(shape)
    R = [operators.restriction(tuple(s // 2 ** l for s in shape)) for l in range(grids - 1)]
    A = operators.coeffecientList(A0, R)
    rng = np.random.default_rng(41)
    out = {}
    with _hip.Hierarchy(A, R, smoother="colour") as h:
        for l in range(grids - 1):
            n = A[l].shape[0]
            b, x = rng.random(n), rng.random(n)
            out["fmt%d" % l] = np.array([h.format_info(l)[k] for k in ("blocks", "pattern_rows")], dtype=float)
            out["res%d" % l] = h.residual(l, b, x)
            xs = x.copy()
            h.smooth(l, b, xs, 1)
            out["gs%d" % l] = xs
        b = A0 @ rng.random(A0.shape[0])
        h.resident_load(b)
        for c in range(2):
            out["norm%d" % c] = np.array([h.resident_cycle(1, 1)])
            out["x%d" % c] = h.resident_fetch()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        shape = tuple(int(v) for v in sys.argv[2].split(","))
        np.savez(sys.argv[4], **run(shape, int(sys.argv[3])))
        sys.exit(0)
    shape, grids = sys.argv[1], sys.argv[2]
    res = {}
    for uk in ("0", "1"):
        env = dict(os.environ, OMG_UNION_KERNEL=uk)
        path = "/tmp/union_debug_%s.npz" % uk
        subprocess.run([sys.executable, __file__, "child", shape, grids, path], env=env, check=True)
        res[uk] = np.load(path)
    for k in res["0"].files:
        a, b = res["0"][k], res["1"][k]
        if k.startswith("fmt"):
            print(k, a, b)
            continue
        bad = np.flatnonzero(a != b)
        print("%-8s %s  differing %d of %d%s" % (k, "same" if bad.size == 0 else "DIFF", bad.size, a.size,
                                               "" if bad.size == 0 else "  first %s  max rel %.3e" % (bad[:8], np.abs(a - b).max() / np.abs(a).max())))
