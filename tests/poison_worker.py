"""Run by tests/test_gpu_poison.py in a process of its own with OMG_POISON=1: every fresh device allocation is filled
with NaN patterns (common.h DevBuf::alloc), so a kernel that reads what nobody wrote — harmless in a young process,
where the allocator hands out zeros — shows up as NaNs or as a difference from the set-by-set schedule."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

from openmg_amd import _hip, operators


def aggregation(shape):
    mats = []
    for s in shape:
        m = sp.lil_matrix((s // 2, s))
        for i in range(s // 2):
            m[i, 2 * i] = 0.5
            m[i, 2 * i + 1] = 0.5
        mats.append(sp.csr_matrix(m))
    R = mats[0]
    for m in mats[1:]:
        R = sp.kron(R, m, format="csr")
    R = sp.csr_matrix(R)
    R.sort_indices()
    return R


def hierarchy(A0, shape, grids):
    A = [sp.csr_matrix(A0)]
    R = []
    sh = tuple(shape)
    for _ in range(grids - 1):
        R.append(aggregation(sh))
        Ac = sp.csr_matrix((R[-1] @ A[-1]) @ R[-1].T)
        Ac.sort_indices()
        A.append(Ac)
        sh = tuple(s // 2 for s in sh)
    return A, R


def hierarchy27(shape, grids):
    return hierarchy(operators.stencil27_variable(shape, seed=2024), shape, grids)


def hierarchy7(shape, grids):
    return hierarchy(operators.stencil_poisson(shape), shape, grids)


def both_ways(A, R, dtype, flag, sweeps, smoother="colour"):
    rng = np.random.default_rng(7)
    b = A[0] @ rng.random(A[0].shape[0])
    x0 = rng.standard_normal(A[0].shape[0])
    if dtype == "float32":
        b, x0 = b.astype(np.float32).astype(np.float64), x0.astype(np.float32).astype(np.float64)
    out = []
    with _hip.Hierarchy(A, R, smoother=smoother, dtype=dtype) as h:
        assert h.level_flags(0)[flag], (flag, h.level_flags(0))
        for pre, post in sweeps:
            res = []
            for fused in (True, False):
                h.use_plane(fused)
                h.resident_load(b, x0)
                norms = [h.resident_cycle(pre, post) for _ in range(3)]
                res.append((norms, h.resident_fetch()))
            same = bool(np.array_equal(res[0][1], res[1][1]))
            finite = bool(np.all(np.isfinite(res[0][1])) and np.all(np.isfinite(res[0][0])))
            out.append({"sweeps": [pre, post], "same_bits": same, "finite": finite,
                        "norm_rel": float(abs(res[0][0][-1] - res[1][0][-1]) / abs(res[1][0][-1]))})
    return out


def main():
    report = {}
    for dtype in ("float64", "float32"):
        A, R = hierarchy27((32, 32, 32), 4)
        report["stencil27 32^3 %s" % dtype] = both_ways(A, R, dtype, "stencil27", ((1, 1), (2, 1), (0, 1), (1, 0)))
        A, R = hierarchy27((12, 8, 36), 2)
        report["stencil27 12x8x36 %s" % dtype] = both_ways(A, R, dtype, "stencil27", ((1, 1), (1, 2)))
        A, R = hierarchy7((64, 64, 64), 4)
        report["plane 64^3 %s" % dtype] = both_ways(A, R, dtype, "plane", ((1, 1), (1, 0), (0, 1), (2, 2)))
        A, R = hierarchy7((20, 12, 24), 2)
        report["plane 20x12x24 %s" % dtype] = both_ways(A, R, dtype, "plane", ((1, 1), (2, 1)))
        # 7-point operators with per-row coefficients (var7.hip, round 6): ragged tiles, three levels through the passes
        os.environ["OMG_VAR7_MIN"] = "4096"
        A, R = hierarchy(operators.stencil7_variable((40, 24, 72)), (40, 24, 72), 3)
        report["var7 40x24x72 %s" % dtype] = both_ways(A, R, dtype, "var7", ((1, 1), (1, 0), (0, 1), (2, 1)))
    A, R = hierarchy7((128, 64), 3)
    report["tiles 128x64 red-black"] = both_ways(A, R, "float64", "plane", ((1, 1), (1, 0)))
    report["tiles 128x64 jacobi"] = both_ways(A, R, "float64", "plane", ((1, 1), (2, 1)), smoother="jacobi")
    print(json.dumps(report))


if __name__ == "__main__":
    sys.exit(main())
