#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference; the GPU box never runs
this).  The reference (tsbertalan/openmg) is Python 2, so this script makes a
throw-away translated copy under a temp dir (never committed, never shipped):

    1. copy /root/reference/openmg to $TMP
    2. python3 -m lib2to3 -w -n   (syntax only)
    3. three semantic edits for Python-2 integer division and one for a SciPy
       module path that no longer exists:
         operators.py:52   N / (2 ** alpha)              -> //
         operators.py:131  problemShape / (2 ** level)   -> //
         operators.py:136  same                          -> //
         __init__.py:227   scipy.sparse.base.np.linalg.norm -> np.linalg.norm

and then imports it and records input/output pairs with fixed seeds.  Only
data (arrays, scalars) is written to the .npz files; no reference source.

Usage:  python tests/golden/make_golden.py [--only g1,g3]
"""
import argparse
import importlib
import os
import re
import shutil
import subprocess
import sys
import tempfile
import warnings

import numpy as np
import scipy.sparse as sp

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    tmp = tempfile.mkdtemp(prefix="openmg_ref_py3_")
    shutil.copytree(os.path.join(REFERENCE, "openmg"), os.path.join(tmp, "openmg"))
    for root, _, files in os.walk(tmp):
        os.chmod(root, 0o755)
        for f in files:
            os.chmod(os.path.join(root, f), 0o644)
    subprocess.run([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", "openmg"],
                   cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ops = os.path.join(tmp, "openmg", "operators.py")
    src = open(ops).read()
    src, n1 = re.subn(r"n = N / \(2 \*\* alpha\)", "n = N // (2 ** alpha)", src)
    src, n2 = re.subn(r"np\.array\(problemShape\) / \(2 \*\* level\)",
                      "np.array(problemShape) // (2 ** level)", src)
    assert (n1, n2) == (1, 2), (n1, n2)
    open(ops, "w").write(src)
    ini = os.path.join(tmp, "openmg", "__init__.py")
    src = open(ini).read()
    src, n3 = re.subn(r"scipy\.sparse\.base\.np\.linalg\.norm", "np.linalg.norm", src)
    assert n3 == 1
    open(ini, "w").write(src)
    sys.path.insert(0, tmp)
    ref = importlib.import_module("openmg")
    return ref, tmp


# ---------------------------------------------------------------- inputs --
def lap1d(n):
    return sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")


def lap_nd(shape):
    """Dirichlet (2*dim, -1) Laplacian in C-order numbering (SURVEY 8d inputs)."""
    mats = [lap1d(s) for s in shape]
    eyes = [sp.identity(s, format="csr") for s in shape]
    total = None
    for d in range(len(shape)):
        term = None
        for e in range(len(shape)):
            f = mats[e] if e == d else eyes[e]
            term = f if term is None else sp.kron(term, f, format="csr")
        total = term if total is None else total + term
    out = sp.csr_matrix(total)
    out.sort_indices()
    return out


def parity_perm(shape):
    """Red-first ordering: cells with even coordinate sum first (stable)."""
    idx = np.indices(shape).reshape(len(shape), -1)
    colour = idx.sum(axis=0) % 2
    return np.argsort(colour, kind="stable"), colour


def csr_parts(prefix, M):
    M = sp.csr_matrix(M)
    return {prefix + "_indptr": M.indptr.astype(np.int32),
            prefix + "_indices": M.indices.astype(np.int32),
            prefix + "_data": M.data.astype(np.float64),
            prefix + "_shape": np.array(M.shape, dtype=np.int64)}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def run_cycles(ref, A_in, b, params, snapshots):
    """mgSolve's setup followed by explicit mgCycle calls, snapshotting x."""
    p = dict(params)
    p["cycles"] = 1
    p["threshold"] = 0
    p["giveInfo"] = True
    _, info = ref.mgSolve(A_in, b.copy(), p)
    A, R = info["A"], info["R"]
    xs, norms = {}, []
    x = None
    for c in range(1, max(snapshots) + 1):
        x, inf = ref.mgCycle(A, b, 0, R, p, initial=x)
        norms.append(float(inf["norm"]))
        if c in snapshots:
            xs[c] = np.array(x, dtype=np.float64).ravel().copy()
    return A, R, xs, np.array(norms)


# --------------------------------------------------------------- fixtures --
def g1(ref):
    """simpleDemo known-answer trace (openmg_usage_demo.py:27-67)."""
    N = 100
    u_true = np.array([np.sin(x / 10.0) for x in np.linspace(0, 20, N)])
    A = ref.operators.poisson(N, sparse=True)
    b = ref.tools.flexibleMmult(A, u_true)
    out = {"b": np.asarray(b).ravel(), "u_true": u_true}
    out.update(csr_parts("A", A))
    for gl in (2, 3):
        for dense in (True, False):
            norms = []
            p = {"problemShape": (N,), "gridLevels": gl, "cycles": 10, "iterations": 2,
                 "verbose": False, "dense": dense, "threshold": 1e-2, "giveInfo": True}
            u, info = ref.mgSolve(A, np.asarray(b).ravel().copy(), p)
            # re-run cycle by cycle to record the whole trace
            pp = dict(p)
            x = None
            for _ in range(info["cycle"]):
                x, inf = ref.mgCycle(info["A"], np.asarray(b).ravel(), 0, info["R"], pp, initial=x)
                norms.append(float(inf["norm"]))
            tag = "gl%d_%s" % (gl, "dense" if dense else "sparse")
            out[tag + "_norms"] = np.array(norms)
            out[tag + "_u"] = np.asarray(u, dtype=np.float64).ravel()
            out[tag + "_cycle"] = np.array(info["cycle"])
    save("g1_simple_demo", **out)


def g2(ref):
    """1-D sparse (4,-1) N=4096, gridLevels=2 (3 grids), V(1,0) and V(1,1)."""
    N = 4096
    A = sp.csr_matrix(ref.operators.poisson(N, sparse=True))
    u_true = np.random.default_rng(12345).random(N)
    b = A @ u_true
    out = {"b": b, "u_true": u_true}
    for post in (0, 1):
        p = {"problemShape": (N,), "gridLevels": 2, "preIterations": 1, "postIterations": post,
             "verbose": False}
        _, _, xs, norms = run_cycles(ref, A, b, p, (1, 2, 5))
        for c, x in xs.items():
            out["v1%d_x_c%d" % (post, c)] = x
        out["v1%d_norms" % post] = norms
    save("g2_poisson1d_4096", **out)


def g3(ref):
    """3-D 7-point Dirichlet Laplacian, 16^3 full hierarchy and 32^3 traces."""
    for n, detail in ((16, True), (32, False)):
        shape = (n, n, n)
        A0 = lap_nd(shape)
        u_true = np.random.default_rng(12345).random(n ** 3)
        b = A0 @ u_true
        out = {"b": b, "shape": np.array(shape)}
        for pre, post in ((1, 0), (1, 1)) if detail else ((1, 1),):
            p = {"problemShape": shape, "gridLevels": 2, "preIterations": pre,
                 "postIterations": post, "verbose": False, "minSize": 8}
            A, R, xs, norms = run_cycles(ref, A0, b, p, (1, 3))
            tag = "v%d%d" % (pre, post)
            for c, x in xs.items():
                out["%s_x_c%d" % (tag, c)] = x
            out[tag + "_norms"] = norms
        out["n_levels"] = np.array(len(A))
        if detail:
            for l, M in enumerate(A):
                out.update(csr_parts("A%d" % l, M))
            for l, M in enumerate(R):
                out.update(csr_parts("R%d" % l, M))
        save("g3_poisson3d_%d" % n, **out)


def g4(ref):
    """Red-black pin: the reference's own lexicographic gaussSeidel / mgCycle run on
    red-first permuted operators equals colour-ordered Gauss-Seidel in natural numbering."""
    out = {}
    # (a) single sweeps on permuted 5- and 7-point matrices
    for tag, shape in (("p5", (16, 16)), ("p7", (8, 8, 8))):
        A = lap_nd(shape)
        n = A.shape[0]
        perm, colour = parity_perm(shape)
        Ap = sp.csr_matrix(A[perm][:, perm])
        rng = np.random.default_rng(777)
        b = rng.random(n)
        x0 = rng.random(n)
        xp = ref.solvers.gaussSeidel(Ap, b[perm].copy(), x0[perm].copy(), iterations=2)
        x = np.empty(n)
        x[perm] = xp
        out[tag + "_b"] = b
        out[tag + "_x0"] = x0
        out[tag + "_x_after2"] = x
        out[tag + "_shape"] = np.array(shape)
    # (b) whole V-cycles with a permuted hierarchy, 16^3, 3 grids, V(1,1)
    shape = (16, 16, 16)
    A0 = lap_nd(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    p = {"problemShape": shape, "gridLevels": 2, "preIterations": 1, "postIterations": 1,
         "verbose": False, "minSize": 8, "cycles": 1, "threshold": 0, "giveInfo": True}
    _, info = ref.mgSolve(A0, b.copy(), dict(p))
    A, R = info["A"], info["R"]
    shapes = [tuple(s // (2 ** l) for s in shape) for l in range(len(A))]
    perms = [parity_perm(s)[0] for s in shapes]
    Ap = [sp.csr_matrix(M[q][:, q]) for M, q in zip(A, perms)]
    Rp = [sp.csr_matrix(M[perms[l + 1]][:, perms[l]]) for l, M in enumerate(R)]
    pp = dict(p)
    pp["coarsestLevel"] = len(R)
    x = None
    norms = []
    for c in range(3):
        x, inf = ref.mgCycle(Ap, b[perms[0]], 0, Rp, pp, initial=x)
        norms.append(float(inf["norm"]))
    xn = np.empty_like(x)
    xn[perms[0]] = x
    out["vc_b"] = b
    out["vc_x_c3"] = xn
    out["vc_norms"] = np.array(norms)
    out["vc_shape"] = np.array(shape)
    save("g4_redblack_pin", **out)


def g5(ref):
    """restriction() outputs, incl. the unequal-extent stride quirk (operators.py:46,78-81)
    and odd-extent truncation (operators.py:74)."""
    out = {}
    cases = [(8,), (8, 8), (8, 8, 8), (4, 8), (8, 4), (5,), (6, 6, 6), (16,), (4, 4, 8)]
    names = []
    for shape in cases:
        tag = "s" + "x".join(str(s) for s in shape)
        try:
            R = ref.operators.restriction(shape)
        except Exception as e:  # record which inputs the reference rejects
            out[tag + "_error"] = np.array(type(e).__name__)
            names.append(tag)
            continue
        out.update(csr_parts(tag, R))
        names.append(tag)
    out["cases"] = np.array(names)
    # restrictionList depth rule (D5) and minSize rule
    rl = []
    for shape, coarsest, minsize in (((64,), 1, 8), ((64,), 3, 8), ((64,), 5, 8), ((1024,), 23, 23),
                                     ((16, 16), 2, 8), ((16, 16, 16), 3, 8), ((200,), 2, 30)):
        R = ref.operators.restrictionList(shape, coarsest, minsize)
        rl.append([len(shape), coarsest, minsize, len(R)] + list(R[-1].shape) + list(shape) + [0] * (3 - len(shape)))
    out["restrictionList_cases"] = np.array(rl, dtype=np.int64)
    save("g5_restriction", **out)


def g7(ref):
    """Stop rules, dict mutation, generators, standalone smoother."""
    out = {}
    size = 36
    u_actual = np.sin(np.array(range(int(size))) * 3.0 / size).T
    A = ref.operators.poisson((size,))
    b = ref.tools.flexibleMmult(A, u_actual)
    out["stop_A"] = np.asarray(A)
    out["stop_b"] = np.asarray(b).ravel()
    # tests.py:502-515
    p = {"problemShape": (size,), "gridLevels": 2, "threshold": 8e-3, "giveInfo": True}
    keys_before = sorted(p.keys())
    u, info = ref.mgSolve(A, np.asarray(b).ravel().copy(), p)
    out["thresh_u"] = np.asarray(u).ravel()
    out["thresh_cycle"] = np.array(info["cycle"])
    out["thresh_norm"] = np.array(info["norm"])
    out["thresh_keys_before"] = np.array(keys_before)
    out["thresh_keys_after"] = np.array(sorted(p.keys()))
    out["thresh_coarsestLevel_after"] = np.array(p["coarsestLevel"])
    # tests.py:517-531
    p = {"problemShape": (size,), "gridLevels": 2, "cycles": 3, "threshold": 1e-10, "giveInfo": True}
    u, info = ref.mgSolve(A, np.asarray(b).ravel().copy(), p)
    out["cyc_u"] = np.asarray(u).ravel()
    out["cyc_cycle"] = np.array(info["cycle"])
    out["cyc_norm"] = np.array(info["norm"])
    # tests.py:558-570 (minSize)
    shape = (1024,)
    u_actual = np.random.default_rng(5).random(shape).ravel()
    A_in = ref.operators.poisson(shape)
    bb = np.asarray(ref.tools.flexibleMmult(A_in, u_actual)).ravel()
    p = {"problemShape": shape, "gridLevels": 24, "iterations": 1, "verbose": False,
         "threshold": 4, "giveInfo": True, "minSize": 23}
    soln, info = ref.mgSolve(A_in, bb.copy(), p)
    out["minsize_b"] = bb
    out["minsize_R_shapes"] = np.array([r.shape for r in info["R"]], dtype=np.int64)
    out["minsize_cycle"] = np.array(info["cycle"])
    out["minsize_norm"] = np.array(info["norm"])
    out["minsize_soln"] = np.asarray(soln).ravel()
    out["minsize_coarsestLevel_after"] = np.array(p["coarsestLevel"])
    # tests.py:58-81 (test_a): 1-D dense operator with a 3-D problemShape
    ps = 12
    sz = ps ** 3
    u_actual = np.sin(np.array(range(int(sz))) * 3.0 / sz).T
    A3 = ref.operators.poisson((sz,))
    b3 = np.asarray(ref.tools.flexibleMmult(A3, u_actual)).ravel()
    p = {"coarsestLevel": 3, "problemShape": (ps, ps, ps), "gridLevels": 4, "threshold": 8e-3,
         "giveInfo": True}
    u, info = ref.mgSolve(A3, b3.copy(), p)
    out["testa_b"] = b3
    out["testa_u"] = np.asarray(u).ravel()
    out["testa_cycle"] = np.array(info["cycle"])
    out["testa_norm"] = np.array(info["norm"])
    # generators with their quirks (operators.py:191-279)
    out["gen_p1sparse_8"] = ref.operators.poisson(8, sparse=True).toarray()
    out["gen_p1dense_8"] = np.asarray(ref.operators.poisson((8,)))
    out["gen_p2dense_3x4"] = np.asarray(ref.operators.poisson((3, 4)))
    out["gen_p2dense_4x4"] = np.asarray(ref.operators.poisson((4, 4)))
    out["gen_p3dense_2x3x4"] = np.asarray(ref.operators.poisson((2, 3, 4)))
    out["gen_p3dense_3x3x3"] = np.asarray(ref.operators.poisson((3, 3, 3)))
    # standalone smoother: iterations only, threshold only (tests.py:342-365)
    NX = 12
    A2 = ref.operators.poisson((NX, NX))
    rng = np.random.default_rng(99)
    b2 = rng.random(NX * NX)
    x = ref.solvers.smoothToThreshold(A2, b2.copy(), np.zeros(NX * NX), 1e-4)
    out["gs_thresh_b"] = b2
    out["gs_thresh_x"] = np.asarray(x).ravel()
    A1 = sp.csr_matrix(ref.operators.poisson(64, sparse=True))
    b1 = rng.random(64)
    x0 = rng.random(64)
    out["gs_b"] = b1
    out["gs_x0"] = x0
    out["gs_x_it1"] = ref.solvers.gaussSeidel(A1, b1.copy(), x0.copy()).copy()
    out["gs_x_it3"] = ref.smooth(A1, b1.copy(), x0.copy(), 3).copy()
    xs = ref.smoothToThreshold(A1, b1.copy(), x0.copy(), 1e-6)
    out["gs_x_thr"] = xs.copy()
    # unsorted-column CSR (what RAP produces): sums run in stored order (solvers.py:63-65)
    Au = lap_nd((6, 6))
    Au = sp.csr_matrix(Au)
    rs = np.random.default_rng(3)
    for i in range(Au.shape[0]):
        s, e = Au.indptr[i], Au.indptr[i + 1]
        q = rs.permutation(e - s)
        Au.indices[s:e] = Au.indices[s:e][q]
        Au.data[s:e] = Au.data[s:e][q]
    Au.has_sorted_indices = False
    bu = rs.random(36)
    xu = ref.solvers.gaussSeidel(Au, bu.copy(), np.zeros(36), iterations=2)
    out.update(csr_parts("unsorted_A", Au))
    out["unsorted_b"] = bu
    out["unsorted_x"] = xu.copy()
    # coarse solve
    out["coarse_x"] = ref.coarseSolve(A1, b1.reshape(-1, 1))
    save("g7_stop_rules_misc", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    warnings.simplefilter("ignore")
    ref, tmp = load_reference()
    todo = {"g1": g1, "g2": g2, "g3": g3, "g4": g4, "g5": g5, "g7": g7}
    pick = [s for s in args.only.split(",") if s] or list(todo)
    try:
        for k in pick:
            todo[k](ref)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
