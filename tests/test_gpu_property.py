"""Property-based parity (hypothesis): arbitrary small CSR operators — empty rows, duplicate
column entries, unsorted columns, unsymmetric patterns, 1x1 — through the C ABI against the
oracle's C loops.  Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest
import scipy.sparse as sp
from hypothesis import HealthCheck, given, settings, strategies as st

from openmg_amd import _hip
from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu

COMMON = dict(deadline=None, max_examples=60, suppress_health_check=list(HealthCheck), derandomize=True)


@st.composite
def csr_matrices(draw, square=False, dominant=False):
    n = draw(st.integers(1, 70))
    m = n if square else draw(st.integers(1, 70))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    density = draw(st.sampled_from([0.0, 0.02, 0.1, 0.4, 1.0]))
    rng = np.random.default_rng(seed)
    counts = rng.binomial(m, density, size=n)
    if draw(st.booleans()) and n > 2:
        counts[rng.integers(n)] = 0                                   # an empty row
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    # columns drawn WITH replacement: duplicates stay as separate stored entries
    indices = rng.integers(0, m, size=int(indptr[-1])).astype(np.int32)
    data = rng.standard_normal(int(indptr[-1]))
    A = sp.csr_matrix((data, indices, indptr), shape=(n, m))
    if dominant:
        rows = np.repeat(np.arange(n), np.diff(indptr))
        absum = np.bincount(rows, weights=np.abs(data), minlength=n)
        # append a dominant diagonal as one more stored entry of every row (position random)
        parts_i, parts_d, ptr = [], [], [0]
        for i in range(n):
            s, e = indptr[i], indptr[i + 1]
            ci = np.concatenate([indices[s:e], [i]])
            cd = np.concatenate([data[s:e], [absum[i] + 1.0 + rng.random()]])
            q = rng.permutation(ci.size)
            parts_i.append(ci[q]); parts_d.append(cd[q]); ptr.append(ptr[-1] + ci.size)
        A = sp.csr_matrix((np.concatenate(parts_d), np.concatenate(parts_i).astype(np.int32),
                           np.array(ptr, dtype=np.int32)), shape=(n, n))
    A.has_sorted_indices = False
    return A, seed


def oracle_csr(A):
    return (np.ascontiguousarray(A.indptr, np.int32), np.ascontiguousarray(A.indices, np.int32),
            np.ascontiguousarray(A.data, np.float64))


@settings(**COMMON)
@given(csr_matrices())
def test_spmv_any_csr(case):
    A, seed = case
    x = np.random.default_rng(seed + 1).standard_normal(A.shape[1])
    want = np.empty(A.shape[0])
    orc._clib().oracle_spmv(A.shape[0], *oracle_csr(A), x, want)
    np.testing.assert_allclose(_hip.spmv(A, x), want, rtol=1e-12, atol=1e-12)


@settings(**COMMON)
@given(csr_matrices(square=True))
def test_residual_and_norm_any_csr(case):
    A, seed = case
    rng = np.random.default_rng(seed + 2)
    x, b = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    want = np.empty(A.shape[0])
    orc._clib().oracle_residual(A.shape[0], *oracle_csr(A), b, x, want)
    r, norm = _hip.residual(A, b, x, want_norm=True)
    np.testing.assert_allclose(r, want, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(norm, np.linalg.norm(want), rtol=1e-12, atol=1e-14)


@settings(**COMMON)
@given(csr_matrices(square=True, dominant=True), st.sampled_from(["gs", "colour", "jacobi"]), st.integers(1, 3))
def test_smoothers_any_dominant_csr(case, smoother, sweeps):
    """Lexicographic (level-scheduled), greedy multi-colour and Jacobi sweeps on arbitrary
    diagonally dominant operators — unsymmetric patterns and duplicate entries included."""
    A, seed = case
    n = A.shape[0]
    rng = np.random.default_rng(seed + 3)
    b, x0 = rng.standard_normal(n), rng.standard_normal(n)
    x = x0.copy()
    assert _hip.gauss_seidel(A, b, x, smoother=smoother, omega=0.7, iterations=sweeps) == sweeps
    if smoother == "gs":
        want = orc.gauss_seidel(A, b, x0.copy(), iterations=sweeps)
    elif smoother == "colour":
        want = orc.gs_ordered(A, b, x0.copy(), orc.colour_order(orc.greedy_colouring(A)), sweeps)
    else:
        want = orc.jacobi(A, b, x0.copy(), sweeps, 0.7)
    np.testing.assert_allclose(x, want, rtol=1e-10, atol=1e-12)


@settings(**dict(COMMON, max_examples=25))
@given(csr_matrices(), st.integers(0, 2 ** 31 - 1))
def test_spgemm_any_csr(case, seed2):
    X, seed = case
    rng = np.random.default_rng(seed2)
    k = int(rng.integers(1, 40))
    Y = sp.random(X.shape[1], k, density=0.2, random_state=np.random.RandomState(seed2 % (2 ** 31)), format="csr")
    C = _hip.spgemm(X, Y)
    W = sp.csr_matrix(X @ Y)
    assert C.shape == W.shape
    assert abs(C - W).max() <= 1e-12 * max(1.0, abs(W).max() if W.nnz else 1.0)


# ------------------------------------------------ structured operators: every coding, same bits --
@st.composite
def banded_operators(draw):
    """Translation-invariant-ish operators: a random set of (column - row) offsets (1..40 of them:
    rows below and above the 16-entry association threshold), values constant per offset,
    random per entry, or a mix; optionally a few rows knocked out or perturbed."""
    n = draw(st.integers(300, 3000))
    k = draw(st.integers(1, 40))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    values = draw(st.sampled_from(["constant", "variable", "mixed"]))
    rng = np.random.default_rng(seed)
    offs = np.unique(np.concatenate([[0], rng.integers(-min(n - 1, 200), min(n - 1, 200) + 1, size=k)]))
    rng.shuffle(offs)                                                # stored order != sorted order
    rows, cols, vals = [], [], []
    for o in offs:
        i = np.arange(max(0, -o), min(n, n - o))
        rows.append(i); cols.append(i + o)
        if values == "constant" or (values == "mixed" and rng.random() < 0.5):
            vals.append(np.full(i.size, rng.standard_normal() if o else 4.0 + len(offs)))
        else:
            vals.append(rng.standard_normal(i.size) if o else 4.0 + len(offs) + rng.random(i.size))
    order = np.argsort(np.concatenate(rows), kind="stable")           # row-major, offsets in the shuffled order
    r, c, v = (np.concatenate(a)[order] for a in (rows, cols, vals))
    indptr = np.searchsorted(r, np.arange(n + 1)).astype(np.int32)
    A = sp.csr_matrix((v, c.astype(np.int32), indptr), shape=(n, n))
    A.has_sorted_indices = False
    if draw(st.booleans()):                                          # a few irregular rows
        for i in rng.integers(0, n, size=3):
            s, e = A.indptr[i], A.indptr[i + 1]
            A.data[s:e] *= 1.0 + rng.random(e - s)
    return A, seed


@settings(deadline=None, max_examples=25, suppress_health_check=list(HealthCheck), derandomize=True)
@given(banded_operators(), st.sampled_from(["float64", "float32"]), st.sampled_from(["colour", "jacobi"]))
def test_every_device_coding_gives_the_same_bits(case, dtype, smoother):
    """Two-level cycles on random banded operators under OMG_COMPRESS = 0 (plain CSR), 3 (entry
    dictionaries), 7 (+ row patterns), 15 (+ offset patterns with ELL values): identical iterates,
    and the fp64 plain one agrees with SciPy to rounding."""
    import os
    A0, seed = case
    n = A0.shape[0]
    rng = np.random.default_rng(seed + 7)
    pairs = np.arange(n) // 2
    R = sp.csr_matrix((np.full(n, 0.5), (pairs, np.arange(n))), shape=(int(pairs[-1]) + 1, n))
    Ac = sp.csr_matrix(R @ A0 @ R.T)
    if Ac.shape[0] > 2000:
        pytest.skip("coarse level too large for the dense inverse in a property test")
    b, x0 = rng.standard_normal(n), rng.standard_normal(n)
    out = {}
    keep = os.environ.get("OMG_COMPRESS")
    try:
        for mode in ("0", "3", "7", "15"):
            os.environ["OMG_COMPRESS"] = mode
            with _hip.Hierarchy([A0, Ac], [R], smoother=smoother, omega=0.5, dtype=dtype) as h:
                r = h.residual(0, b, x0)
                h.resident_load(b, x0)
                norms = [h.resident_cycle(1, 1) for _ in range(2)]
                out[mode] = (r, h.resident_fetch(), norms)
    finally:
        if keep is None:
            os.environ.pop("OMG_COMPRESS", None)
        else:
            os.environ["OMG_COMPRESS"] = keep
    for mode in ("3", "7", "15"):
        assert np.array_equal(out[mode][0], out["0"][0]), mode
        assert np.array_equal(out[mode][1], out["0"][1]), mode
        np.testing.assert_allclose(out[mode][2], out["0"][2], rtol=1e-12)
    if dtype == "float64":
        scale = abs(A0).sum(axis=1).max() * max(1.0, np.abs(x0).max()) + np.abs(b).max()
        np.testing.assert_allclose(out["0"][0], b - A0 @ x0, rtol=0, atol=1e-13 * scale)
