"""Host-side logic of openmg_amd that needs no GPU: generators (with the reference's
quirks), dict helpers, argument validation, smoother-name table.  CPU only."""
import doctest

import numpy as np
import pytest
import scipy.sparse as sp

import openmg_amd
from openmg_amd import _hip, operators, tools
from oracle import mg_oracle as orc


def test_generators_match_reference_fixtures(golden):
    d = golden("g7_stop_rules_misc")
    np.testing.assert_array_equal(operators.poisson(8, sparse=True).toarray(), d["gen_p1sparse_8"])
    np.testing.assert_array_equal(operators.poisson((8,)), d["gen_p1dense_8"])
    np.testing.assert_array_equal(operators.poisson((3, 4)), d["gen_p2dense_3x4"])
    np.testing.assert_array_equal(operators.poisson((4, 4)), d["gen_p2dense_4x4"])
    np.testing.assert_array_equal(operators.poisson((2, 3, 4)), d["gen_p3dense_2x3x4"])
    np.testing.assert_array_equal(operators.poisson((3, 3, 3)), d["gen_p3dense_3x3x3"])
    with pytest.raises(ValueError):                      # tests.py:533-536
        operators.poisson((1, 2, 3, 4))
    with pytest.raises(NotImplementedError):             # operators.py:224,247
        operators.poisson((4, 4), sparse=True)
    with pytest.raises(NotImplementedError):
        operators.poisson((4, 4, 4), sparse=True)


def test_restriction_argument_errors_precede_device_work():
    with pytest.raises(ValueError):                      # tests.py:544-548
        operators.restriction((4, 4, 4, 4))
    with pytest.raises(ValueError):                      # coarse set of 0 or 1 points
        operators.restriction((2,))
    with pytest.raises(ValueError):
        operators.restriction((1,))


def test_stencil_poisson_shapes():
    A = operators.stencil_poisson((4, 5, 6))
    assert A.shape == (120, 120) and A.nnz == 7 * 120 - 2 * (5 * 6 + 4 * 6 + 4 * 5)
    assert (A.diagonal() == 6).all() and A.has_sorted_indices
    assert abs(A - A.T).max() == 0


def test_dict_helpers_and_defaults():
    assert doctest.testmod(tools).failed == 0
    assert tools.product((3, 4, 5)) == 60
    assert sorted(openmg_amd.defaults) == sorted(
        ["problemShape", "gridLevels", "verbose", "threshold", "cycles", "preIterations",
         "postIterations", "dense", "giveInfo", "minSize"] + (["coarsestLevel"] if "coarsestLevel" in openmg_amd.defaults else []))
    assert openmg_amd.defaults["preIterations"] == 1 and openmg_amd.defaults["postIterations"] == 0
    assert openmg_amd.mg_cycle is openmg_amd.mgCycle


def test_smoother_names():
    assert _hip.smoother_code("gs") == _hip.SMOOTH_GS_LEX
    assert _hip.smoother_code("red-black") == _hip.SMOOTH_GS_COLOUR
    assert _hip.smoother_code("jacobi") == _hip.SMOOTH_JACOBI
    with pytest.raises(ValueError):
        _hip.smoother_code("sor")


def test_as_csr_keeps_stored_order():
    import scipy.sparse as sp
    M = sp.csr_matrix((np.array([1.0, 2.0, 3.0]), np.array([2, 0, 1]), np.array([0, 2, 3])), shape=(2, 3))
    out = _hip.as_csr(M)
    assert out.indices.tolist() == [2, 0, 1] and out.indices.dtype == np.int32
    M64 = sp.csr_matrix((M.data, M.indices.astype(np.int64), M.indptr.astype(np.int64)), shape=(2, 3))
    assert _hip.as_csr(M64).indices.dtype == np.int32


def test_drop_in_alias_package():
    import openmg
    import openmg.tools
    from openmg import operators as ops
    assert openmg.mgSolve is openmg_amd.mgSolve and openmg.mg_cycle is openmg_amd.mgCycle
    assert ops is openmg_amd.operators and openmg.tools is openmg_amd.tools
    assert openmg.defaults is openmg_amd.defaults


# ------------------------------------------------ device format: host-side coder (no GPU) --
def _shuffled_rows(A, rng):
    A = sp.csr_matrix(A)
    for i in range(A.shape[0]):
        s, e = A.indptr[i], A.indptr[i + 1]
        q = rng.permutation(e - s)
        A.indices[s:e] = A.indices[s:e][q]
        A.data[s:e] = A.data[s:e][q]
    A.has_sorted_indices = False
    return A


def test_device_format_is_lossless_and_picks_the_expected_coding(monkeypatch):
    """csrc/setup_host.cpp:encode_csr codes every row block as row patterns, per-entry
    dictionaries or plain CSR; omg_format_selftest decodes the result like the kernels do and
    compares every column and value bit with the input.  Runs on the host."""
    rng = np.random.default_rng(5)
    n1 = 12
    T = sp.diags([np.ones(n1 - 1), np.ones(n1), np.ones(n1 - 1)], [-1, 0, 1])
    stencil27 = sp.csr_matrix(-sp.kron(sp.kron(T, T), T) + sp.diags(np.full(n1 ** 3, 28.0)))
    poisson = orc.stencil_poisson((20, 16, 24))
    variable = sp.csr_matrix(poisson.multiply(sp.csr_matrix((rng.random(poisson.nnz) + 0.5, poisson.indices, poisson.indptr),
                                                           shape=poisson.shape)))
    irregular = sp.random(3000, 3000, density=0.004, random_state=np.random.RandomState(3), format="csr") \
        + sp.diags(np.arange(1.0, 3001.0))
    irregular = _shuffled_rows(irregular, rng)
    long_row = sp.lil_matrix((500, 3000))
    long_row[7, :] = rng.standard_normal(3000)                   # > ROWBLK_NNZ entries: a block of its own
    long_row[8, 3] = -0.0                                        # the sign of a zero is a bit too
    long_row[9, 5] = np.nan
    long_row = sp.csr_matrix(long_row)
    R = orc.restriction((16, 16, 16))
    empty = sp.csr_matrix((40, 40))
    for mode in ("7", "3", "4", "2", "1", "0"):
        monkeypatch.setenv("OMG_COMPRESS", mode)
        for dtype, w in (("float64", 8), ("float32", 4)):
            got = {name: _hip.format_selftest(M, dtype) for name, M in
                   (("poisson", poisson), ("stencil27", stencil27), ("variable", variable), ("irregular", irregular),
                    ("long_row", long_row), ("R", R), ("P", sp.csr_matrix(R.T)), ("empty", empty))}
            for name, f in got.items():
                assert f["csr_bytes"] == f["nnz"] * (4 + w) + 4 * f["rows"], (name, mode)
            if mode == "0":
                for name, f in got.items():
                    assert f["pattern_rows"] == f["coldict_nnz"] == f["valdict_nnz"] == 0
                    assert f["format_bytes"] == f["csr_bytes"] + 32 * f["blocks"] - (4 * f["rows"] if name == "P" else 0)
            if mode == "7":
                assert got["poisson"]["pattern_rows"] == got["poisson"]["rows"]
                assert got["stencil27"]["pattern_rows"] == got["stencil27"]["rows"]
                assert got["poisson"]["format_bytes"] < 0.2 * got["poisson"]["csr_bytes"]
                assert got["variable"]["pattern_rows"] == 0 and got["variable"]["valdict_nnz"] == 0
                assert got["variable"]["coldict_nnz"] == got["variable"]["nnz"]
                assert got["irregular"]["pattern_rows"] == 0
                assert got["R"]["valdict_nnz"] + got["R"]["pattern_nnz"] == got["R"]["nnz"]      # all entries 0.125
                assert got["P"]["valdict_nnz"] == got["P"]["nnz"] and got["P"]["pattern_rows"] == 0
                assert got["empty"]["nnz"] == 0
    with pytest.raises(_hip.HipError):                          # a bad dtype is an argument error, not a crash
        out = (_hip.ctypes.c_int64 * 10)()
        v = _hip.csr_view(_hip.as_csr(poisson))
        _hip.check(_hip.lib().omg_format_selftest(_hip.ctypes.byref(v), 9, out))


def test_stencil27_variable_is_the_q1_stiffness_survey_8d_specifies():
    """configs[4]'s input: assembled directly in CSR; compared here with a cell-by-cell
    assembly of kappa_c * K_e over all cells touching an interior node."""
    K, corners = operators._q1_element_stiffness()
    np.testing.assert_allclose(K[0] * 12, [4, 0, 0, -1, 0, -1, -1, -1], atol=1e-12)
    np.testing.assert_allclose(K.sum(axis=1), 0, atol=1e-12)
    shape = (4, 3, 5)
    N = int(np.prod(shape))
    kappa = np.exp(np.random.default_rng(2024).uniform(-1, 1, size=tuple(s + 1 for s in shape)) * np.log(10))
    want = np.zeros((N, N))
    flat = lambda p: (p[0] * shape[1] + p[1]) * shape[2] + p[2]
    inside = lambda p: all(0 <= p[k] < shape[k] for k in range(3))
    for c in np.ndindex(*kappa.shape):
        nodes = [(c[0] - 1 + a[0], c[1] - 1 + a[1], c[2] - 1 + a[2]) for a in corners]
        for ia, na in enumerate(nodes):
            for ib, nb in enumerate(nodes):
                if inside(na) and inside(nb):
                    want[flat(na), flat(nb)] += kappa[c] * K[ia, ib]
    A = operators.stencil27_variable(shape)
    np.testing.assert_allclose(A.toarray(), want, rtol=0, atol=1e-13)
    assert A.has_sorted_indices and A.indices.dtype == np.int32
    assert abs(A - A.T).max() == 0
    assert np.linalg.eigvalsh(A.toarray()).min() > 0
    B = operators.stencil27_variable((9, 8, 10))
    assert np.bincount(np.diff(B.indptr))[27] == 7 * 6 * 8          # interior nodes hold all 27 couplings
    with pytest.raises(ValueError):
        operators.stencil27_variable((8, 8))


def test_device_format_selftest_property():
    """Random structure: banded / repeated-value / random matrices with empty rows, duplicate
    columns and unsorted rows survive the coder bit for bit in both precisions."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(n=st.integers(1, 700), m=st.integers(1, 700), kind=st.sampled_from(["banded", "few_values", "random", "dup"]),
           seed=st.integers(0, 2 ** 31 - 1), dtype=st.sampled_from(["float64", "float32"]))
    def run(n, m, kind, seed, dtype):
        rng = np.random.default_rng(seed)
        if kind == "banded":
            offs = sorted(set(int(o) for o in rng.integers(-min(n, m) + 1, min(n, m), size=5)))
            M = sp.diags([rng.choice([-1.0, 2.0, 0.5], size=1)[0] * np.ones(min(n, m))] * len(offs), offs, shape=(n, m), format="csr")
        elif kind == "few_values":
            M = sp.random(n, m, density=0.05, random_state=np.random.RandomState(seed % (2 ** 31)), format="csr")
            M.data = rng.choice([1.0, -1.0, 0.125, 3.0], size=M.nnz)
        elif kind == "random":
            M = sp.random(n, m, density=0.03, random_state=np.random.RandomState(seed % (2 ** 31)), format="csr")
        else:                                                    # duplicate column entries inside rows, unsorted
            k = int(rng.integers(0, 4 * n + 1))
            rows = np.sort(rng.integers(0, n, size=k))
            cols = rng.integers(0, m, size=k)
            indptr = np.searchsorted(rows, np.arange(n + 1)).astype(np.int32)
            M = sp.csr_matrix((rng.standard_normal(k), cols.astype(np.int32), indptr), shape=(n, m))
        info = _hip.format_selftest(M, dtype)                    # raises on any mismatch
        assert info["rows"] == n and info["nnz"] == M.nnz

    run()


def test_mgcycle_cache_key_follows_content_not_object_identity():
    """the device-hierarchy cache of mgCycle was keyed on id() / buffer
    addresses, which CPython and malloc recycle once the caller frees its matrices.  The key is
    a checksum of the stored arrays now: same-shape operators with different values, built in a
    loop with the previous ones freed, must all get distinct keys; equal content gives an equal
    key; an in-place edit changes it."""
    import gc
    keys = set()
    R = [sp.csr_matrix((np.full(64, 0.5), (np.arange(64) // 2, np.arange(64))), shape=(32, 64))]
    A = None
    for k in range(8):
        A = None
        gc.collect()
        T = sp.diags([-np.ones(63), (2.0 + k) * np.ones(64), -np.ones(63)], [-1, 0, 1], format="csr")
        A = [T, sp.csr_matrix(R[0] @ T @ R[0].T)]
        keys.add(openmg_amd._fingerprint(A, R, 2, 0, 1.0, 0))
    assert len(keys) == 8
    same = [sp.csr_matrix(M.copy()) for M in A]
    assert openmg_amd._fingerprint(same, R, 2, 0, 1.0, 0) == openmg_amd._fingerprint(A, R, 2, 0, 1.0, 0)
    before = openmg_amd._fingerprint(A, R, 2, 0, 1.0, 0)
    A[0].data[5] += 1.0
    assert openmg_amd._fingerprint(A, R, 2, 0, 1.0, 0) != before
    assert openmg_amd._fingerprint(A, R, 2, 1, 1.0, 0) != openmg_amd._fingerprint(A, R, 2, 0, 1.0, 0)
    # large arrays are hashed WHOLE: an edit of one entry anywhere changes the key
    big = np.arange((1 << 19) + 1000, dtype=np.float64)
    c0 = openmg_amd._array_checksum(big)
    for where in (-1, 0, 123457, big.size // 2 + 3):
        keep = big[where]
        big[where] = -1.0
        assert openmg_amd._array_checksum(big) != c0
        big[where] = keep
    assert openmg_amd._array_checksum(big) == c0
    # a member that is not CSR is converted once while it lives, and follows in-place edits of the ORIGINAL
    csc = sp.csc_matrix(A[0])
    k1 = openmg_amd._fingerprint([csc, A[1]], R, 2, 0, 1.0, 0)
    assert openmg_amd._as_csr_cached(csc) is openmg_amd._as_csr_cached(csc)
    assert k1 == openmg_amd._fingerprint([csc, A[1]], R, 2, 0, 1.0, 0)
    csc.data[3] += 0.5
    assert openmg_amd._fingerprint([csc, A[1]], R, 2, 0, 1.0, 0) != k1


def test_union_partition_is_chosen_and_decodes(monkeypatch):
    """Round 2: level operators whose row patterns are all subsequences of one short sequence are
    blocked several-rows-per-thread x 256 rows for rows_union_kernel (csrc/common.h UNION_MAX);
    omg_format_selftest also expands every row from the block's union and its pattern's mask and
    compares it with the stored pattern.  Host only."""
    shape = (32, 32, 32)
    A = orc.stencil_poisson(shape)
    order = orc.colour_order(orc.parity_colouring(shape))
    P = sp.csr_matrix(A[order][:, order])                         # red-black permuted: two interior row patterns per colour
    natural2d = orc.stencil_poisson((256, 256))
    galerkin = sp.csr_matrix(orc.restriction(shape) @ A @ orc.restriction(shape).T)
    galerkin.sort_indices()
    blocks = {}
    for rows in ("1", "2", "4"):
        monkeypatch.setenv("OMG_UNION_ROWS", rows)
        for name, M in (("rb", P), ("2d", natural2d), ("natural", A), ("galerkin", galerkin)):
            for dtype in ("float64", "float32"):
                f = _hip.format_selftest(M, dtype)                # raises if a union / mask does not reproduce a row
                assert f["pattern_rows"] == f["rows"], (name, rows)
                blocks[(name, rows, dtype)] = f["blocks"]
    for name, M in (("rb", P), ("2d", natural2d), ("natural", A)):
        n = M.shape[0]
        assert blocks[(name, "1", "float64")] == -(-n // 256)
        assert blocks[(name, "2", "float64")] == -(-n // 512)
        assert blocks[(name, "4", "float64")] == -(-n // 1024)
    assert blocks[("galerkin", "2", "float64")] == -(-galerkin.shape[0] // 512)     # 4096 rows: still >= 4 blocks
    monkeypatch.setenv("OMG_UNION_KERNEL", "0")                   # the kernel off: 256-row blocks as in round 1
    assert _hip.format_selftest(P)["blocks"] == -(-P.shape[0] // 256)
    monkeypatch.delenv("OMG_UNION_KERNEL")
    monkeypatch.setenv("OMG_UNION_ROWS", "2")
    # operators without a short common supersequence keep the standard partition: 27-point variable
    # coefficients (values differ row by row), irregular sparsity, restriction (not a level operator)
    assert _hip.format_selftest(operators.stencil27_variable((12, 12, 12)))["pattern_rows"] == 0
    R = orc.restriction(shape)
    assert _hip.format_selftest(R)["blocks"] == -(-R.shape[0] // 256)
