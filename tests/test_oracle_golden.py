"""Pin the CPU oracle (oracle/mg_oracle.py) against vectors recorded from the REAL
reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import mg_oracle as orc

TIGHT = dict(rtol=1e-12, atol=1e-13)


def csr_from(d, prefix):
    return sp.csr_matrix((d[prefix + "_data"], d[prefix + "_indices"], d[prefix + "_indptr"]),
                         shape=tuple(d[prefix + "_shape"]))


def test_g1_simple_demo_known_answer(golden):
    """openmg_usage_demo.py:48-66: 0.805398, 0.107866, 0.018650, 0.003405 (gridLevels=2)."""
    d = golden("g1_simple_demo")
    A = csr_from(d, "A")
    printed = [0.805398, 0.107866, 0.018650, 0.003405]
    for gl in (2, 3):
        for dense in (True, False):
            tag = "gl%d_%s" % (gl, "dense" if dense else "sparse")
            p = {"problemShape": (100,), "gridLevels": gl, "cycles": 10, "iterations": 2,
                 "verbose": False, "dense": dense, "threshold": 1e-2, "giveInfo": True}
            u, info = orc.mg_solve(A, d["b"].copy(), p)
            assert info["cycle"] == int(d[tag + "_cycle"])
            np.testing.assert_allclose(info["norm"], d[tag + "_norms"][-1], rtol=1e-9)
            np.testing.assert_allclose(u, d[tag + "_u"], rtol=1e-10, atol=1e-12)
    # the reference's own printed numbers
    np.testing.assert_allclose(d["gl2_dense_norms"], printed, atol=5e-7)


@pytest.mark.parametrize("post", [0, 1])
def test_g2_poisson1d_4096(golden, post):
    d = golden("g2_poisson1d_4096")
    A = orc.poisson(4096, sparse=True)
    b = d["b"]
    np.testing.assert_allclose(A @ d["u_true"], b, **TIGHT)
    p = {"problemShape": (4096,), "gridLevels": 2, "preIterations": 1, "postIterations": post,
         "cycles": 1, "threshold": 0, "giveInfo": True}
    _, info = orc.mg_solve(A, b.copy(), p)
    x = None
    norms = []
    for c in range(1, 6):
        x, inf = orc.mg_cycle(info["A"], b, 0, info["R"], p, initial=x)
        norms.append(inf["norm"])
        if c in (1, 2, 5):
            np.testing.assert_allclose(x, d["v1%d_x_c%d" % (post, c)], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(norms, d["v1%d_norms" % post], rtol=1e-10)


def test_g3_poisson3d_16_hierarchy_and_cycles(golden):
    d = golden("g3_poisson3d_16")
    shape = tuple(int(s) for s in d["shape"])
    A0 = orc.stencil_poisson(shape)
    assert abs(A0 - csr_from(d, "A0")).max() == 0
    R = orc.restriction_list(shape, 1, 8)
    A = orc.coefficient_list(A0, R)
    assert len(A) == int(d["n_levels"])
    for l, M in enumerate(A):
        G = csr_from(d, "A%d" % l)
        assert M.shape == G.shape
        assert abs(sp.csr_matrix(M) - G).max() < 1e-15
    for l, M in enumerate(R):
        assert abs(M - csr_from(d, "R%d" % l)).max() == 0
    # Galerkin of the constant 7-point stencil is lap3(n/2)/16
    assert abs(sp.csr_matrix(A[1]) - orc.stencil_poisson((8, 8, 8)) / 16.0).max() < 1e-15
    for pre, post in ((1, 0), (1, 1)):
        p = {"problemShape": shape, "gridLevels": 2, "preIterations": pre, "postIterations": post,
             "coarsestLevel": len(R)}
        x = None
        norms = []
        for c in range(1, 4):
            x, inf = orc.mg_cycle(A, d["b"], 0, R, p, initial=x)
            norms.append(inf["norm"])
            if c in (1, 3):
                np.testing.assert_allclose(x, d["v%d%d_x_c%d" % (pre, post, c)], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(norms, d["v%d%d_norms" % (pre, post)], rtol=1e-10)


def test_g3_poisson3d_32_trace(golden):
    d = golden("g3_poisson3d_32")
    shape = (32, 32, 32)
    A0 = orc.stencil_poisson(shape)
    p = {"problemShape": shape, "gridLevels": 2, "preIterations": 1, "postIterations": 1,
         "cycles": 3, "threshold": 0, "giveInfo": True, "minSize": 8}
    x, info = orc.mg_solve(A0, d["b"].copy(), p)
    assert info["cycle"] == 3
    np.testing.assert_allclose(x, d["v11_x_c3"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(info["norm"], d["v11_norms"][2], rtol=1e-10)


def test_g4_redblack_is_permuted_lexicographic(golden):
    """The pin for colour-ordered Gauss-Seidel: the real reference's gaussSeidel on the
    red-first permuted matrix (recorded) == ordered sweep in natural numbering."""
    d = golden("g4_redblack_pin")
    for tag in ("p5", "p7"):
        shape = tuple(int(s) for s in d[tag + "_shape"])
        A = orc.stencil_poisson(shape)
        colour = orc.greedy_colouring(A)
        parity = np.indices(shape).reshape(len(shape), -1).sum(axis=0) % 2
        assert np.array_equal(colour, parity)            # greedy == red-black on 5/7-point
        x = d[tag + "_x0"].copy()
        orc.gs_ordered(A, d[tag + "_b"], x, orc.colour_order(colour), iterations=2)
        np.testing.assert_allclose(x, d[tag + "_x_after2"], **TIGHT)
    shape = tuple(int(s) for s in d["vc_shape"])
    A0 = orc.stencil_poisson(shape)
    R = orc.restriction_list(shape, 1, 8)
    A = orc.coefficient_list(A0, R)
    sm = orc.make_smoother("colour", A)
    p = {"preIterations": 1, "postIterations": 1, "coarsestLevel": len(R)}
    x = None
    norms = []
    for _ in range(3):
        x, inf = orc.mg_cycle(A, d["vc_b"], 0, R, p, initial=x, smoother=sm)
        norms.append(inf["norm"])
    np.testing.assert_allclose(x, d["vc_x_c3"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(norms, d["vc_norms"], rtol=1e-10)


def test_g5_restriction(golden):
    d = golden("g5_restriction")
    for tag in d["cases"]:
        tag = str(tag)
        shape = tuple(int(s) for s in tag[1:].split("x"))
        if tag + "_error" in d.files:
            with pytest.raises(IndexError):
                orc.restriction(shape)
            continue
        G = csr_from(d, tag)
        R = orc.restriction(shape)
        assert R.shape == G.shape
        assert abs(R - G).max() == 0
        assert R.nnz == G.nnz
    for row in d["restrictionList_cases"]:
        dim, coarsest, minsize, nR, last0, last1 = (int(v) for v in row[:6])
        shape = tuple(int(v) for v in row[6:6 + dim])
        R = orc.restriction_list(shape, coarsest, minsize)
        assert len(R) == nR
        assert R[-1].shape == (last0, last1)
    with pytest.raises(ValueError):
        orc.restriction((4, 4, 4, 4))        # tests.py:544-548
    with pytest.raises(ValueError):
        orc.restriction((2,))                # coarse set of one point, operators.py:53-56


def test_g7_stop_rules_and_dict_mutation(golden):
    d = golden("g7_stop_rules_misc")
    A, b = d["stop_A"], d["stop_b"]
    np.testing.assert_array_equal(A, orc.poisson((36,)))
    p = {"problemShape": (36,), "gridLevels": 2, "threshold": 8e-3, "giveInfo": True}
    u, info = orc.mg_solve(A, b.copy(), p)
    assert info["cycle"] == int(d["thresh_cycle"])
    np.testing.assert_allclose(info["norm"], float(d["thresh_norm"]), rtol=1e-9)
    np.testing.assert_allclose(u, d["thresh_u"], rtol=1e-10, atol=1e-12)
    assert sorted(p.keys()) == [str(k) for k in d["thresh_keys_after"]]
    assert p["coarsestLevel"] == int(d["thresh_coarsestLevel_after"])
    assert np.linalg.norm(A @ u - b) < 8e-3          # tests.py:514-515
    p = {"problemShape": (36,), "gridLevels": 2, "cycles": 3, "threshold": 1e-10, "giveInfo": True}
    u, info = orc.mg_solve(A, b.copy(), p)
    assert info["cycle"] == 3 == int(d["cyc_cycle"])  # tests.py:531
    np.testing.assert_allclose(u, d["cyc_u"], rtol=1e-10, atol=1e-12)
    # minSize (tests.py:558-570)
    p = {"problemShape": (1024,), "gridLevels": 24, "iterations": 1, "verbose": False,
         "threshold": 4, "giveInfo": True, "minSize": 23}
    soln, info = orc.mg_solve(orc.poisson((1024,)), d["minsize_b"].copy(), p)
    assert [list(r.shape) for r in info["R"]] == d["minsize_R_shapes"].tolist()
    assert min(info["R"][-1].shape) > 23
    assert info["cycle"] == int(d["minsize_cycle"])
    assert p["coarsestLevel"] == int(d["minsize_coarsestLevel_after"])
    np.testing.assert_allclose(soln, d["minsize_soln"], rtol=1e-10, atol=1e-12)
    # neither stop rule -> ValueError after the first cycle (tests.py:550-556)
    with pytest.raises(ValueError):
        orc.mg_solve(orc.poisson((64,)), np.ones(64), {"problemShape": (64,), "gridLevels": 2,
                                                      "cycles": 0, "threshold": 0})


def test_g7_test_a_1d_operator_3d_shape(golden):
    d = golden("g7_stop_rules_misc")
    p = {"coarsestLevel": 3, "problemShape": (12, 12, 12), "gridLevels": 4, "threshold": 8e-3,
         "giveInfo": True}
    u, info = orc.mg_solve(orc.poisson((1728,)), d["testa_b"].copy(), p)
    assert info["cycle"] == int(d["testa_cycle"])
    np.testing.assert_allclose(info["norm"], float(d["testa_norm"]), rtol=1e-8)
    np.testing.assert_allclose(u, d["testa_u"], rtol=1e-9, atol=1e-11)


def test_g7_generators(golden):
    d = golden("g7_stop_rules_misc")
    np.testing.assert_array_equal(orc.poisson(8, sparse=True).toarray(), d["gen_p1sparse_8"])
    np.testing.assert_array_equal(orc.poisson((8,)), d["gen_p1dense_8"])
    np.testing.assert_array_equal(orc.poisson((3, 4)), d["gen_p2dense_3x4"])
    np.testing.assert_array_equal(orc.poisson((4, 4)), d["gen_p2dense_4x4"])
    np.testing.assert_array_equal(orc.poisson((2, 3, 4)), d["gen_p3dense_2x3x4"])
    np.testing.assert_array_equal(orc.poisson((3, 3, 3)), d["gen_p3dense_3x3x3"])
    with pytest.raises(ValueError):
        orc.poisson((1, 2, 3, 4))                     # tests.py:533-536
    with pytest.raises(NotImplementedError):
        orc.poisson((4, 4), sparse=True)              # operators.py:224


def test_g7_smoother_standalone(golden):
    d = golden("g7_stop_rules_misc")
    A1 = orc.poisson(64, sparse=True)
    x = d["gs_x0"].copy()
    out = orc.gauss_seidel(A1, d["gs_b"], x)
    assert out is x                                   # in place (Q2)
    np.testing.assert_allclose(x, d["gs_x_it1"], **TIGHT)
    np.testing.assert_allclose(orc.smooth(A1, d["gs_b"], d["gs_x0"].copy(), 3), d["gs_x_it3"], **TIGHT)
    np.testing.assert_allclose(orc.smooth_to_threshold(A1, d["gs_b"], d["gs_x0"].copy(), 1e-6),
                               d["gs_x_thr"], **TIGHT)
    A2 = orc.poisson((12, 12))
    np.testing.assert_allclose(orc.smooth_to_threshold(A2, d["gs_thresh_b"], np.zeros(144), 1e-4),
                               d["gs_thresh_x"], rtol=1e-9, atol=1e-11)
    Au = csr_from(d, "unsorted_A")
    np.testing.assert_allclose(orc.gauss_seidel(Au, d["unsorted_b"], np.zeros(36), iterations=2),
                               d["unsorted_x"], **TIGHT)
    np.testing.assert_allclose(orc.coarse_solve(A1, d["gs_b"].reshape(-1, 1)), d["coarse_x"], **TIGHT)


def test_jacobi_unpinned_self_consistency():
    """Weighted Jacobi has no reference counterpart; check the C loop against NumPy."""
    A = orc.stencil_poisson((6, 5))
    rng = np.random.default_rng(0)
    b, x0 = rng.random(30), rng.random(30)
    x = orc.jacobi(A, b, x0.copy(), iterations=2, omega=0.8)
    ref = x0.copy()
    for _ in range(2):
        ref = ref + 0.8 * (b - A @ ref) / A.diagonal()
    np.testing.assert_allclose(x, ref, **TIGHT)
