"""Host-side logic of openmg_amd that needs no GPU: generators (with the reference's
quirks), dict helpers, argument validation, smoother-name table.  CPU only."""
import doctest

import numpy as np
import pytest

import openmg_amd
from openmg_amd import _hip, operators, tools


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
