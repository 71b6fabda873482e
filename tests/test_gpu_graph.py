"""hipGraph replay of a cycle (omg_resident_use_graph) when the cycle leaves the levels' current / scratch vectors
swapped: the 27-point pair launches swap once per sweep, the plane passes once per pass that relaxes, Jacobi once per
sweep.  A captured graph holds the pointers of the phase it was captured in; replayed from the other phase it would
recompute the same cycle from stale data.  Every cycle of a replayed run must have the bits of the
eager run (openmg/__init__.py:112-138: mgSolve's loop feeds every cycle the previous cycle's iterate)."""
import numpy as np
import pytest
import scipy.sparse as sp

from openmg_amd import _hip, operators

pytestmark = pytest.mark.gpu


def run(h, b, pre, post, cycles, x0=None):
    h.resident_load(b, x0)
    norms = [h.resident_cycle(pre, post) for _ in range(cycles)]
    return norms, h.resident_fetch()


def galerkin(A0, shape, grids):
    R = operators.restrictionList(shape, grids - 2, 1)
    assert len(R) == grids - 1
    return operators.coeffecientList(A0, R), R


CASES = [
    ("s27", (16, 16, 16), 3, "colour", [(1, 0), (2, 1), (0, 1), (1, 1)]),
    ("plane", (32, 32, 32), 3, "colour", [(0, 1), (1, 0), (2, 1), (1, 1), (0, 0)]),
    ("plane2d", (64, 64), 3, "colour", [(1, 0), (2, 0), (1, 1)]),
    ("jacobi2d", (64, 64), 3, "jacobi", [(2, 0), (1, 0), (2, 1), (1, 1)]),
    ("sets", (8, 8, 8), 2, "jacobi", [(1, 0), (2, 1)]),
]


@pytest.mark.parametrize("name,shape,grids,smoother,sweeps", CASES, ids=[c[0] for c in CASES])
def test_graph_replay_has_the_bits_of_eager_cycles_for_every_swap_parity(name, shape, grids, smoother, sweeps):
    if name == "s27":
        A0 = operators.stencil27_variable(shape)
    else:
        A0 = operators.stencil_poisson(shape)
    A, R = galerkin(A0, shape, grids)
    rng = np.random.default_rng(5)
    b = A0 @ rng.random(A0.shape[0])
    x0 = rng.standard_normal(A0.shape[0])
    with _hip.Hierarchy(A, R, smoother=smoother, omega=0.8) as h:
        flags = h.level_flags(0)
        if name == "s27":
            assert flags["stencil27"]
        elif name in ("plane", "plane2d", "jacobi2d"):
            assert flags["plane"]
        else:
            assert not flags["plane"] and not flags["stencil27"]
        for pre, post in sweeps:
            h.use_graph(False)
            eager = run(h, b, pre, post, 5, x0)
            # strictly decreasing norms: a replay from the wrong phase shows as a norm that stops moving
            if pre + post:
                assert all(v < u for u, v in zip(eager[0], eager[0][1:])), (name, pre, post, eager[0])
            h.use_graph(True)
            graph = run(h, b, pre, post, 5, x0)
            assert graph[0] == eager[0], (name, pre, post, graph[0], eager[0])
            assert np.array_equal(graph[1], eager[1]), (name, pre, post)
            # an eager cycle in between leaves the vectors in the phase no graph may have been captured from
            h.resident_load(b, x0)
            mixed = [h.resident_cycle(pre, post)]
            h.use_graph(False)
            mixed.append(h.resident_cycle(pre, post))
            h.use_graph(True)
            mixed += [h.resident_cycle(pre, post) for _ in range(3)]
            assert mixed == eager[0], (name, pre, post, mixed, eager[0])
            assert np.array_equal(h.resident_fetch(), eager[1])
            # omg_solve runs its cycles through the same replay
            x = x0.copy()
            assert h.solve(b, x, pre, post, 5, 0.0) == (5, eager[0][-1])
            assert np.array_equal(x, eager[1])
        h.use_graph(False)
