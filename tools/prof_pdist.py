"""One slab (the whole 256^3 grid) through the plane-slab runner, in this process: for rocprofv3 --kernel-trace.
argv: mode (0 copies / RCCL path, 1 peer mode with waiting passes, 2 peer mode with wait launches), cycles."""
import sys
import time

import numpy as np

from openmg_amd import _hip, _hip_dist, operators


def main():
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    shape, n_levels, tail_grids = (256, 256, 256), 3, 3
    coef = [[v / 16.0 ** l for v in (-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0)] for l in range(n_levels)]
    tshape = tuple(s >> n_levels for s in shape)
    Rt = operators.restrictionList(tshape, tail_grids - 2, 8)
    At = operators.coeffecientList(operators.stencil_poisson(tshape) / 16.0 ** n_levels, Rt)
    b = np.random.default_rng(5).random(shape[0] * shape[1] * shape[2])
    d = _hip_dist.PlaneDistRank(0, 1, shape, coef, 0.125, _hip.Hierarchy(At, Rt, smoother="colour"))
    d.load(b)
    if mode:
        d.p2p_enable(mode)
    d.cycles(3)
    t = time.perf_counter()
    d.cycles(n)
    print("mode %d: %.1f us per cycle" % (mode, (time.perf_counter() - t) / n * 1e6))
    d.close()


if __name__ == "__main__":
    main()
