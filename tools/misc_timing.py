#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md: host-buffer (PCIe-inclusive) cycle rate through
omg_vcycle, and hipGraph replay for the many-set lexicographic ordering."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openmg_amd import _hip, operators  # noqa: E402


def build(shape, grids, smoother):
    A0 = operators.stencil_poisson(shape)
    b = A0 @ np.random.default_rng(12345).random(A0.shape[0])
    R = operators.restrictionList(shape, grids - 2, 8)
    A = operators.coeffecientList(A0, R)
    return _hip.Hierarchy(A, R, smoother=smoother), b


def rate(fn, steps):
    fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    return steps / (time.perf_counter() - t0)


def main():
    h, b = build((256, 256, 256), 5, "colour")
    x = np.zeros(b.size)
    print("host-buffer path (omg_vcycle: H2D b, x + cycle + D2H x), 256^3 red-black: %.1f cycles/s"
          % rate(lambda: h.vcycle(b, x, 1, 1), 10))
    h.close()
    for shape, grids in (((256, 256, 256), 5), ((128, 128, 128), 4)):
        h, b = build(shape, grids, "gs")
        h.resident_load(b)
        def cyc():
            h.resident_cycle(1, 1, want_norm=False)
        def cyc_sync():
            cyc(); h.sync()
        r0 = rate(cyc_sync, 10)
        h.use_graph(True)
        r1 = rate(cyc_sync, 10)
        print("lexicographic GS %s: %.1f cycles/s eager, %.1f cycles/s hipGraph replay" % (shape, r0, r1))
        h.close()
    A1 = operators.poisson(4096, sparse=True)
    R = operators.restrictionList((4096,), 1, 8)
    A = operators.coeffecientList(A1, R)
    h = _hip.Hierarchy(A, R, smoother="gs")
    h.resident_load(A1 @ np.ones(4096))
    def c1():
        h.resident_cycle(1, 1, want_norm=False); h.sync()
    r0 = rate(c1, 10)
    h.use_graph(True)
    r1 = rate(c1, 10)
    print("1-D N=4096 lexicographic GS: %.1f cycles/s eager, %.1f hipGraph" % (r0, r1))


if __name__ == "__main__":
    main()
