#!/usr/bin/env python3
"""Torch-free driver for rocprofv3: build the bench problem, run K resident V-cycles.

    rocprofv3 --kernel-trace --stats ... -- python3 tools/prof_cycle.py --size 256 --steps 10
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--grids", type=int, default=5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--smoother", default="colour")
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--dtype", default="float64")
    args = ap.parse_args()
    h, b, meta = bench.build_problem(args.size, args.grids, args.smoother, args.dtype)
    h.resident_load(b)
    if args.graph:
        h.use_graph(True)
    for _ in range(2):
        h.resident_cycle(1, 1, want_norm=False)
    h.sync()
    h.profile_enable(False if args.graph else True)
    for _ in range(args.steps):
        h.resident_cycle(1, 1, want_norm=False)
    h.sync()
    if not args.graph:
        for name, (cnt, ms) in h.profile_read().items():
            if cnt:
                print("%-20s %6d launches  avg %9.2f us" % (name, cnt, 1e3 * ms / cnt))
    print("norm", h.resident_cycle(1, 1, want_norm=True))
    h.close()


if __name__ == "__main__":
    main()
