#!/usr/bin/env python3
"""Print the last N kernels of a rocprofv3 kernel_trace.csv with start, gap and duration (us).

    python tools/trace_dump.py <kernel_trace.csv> [N]"""
import csv
import re
import sys


def short(n):
    m = re.search(r"s27_sweep_kernel<(\w+), (\d+), (\d+), (true|false), (true|false), (true|false)>", n)
    if m:
        return "s27_sweep<%s rg%s pair%s%s%s%s>" % (m.group(1), m.group(2), m.group(3), " x=0" if m.group(4) == "true" else "",
                                                    " +old-norm" if m.group(5) == "true" else "", " +res67" if m.group(6) == "true" else "")
    m = re.search(r"s27_residual_kernel<(\w+), (\d+), (\d+), (\d+)>", n)
    if m:
        return "s27_residual<%s rg%s %s colours %s>" % (m.group(1), m.group(2), m.group(3), "restrict" if m.group(4) == "0" else "norm")
    m = re.search(r"(s27_\w+_kernel|tile2d_kernel|plane_kernel|block_kernel|\w+_kernel|__amd\w+)", n)
    return m.group(1) if m else n[:40]


def main(path, count=60):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]), int(r.get("Workgroup_Size_X", 256) or 256))
                for r in rows)
    ev = ev[-count:]
    t0, prev = ev[0][0], ev[0][0]
    for s, e, n, g, w in ev:
        print("%9.1f gap %6.1f dur %8.1f  %-52s wgs %d x %d" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, short(n), g // max(w, 1), w))
        prev = e


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
