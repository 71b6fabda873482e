#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> calls / avg / min / max duration per (kernel, grid size).

    python tools/trace_stats.py <kernel_trace.csv>

rocprofv3's own --stats table averages a kernel over ALL its launches; the row kernels run on
every level of the hierarchy, so the fine-grid launch bench.py times is the one with the
largest grid (32768 workgroups for a colour of 256^3)."""
import collections
import csv
import re
import sys

MODES = {0: "spmv", 1: "residual", 2: "resnorm", 3: "gs", 4: "jacobi", 5: "axpy", 6: "norm_only", 7: "gs+res", 8: "gs+norm", 9: "scatter", 10: "gs+prenorm", 11: "jacobi+prenorm"}


def short(name):
    m = re.search(r"plane_kernel<(\w+), (\d+), (true|false), (true|false), (\d+)>", name)
    if m:
        return "plane_kernel<%s, %s%s%s>" % (m.group(1), "down" if m.group(2) == "0" else "up", ", norm" if m.group(3) == "true" else "",
                                             ", x=0" if m.group(4) == "true" else "")
    m = re.search(r"rows_union_kernel<(\d+), (\d+), (\d+), (\w+)>", name)
    if m:
        return "rows_union_kernel<%s, U%s, %s>" % (MODES[int(m.group(1))], m.group(2), m.group(4))
    m = re.search(r"rows_pattern_kernel<(\d+), (?:true|false), (?:true|false), (\w+)>", name)
    if m:
        return "rows_pattern_kernel<%s, %s>" % (MODES[int(m.group(1))], m.group(2))
    m = re.search(r"rows_kernel<(\d+), (true|false), (true|false), (\d+), (?:true|false), (\w+)>", name)
    if m:
        return "rows_kernel<%s, nt=%s, %s, %s>" % (MODES[int(m.group(1))], m.group(2)[0],
                                                   "short" if m.group(3) == "true" else "lpr" + m.group(4), m.group(5))
    m = re.search(r"s27_sweep_kernel<(\w+), (\d+), (\d+), (true|false), (true|false), (true|false)>", name)
    if m:
        return "s27_sweep<%s rg%s pair%s%s%s%s>" % (m.group(1), m.group(2), m.group(3), " x=0" if m.group(4) == "true" else "",
                                                    " +old-norm" if m.group(5) == "true" else "", " +res67" if m.group(6) == "true" else "")
    m = re.search(r"s27_residual_kernel<(\w+), (\d+), (\d+), (\d+)>", name)
    if m:
        return "s27_residual<%s rg%s %s colours %s>" % (m.group(1), m.group(2), m.group(3), "restrict" if m.group(4) == "0" else "norm")
    m = re.search(r"tile2d_kernel<(\w+), (\d+), (true|false), (true|false), (true|false), (\d+), (\d+), (true|false)>", name)
    if m:
        return "tile2d_kernel<%s %s%s%s %sx%s%s>" % (m.group(1), "down" if m.group(2) == "0" else "up", ", norm" if m.group(4) == "true" else "",
                                                     "" if m.group(5) == "true" else ", no sweep", m.group(6), m.group(7), ", jacobi" if m.group(8) == "true" else "")
    m = re.search(r"(\w+_kernel|__amd\w+)", name)
    return m.group(1) if m else name[:40]


def main():
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[1])):
        wg = max(int(r.get("Workgroup_Size_X", 256) or 256), 1)
        agg[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("%-52s %9s %7s %10s %10s %10s" % ("kernel", "wgs", "calls", "avg us", "min us", "max us"))
    for (name, wgs), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if "eliminate" in name or "pivot" in name or "column_kernel" in name or "swap_scale" in name:
            wgs = -1
        print("%-52s %9d %7d %10.2f %10.2f %10.2f" % (name, wgs, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
