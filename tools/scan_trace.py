#!/usr/bin/env python3
"""Per kernel and launch shape: average duration of the lexicographic sweep kernels (march_gs_kernel, scan_gs_kernel) in a
rocprofv3 kernel_trace.csv."""
import collections
import csv
import sys

d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "march_gs_kernel" not in name and "scan_gs_kernel" not in name:
        continue
    short = "scan" if "scan_gs" in name else "march"
    g = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    d.setdefault((short, g), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (short, g), v in d.items():
    print("%-6s %4d workgroups: %8.1f us (min %.1f, %d launches)" % (short, g, sum(v) / len(v), min(v), len(v)))
