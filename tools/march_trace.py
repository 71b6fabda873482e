#!/usr/bin/env python3
"""Per launch shape: average duration of march_gs_kernel in a rocprofv3 kernel_trace.csv.
A grid of 256 workgroups is either a level of 256 tiles or 256 persistent workers over a larger level (march.hip
sweep()); the trace does not say which, so a bucket whose durations fall into two clusters is reported as two."""
import collections
import csv
import sys

d = collections.OrderedDict()
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    if "march_gs_kernel" not in r["Kernel_Name"]:
        continue
    g = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    d.setdefault(g, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for g, v in d.items():
    s = sorted(v)
    cut = max(range(1, len(s)), key=lambda i: s[i] / s[i - 1]) if len(s) > 1 else 0
    if cut and s[cut] / s[cut - 1] > 1.5 and g == 256:
        lo, hi = s[:cut], s[cut:]
        out.append("256 workgroups = 256 tiles: %.1f us (min %.1f, %d launches)" % (sum(lo) / len(lo), lo[0], len(lo)))
        out.append("256 workgroups = persistent workers over a larger level (256^3: 1024 tiles): %.1f us (min %.1f, %d launches)"
                   % (sum(hi) / len(hi), hi[0], len(hi)))
    else:
        out.append("%d tiles: %.1f us (min %.1f, %d launches)" % (g, sum(v) / len(v), min(v), len(v)))
print(" | ".join(out))
