#!/usr/bin/env python3
"""Per launch shape: average duration of march_gs_kernel in a rocprofv3 kernel_trace.csv (launch order)."""
import collections
import csv
import sys

d = collections.OrderedDict()
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
run, prev = 0, None
for r in rows:
    if "march_gs_kernel" not in r["Kernel_Name"]:
        if prev is not None:
            run += 1
            prev = None
        continue
    g = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    # (a launch of persistent workers — more tiles than workers — carries a pad of dynamic LDS: march.hip sweep())
    g = "%d workers (persistent: more tiles than that)" % g if int(r["LDS_Block_Size"]) > 65536 else "%d tiles" % g
    prev = g
    d.setdefault((run if "--runs" in sys.argv else 0, g), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(" ".join(a for a in sys.argv[2:] if a != "--runs"),
      " | ".join("%s: %.1f us (min %.1f, %d launches)" % (g, sum(v) / len(v), min(v), len(v)) for (_, g), v in d.items()))
