#!/usr/bin/env python3
"""Per-kernel averages of every counter found under a tools/pmc_passes.sh output directory.

    python tools/pmc_table.py gpurun_out/pmc [min_blocks]

Rows: rows_kernel instantiations (mode, value type) at one grid size; columns: counters
(averaged over the launches of that kernel in the pass that collected them)."""
import collections
import csv
import glob
import os
import re
import sys

MODES = {0: "spmv", 1: "residual", 2: "resnorm", 3: "gs", 4: "jacobi", 5: "axpy", 6: "norm_only", 7: "gs+res", 8: "gs+norm", 9: "scatter", 10: "gs+prenorm", 11: "jacobi+prenorm"}


def main():
    root = sys.argv[1]
    min_blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    table = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            m = re.search(r"rows_kernel<(\d+), (true|false), (true|false), (\d+), (?:true|false), (\w+)>", r["Kernel_Name"])
            mp = re.search(r"rows_pattern_kernel<(\d+), (?:true|false), (?:true|false), (\w+)>", r["Kernel_Name"])
            mu = re.search(r"rows_union_kernel<(\d+), (\d+), (\d+), (\w+)>", r["Kernel_Name"])
            if not m and not mp and not mu:
                continue
            blocks = int(r["Grid_Size"]) // 256
            if blocks < min_blocks:
                continue
            if mu:
                key = (MODES.get(int(mu.group(1)), mu.group(1)), mu.group(4), "union U" + mu.group(2), blocks)
            elif mp:
                key = (MODES.get(int(mp.group(1)), mp.group(1)), mp.group(2), "pattern", blocks)
            else:
                key = (MODES.get(int(m.group(1)), m.group(1)), m.group(5), "short" if m.group(3) == "true" else "lpr" + m.group(4), blocks)
            table[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r:
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for key in sorted(table):
        print("%s %s %s blocks=%d   (avg duration under collection %.1f us)" % (key + (sum(dur[key]) / max(len(dur[key]), 1),)))
        for name in sorted(table[key]):
            v = table[key][name]
            print("    %-40s %16.1f   (n=%d)" % (name, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
