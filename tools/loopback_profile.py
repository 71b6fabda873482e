#!/usr/bin/env python3
"""Aggregate one loopback cycle from a rocprofv3 kernel trace of tools/loopback_overhead.py."""
import collections
import csv
import re
import sys


def main(path, phase=10):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"])) for r in rows)
    idx = [i for i, e in enumerate(ev) if "sqrt_kernel" in e[2]]
    pairs = [k for k in range(len(idx) - 1) if idx[k + 1] == idx[k] + 1]
    a, b = idx[pairs[phase] + 1], idx[pairs[phase + 1] + 1]
    cyc = ev[a + 1:b + 1]
    print("cycle wall us %.1f kernels %d" % ((cyc[-1][1] - ev[a][1]) / 1e3, len(cyc)))
    agg = collections.defaultdict(lambda: [0, 0.0])
    tot, prev, gaps = 0, ev[a][1], 0
    for s, e, n, g in cyc:
        m = re.search(r"(rows_kernel<\d+, \w+, \w+, \d+>|\w+_kernel|__amd\w+)", n)
        name = m.group(1) if m else n[:30]
        big = "big" if g >= 256 * 1000 else "small"
        agg[(name, big)][0] += 1
        agg[(name, big)][1] += (e - s) / 1e3
        tot += e - s
        gaps += max(0, s - prev)
        prev = e
    print("sum dur %.1f gaps %.1f" % (tot / 1e3, gaps / 1e3))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-40s %-6s n %3d  total %8.1f us" % (k[0], k[1], v[0], v[1]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
