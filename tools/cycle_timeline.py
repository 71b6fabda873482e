#!/usr/bin/env python3
"""Print the kernel timeline of one V-cycle from a rocprofv3 kernel_trace.csv."""
import csv
import re
import sys


def main(path, which=-4):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]))
                for r in rows)
    idx = [i for i, e in enumerate(ev) if "sum_kernel" in e[2]]
    a, b = idx[which], idx[which + 1]
    cyc = ev[a + 1:b + 1]
    t0 = cyc[0][0]
    tot = 0
    prev_end = ev[a][1]
    print("cycle wall us %.1f, %d kernels" % ((cyc[-1][1] - ev[a][1]) / 1e3, len(cyc)))
    gaps = 0
    for s, e, n, g in cyc:
        m = re.search(r"(rows_kernel<\d|rows_pattern_kernel<\d|rows_union_kernel<\d|rows_serial_kernel<\d|\w+_kernel|__amd\w+)", n)
        label = m.group(1) if m else n[:30]
        pm = re.search(r"plane_kernel<(\w+), (\d+), (true|false), (true|false)", n)
        if pm:
            label = "plane_kernel<%s%s%s>" % ("down" if pm.group(2) == "0" else "up", ", norm" if pm.group(3) == "true" else "",
                                              ", x=0" if pm.group(4) == "true" else "")
        print("%8.1f gap %6.1f dur %7.1f  %s grid %d" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, label, g))
        tot += e - s
        gaps += max(0, s - prev_end)
        prev_end = e
    print("sum of kernel durations us %.1f, gaps us %.1f" % (tot / 1e3, gaps / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -4)
