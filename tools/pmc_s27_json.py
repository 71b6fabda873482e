#!/usr/bin/env python3
"""profiles/rNN_pmc_s27_sweep.json from a tools/pmc_s27.sh directory (FETCH_SIZE / WRITE_SIZE passes): HBM traffic of ONE
fine-grid sweep of the 27-point kernels = the four pair launches of the largest grid (plain variants: no norm, no residuals).

    python tools/pmc_s27_json.py <pmc dir> <out.json> <source note>"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def counter(root, name):
    per_pair = collections.defaultdict(list)
    durs = collections.defaultdict(list)
    grid = 0
    rows = []
    for path in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != name:
                continue
            m = re.search(r"s27_sweep_kernel<float, 4, (\d), false, false, false>", r["Kernel_Name"])
            if not m:
                continue
            rows.append((int(m.group(1)), int(r["Grid_Size"]), float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
            grid = max(grid, int(r["Grid_Size"]))
    for pair, g, v, d in rows:
        if g == grid:
            per_pair[pair].append(v)
            durs[pair].append(d)
    return {p: sum(v) / len(v) for p, v in per_pair.items()}, {p: sum(v) / len(v) for p, v in durs.items()}, grid


def main():
    root, out, note = sys.argv[1:4]
    fetch, dur, grid = counter(root, "FETCH_SIZE")
    write, _, _ = counter(root, "WRITE_SIZE")
    assert sorted(fetch) == [0, 1, 2, 3], fetch
    read_b = int(round(sum(2 * fetch[p] * 1024 for p in range(4))))
    write_b = int(round(sum(write[p] * 1024 for p in range(4))))
    n = grid // 256 * 64 * 4 * 8 if grid else 0        # (not used: the bench line carries the unknowns)
    doc = collections.OrderedDict([
        ("kernel", "s27_sweep_kernel<float, 4, pair 0..3> x 4: one 8-colour Gauss-Seidel sweep of the fine grid (grid %d threads per launch)" % grid),
        ("fetch_size_KiB_per_pair", {str(p): round(fetch[p], 1) for p in range(4)}),
        ("fetch_correction", "x2 on gfx950 (MI355X_MICROARCH.md, HBM): reads = 2 * FETCH_SIZE"),
        ("write_size_KiB_per_pair", {str(p): round(write[p], 1) for p in range(4)}),
        ("read_bytes", read_b), ("write_bytes", write_b), ("traffic_bytes", read_b + write_b),
        ("avg_duration_under_collection_us_per_pair", {str(p): round(dur[p], 1) for p in range(4)}),
        ("source", note),
        ("kernel_src_sha", bench.kernel_source_hash()),
        ("git_head_at_collection", os.environ.get("OMG_GIT_HEAD") or bench.git_head() or "unknown"),
    ])
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
