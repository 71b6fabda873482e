#!/usr/bin/env python3
"""profiles/rNN_pmc_plane_down.json from a tools/pmc_plane.sh directory (FETCH_SIZE / WRITE_SIZE passes) and a bench.py line.

    python tools/pmc_plane_json.py <pmc dir> <bench.json> <out.json> <source note>

Records the hash of the kernel sources (bench.kernel_source_hash) and the commit of the collection:
bench.py copies traffic_bytes into roofline.traffic only while the sources still hash to it."""
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


def counter(root, name, grid):
    vals, durs = [], []
    for path in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != name:
                continue
            m = re.search(r"plane_kernel<double, 0, false, false", r["Kernel_Name"])     # down pass that reads x
            if not m or int(r["Grid_Size"]) != grid:
                continue
            vals.append(float(r["Counter_Value"]))
            durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return sum(vals) / len(vals), sum(durs) / len(durs), len(vals)


def main():
    root, bench_json, out, note = sys.argv[1:5]
    line = json.loads(open(bench_json).read().strip().splitlines()[-1])
    roof = line["roofline"]
    grid = roof["tiling"]["workgroups"] * roof["tiling"]["threads"]
    fetch, dur, n = counter(root, "FETCH_SIZE", grid)
    write, _, _ = counter(root, "WRITE_SIZE", grid)
    read_b, write_b = int(round(2 * fetch * 1024)), int(round(write * 1024))
    doc = collections.OrderedDict([
        ("kernel", "plane_kernel<double, down>, %d workgroups of %d threads (256^3 level 0: red-black sweep + residual + restriction)"
         % (roof["tiling"]["workgroups"], roof["tiling"]["threads"])),
        ("bytes_per_launch", roof["bytes_per_launch"]),
        ("csr_equiv_bytes", roof["csr_equiv_bytes"]),
        ("fetch_size_KiB", round(fetch, 1)),
        ("fetch_correction", "x2 on gfx950 (MI355X_MICROARCH.md, HBM): reads = 2 * FETCH_SIZE"),
        ("write_size_KiB", round(write, 1)),
        ("read_bytes", read_b), ("write_bytes", write_b), ("traffic_bytes", read_b + write_b),
        ("launches_averaged", n),
        ("avg_duration_under_collection_us", round(dur, 1)),
        ("source", note),
        ("kernel_src_sha", bench.kernel_source_hash()),
        ("git_head_at_collection", os.environ.get("OMG_GIT_HEAD") or bench.git_head() or "unknown"),
    ])
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
