#!/usr/bin/env python3
"""Per (kernel, grid size) averages of every counter under a pmc_*.sh output directory.
    python tools/pmc_any.py gpurun_out/pmc_plane [name-regex]"""
import collections, csv, glob, os, re, sys


def short(name):
    m = re.search(r"plane_kernel<(\w+), (\d+), (true|false), (true|false)>", name)
    if m:
        return "plane_kernel<%s, %s%s%s>" % (m.group(1), "down" if m.group(2) == "0" else "up", ", norm" if m.group(3) == "true" else "",
                                             ", x=0" if m.group(4) == "true" else "")
    m = re.search(r"s27_sweep_kernel<(\w+), (\d+), (\d+), (true|false), (true|false), (true|false)>", name)
    if m:
        return "s27_sweep<%s rg%s pair%s%s%s%s>" % (m.group(1), m.group(2), m.group(3), " x=0" if m.group(4) == "true" else "",
                                                    " +old-norm" if m.group(5) == "true" else "", " +res67" if m.group(6) == "true" else "")
    m = re.search(r"s27_residual_kernel<(\w+), (\d+), (\d+), (\d+)>", name)
    if m:
        return "s27_residual<%s rg%s %s colours %s>" % (m.group(1), m.group(2), m.group(3), "restrict" if m.group(4) == "0" else "norm")
    return re.sub(r"\(.*", "", name)[:80]


def main():
    root = sys.argv[1]
    rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    table = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            if rx and not rx.search(r["Kernel_Name"]):
                continue
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r.get("Workgroup_Size", 0) or 0))
            table[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r:
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for key in sorted(table, key=lambda k: (-k[1], k[0])):
        print("%s grid %d wg %d   (avg duration under collection %.1f us)" % (key + (sum(dur[key]) / max(len(dur[key]), 1),)))
        for name in sorted(table[key]):
            v = table[key][name]
            extra = ""
            if name == "FETCH_SIZE":
                extra = "   -> read bytes (x 2, gfx950) %.1f MB" % (2 * 1024 * sum(v) / len(v) / 1e6)
            if name == "WRITE_SIZE":
                extra = "   -> written bytes %.1f MB" % (1024 * sum(v) / len(v) / 1e6)
            print("    %-40s %16.1f   (n=%d)%s" % (name, sum(v) / len(v), len(v), extra))


if __name__ == "__main__":
    main()
