"""profiles/rNN_peer_mode.txt: kernel durations of one slab (no neighbour) through the plane-slab runner with the
copy / RCCL path (mode 0) and in peer mode (mode 1), from two rocprofv3 --kernel-trace runs of tools/prof_pdist.py.
usage: peer_mode_table.py <trace dir of mode 0> <trace dir of mode 1>"""
import collections
import csv
import glob
import re
import sys


def main():
    print("Peer mode's price on one rank (no neighbour): tools/prof_pdist.py <mode> 20 under rocprofv3 --kernel-trace, 256^3, "
          "three slab levels over a 32^3 tail.")
    print("mode 0 = copies / RCCL path, mode 1 = peer stores with waiting passes.  Average kernel duration in us per "
          "(instantiation, grid, workgroup).")
    print("plane_kernel<V, MODE (0 down / 1 up), NORM, XZ (iterate zero), LA, PEER, max threads, FIRST>")
    print()
    for m, d in enumerate(sys.argv[1:3]):
        path = glob.glob(d + "/*kernel_trace.csv")[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"]
            k = re.search(r"(plane_kernel<[^>]*>|block_kernel<[^>]*>|pd_\w+|sine_solve_kernel<[^>]*>)", n)
            if not k:
                continue
            acc[(k.group(1), r["Grid_Size_X"], r["Workgroup_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print("mode %d" % m)
        for k, v in sorted(acc.items()):
            print("  %-62s grid %-7s wg %-4s n %3d avg %7.1f us" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
