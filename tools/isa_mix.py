#!/usr/bin/env python3
"""Instruction mix of the hot loop of each kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iopenmg_amd/csrc -Iinclude --cuda-device-only -S -o /tmp/plane.s openmg_amd/csrc/plane.hip
    python tools/isa_mix.py /tmp/plane.s 'plane_kernel<double, 0, false, false, 1, false, 512, false, true>'

The hot loop is taken to be the largest backward branch target range (label .. s_cbranch back to it)."""
import collections
import re
import subprocess
import sys


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def classify(op):
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "lane<->sgpr"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if re.match(r"v_(fma|mul|add|sub|rcp|div|max|min|fmac)_f64", op):
        return "fp64"
    if re.match(r"v_(fma|mul|add|sub|rcp|fmac|pk_fma|pk_mul|pk_add)_f32", op):
        return "fp32"
    if op.startswith("v_mov") or op.startswith("v_pk_mov") or op.startswith("v_accvgpr"):
        return "v_mov"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem" if not op.startswith("scratch_") else "scratch"
    if op.startswith("v_"):
        return "valu other"
    return "other"


def main():
    path, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
    lines = open(path).read().split("\n")
    starts = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    names = demangle([n for _, n in starts])
    for (i, n), nxt in zip(starts, [s for s, _ in starts[1:]] + [len(lines)]):
        dn = names[n]
        if want and want not in dn:
            continue
        body = lines[i:nxt]
        end = next((k for k, l in enumerate(body) if l.strip().startswith("s_endpgm")), len(body))
        meta = {}
        for l in lines[i:nxt]:
            m = re.match(r"\s*;\s*(NumVgprs|NumAgprs|ScratchSize|NumSgprs|Occupancy|LDSByteSize|codeLenInByte):\s*(\d+)", l)
            if m:
                meta[m.group(1)] = int(m.group(2))
        body = body[:end]
        label_at = {}
        for k, l in enumerate(body):
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                label_at[m.group(1)] = k
        best = None
        for k, l in enumerate(body):
            m = re.match(r"\s+s_cbranch\w*\s+(\.LBB\w+)", l) or re.match(r"\s+s_branch\s+(\.LBB\w+)", l)
            if m and m.group(1) in label_at and label_at[m.group(1)] < k:
                span = (label_at[m.group(1)], k)
                if best is None or span[1] - span[0] > best[1] - best[0]:
                    best = span
        print(dn[:170])
        print("   ", meta)
        if not best:
            print("    no loop")
            continue
        mix = collections.Counter()
        ops = collections.Counter()
        for l in body[best[0]:best[1] + 1]:
            m = re.match(r"^\s+([a-z]\w+)", l)
            if not m:
                continue
            mix[classify(m.group(1))] += 1
            ops[m.group(1)] += 1
        tot = sum(mix.values())
        print("    loop: %d instructions  " % tot + "  ".join("%s %d" % kv for kv in mix.most_common()))
        if want:
            print("    " + "  ".join("%s %d" % kv for kv in ops.most_common(45)))


if __name__ == "__main__":
    main()
