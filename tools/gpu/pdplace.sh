#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_plane_dist.py tests/test_gpu_dist.py -x -q > $o/pdplace_tests.log 2>&1
: > $o/pdplace.txt
for t in 5 1 5 1 5 1; do
echo "== OMG_PDIST_TRIALS=$t" >> $o/pdplace.txt
OMG_PDIST_TRIALS=$t OMG_DIST_P2P=0 OMG_SETUP_TIMING=1 timeout 300 python bench.py --dist 1 --no-cpu 2> $o/pdplace.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $o/pdplace.txt
grep -E "slab vectors" $o/pdplace.err >> $o/pdplace.txt
done
