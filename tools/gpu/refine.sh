#!/bin/bash
# the search vector by vector alone (no pool candidates): what it rescues from an ordinary hipMalloc'ed pool
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/refine_$1.txt
for route in lists from_fine lists from_fine; do
OMG_POOL_TRIALS=1 OMG_POOL_REFINE=4 OMG_SETUP_TIMING=1 timeout 300 python tools/population_probe.py $route 2>&1 | grep -E "candidate|pid" >> $o/refine_$1.txt
echo "--" >> $o/refine_$1.txt
done
