#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population4.txt
for what in 1 2 3 1 2 3; do
OMG_POOL_TRIALS=8 OMG_POOL_TRIALS_WHAT=$what OMG_SETUP_TIMING=1 timeout 300 python tools/population_probe.py 2>&1 | grep -E "WHAT|pid" >> $o/population4.txt
echo "--" >> $o/population4.txt
done
