#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population2.txt
for i in 1 2 3; do
OMG_POOL_TRIALS=12 OMG_SETUP_TIMING=1 timeout 300 python tools/population_probe.py 2>&1 | grep -E "trial|pid|level 0 vectors" >> $o/population2.txt
echo "--" >> $o/population2.txt
done
