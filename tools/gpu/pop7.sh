#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population7.txt
for place in 2 0 4 8 32 2 0 1; do
  echo "== OMG_POOL_PLACE=$place" >> $o/population7.txt
  OMG_POOL_PLACE=$place OMG_POOL_TRIALS=4 OMG_SETUP_TIMING=1 timeout 300 python tools/population_probe.py 2>&1 | grep -E "trial|pid|placement of" >> $o/population7.txt
done
