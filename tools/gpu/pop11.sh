#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population11.txt
for shp in "252,256,252 2" "256,256,256 4" "240,256,240 4" "256,250,256 2"; do
  set -- $shp
  for place in 1 0 2; do
    echo "== shape $1 OMG_POOL_PLACE=$place" >> $o/population11.txt
    PROBE_SHAPE=$1 PROBE_RESTRICTIONS=$2 OMG_POOL_PLACE=$place OMG_POOL_TRIALS=5 OMG_SETUP_TIMING=1 timeout 300 python tools/population_probe.py 2>&1 | grep -E "candidate|pid|Error|error" | head -16 >> $o/population11.txt
  done
done
