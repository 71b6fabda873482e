#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population9.txt
for cfg in "2 from_fine" "2 lists" "0 from_fine" "0 lists" "2 from_fine" "2 lists" "32 from_fine" "8 from_fine"; do
  set -- $cfg
  echo "== OMG_POOL_PLACE=$1 $2" >> $o/population9.txt
  OMG_POOL_PLACE=$1 OMG_POOL_TRIALS=4 OMG_SETUP_TIMING=1 timeout 300 python tools/population_probe.py $2 2>&1 | grep -E "trial|pid|placement of" >> $o/population9.txt
done
