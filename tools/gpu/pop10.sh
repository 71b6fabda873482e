#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population10.txt
for route in from_fine lists from_fine lists from_fine lists; do
  OMG_SETUP_TIMING=1 timeout 300 python tools/population_probe.py $route 2>&1 | grep -E "candidate|pid|placement of" >> $o/population10.txt
  echo "--" >> $o/population10.txt
done
for i in 1 2 3 4 5 6; do OMG_POOL_TRIALS=1 timeout 200 python tools/population_probe.py $( [ $((i%2)) = 0 ] && echo lists ) >> $o/population10.txt 2>&1; done
