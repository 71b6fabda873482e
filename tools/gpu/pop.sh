#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population.txt
for route in lists from_fine; do
for mode in new old new old; do
  echo "== $route $mode" >> $o/population.txt
  for i in 1 2 3 4 5; do
    if [ $mode = old ]; then OMG_POOL_TRIALS=1 OMG_POOL_EARLY=0 timeout 200 python tools/population_probe.py $route >> $o/population.txt 2>&1
    else timeout 200 python tools/population_probe.py $route >> $o/population.txt 2>&1; fi
  done
done
done
OMG_SETUP_TIMING=1 timeout 200 python tools/population_probe.py lists 2>&1 | grep -E "trial|placement|pid" >> $o/population.txt
