#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population5.txt
for c in 1 0 1 0; do
  echo "== OMG_POOL_CONTIG=$c (OMG_POOL_TRIALS=1)" >> $o/population5.txt
  for i in 1 2 3 4 5; do OMG_POOL_CONTIG=$c OMG_POOL_TRIALS=1 timeout 200 python tools/population_probe.py >> $o/population5.txt 2>&1; done
done
OMG_POOL_CONTIG=1 OMG_POOL_TRIALS=8 OMG_SETUP_TIMING=1 timeout 200 python tools/population_probe.py 2>&1 | grep -E "trial|pid" >> $o/population5.txt
