#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population8.txt
timeout 300 python tools/prof_config4.py --steps 1 > /dev/null 2>&1
for cfg in "0 0" "2 0" "2 2" "2 8" "2 32" "0 2" "2 2" "0 0"; do
  set -- $cfg
  echo "== OMG_POOL_PLACE=$1 OMG_S27_PLACE=$2" >> $o/population8.txt
  OMG_POOL_PLACE=$1 OMG_S27_PLACE=$2 timeout 600 python tools/config4_probe.py --size 256 --cache /tmp/cfg4 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['vcycles_per_s'], d['ms_per_cycle'], {k: v['avg_us'] for k, v in d['kernels'].items()})" >> $o/population8.txt 2>&1
done
for i in 1 2 3 4; do timeout 200 python tools/population_probe.py lists >> $o/population8.txt 2>&1; done
for i in 1 2 3 4; do timeout 200 python tools/population_probe.py >> $o/population8.txt 2>&1; done
