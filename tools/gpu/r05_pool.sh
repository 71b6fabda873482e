#!/bin/bash
out=gpurun_out/r05_pool_tmp.txt
: > $out
for i in 1 2 3 4; do
  for m in 0 1; do
    OMG_POOL_TMP_OWN=$m timeout 200 python bench.py --no-cpu --no-plain --no-lex --no-sets --no-dropin --no-config4 --no-config1 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('tmp_own=$m', d['value'], d['ms_per_step'], d['roofline']['level0_kernels']['plane_down']['avg_us'], d['roofline']['level0_kernels']['plane_up']['avg_us'], d['fine_grid_spmv']['frac'])" >> $out
  done
done
