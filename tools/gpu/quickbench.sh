#!/bin/bash
# the headline leg of bench.py alone, twice in a row on one box (profiles/r05_bench_repeats.txt: one gpurun call per line pair)
cd "$(dirname "$0")/../.."; o=gpurun_out
for i in 1 2; do
timeout 300 python bench.py --no-cpu --no-plain --no-lex --no-sets --no-dropin --no-config4 --no-config1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['config']['process_population']
print('%.1f V-cycles/s  %.4f ms  frac %.3f  %s  spmv %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], p['which'], d['fine_grid_spmv']['frac']))" >> $o/quickbench_$1.txt
done
