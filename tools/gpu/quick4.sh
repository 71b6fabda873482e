#!/bin/bash
# headline + configs[4] legs of bench.py, twice in a row on one box; first call also the tests of the searched paths
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/quick4_$1.txt
if [ "$1" = 1 ]; then timeout 1500 python -m pytest tests/test_gpu_placement.py tests/test_gpu_stencil27.py tests/test_gpu_dist27.py tests/test_gpu_plane_dist.py tests/test_gpu_update.py -x -q 2>&1 | tail -1 >> $o/quick4_$1.txt; fi
for i in 1 2; do
timeout 600 python bench.py --no-cpu --no-plain --no-lex --no-sets --no-dropin --no-config1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d['config']['process_population']; c = d['config4']
print('%.1f V-cycles/s  frac %.3f  %s  spmv %.3f | configs[4] %.1f V-cycles/s  %.4f ms  sweep frac %.3f' % (d['value'], d['roofline']['frac'], p['which'], d['fine_grid_spmv']['frac'], c['vcycles_per_s'], c['ms_per_step'], c['roofline']['frac']))" >> $o/quick4_$1.txt
done
