#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_stencil27.py tests/test_gpu_update.py tests/test_gpu_configs.py -x -q > $o/s27place_tests.log 2>&1
: > $o/s27place.txt
for t in 4 1 4 1 4 1; do
echo "== OMG_S27_TRIALS=$t" >> $o/s27place.txt
OMG_S27_TRIALS=$t OMG_SETUP_TIMING=1 timeout 600 python tools/update_probe.py 256 5 2>&1 | grep -E "27-point tiles|placement of the finest 27|^update" | head -8 >> $o/s27place.txt
OMG_S27_TRIALS=$t timeout 600 python tools/config4_probe.py --size 256 --cache /tmp/cfg4 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('config4_probe (host-list route):', d['vcycles_per_s'], d['ms_per_cycle'], {k: v['avg_us'] for k, v in d['kernels'].items()})" >> $o/s27place.txt 2>&1
done
