#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_placement.py tests/test_gpu_dist27.py tests/test_gpu_stencil27.py -x -q > $o/placetest.log 2>&1
OMG_SETUP_TIMING=1 timeout 600 python bench.py --dist 1 --stencil 27var --dtype f32 --steps 20 --repeats 5 2> $o/placetest.err | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('dist1 27var', d['value'], d['ms_per_step'])" >> $o/placetest.log
grep -E "27-point tiles|placement of a large" $o/placetest.err >> $o/placetest.log
