#!/bin/bash
python -m pytest tests/test_gpu_coarse.py tests/test_gpu_update.py tests/test_gpu_stencil27.py tests/test_gpu_dist27.py -x -q 2>&1 | tail -6 > gpurun_out/r05_t8.log
OMG_SETUP_TIMING=1 python tools/update_probe.py 256 5 2>&1 | grep -E "update|coarse fact|norm" | tail -8 > gpurun_out/r05_update_probe2.txt
