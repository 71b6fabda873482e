#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
timeout 900 python -m pytest tests/test_gpu_coarse.py tests/test_gpu_update.py -x -q > $o/r5g_tests.log 2>&1
