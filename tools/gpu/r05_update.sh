#!/bin/bash
python -m pytest tests/test_gpu_update.py -x -q 2>&1 | tail -12 > gpurun_out/r05_t7.log
OMG_SETUP_TIMING=1 python tools/update_probe.py 256 5 > gpurun_out/r05_update_probe.txt 2>&1
