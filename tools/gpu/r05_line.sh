#!/bin/bash
python -m pytest tests/test_gpu_march.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5 > gpurun_out/r05_t6.log
python tools/run_configs.py 0 > gpurun_out/r05_cfg0.txt 2>&1
bash tools/gpu/r05_cfg0_prof.sh
