#!/bin/bash
python tools/gate_debug.py > gpurun_out/r05_gate_debug.txt 2>&1
python -m pytest tests/test_gpu_plane_dist.py -x -q 2>&1 | tail -15 > gpurun_out/r05_t4.log
