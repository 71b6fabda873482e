#!/bin/bash
python -m pytest tests/test_gpu_plane_dist.py -x -q 2>&1 | tail -8 > gpurun_out/r05_t4.log
PYTHONPATH=. python tools/pdist_loopback_time.py 2 4 8 2>&1 | grep -E "world|Error" > gpurun_out/r05_loop_gate1.txt
OMG_PDIST_GATE=0 PYTHONPATH=. python tools/pdist_loopback_time.py 2 4 8 2>&1 | grep -E "world|Error" > gpurun_out/r05_loop_gate0.txt
