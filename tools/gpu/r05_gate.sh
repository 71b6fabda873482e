#!/bin/bash
# gated slab passes: tests, device cost of the one-launch inner + edge form (one slab, OMG_PDIST_GATE=0/1/2), loopback timing
python -m pytest tests/test_gpu_plane_dist.py -x -q 2>&1 | tail -15 > gpurun_out/r05_t4.log; tail -4 gpurun_out/r05_t4.log
for m in 0 1 2; do OMG_DIST_P2P=0 OMG_PDIST_GATE=$m timeout 300 python bench.py --dist 1 --no-cpu 2>/dev/null | tail -1 > gpurun_out/r05_dist1_gate$m.json; done
PYTHONPATH=. python tools/pdist_loopback_time.py 1 2 8 2>&1 | grep world > gpurun_out/r05_loop_gate1.txt
OMG_PDIST_GATE=0 PYTHONPATH=. python tools/pdist_loopback_time.py 2 8 2>&1 | grep world > gpurun_out/r05_loop_gate0.txt
