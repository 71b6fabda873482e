#!/bin/bash
out=gpurun_out/r05_gate_w8.txt
: > $out
run() { echo "== $*" >> $out; env "$@" PYTHONPATH=. timeout 120 python tools/pdist_loopback_time.py 8 2>&1 | grep -E "world|Error|gave up" | tail -2 >> $out; }
run OMG_PDIST_GATE=1
run OMG_PDIST_GATE=1 OMG_PLANE_GATE_LZ=28
run OMG_PDIST_GATE=1 GPU_MAX_HW_QUEUES=8
run OMG_PDIST_GATE=1 OMG_PLANE_TILE=128,22,16
run OMG_PDIST_GATE=0
echo "== world 4 gate 1" >> $out; OMG_PDIST_GATE=1 PYTHONPATH=. timeout 120 python tools/pdist_loopback_time.py 4 2>&1 | grep -E "world|Error|gave up" | tail -2 >> $out
