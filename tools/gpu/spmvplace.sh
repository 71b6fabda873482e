#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/spmvplace.txt
for t in 5 1 5 1 5 1 5 1; do
echo "== OMG_SPMV_TRIALS=$t" >> $o/spmvplace.txt
OMG_SPMV_TRIALS=$t OMG_SETUP_TIMING=1 timeout 300 python tools/spmv_probe.py 256 2>&1 | grep -E "SpMV destination|per launch" >> $o/spmvplace.txt
done
timeout 600 python -m pytest tests/test_gpu_plane.py -x -q 2>&1 | tail -2 >> $o/spmvplace.txt
