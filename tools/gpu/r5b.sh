#!/bin/bash
# round 5, second half: the blocked inverse's new kernels + tiling experiments
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out
timeout 900 python -m pytest tests/test_gpu_coarse.py tests/test_gpu_update.py tests/test_gpu_fp32.py -x -q > $o/r5b_tests.log 2>&1
OMG_SETUP_TIMING=1 timeout 600 python tools/update_probe.py 256 5 > $o/r5b_update.txt 2>&1
OMG_PLANE_TUNE_DEBUG=1 OMG_PLANE_TUNE_EXTRA="64,20,30;64,20,26;64,20,32;64,20,22;128,6,26" timeout 300 python tools/prof_cycle.py --steps 4 > $o/r5b_tiles.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/r5b_trace -o u -- python3 $root/tools/update_probe.py 256 5 > /dev/null 2>&1
grep -E "gjb|s27_rap|s27_build|extract|Name" $o/r5b_trace/u_kernel_stats.csv > $o/r5b_update_kernels.csv
rm -rf $o/r5b_trace
