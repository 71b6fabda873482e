#!/bin/bash
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out
timeout 900 python -m pytest tests/test_gpu_coarse.py tests/test_gpu_update.py tests/test_gpu_dist27.py tests/test_gpu_fp32.py -x -q > $o/r5c_tests.log 2>&1
OMG_SETUP_TIMING=1 timeout 600 python tools/update_probe.py 256 5 > $o/r5c_update.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/r5c_trace -o u -- python3 $root/tools/update_probe.py 256 5 > /dev/null 2>&1
grep -E "gjb|s27_rap|s27_build|extract|Name|fill_aug|copy_block|narrow|scatter" $o/r5c_trace/u_kernel_stats.csv > $o/r5c_update_kernels.csv
rm -rf $o/r5c_trace
