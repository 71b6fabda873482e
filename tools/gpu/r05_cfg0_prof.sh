#!/bin/bash
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r05_cfg0_prof -o c -- python3 $root/tools/run_configs.py 0 > /dev/null 2>&1
python3 $root/tools/trace_stats.py $root/gpurun_out/r05_cfg0_prof/c_kernel_trace.csv > $root/gpurun_out/r05_cfg0_kernels.txt 2>&1
python3 $root/tools/trace_dump.py $root/gpurun_out/r05_cfg0_prof/c_kernel_trace.csv 40 > $root/gpurun_out/r05_cfg0_timeline.txt 2>&1
rm -rf $root/gpurun_out/r05_cfg0_prof
