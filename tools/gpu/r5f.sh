#!/bin/bash
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out
timeout 900 python tools/ab_libs.py openmg_amd/lib/libopenmg_hip.so openmg_amd/lib/libopenmg_hip_prio3.so 5 > $o/r5f_setprio.txt 2>&1
OMG_SETUP_TIMING=1 timeout 600 python tools/config4_probe.py --size 256 --cache /tmp/cfg4 > $o/r5f_config4_setup.txt 2>&1
