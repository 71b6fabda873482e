#!/bin/bash
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out
: > $o/r5h_dbg.txt
for d in 0 16 0 16 12 1 13 29; do
  OMG_LIB_PATH=$root/openmg_amd/lib/libopenmg_dbg.so OMG_PLANE_DBG=$d timeout 200 python tools/plane_dbg_times.py >> $o/r5h_dbg.txt 2>&1
done
