#!/bin/bash
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out
: > $o/r5e_rap_variants.txt
for lib in openmg_amd/lib/libopenmg_hip_g*.so openmg_amd/lib/libopenmg_hip_g*.so; do
  echo "== $lib" >> $o/r5e_rap_variants.txt
  OMG_LIB_PATH=$root/$lib OMG_SETUP_TIMING=1 timeout 300 python tools/update_probe.py 256 5 2>&1 | grep -E "Galerkin product of a level|^update" | tail -9 >> $o/r5e_rap_variants.txt
done
