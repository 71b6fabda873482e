#!/bin/bash
root=$(pwd)
out=$root/gpurun_out/r05_spmv2.txt
: > $out
OMG_PLANE_SPMV_T=512 python tools/spmv_probe.py 256 >> $out 2>&1
OMG_PLANE_SPMV_T=512 OMG_PLANE_SPMV_LZ=64 python tools/spmv_probe.py 256 >> $out 2>&1
OMG_PLANE_SPMV_LZ=64 python tools/spmv_probe.py 256 >> $out 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-include-regex "plane_spmv" --output-format csv -d $root/gpurun_out/r05_spmv_pmc/p$i -- python3 $root/tools/spmv_probe.py 256 > /dev/null 2>&1
  OMG_PLANE_SPMV=0 timeout 200 rocprofv3 --pmc $set --kernel-include-regex "plane_spmv" --output-format csv -d $root/gpurun_out/r05_spmv_pmc0/p$i -- python3 $root/tools/spmv_probe.py 256 > /dev/null 2>&1
done
cd $root
echo "== z-marching kernel" >> $out; python tools/pmc_any.py gpurun_out/r05_spmv_pmc >> $out 2>&1
echo "== one thread per pair" >> $out; python tools/pmc_any.py gpurun_out/r05_spmv_pmc0 >> $out 2>&1
rm -rf gpurun_out/r05_spmv_pmc gpurun_out/r05_spmv_pmc0
