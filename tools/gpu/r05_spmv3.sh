#!/bin/bash
out=gpurun_out/r05_spmv3.txt
: > $out
for rep in 1 2; do
for lz in 16 32 64 128 256; do OMG_PLANE_SPMV_LZ=$lz python tools/spmv_probe.py 256 2>&1 | head -1 >> $out; done
done
OMG_PLANE_SPMV_LZ=64 python tools/spmv_probe.py 256 float32 2>&1 | head -1 >> $out
OMG_PLANE_SPMV_LZ=64 python tools/spmv_probe.py 128 2>&1 | head -1 >> $out
OMG_PLANE_SPMV_LZ=32 python tools/spmv_probe.py 128 2>&1 | head -1 >> $out
OMG_PLANE_SPMV_LZ=64 python tools/spmv_probe.py 512 2>&1 | head -1 >> $out
OMG_PLANE_SPMV_LZ=128 python tools/spmv_probe.py 512 2>&1 | head -1 >> $out
