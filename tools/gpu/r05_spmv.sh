#!/bin/bash
out=gpurun_out/r05_spmv.txt
: > $out
OMG_PLANE_SPMV=0 python tools/spmv_probe.py 256 >> $out 2>&1
for lz in 8 16 32 64; do OMG_PLANE_SPMV_LZ=$lz python tools/spmv_probe.py 256 >> $out 2>&1; done
python tools/spmv_probe.py 256 float32 >> $out 2>&1
OMG_PLANE_SPMV=0 python tools/spmv_probe.py 256 float32 >> $out 2>&1
python tools/spmv_probe.py 128 >> $out 2>&1
python -m pytest tests/test_gpu_plane.py -x -q 2>&1 | tail -3 >> $out
