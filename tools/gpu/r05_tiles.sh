#!/bin/bash
out=gpurun_out/r05_tiles.txt
: > $out
OMG_PLANE_TUNE_DEBUG=1 OMG_PLANE_TUNE_EXTRA="64,44,26;64,44,28;64,44,30;64,44,32;32,88,26;128,22,26;128,20,26;64,36,26;64,42,30;96,30,30" timeout 300 python tools/spmv_probe.py 256 2>&1 | grep "plane tune" >> $out
