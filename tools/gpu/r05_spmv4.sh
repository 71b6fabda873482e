#!/bin/bash
for i in 1 2 3 4 5 6; do OMG_PLANE_SPMV_LZ=64 python tools/spmv_place.py 2>&1 | tail -1; done > gpurun_out/r05_spmv_place.txt
