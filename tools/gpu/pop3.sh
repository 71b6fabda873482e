#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population3.txt
for i in 1 2 3 4 5 6; do timeout 300 python tools/population_smi.py >> $o/population3.txt 2>&1; done
