#!/bin/bash
cd "$(dirname "$0")/../.."; o=gpurun_out
: > $o/population6.txt
for pad in 0 256 512 1024 2048 4096 8192 16384 32768 65536 131072 262144 524288 1048576 4352 20736 86272 348416 1396992 12288 49152 196608 786432 3072; do
  echo "== contiguous, OMG_POOL_PAD=$pad" >> $o/population6.txt
  OMG_POOL_PAD=$pad OMG_POOL_CONTIG=1 OMG_POOL_TRIALS=2 OMG_SETUP_TIMING=1 timeout 200 python tools/population_probe.py 2>&1 | grep -E "trial|pid" | head -3 >> $o/population6.txt
done
