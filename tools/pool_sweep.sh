#!/bin/bash
# offsets (bytes) of tmp and b inside the pooled allocation of the finest level's vectors: steady ms per cycle for each pair
for pair in "256 512" "256 768" "512 256" "192 384" "320 640" "256 1280" "768 1536" "256 2304" "2304 4608" "256 131328"; do
  set -- $pair
  python tools/ab_libs.py OMG_VEC_POOL=1+OMG_POOL_OFF1=$1+OMG_POOL_OFF2=$2 OMG_VEC_POOL=1+OMG_POOL_OFF1=$1+OMG_POOL_OFF2=$2 1 | head -1
done
