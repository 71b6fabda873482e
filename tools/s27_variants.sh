#!/bin/bash
# A/B of stencil27.hip builds (openmg_amd/lib/libopenmg_hip_<name>.so, made with make EXTRA="-DOMG_EXPERIMENTS -DS27_..."): configs[4] at 256^3
out=${1:-gpurun_out/s27_variants.txt}
: > $out
for lib in openmg_amd/lib/libopenmg_hip_*.so; do
    name=$(basename $lib .so); name=${name#libopenmg_hip_}
    echo "== $name" >> $out
    OMG_LIB_PATH=$PWD/$lib timeout 600 python tools/config4_probe.py --size 256 --cache /tmp/cfg4 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['vcycles_per_s'], d['ms_per_cycle'], {k: v['avg_us'] for k, v in d['kernels'].items()})" >> $out 2>&1
done
cat $out
