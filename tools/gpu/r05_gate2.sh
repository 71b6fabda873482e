#!/bin/bash
python tools/gate_debug.py > gpurun_out/r05_gate_debug.txt 2>&1
for lz in 26 28 32; do
  echo "forced gate, one slab, inner chunk $lz planes" >> gpurun_out/r05_gate_lz.txt
  OMG_PDIST_GATE=2 OMG_PLANE_GATE_LZ=$lz PYTHONPATH=. python tools/pdist_loopback_time.py 1 2>&1 | grep world >> gpurun_out/r05_gate_lz.txt
done
echo "no gate" >> gpurun_out/r05_gate_lz.txt
OMG_PDIST_GATE=0 PYTHONPATH=. python tools/pdist_loopback_time.py 1 2>&1 | grep world >> gpurun_out/r05_gate_lz.txt
python -m pytest tests/test_gpu_march.py -x -q 2>&1 | tail -5 > gpurun_out/r05_t5.log
python tools/run_configs.py 0 > gpurun_out/r05_cfg0.txt 2>&1
