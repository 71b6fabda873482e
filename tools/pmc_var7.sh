#!/bin/bash
# Hardware-counter passes over the var7 passes (var7.hip), one rocprofv3 run per counter set.
#     bash tools/pmc_var7.sh gpurun_out/pmc_var7
out=${1:-gpurun_out/pmc_var7}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "var7_pass" --output-format csv \
      -d "$root/$out/p$i" -- python3 "$root/tools/var7_probe.py" 256 > "$root/$out/p$i.log" 2>&1
  echo "pass $i ($set): rc=$?"
done <<'SETS'
FETCH_SIZE
WRITE_SIZE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY
SETS
