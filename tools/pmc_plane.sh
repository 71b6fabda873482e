#!/bin/bash
# Hardware-counter passes over the plane-pipelined kernels (plane.hip), one rocprofv3 run per counter set.
#     [PASSES="1 5 6"] bash tools/pmc_plane.sh gpurun_out/pmc_plane [prof_cycle.py arguments]
out=${1:-gpurun_out/pmc_plane}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "plane_kernel" --output-format csv \
      -d "$root/$out/p$i" -- python3 "$root/tools/prof_cycle.py" --steps 2 "$@" > "$root/$out/p$i.log" 2>&1
  echo "pass $i ($set): rc=$?"
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_SMEM
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
FETCH_SIZE
WRITE_SIZE
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
SETS
