#!/bin/bash
# Hardware-counter passes over the V-cycle's row kernels, one rocprofv3 run per counter set
# (counters of different blocks cannot all be collected at once).  Run on the GPU box:
#     [PASSES="1 7 8"] bash tools/pmc_passes.sh gpurun_out/pmc [prof_cycle.py arguments]
# --kernel-include-regex keeps the 16k tiny Gauss-Jordan dispatches of the setup out of the
# collection (they make it take minutes); every pass has its own timeout.
out=${1:-gpurun_out/pmc}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  timeout 240 rocprofv3 --pmc $set --kernel-include-regex "rows_(pattern_|union_)?kernel" --output-format csv \
      -d "$root/$out/p$i" -- python3 "$root/tools/prof_cycle.py" --steps 2 "$@" > "$root/$out/p$i.log" 2>&1
  echo "pass $i ($set): rc=$?"
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_SMEM
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
FETCH_SIZE
WRITE_SIZE
SETS
