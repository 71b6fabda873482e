#!/bin/bash
# counters of the closed-form Galerkin kernel during update_fine (256^3, fp32 levels)
cd "$(dirname "$0")/../.."; root=$(pwd); o=$root/gpurun_out
OMG_SETUP_TIMING=1 timeout 600 python tools/update_probe.py 256 5 > $o/r5d_update.txt 2>&1
mkdir -p $o/r5d_pmc
cd /tmp && export TMPDIR=/tmp
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "s27_rap" --output-format csv -d "$o/r5d_pmc/p$i" -- python3 "$root/tools/update_probe.py" 256 5 > "$o/r5d_pmc/p$i.log" 2>&1
  echo "pass $i ($set): rc=$?"
done <<'SETS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
FETCH_SIZE
WRITE_SIZE
SETS
cd $root; python tools/pmc_any.py gpurun_out/r5d_pmc > $o/r5d_pmc.txt 2>&1
rm -rf $o/r5d_pmc
