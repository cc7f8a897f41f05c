#!/bin/bash
TAG=${1:-p2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum" \
           "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -o k -- python scripts/kbench.py --reps 2 > $OUT/pmc$i.log 2>&1
  echo "pmc pass $i ($set) exit $?"
  find $OUT/pmc$i -name "*kernel_trace*" -delete
done
