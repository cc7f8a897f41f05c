#!/bin/bash
# kernel micro-benchmarks + PMC passes.  usage: gpurun --timeout 1200 -- 'bash scripts/gpu_kbench.sh tag'
TAG=${1:-k01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/kbench.py --reps 10 > $OUT/kbench.txt 2> $OUT/kbench.err; echo "kbench exit $?"; grep -v "^{" $OUT/kbench.txt; tail -3 $OUT/kbench.err
if [ ! -f gpurun_out/counters_list.txt ]; then rocprofv3 -L > gpurun_out/counters_list.txt 2>&1; fi
if [ -n "$PMC" ]; then
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -o k -- python scripts/kbench.py --reps 2 > $OUT/pmc$i.log 2>&1
    echo "pmc pass $i ($set) exit $?"
    find $OUT/pmc$i -name "*kernel_trace*" -delete
  done
fi
