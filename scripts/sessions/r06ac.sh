#!/bin/bash
# Round 6: which of the two changes to the direct kernels' main loop costs time -- four builds on one box (previous | accumulators pinned |
# prologue loads in order | both).   usage: gpurun --timeout 1200 -- 'bash scripts/sessions/r06ac.sh [tag]'
TAG=${1:-r06ac}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
D=$PWD/gcc_nmf_amd
for rep in 1 2 3; do
  for cfg in "1024 256 1" "1024 256 4" "128 128 1"; do
    set -- $cfg
    for lib in $D/libgccnmf_hip_prev.so $D/libgccnmf_hip_pin.so $D/libgccnmf_hip_order.so $D/libgccnmf_hip.so; do
      [ -f $lib ] || continue
      K=$1 HOP=$2 FILES=$3 GCCNMF_HIP_LIB=$lib timeout 300 python scripts/direct_ab.py 2>&1 | tail -1
    done
  done
done | tee $OUT/direct_ab4.txt
