#!/bin/bash
# fused slab kernel (K3 + K4a): timeline only
TAG=${1:-r04r}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
GCCNMF_TUNE=16=0,17=1 timeout 300 python scripts/kbench.py --K 128 --reps 10 > $OUT/stages.txt 2>&1; grep -E '^K[0-9].* fused' $OUT/stages.txt
GCCNMF_TUNE=17=1 timeout 300 python scripts/ktrace_fused.py --stage 3 > $OUT/ktrace_fused34.txt 2>&1; cat $OUT/ktrace_fused34.txt
