#!/bin/bash
TAG=${1:-r04v}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --tb=short -p no:cacheprovider -k "short_dictionary" > $OUT/pytest_fused.log 2>&1; echo "pytest exit $?"; tail -12 $OUT/pytest_fused.log
for t in 16=0 16=1; do
GCCNMF_TUNE=$t timeout 300 python scripts/kbench.py --K 128 --reps 10 > $OUT/stages_$t.txt 2>&1; grep -E '^K[12].* fused' $OUT/stages_$t.txt
done
GCCNMF_TUNE=16=1 timeout 300 python scripts/ktrace_fused.py --stage 1 --slab > $OUT/ktrace_fused12.txt 2>&1; cat $OUT/ktrace_fused12.txt
