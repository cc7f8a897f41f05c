#!/bin/bash
# Round 6: the direct (one mixture alone) kernels with their accumulators pinned to fixed AGPRs and the prologue's loads in order -- A/B against
# the previous build on one box, the kernels' tests, timelines.   usage: gpurun --timeout 1500 -- 'bash scripts/sessions/r06ab.sh [tag]'
TAG=${1:-r06ab}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
NEW=$PWD/gcc_nmf_amd/libgccnmf_hip.so
OLD=$PWD/gcc_nmf_amd/libgccnmf_hip_prev.so
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
echo "== A/B (old | new, twice)"
for rep in 1 2; do
  for cfg in "1024 256 1" "128 128 1" "128 256 1" "256 256 1" "1024 256 4"; do
    set -- $cfg
    for lib in $OLD $NEW; do
      [ -f $lib ] || continue
      K=$1 HOP=$2 FILES=$3 GCCNMF_HIP_LIB=$lib timeout 300 python scripts/direct_ab.py 2>&1 | tail -1
    done
  done
done | tee $OUT/direct_ab.txt
echo "== tests of the direct kernels"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "direct or small_batches or alone" > $OUT/pytest_direct.log 2>&1; echo "exit $? $(grep -E 'passed|failed' $OUT/pytest_direct.log | tail -1)"
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "alone or direct" > $OUT/pytest_direct2.log 2>&1; echo "exit $? $(grep -E 'passed|failed' $OUT/pytest_direct2.log | tail -1)"
echo "== timelines (lab build)"
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace_single.py > $OUT/ktrace_single.txt 2>&1; echo "ktrace exit $?"; cat $OUT/ktrace_single.txt
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace_single.py --K 128 > $OUT/ktrace_single_K128.txt 2>&1; echo "ktrace exit $?"; grep -E "^stage|main loop t2|end " $OUT/ktrace_single_K128.txt
echo "== rocprofv3 kernel stats, one mixture"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_single -o single -- python scripts/single_file.py --profile > $OUT/prof_single.out 2>&1
f=$(find $OUT/prof_single -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp $f $OUT/single_file_kernel_stats.csv && head -7 $f | cut -c1-160
rm -rf $OUT/prof_single
