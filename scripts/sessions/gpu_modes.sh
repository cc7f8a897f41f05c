#!/bin/bash
# Kept evidence for the two secondary bench modes (shared dictionary = BASELINE config 4 on one GPU, streaming = config 5):
# the un-profiled JSON line and the rocprofv3 kernel stats of the same command.
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_modes.sh [tag]'
TAG=${1:-r02}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for mode in shared-dictionary streaming; do
  timeout 300 python bench.py --mode $mode --steps 3 --warmup 1 > $OUT/${mode}_bench.json 2> $OUT/${mode}_bench.err
  echo "$mode bench exit $?"; cat $OUT/${mode}_bench.json; tail -3 $OUT/${mode}_bench.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$mode -o bench -- python bench.py --mode $mode --steps 2 --warmup 1 > $OUT/prof_${mode}.json 2> $OUT/prof_${mode}.err
  echo "$mode rocprof exit $?"
  find $OUT/prof_$mode -name "*kernel_stats*.csv" | head -1 | xargs -r head -12
  find $OUT/prof_$mode -name "*kernel_trace*.csv" -size +10M -delete
done
