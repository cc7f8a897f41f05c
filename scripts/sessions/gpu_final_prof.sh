#!/bin/bash
# rocprofv3 kernel stats of the bench command on HEAD: as the timed steps run (two file groups) and as `roofline` is timed (--nmf-groups 1)
OUT=gpurun_out/r05_final; mkdir -p $OUT; export TMPDIR=/tmp
for g in default g1; do
  extra=""; [ $g = g1 ] && extra="--nmf-groups 1"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$g -o bench -- python bench.py --gpus 1 --steps 3 --warmup 1 --skip-extras --skip-config-lines --skip-cpu-baseline --no-live-traffic $extra > $OUT/prof_$g.json 2> $OUT/prof_$g.err
  echo "prof $g exit $?"
  f=$(find $OUT/prof_$g -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/${g}_bench_kernel_stats.csv; head -5 $OUT/${g}_bench_kernel_stats.csv | cut -c1-160
  rm -rf $OUT/prof_$g
done
