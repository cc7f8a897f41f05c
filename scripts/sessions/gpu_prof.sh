#!/bin/bash
# rocprofv3 kernel stats of a few commands.  usage: gpurun -- 'bash scripts/gpu_prof.sh TAG'   -> gpurun_out/TAG/*_kernel_stats.csv
TAG=${1:-prof}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
prof() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/prof_$name.out 2> $OUT/prof_$name.err
  echo "rocprof $name exit $?"
  f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv && head -12 $f | cut -c1-200
  rm -rf $OUT/prof_$name
}
prof single_file python scripts/single_file.py --profile
K=128 HOP=128 prof single_file_K128_hop128 python scripts/single_file.py --profile
prof bench_K128_g1 python bench.py --dictionary-size 128 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
prof bench_K128_hop128_g1 python bench.py --dictionary-size 128 --hop 128 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
