#!/bin/bash
# Round 4, session after the short-dictionary slab launch: parity tests, smoke, the driver-contract bench line, the K = 128 lines
# (BASELINE config 1 / the reference driver's own parameters), rocprofv3 kernel stats of both.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_r04u.sh [tag]'
TAG=${1:-r04u}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 python bench.py --dictionary-size 128 --steps 5 --warmup 1 > $OUT/bench_K128.json 2> $OUT/bench_K128.err; echo "bench K128 exit $?"; cut -c1-200 $OUT/bench_K128.json
timeout 600 python bench.py --dictionary-size 128 --hop 128 --steps 5 --warmup 1 > $OUT/bench_K128_hop128.json 2> $OUT/bench_K128_hop128.err; echo "bench K128 hop128 exit $?"; cut -c1-200 $OUT/bench_K128_hop128.json
GCCNMF_TUNE=16=0,17=0 timeout 600 python bench.py --dictionary-size 128 --steps 5 --warmup 1 --skip-extras > $OUT/bench_K128_four_launches.json 2> /dev/null; echo "bench K128 (keys 16, 17 = 0) exit $?"; cut -c1-200 $OUT/bench_K128_four_launches.json
echo "== K = 128 stage times (kbench) and the slab launch's workgroup timeline"
timeout 300 python scripts/kbench.py --K 128 --reps 10 > $OUT/kbench_K128.txt 2>&1; grep -E '^K[0-9].* fused|^K4b' $OUT/kbench_K128.txt
timeout 300 python scripts/ktrace_fused.py --stage 3 > $OUT/ktrace_slab_K128.txt 2>&1; cat $OUT/ktrace_slab_K128.txt
timeout 300 python scripts/ktrace_fused.py --stage 1 --slab > $OUT/ktrace_column_tiles_K128.txt 2>&1; cat $OUT/ktrace_column_tiles_K128.txt
echo "== rocprofv3 kernel stats"
prof() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/prof_$name.out 2> $OUT/prof_$name.err
  echo "rocprof $name exit $?"
  f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv && head -6 $f | cut -c1-170
  rm -rf $OUT/prof_$name
}
prof bench python bench.py --gpus 1 --steps 2 --warmup 1 --skip-extras
prof K128_g1_bench python bench.py --dictionary-size 128 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
prof K128_hop128_g1_bench python bench.py --dictionary-size 128 --hop 128 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
