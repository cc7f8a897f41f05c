#!/bin/bash
# Round 4, first GPU session: the direct latency path (csrc/direct.hip) -- parity tests, per-stage times against the split-K path,
# K = 128 baselines.   usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r04a.sh [tag]'
TAG=${1:-r04a}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest direct + klnmf"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -x -k "direct or klnmf" > $OUT/pytest_direct.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/pytest_direct.log
echo "== direct_bench"
for cfg in "1024 256 1" "128 128 1" "128 256 1" "1024 256 4"; do
  timeout 300 python scripts/direct_bench.py $cfg > $OUT/direct_bench_$(echo $cfg | tr ' ' '_').txt 2>&1; echo "direct_bench $cfg exit $?"
  grep -v "^{" $OUT/direct_bench_$(echo $cfg | tr ' ' '_').txt | tail -14
done
echo "== pipeline tests"
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider -k "dev1 or repeats or all_reference or hop128" > $OUT/pytest_pipeline.log 2>&1
echo "pytest pipeline exit $?"; tail -5 $OUT/pytest_pipeline.log
echo "== bench K=128 (batch 64, hop 256)"
timeout 600 python bench.py --dictionary-size 128 --steps 3 --warmup 1 --skip-cpu-baseline > $OUT/bench_K128.json 2> $OUT/bench_K128.err; echo "bench K128 exit $?"; cut -c1-1500 $OUT/bench_K128.json
