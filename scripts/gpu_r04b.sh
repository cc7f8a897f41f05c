#!/bin/bash
TAG=${1:-r04b}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "direct or klnmf" > $OUT/pytest_direct.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/pytest_direct.log
for cfg in "1024 256 1" "128 128 1" "1024 256 8" "1024 256 12"; do
  timeout 300 python scripts/direct_bench.py $cfg > $OUT/direct_bench_$(echo $cfg | tr ' ' '_').txt 2>&1; echo "direct_bench $cfg exit $?"
  grep -v "^{" $OUT/direct_bench_$(echo $cfg | tr ' ' '_').txt | grep -v "^tile" | tail -8
done
