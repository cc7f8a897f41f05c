#!/bin/bash
TAG=${1:-r03d}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s --durations=8 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python -c "
import json;b=json.load(open('$OUT/bench.json'))
for k in ('value','ms_per_step','sec8d_host_to_host','single_file','dropin_performKLNMF','cpu_baseline','roofline'): print(k, json.dumps(b.get(k))[:300])"
