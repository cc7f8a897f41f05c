#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_round.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname --showmeminfo vram > $OUT/rocm-smi.txt 2>&1
echo "== pytest -m gpu" 
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprofv3 kernel stats"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --gpus 1 --steps 2 --warmup 1 --skip-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof exit $?"; ls -R $OUT/prof | head -20
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -r head -30
# keep the merged-back payload small: drop the raw per-dispatch trace if it is huge
find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
