#!/bin/bash
# Round 6: full GPU suite (product + lab build), smoke and the driver's bench command after the FFT kernels' batched loads.
# usage: gpurun --timeout 2400 -- 'bash scripts/sessions/r06am.sh [tag]'
TAG=${1:-r06am}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 1700 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
GCCNMF_HIP_LIB=$EXP timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_gpu_exp.log 2>&1; echo "kernel + pipeline tests on the lab build: exit $? $(grep -E 'passed|failed' $OUT/pytest_gpu_exp.log | tail -1)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -n 1 $OUT/smoke.log
S0=$(date +%s); timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $? ($(( $(date +%s) - S0 )) s wall)"; cut -c1-300 $OUT/bench.json
timeout 300 python scripts/stage_times.py 2>&1 | tail -n 1 | tee $OUT/stage_times.txt
timeout 600 python bench.py --dictionary-size 128 --steps 5 --warmup 2 --skip-extras --skip-roofline --skip-cpu-baseline 2>/dev/null | cut -c1-220 | tee $OUT/bench_K128.txt
