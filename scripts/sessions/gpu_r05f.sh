#!/bin/bash
# Round 5: epilogue per column block (16 independent quotients, then 16 stores) + tuning snapshot + product/experiment split, vs round 4.
TAG=${1:-r05g}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
OLD=$PWD/gcc_nmf_amd/libgccnmf_hip_r04.so
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
kb() { local name=$1; shift
  env "$@" timeout 300 python scripts/kbench.py --reps 8 ${KARGS} > $OUT/kbench_$name.txt 2> $OUT/kbench_$name.err
  echo "kbench $name exit $?"; grep -E "^K[1-4]a? fused|512 blocks" $OUT/kbench_$name.txt | cut -c1-110
}
kb r04 GCCNMF_HIP_LIB=$OLD
kb new GCCNMF_TUNE=
kb new_rcp GCCNMF_TUNE=7=0
kb r04_2 GCCNMF_HIP_LIB=$OLD
kb new_2 GCCNMF_TUNE=
b() { local name=$1; shift; env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --skip-extras ${EXTRA} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name exit $?"; python - $OUT/bench_$name.json <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=b.get('roofline',{}); print('  value %.0f  ms %.2f  k3 %.4f ms frac %.3f  iter %.3f ms' % (b['value'], b['ms_per_step'], r.get('avg_launch_ms',0), r.get('frac',0), b.get('nmf_iteration_one_stream',{}).get('ms',0)))
except Exception as e: print('  parse failed', e)
PY
}
b r04 GCCNMF_HIP_LIB=$OLD
b new GCCNMF_TUNE=
b new_rcp GCCNMF_TUNE=7=0
b r04_2 GCCNMF_HIP_LIB=$OLD
b new_2 GCCNMF_TUNE=
for f in 64; do
  GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace.py --files $f --stage 3 --repeat 3 > $OUT/ktrace_new_f${f}.txt 2>&1; sed -n 2,5p $OUT/ktrace_new_f${f}.txt
  GCCNMF_HIP_LIB=$EXP GCCNMF_TUNE=7=0 timeout 300 python scripts/ktrace.py --files $f --stage 3 --repeat 3 > $OUT/ktrace_new_rcp_f${f}.txt 2>&1; sed -n 2,5p $OUT/ktrace_new_rcp_f${f}.txt
  GCCNMF_HIP_LIB=$OLD timeout 300 python scripts/ktrace.py --files $f --stage 3 --repeat 3 > $OUT/ktrace_r04_f${f}.txt 2>&1; sed -n 2,5p $OUT/ktrace_r04_f${f}.txt
done
FILES="32 51 80" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
GCCNMF_HIP_LIB=$OLD FILES="32 51 80" bash scripts/files_sweep.sh > $OUT/files_sweep_r04.txt 2>&1; cat $OUT/files_sweep_r04.txt
echo "== full GPU suite (product build)"
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
echo "== kernel tests on the experiment build"
GCCNMF_HIP_LIB=$EXP timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu_exp.log 2>&1
echo "pytest exp exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu_exp.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu_exp.log | head -20
