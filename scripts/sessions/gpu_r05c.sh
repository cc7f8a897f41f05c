#!/bin/bash
# Round 5, third session: the final launcher policy (round-4 cost model of full / half-height tiles + ragged narrow items + half-height tiles
# for outputs of <= 256 rows) against the round-4 library and a build without the narrow loop; IEEE division as the default; K = 256.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_r05c.sh [tag]'
TAG=${1:-r05c}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
OLD=$PWD/gcc_nmf_amd/libgccnmf_hip_r04.so
NON=$PWD/gcc_nmf_amd/libgccnmf_hip_vnonarrow.so
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
echo "== form tests (product, then the experiment build)"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu -x -k "narrow_items" --tb=short -p no:cacheprovider > $OUT/pytest_forms.log 2>&1; echo "forms exit $?"; tail -2 $OUT/pytest_forms.log
GCCNMF_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "narrow_items" --tb=short -p no:cacheprovider > $OUT/pytest_forms_exp.log 2>&1; echo "forms exp exit $?"; tail -2 $OUT/pytest_forms_exp.log
kb() { local name=$1; shift
  env "$@" timeout 300 python scripts/kbench.py --reps 8 ${KARGS} > $OUT/kbench_$name.txt 2> $OUT/kbench_$name.err
  echo "kbench $name exit $?"; grep -E "^K[1-4]a? fused|512 blocks" $OUT/kbench_$name.txt | cut -c1-110
}
kb r04 GCCNMF_HIP_LIB=$OLD
kb new GCCNMF_TUNE=
kb nonarrow GCCNMF_HIP_LIB=$NON
kb new_rcp GCCNMF_TUNE=7=0
kb r04_again GCCNMF_HIP_LIB=$OLD
KARGS="--K 256" kb K256_r04 GCCNMF_HIP_LIB=$OLD
KARGS="--K 256" kb K256_new GCCNMF_TUNE=
KARGS="--K 512" kb K512_r04 GCCNMF_HIP_LIB=$OLD
KARGS="--K 512" kb K512_new GCCNMF_TUNE=
b() { local name=$1; shift; env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --skip-extras ${EXTRA} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name exit $?"; python - $OUT/bench_$name.json <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=b.get('roofline',{}); print('  value %.0f  ms %.2f  k3 %.4f ms frac %.3f  iter %.3f ms' % (b['value'], b['ms_per_step'], r.get('avg_launch_ms',0), r.get('frac',0), b.get('nmf_iteration_one_stream',{}).get('ms',0)))
except Exception as e: print('  parse failed', e)
PY
}
b r04 GCCNMF_HIP_LIB=$OLD
b new GCCNMF_TUNE=
b nonarrow GCCNMF_HIP_LIB=$NON
b new_rcp GCCNMF_TUNE=7=0
b r04_2 GCCNMF_HIP_LIB=$OLD
b new_2 GCCNMF_TUNE=
EXTRA="--dictionary-size 256" b K256_r04 GCCNMF_HIP_LIB=$OLD
EXTRA="--dictionary-size 256" b K256_new GCCNMF_TUNE=
SIZES="16 24 25 26 32 40 48 51 52 64 72 76 77 80 88 96 102 104"
echo "== files sweep (new)"
FILES="$SIZES" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
echo "== full GPU suite"
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
