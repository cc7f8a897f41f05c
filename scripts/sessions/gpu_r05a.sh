#!/bin/bash
# Round 5, first session: the persistent / narrow-item throughput tile against the round-4 library on ONE box.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_r05a.sh [tag]'
TAG=${1:-r05a}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
OLD=$PWD/gcc_nmf_amd/libgccnmf_hip_r04.so
echo "== new bitwise-form tests first"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "narrow_items or batch_is_file_independent or repeats_are_bitwise" --tb=short -p no:cacheprovider > $OUT/pytest_forms.log 2>&1
echo "forms exit $?"; tail -5 $OUT/pytest_forms.log
echo "== kbench variants"
kb() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python scripts/kbench.py --reps 8 > $OUT/kbench_$name.txt 2> $OUT/kbench_$name.err
  echo "kbench $name exit $?"; grep -E "^K[1-4]|mfma-only" $OUT/kbench_$name.txt | cut -c1-120
}
kb r04 GCCNMF_HIP_LIB=$OLD
kb new_default GCCNMF_TUNE=
kb wide_classic GCCNMF_TUNE=9=0,18=0
kb wide_resident GCCNMF_TUNE=9=0,18=1,19=1
kb wide_resident_nopf GCCNMF_TUNE=9=0,18=1,19=0
kb narrow_classic GCCNMF_TUNE=9=1,18=0
kb all_narrow GCCNMF_TUNE=9=2,18=1
echo "== bench"
b() { local name=$1; shift; env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --skip-extras ${EXTRA} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name exit $?"; python - $OUT/bench_$name.json <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=b.get('roofline',{}); print('  value %.0f  ms %.2f  k3 %.4f ms frac %.3f  iter %.3f ms' % (b['value'], b['ms_per_step'], r.get('avg_launch_ms',0), r.get('frac',0), b.get('nmf_iteration_one_stream',{}).get('ms',0)))
except Exception as e: print('  parse failed', e)
PY
}
b r04 GCCNMF_HIP_LIB=$OLD
b new GCCNMF_TUNE=
EXTRA="--nmf-groups 1" b r04_g1 GCCNMF_HIP_LIB=$OLD
EXTRA="--nmf-groups 1" b new_g1 GCCNMF_TUNE=
b new_classic GCCNMF_TUNE=18=0
b new_wide GCCNMF_TUNE=9=0
b new_nopf GCCNMF_TUNE=19=0
echo "== files sweep (new)"
FILES="16 24 25 26 32 40 48 51 52 64 72 76 77 80 88 96 102 104" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
echo "== files sweep (r04 library, same box)"
GCCNMF_HIP_LIB=$OLD FILES="16 26 32 40 52 64 72 77 96 104" bash scripts/files_sweep.sh > $OUT/files_sweep_r04.txt 2>&1; cat $OUT/files_sweep_r04.txt
echo "== full GPU suite"
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head -20
