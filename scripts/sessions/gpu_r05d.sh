#!/bin/bash
# Round 5, fourth session: lean epilogue with the compute -> store order pinned, against the round-4 library on ONE box.
TAG=${1:-r05d}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
OLD=$PWD/gcc_nmf_amd/libgccnmf_hip_r04.so
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu -x -k "narrow_items or klnmf" --tb=short -p no:cacheprovider > $OUT/pytest_forms.log 2>&1; echo "klnmf tests exit $?"; tail -2 $OUT/pytest_forms.log
kb() { local name=$1; shift
  env "$@" timeout 300 python scripts/kbench.py --reps 8 ${KARGS} > $OUT/kbench_$name.txt 2> $OUT/kbench_$name.err
  echo "kbench $name exit $?"; grep -E "^K[1-4]a? fused|512 blocks" $OUT/kbench_$name.txt | cut -c1-110
}
kb r04 GCCNMF_HIP_LIB=$OLD
kb new GCCNMF_TUNE=
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
b r04_2 GCCNMF_HIP_LIB=$OLD
b new_2 GCCNMF_TUNE=
b new_wide GCCNMF_TUNE=9=0
echo "== files sweep (new / r04, the sizes that differed)"
FILES="16 26 32 51 80" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
GCCNMF_HIP_LIB=$OLD FILES="16 26 32 51 80" bash scripts/files_sweep.sh > $OUT/files_sweep_r04.txt 2>&1; cat $OUT/files_sweep_r04.txt
