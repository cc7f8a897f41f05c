#!/bin/bash
# Round 5, second session: narrow items on the classic grid (the hardware dispatcher hands the ordered items out) against the round-4 library,
# IEEE division cost, the default bench line with its live sub-records.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_r05b.sh [tag]'
TAG=${1:-r05b}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
OLD=$PWD/gcc_nmf_amd/libgccnmf_hip_r04.so
echo "== new tests"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_realtime.py tests/test_gpu_pipeline.py -q -m gpu -x -k "narrow_items or tracked_index or largest_windows" --tb=short -p no:cacheprovider > $OUT/pytest_new.log 2>&1
echo "new tests exit $?"; tail -4 $OUT/pytest_new.log
kb() { local name=$1; shift
  env "$@" timeout 300 python scripts/kbench.py --reps 8 > $OUT/kbench_$name.txt 2> $OUT/kbench_$name.err
  echo "kbench $name exit $?"; grep -E "^K[1-4] fused|512 blocks" $OUT/kbench_$name.txt | cut -c1-110
}
kb r04 GCCNMF_HIP_LIB=$OLD
kb new GCCNMF_TUNE=
kb new_wide GCCNMF_TUNE=9=0
kb new_exact_div GCCNMF_TUNE=7=1
kb r04_exact_div GCCNMF_HIP_LIB=$OLD GCCNMF_TUNE=7=1
b() { local name=$1; shift; env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --skip-extras ${EXTRA} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name exit $?"; python - $OUT/bench_$name.json <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=b.get('roofline',{}); print('  value %.0f  ms %.2f  k3 %.4f ms frac %.3f  iter %.3f ms' % (b['value'], b['ms_per_step'], r.get('avg_launch_ms',0), r.get('frac',0), b.get('nmf_iteration_one_stream',{}).get('ms',0)))
except Exception as e: print('  parse failed', e)
PY
}
b r04 GCCNMF_HIP_LIB=$OLD
b new GCCNMF_TUNE=
b new_wide GCCNMF_TUNE=9=0
b new_exact_div GCCNMF_TUNE=7=1
EXTRA="--nmf-groups 1" b r04_g1 GCCNMF_HIP_LIB=$OLD
EXTRA="--nmf-groups 1" b new_g1 GCCNMF_TUNE=
SIZES="16 24 25 26 32 40 48 51 52 64 72 76 77 80 88 96 102 104"
echo "== files sweep (new)"
FILES="$SIZES" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
echo "== files sweep (new, wide tiles only)"
GCCNMF_TUNE=9=0 FILES="$SIZES" bash scripts/files_sweep.sh > $OUT/files_sweep_wide.txt 2>&1; cat $OUT/files_sweep_wide.txt
echo "== files sweep (r04 library, same box)"
GCCNMF_HIP_LIB=$OLD FILES="$SIZES" bash scripts/files_sweep.sh > $OUT/files_sweep_r04.txt 2>&1; cat $OUT/files_sweep_r04.txt
echo "== default bench line with the live sub-records"
timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench full exit $?"; tail -3 $OUT/bench_full.err
python - <<'PY'
import json, os
b = json.loads(open(os.path.join(os.environ['OUT'], 'bench_full.json')).read().strip().splitlines()[-1])
print('value', b['value'], 'frac', b['roofline']['frac'], 'config_lines_seconds', b.get('config_lines_seconds'))
for k in ('k128_batch', 'k_sweep', 'it200', 'shared_dictionary_n1', 'streaming', 'big_matrix_n80000'):
    print(k, json.dumps(b.get(k))[:600])
PY
