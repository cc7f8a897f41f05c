#!/bin/bash
# Round 6, session y: chained launches on spread lists with agent-scope hand-over (key 23 = 2): parity, 64 files A/B, the small / unbalanced batch sizes
TAG=${1:-r06y}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "chained_iteration_is_bitwise" > $OUT/pytest_chain.log 2>&1; echo "chain tests exit $?"; tail -3 $OUT/pytest_chain.log | cut -c1-200
run() { local name=$1; shift; timeout 300 python bench.py --steps 5 --warmup 2 --skip-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name exit $?"; }
run f64_whole_a
run f64_spread_a --tune 23=2
run f64_whole_b
run f64_spread_b --tune 23=2
python - <<'PY'
import json, os, glob
for f in sorted(glob.glob(os.path.join(os.environ['OUT'], 'bench_f64*.json'))):
    b = json.loads(open(f).read().strip().splitlines()[-1])
    print('%-14s value %.0f  iter %.4f ms (%.3f of peak)  call %.2f ms' % (os.path.basename(f)[6:-5], b['value'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['roofline']['avg_launch_ms']))
PY
FILES="16 20 24 25 26 32 40 51 52 77" bash scripts/files_sweep.sh > $OUT/files_sweep_default.txt 2>&1
FILES="16 20 24 25 26 32 40 51 52 77" TUNE="23=2" bash scripts/files_sweep.sh > $OUT/files_sweep_spread.txt 2>&1
paste -d'\n' $OUT/files_sweep_default.txt $OUT/files_sweep_spread.txt | cut -c1-190
