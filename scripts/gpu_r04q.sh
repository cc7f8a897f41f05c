#!/bin/bash
# Round 4, fused short-dictionary launches: parity test, K = 128 stage times with tuning key 16 off / on, workgroup timeline.
TAG=${1:-r04q}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --tb=short -p no:cacheprovider -k "short_dictionary" > $OUT/pytest_fused.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_fused.log
for t in 16=0 17=1 16=1,17=1; do
  GCCNMF_TUNE=$t timeout 300 python scripts/kbench.py --K 128 --reps 10 > $OUT/stages_K128_$t.txt 2>&1; echo "stages $t exit $?"; grep -E '^K[0-9].* fused' $OUT/stages_K128_$t.txt
done
GCCNMF_TUNE=16=1,17=1 timeout 300 python scripts/ktrace_fused.py > $OUT/ktrace_fused.txt 2>&1; cat $OUT/ktrace_fused.txt
GCCNMF_TUNE=16=1,17=1 timeout 300 python scripts/ktrace_fused.py --stage 3 > $OUT/ktrace_fused34.txt 2>&1; cat $OUT/ktrace_fused34.txt
if [ -n "$BENCH" ]; then
for t in 16=0 16=1; do
  GCCNMF_TUNE=$t timeout 300 python bench.py --dictionary-size 128 --skip-extras --nmf-groups 1 --steps 3 > $OUT/bench_K128_$t.json 2> $OUT/bench_K128_$t.err; echo "bench $t exit $?"
  python - "$OUT/bench_K128_$t.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d.get('nmf_iteration_one_stream')))
PY
done
fi
