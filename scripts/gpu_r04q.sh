#!/bin/bash
# Round 4, fused short-dictionary launches: parity test, then the K = 128 bench with tuning key 16 off / on.
TAG=${1:-r04q}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --tb=short -p no:cacheprovider -k "short_dictionary or batch_is_file_independent" > $OUT/pytest_fused.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest_fused.log
for t in 16=0 16=1; do
  GCCNMF_TUNE=$t timeout 300 python bench.py --dictionary-size 128 --skip-extras --nmf-groups 1 --steps 3 > $OUT/bench_K128_$t.json 2> $OUT/bench_K128_$t.err; echo "bench $t exit $?"
  python - "$OUT/bench_K128_$t.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d.get('nmf_iteration_one_stream')), json.dumps(d['roofline'])[:300])
PY
done
for t in 16=0 16=1; do
  GCCNMF_TUNE=$t timeout 300 python scripts/kbench.py --K 128 --reps 10 > $OUT/stages_K128_$t.txt 2>&1; echo "stages $t exit $?"; grep -E 'fused' $OUT/stages_K128_$t.txt
done
