#!/bin/bash
# fused K1 + K2 column-tile launch: tests, K = 128 bench A/B, stage times over file counts (cost-model check)
TAG=${1:-r04w}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --tb=short -p no:cacheprovider -k "klnmf" > $OUT/pytest_klnmf.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_klnmf.log
for t in 16=0 16=1; do
  GCCNMF_TUNE=$t timeout 300 python bench.py --dictionary-size 128 --skip-extras --nmf-groups 1 --steps 3 > $OUT/bench_K128_$t.json 2> $OUT/bench_K128_$t.err; echo "bench $t exit $?"
  python - "$OUT/bench_K128_$t.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d.get('nmf_iteration_one_stream')))
PY
done
for f in 20 25 26 40 50 64 96 128; do
  for t in 16=0 16=2; do
    GCCNMF_TUNE=$t timeout 300 python scripts/kbench.py --K 128 --files $f --reps 6 2>/dev/null | grep -E '^K1 fused|^K2 fused' | awk -v f=$f -v t=$t '{print "files", f, t, $1, $2, $3, $5, $6}' | tr '\n' ' '; echo
  done
done
for t in 16=0 16=2; do
GCCNMF_TUNE=$t timeout 300 python scripts/kbench.py --K 128 --T 1243 --files 64 --reps 6 2>/dev/null | grep -E '^K1 fused|^K2 fused' | awk -v t=$t '{print "hop128 files 64", t, $1, $2, $3, $5, $6}' | tr '\n' ' '; echo
GCCNMF_TUNE=$t timeout 300 python scripts/kbench.py --K 64 --files 64 --reps 6 2>/dev/null | grep -E '^K1 fused|^K2 fused' | awk -v t=$t '{print "K=64 files 64", t, $1, $2, $3, $5, $6}' | tr '\n' ' '; echo
done
