#!/bin/bash
# Round 3, session A: parity tests + the collective-mode benches (C-side iteration loop) + mask-flip experiment.
TAG=${1:-r03a}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s --durations=15 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -5
echo "== time-sharded 160 s x5 processes"
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --mode time-sharded --steps 5 --warmup 2 > $OUT/ts160_$i.json 2> $OUT/ts160_$i.err; echo "ts160 $i exit $?"; python -c "import json;d=json.load(open('$OUT/ts160_$i.json'));print(d['value'],d['ms_per_step'],d['column_blocks'])"
done
timeout 300 python bench.py --mode time-sharded --seconds 640 --steps 3 --warmup 1 > $OUT/ts640.json 2> $OUT/ts640.err; echo "ts640 exit $?"; python -c "import json;d=json.load(open('$OUT/ts640.json'));print(d['value'],d['ms_per_step'],d['column_blocks'])"
timeout 300 python bench.py --mode time-sharded --steps 5 --warmup 2 --tune 2=2 > $OUT/ts160_small.json 2> $OUT/ts160_small.err; echo "ts160 small tile exit $?"; python -c "import json;d=json.load(open('$OUT/ts160_small.json'));print(d['value'],d['ms_per_step'])"
echo "== shared-dictionary"
timeout 300 python bench.py --mode shared-dictionary --steps 3 --warmup 1 > $OUT/shared.json 2> $OUT/shared.err; echo "shared exit $?"; cut -c1-300 $OUT/shared.json
echo "== default bench"
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== mask flips"
timeout 600 python scripts/mask_flips.py > $OUT/mask_flips.json 2> $OUT/mask_flips.err; echo "mask_flips exit $?"; tail -12 $OUT/mask_flips.json
