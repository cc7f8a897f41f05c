#!/bin/bash
# Live PMC traffic leg of bench.py: does it run beside the idle parent, how long does it take, what does it read?
OUT=gpurun_out/r05m; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --skip-config-lines --skip-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/time.txt
echo "exit $?"; tail -3 $OUT/time.txt; tail -5 $OUT/bench.err
python - <<'PY'
import json
b = json.loads(open('gpurun_out/r05m/bench.json').read().strip().splitlines()[-1])
r = b['roofline']
print(r['frac'], r['traffic'], r.get('traffic_recorded'), r.get('traffic_live'))
print(r['traffic_source'][:300])
PY
