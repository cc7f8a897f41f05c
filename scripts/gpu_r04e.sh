#!/bin/bash
TAG=${1:-r04e}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
for t in "14=0" "14=1" "14=1 --tune 2=2" "14=0 --tune 2=2"; do
  for hop in 256 128; do
  echo "== K=128 hop $hop tune $t"
  timeout 300 python bench.py --dictionary-size 128 --hop $hop --steps 3 --warmup 1 --skip-extras --nmf-groups 1 --tune $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.0f ms_per_step %.2f iter_ms %.4f k3_ms %.4f' % (d['value'], d['ms_per_step'], d['nmf_iteration_one_stream']['ms'], d['roofline']['avg_launch_ms']))"
  done
done
