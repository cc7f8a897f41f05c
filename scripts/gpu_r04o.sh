#!/bin/bash
export TMPDIR=/tmp
for t in 1 3; do
echo "== key 9 = $t"
for f in 24 26 28 32 40 48 52 56 64 72 77 80 90 104; do
python bench.py --gpus 1 --steps 2 --warmup 1 --files $f --skip-extras --nmf-groups 1 --tune 9=$t 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']; print('files %3d: K3 %.4f ms  roofline %.3f   one-stream iteration %.3f ms (%.3f of peak)  e2e %.0f frames/s  tdoa %s' % ($f, r['avg_launch_ms'], r['frac'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['value'], b['tdoa_indexes_as_expected']))"
done
done
