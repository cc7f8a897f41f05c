#!/bin/bash
# Short reductions (K = 256 / 512): are half-height tiles (three workgroups per CU) better for K1 / K3 than the full tile?
TAG=${1:-r05j}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
for K in 256 512 1024; do
  for t in "" "9=3"; do
    GCCNMF_TUNE=$t timeout 300 python scripts/kbench.py --reps 8 --K $K > $OUT/kbench_K${K}_t${t}.txt 2> $OUT/kbench_K${K}_t${t}.err
    echo "K $K tune '$t' exit $?"; grep -E "^K[1-4]a? fused" $OUT/kbench_K${K}_t${t}.txt | cut -c1-110
  done
done
for K in 256 512; do
  for t in "" "9=3"; do
    GCCNMF_TUNE=$t timeout 300 python bench.py --dictionary-size $K --steps 3 --warmup 1 --skip-extras > $OUT/bench_K${K}_t${t}.json 2>/dev/null
    python - $OUT/bench_K${K}_t${t}.json <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], '%.0f frames/s  iter %.4f ms (%.3f)  k3 %.4f' % (b['value'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['roofline']['avg_launch_ms']))
PY
  done
done
