#!/bin/bash
# Round 6: big batches (the 288 GB of HBM are there to be used): 128 / 256 / 512 / 1024 files of 10 s per step, and 64 files of 60 s, through the default rule.
# usage: gpurun --timeout 1500 -- 'bash scripts/sessions/r06ag.sh [tag]'
TAG=${1:-r06ag}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for files in 128 256 512 1024; do
  timeout 600 python bench.py --files $files --steps 2 --warmup 1 --skip-extras --skip-cpu-baseline --no-live-traffic > $OUT/bench_files$files.json 2> $OUT/bench_files$files.err
  echo "files $files exit $?"; python - $OUT/bench_files$files.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('   value %.0f frames/s  %.1f ms per step  roofline frac %.3f (%s ms per launch)  iteration %.3f  tdoa ok %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], round(d['roofline']['avg_launch_ms'], 2), d.get('nmf_iteration_one_stream', {}).get('frac_of_peak', 0), d.get('tdoa_indexes_as_expected')))
except Exception as e:
    print('   no line:', e)
PY
  tail -n 2 $OUT/bench_files$files.err | cut -c1-200
done 2>&1 | tee $OUT/big_batches.txt
timeout 900 python bench.py --files 64 --seconds 60 --steps 2 --warmup 1 --skip-extras --skip-cpu-baseline --no-live-traffic > $OUT/bench_60s.json 2> $OUT/bench_60s.err; echo "60 s files exit $?"; cut -c1-260 $OUT/bench_60s.json | tee -a $OUT/big_batches.txt; tail -n 2 $OUT/bench_60s.err | cut -c1-200
rocm-smi --showmemuse 2>/dev/null | head -8
