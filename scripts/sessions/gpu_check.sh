#!/bin/bash
# quick GPU check: the whole -m gpu suite, smoke, and (optionally) the single-file sweep / the headline bench
OUT=gpurun_out/${1:-check}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 800 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -v amdgpu.ids $OUT/pytest_gpu.log | tail -5
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
TUNE=$TUNE timeout 200 python scripts/single_file.py --profile 2>&1 | grep default
if [ -n "$BENCH" ]; then timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 $BENCH > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python -c "
import json; b=json.load(open('$OUT/bench.json')); print('value %.0f  ms/step %.2f  roofline %.3f (%.4f ms)  single file %.2f ms  dropin %.2f ms' % (b['value'], b['ms_per_step'], b.get('roofline',{}).get('frac',0), b.get('roofline',{}).get('avg_launch_ms',0), b['single_file']['ms_per_file'], b['dropin_performKLNMF']['ms']))"; fi
