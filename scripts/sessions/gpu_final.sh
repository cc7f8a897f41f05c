#!/bin/bash
# What the driver runs at round end, on HEAD: the GPU suite, smoke, the bench command.
TAG=${1:-r05_final}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt; echo "bench exit $?"; cat $OUT/bench_time.txt | tail -3
python - <<'PY'
import json, os
b = json.loads(open(os.path.join(os.environ['OUT'], 'bench.json')).read().strip().splitlines()[-1])
print('value', b['value'], 'ms', b['ms_per_step'], 'frac', b['roofline']['frac'], 'rms', b.get('gpu_vs_cpu_waveform_rms'), b.get('gpu_vs_cpu_tdoa_equal'))
print('k128', b['k128_batch']['hop256']['frames_per_s'], b['k128_batch']['hop256']['iteration_frac'], 'k256', b['k_sweep']['256']['frames_per_s'], b['k_sweep']['256']['iteration_frac'], 'cfg s', b['config_lines_seconds'])
PY
