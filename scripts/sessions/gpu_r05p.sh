#!/bin/bash
OUT=gpurun_out/r05p; mkdir -p $OUT; export TMPDIR=/tmp
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/time.txt
echo "exit $?"; tail -3 $OUT/time.txt; tail -3 $OUT/bench.err
python - <<'PY'
import json
b = json.loads(open('gpurun_out/r05p/bench.json').read().strip().splitlines()[-1])
print('value', b['value'], 'frac', b['roofline']['frac'], 'traffic', b['roofline']['traffic'], b['roofline'].get('traffic_live', {}).get('seconds'))
print('iteration', b['nmf_iteration_one_stream'])
print('k128', {h: (v['frames_per_s'], v['iteration_ms_one_stream'], v['iteration_frac']) for h, v in b['k128_batch'].items() if isinstance(v, dict)})
print('ksweep', {k: (v['frames_per_s'], v['iteration_ms_one_stream'], v['iteration_frac']) for k, v in b['k_sweep'].items()})
print('rms', b['gpu_vs_cpu_waveform_rms'], b['gpu_vs_cpu_tdoa_equal'], b['tdoa_indexes_as_expected'])
PY
