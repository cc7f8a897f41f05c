#!/bin/bash
# Round 6, session b: per-function host-to-host times of the drop-in sequence (this revision, copying / resident, against the round-5 host
# layer under lab_old/), the unmodified driver's wall-time split, the unmodified-driver test.
TAG=${1:-r06b}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/dropin_times.py --old lab_old > $OUT/dropin_times.json 2> $OUT/dropin_times.err; echo "dropin_times exit $?"; tail -5 $OUT/dropin_times.err
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ['OUT'], 'dropin_times.json')))
for shape, modes in d.items():
    for mode, r in modes.items():
        print(shape, mode, 'sum %.2f ms  whole %.2f ms  numpy-between %.2f' % (r['sum_of_the_eight_ms'], r['whole_sequence_ms'], r['driver_numpy_ms']), ' '.join('%.2f' % v for v in r['ms'].values()))
PY
if [ -d oracle/_ref/reference_checkout/gccNMF ]; then
  timeout 300 python scripts/driver_split.py oracle/_ref/reference_checkout > $OUT/driver_split.json 2> $OUT/driver_split.err; echo "driver_split exit $?"; tail -3 $OUT/driver_split.err
  timeout 300 python scripts/driver_split.py oracle/_ref/reference_checkout --resident > $OUT/driver_split_resident.json 2> $OUT/driver_split_resident.err; echo "driver_split resident exit $?"
  python - <<'PY'
import json, os
for f in ('driver_split.json', 'driver_split_resident.json'):
    d = json.load(open(os.path.join(os.environ['OUT'], f)))
    print(f, d['before_the_driver_ms'])
    for r in d['runs']:
        print('  ', r['run'], 'wall %.1f  eight %.1f  wavread %.1f  wavwrite %.1f  rest %.1f' % (r['wall_ms'], r['the_eight_named_functions_ms'], r['wav_read_ms'], r['wav_writes_ms'], r['driver_numpy_imports_and_rest_ms']))
PY
  timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s --tb=short -p no:cacheprovider -k unmodified_reference_driver > $OUT/dropin_driver.log 2>&1; echo "driver test exit $?"; grep -E "unmodified|passed|failed|skipped" $OUT/dropin_driver.log | head -5
fi
