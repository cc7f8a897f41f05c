#!/bin/bash
TAG=${1:-r04c}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<'PY'
import json, os
d = json.loads(open(os.path.join(os.environ['OUT'], 'bench.json')).read().strip().splitlines()[-1])
for k in ('value', 'ms_per_step', 'roofline', 'single_file', 'dropin_performKLNMF'):
    print(k, d.get(k))
print('cpu_baseline', {k: d['cpu_baseline'][k] for k in ('value', 'cores', 'kind')})
PY
