#!/bin/bash
# Round 6, session r: the whole GPU suite once more (short chain by rule only where the plain call runs the same item programs), smoke with the chained stage
TAG=${1:-r06r}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 1700 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED" $OUT/pytest_gpu.log | head
GCCNMF_HIP_LIB=$EXP timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_gpu_exp.log 2>&1; echo "kernel tests on the lab build: exit $? $(grep -E 'passed|failed' $OUT/pytest_gpu_exp.log | tail -1)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 --skip-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -2 $OUT/bench.err
python - <<'PY'
import json, os
b = json.loads(open(os.path.join(os.environ['OUT'], 'bench.json')).read().strip().splitlines()[-1])
print('value', b['value'], 'iter', b['nmf_iteration_one_stream']['frac_of_peak'], 'roofline', b['roofline']['frac'])
print('k128', {k: (round(v['frames_per_s']), round(v['ms_per_step'], 2), round(v['iteration_frac'], 3), v['klnmf_plan']) for k, v in b['k128_batch'].items() if isinstance(v, dict)})
print('ksweep', {k: (round(v['frames_per_s']), round(v['iteration_frac'], 3), v['klnmf_plan']) for k, v in b['k_sweep'].items() if isinstance(v, dict)})
print('mixed', b['mixed_lengths']['vs_equal_length_rate'])
PY
