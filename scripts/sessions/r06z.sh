#!/bin/bash
# Round 6, session z: HEAD with the list rule (whole files | spread): the whole GPU suite, lab-build kernel tests, smoke, the driver's bench command, the sweep
TAG=${1:-r06z}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 1700 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED" $OUT/pytest_gpu.log | head
GCCNMF_HIP_LIB=$EXP timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_gpu_exp.log 2>&1; echo "kernel tests on the lab build: exit $? $(grep -E 'passed|failed' $OUT/pytest_gpu_exp.log | tail -1)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
S=$(date +%s); timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $? wall $(( $(date +%s) - S )) s"
python - <<'PY'
import json, os
b = json.loads(open(os.path.join(os.environ['OUT'], 'bench.json')).read().strip().splitlines()[-1])
print('value', b['value'], 'iter', b['nmf_iteration_one_stream']['frac_of_peak'], 'roofline', b['roofline']['frac'], b['roofline']['avg_launch_ms'], 'traffic', b['roofline']['traffic'])
print('mixed', b['mixed_lengths']['vs_equal_length_rate'], 'h2h', b['sec8d_host_to_host']['value'], 'cpu', b['cpu_baseline']['value'])
PY
FILES="16 20 24 25 26 32 40 48 51 52 64 72 76 77 80 88 96 102 104" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cut -c1-175 $OUT/files_sweep.txt
