#!/bin/bash
# Round 5, robustness of the final code: the GPU parity suite under forced tunings (GCCNMF_TUNE applies gccnmf_set_tuning pairs at library load),
# the whole suite on the experiment library, the unmodified reference driver (checkout staged by scripts/stage_reference.sh), scale.sh on one GPU,
# and the driver's bench command once more.
TAG=${1:-r05k}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
: > $OUT/tuning_matrix.txt
for t in "9=0" "9=2" "9=3" "2=1" "2=2,8=1" "3=0" "7=0" "16=0,17=0"; do
  GCCNMF_TUNE=$t timeout 900 python -m pytest tests -q -m gpu --tb=line -p no:cacheprovider -x > $OUT/pytest_tune_$t.log 2>&1
  echo "GCCNMF_TUNE=$t: exit $? $(grep -E 'passed|failed' $OUT/pytest_tune_$t.log | tail -1)" | tee -a $OUT/tuning_matrix.txt
done
GCCNMF_HIP_LIB=$EXP timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu_exp_full.log 2>&1; echo "experiment library, whole suite: exit $? $(grep -E 'passed|failed' $OUT/pytest_gpu_exp_full.log | tail -1)" | tee -a $OUT/tuning_matrix.txt
GCCNMF_HIP_LIB=$EXP GCCNMF_TUNE=18=1 timeout 1200 python -m pytest tests -q -m gpu --tb=line -p no:cacheprovider > $OUT/pytest_gpu_exp_resident.log 2>&1; echo "experiment library, resident-workgroup grid (18=1): exit $? $(grep -E 'passed|failed' $OUT/pytest_gpu_exp_resident.log | tail -1)" | tee -a $OUT/tuning_matrix.txt
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s --tb=short -p no:cacheprovider -k unmodified_reference_driver > $OUT/dropin_driver.log 2>&1; echo "driver test exit $?"; grep -E "unmodified|passed|failed|skipped|LSB|wall" $OUT/dropin_driver.log | head
timeout 900 bash scripts/scale.sh $OUT/scale > $OUT/scale.log 2>&1; echo "scale.sh exit $?"; cat $OUT/scale/scale.jsonl | cut -c1-250
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<'PY'
import json, os
b = json.loads(open(os.path.join(os.environ['OUT'], 'bench.json')).read().strip().splitlines()[-1])
print('value', b['value'], 'frac', b['roofline']['frac'], 'rms', b.get('gpu_vs_cpu_waveform_rms'), b.get('gpu_vs_cpu_tdoa_equal'), 'cfg s', b.get('config_lines_seconds'))
PY
