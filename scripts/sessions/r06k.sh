#!/bin/bash
# Round 6, session k: ragged batches -- parity, the whole GPU suite, the bench line with the mixed_lengths sub-record
TAG=${1:-r06k}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "ragged" > $OUT/pytest_ragged.log 2>&1; echo "ragged tests exit $?"; tail -15 $OUT/pytest_ragged.log
timeout 1700 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --timeout 300 --deselect tests/test_gpu_pipeline.py::test_ragged_batch_is_bitwise_the_equal_length_batches > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.log
GCCNMF_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 200 -k "chained" > $OUT/pytest_chain_exp.log 2>&1; echo "chain tests (experiment build) exit $?"; tail -3 $OUT/pytest_chain_exp.log
timeout 900 python bench.py --steps 10 --warmup 3 --skip-config-lines --skip-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -3 $OUT/bench.err
python - <<'PY'
import json, os
b = json.loads(open(os.path.join(os.environ['OUT'], 'bench.json')).read().strip().splitlines()[-1])
print('value', b['value'], 'ms/step', b['ms_per_step'], 'iter', b['nmf_iteration_one_stream'], 'roofline frac', b['roofline']['frac'])
print('mixed_lengths', {k: v for k, v in b.get('mixed_lengths', {}).items() if k != 'what'})
PY
