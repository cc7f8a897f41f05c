#!/bin/bash
# Round 6, session j: chain for K = 256 (half-height K2 tiles) and K = 512, parity; the bench line with the roofline on the chained kernel + live traffic
TAG=${1:-r06j}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 200 -k "chained" > $OUT/pytest_chain.log 2>&1; echo "chain tests exit $?"; tail -5 $OUT/pytest_chain.log
GCCNMF_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 200 -k "chained" > $OUT/pytest_chain_exp.log 2>&1; echo "chain tests (experiment build) exit $?"; tail -3 $OUT/pytest_chain_exp.log
run() { local name=$1; shift; timeout 300 python bench.py --steps 3 --warmup 1 --skip-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name exit $?"; }
for K in 256 384 512; do
  run K${K}_chain0 --dictionary-size $K --tune 21=0
  run K${K}_chain1 --dictionary-size $K
done
python - <<'PY'
import json, os, glob
for f in sorted(glob.glob(os.path.join(os.environ['OUT'], 'bench_K*.json'))):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-16s value %.0f  ms/step %.2f  iter %.4f ms (%.3f of peak)  groups %s tdoa %s' % (os.path.basename(f)[6:-5], b['value'], b['ms_per_step'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['config']['nmf_file_groups_per_gpu'], b['tdoa_indexes_as_expected']))
    except Exception as e:
        print(f, 'failed', e)
PY
timeout 900 python bench.py --steps 10 --warmup 3 --skip-config-lines > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -3 $OUT/bench.err
python - <<'PY'
import json, os
b = json.loads(open(os.path.join(os.environ['OUT'], 'bench.json')).read().strip().splitlines()[-1])
print('value', b['value'], 'ms/step', b['ms_per_step'], 'iter', b['nmf_iteration_one_stream'])
r = b['roofline']
print({k: v for k, v in r.items() if k not in ('traffic_live', 'launch', 'kernel', 'traffic_source')})
print(r.get('traffic_source', '')[:300])
PY
