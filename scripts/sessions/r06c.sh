#!/bin/bash
# Round 6, session c: chained launches (tuning key 21) -- parity, then the iteration and end-to-end times: unchained | K1+K2 chained | whole iteration chained,
# one stream and the engine's two file groups, same box.
TAG=${1:-r06c}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "chained" > $OUT/pytest_chain.log 2>&1; echo "chain tests exit $?"; tail -15 $OUT/pytest_chain.log
GCCNMF_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "chained" > $OUT/pytest_chain_exp.log 2>&1; echo "chain tests (experiment build) exit $?"; tail -5 $OUT/pytest_chain_exp.log
for t in 0 2 4; do
  timeout 300 python bench.py --steps 5 --warmup 2 --skip-extras --nmf-groups 1 --tune 21=$t > $OUT/bench_g1_chain$t.json 2> $OUT/bench_g1_chain$t.err; echo "g1 chain=$t exit $?"
  timeout 300 python bench.py --steps 5 --warmup 2 --skip-extras --tune 21=$t > $OUT/bench_g2_chain$t.json 2> $OUT/bench_g2_chain$t.err; echo "g2 chain=$t exit $?"
done
python - <<'PY'
import json, os
for g in ('g1', 'g2'):
    for t in (0, 2, 4):
        try:
            b = json.loads(open(os.path.join(os.environ['OUT'], 'bench_%s_chain%d.json' % (g, t))).read().strip().splitlines()[-1])
            print(g, 'chain', t, 'value %.0f  ms/step %.2f  iter %.4f ms (%.3f of peak)  K3 %.4f ms  tdoa %s' % (b['value'], b['ms_per_step'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['roofline']['avg_launch_ms'], b['tdoa_indexes_as_expected']))
        except Exception as e:
            print(g, t, 'failed', e)
PY
