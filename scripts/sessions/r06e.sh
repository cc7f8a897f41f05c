#!/bin/bash
# Round 6, session e: the whole KL-NMF call as ONE chained launch (key 21 = 8): parity, iteration and end-to-end times against 0 / 4, timelines
TAG=${1:-r06e}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "chained" > $OUT/pytest_chain.log 2>&1; echo "chain tests exit $?"; tail -15 $OUT/pytest_chain.log
GCCNMF_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "chained" > $OUT/pytest_chain_exp.log 2>&1; echo "chain tests (experiment build) exit $?"; tail -5 $OUT/pytest_chain_exp.log
for t in 0 4 8; do
  timeout 300 python bench.py --steps 5 --warmup 2 --skip-extras --nmf-groups 1 --tune 21=$t > $OUT/bench_g1_chain$t.json 2> $OUT/bench_g1_chain$t.err; echo "g1 chain=$t exit $?"
  timeout 300 python bench.py --steps 5 --warmup 2 --skip-extras --tune 21=$t > $OUT/bench_g2_chain$t.json 2> $OUT/bench_g2_chain$t.err; echo "g2 chain=$t exit $?"
done
python - <<'PY'
import json, os
for g in ('g1', 'g2'):
    for t in (0, 4, 8):
        try:
            b = json.loads(open(os.path.join(os.environ['OUT'], 'bench_%s_chain%d.json' % (g, t))).read().strip().splitlines()[-1])
            print(g, 'chain', t, 'value %.0f  ms/step %.2f  iter %.4f ms (%.3f of peak)  K3 %.4f ms  tdoa %s' % (b['value'], b['ms_per_step'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['roofline']['avg_launch_ms'], b['tdoa_indexes_as_expected']))
        except Exception as e:
            print(g, t, 'failed', e)
PY
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace_chain.py --files 64 --chain 8 > $OUT/ktrace_chain8_64.txt 2>&1; echo "ktrace_chain 8 exit $?"; cut -c1-330 $OUT/ktrace_chain8_64.txt | grep -v "t\[us\]"
