#!/bin/bash
# Round 6, session g: whole-call chain with K3's ragged tiles in their natural place (key 23): parity, times, timeline
TAG=${1:-r06g}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "chained" > $OUT/pytest_chain.log 2>&1; echo "chain tests exit $?"; tail -5 $OUT/pytest_chain.log
GCCNMF_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "chained" > $OUT/pytest_chain_exp.log 2>&1; echo "chain tests (experiment build) exit $?"; tail -5 $OUT/pytest_chain_exp.log
run() { local name=$1; shift; timeout 300 python bench.py --steps 5 --warmup 2 --skip-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name exit $?"; }
run g1_chain0 --nmf-groups 1
run g2_chain0
run g1_chain8_rag0 --nmf-groups 1 --tune 21=8 --tune 23=0
run g1_chain8_rag1 --nmf-groups 1 --tune 21=8 --tune 23=1
run g1_chain9_rag1 --nmf-groups 1 --tune 21=9 --tune 23=1
run g2_chain8_rag1 --tune 21=8 --tune 23=1
run g1_chain0_b --nmf-groups 1
run g1_chain8_rag1_b --nmf-groups 1 --tune 21=8 --tune 23=1
python - <<'PY'
import json, os, glob
for f in sorted(glob.glob(os.path.join(os.environ['OUT'], 'bench_*.json'))):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-22s value %.0f  ms/step %.2f  iter %.4f ms (%.3f of peak)  K3 %.4f ms  tdoa %s' % (os.path.basename(f)[6:-5], b['value'], b['ms_per_step'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['roofline']['avg_launch_ms'], b['tdoa_indexes_as_expected']))
    except Exception as e:
        print(f, 'failed', e)
PY
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace_chain.py --files 64 --chain 8 > $OUT/ktrace_chain8_64.txt 2>&1; echo "ktrace_chain 8 exit $?"; grep -A6 "one chained" $OUT/ktrace_chain8_64.txt | cut -c1-330 | grep -v "t\[us\]"
