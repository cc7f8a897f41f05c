#!/bin/bash
# Round 6, session h: whole-file lists (any batch size), XCD-local ready counters (key 24), batch-size sweep chained against plain
TAG=${1:-r06h}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "chained" > $OUT/pytest_chain.log 2>&1; echo "chain tests exit $?"; tail -5 $OUT/pytest_chain.log
GCCNMF_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "chained" > $OUT/pytest_chain_exp.log 2>&1; echo "chain tests (experiment build) exit $?"; tail -5 $OUT/pytest_chain_exp.log
run() { local name=$1; shift; timeout 300 python bench.py --steps 5 --warmup 2 --skip-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name exit $?"; }
run g1_chain0 --nmf-groups 1
run g2_chain0
run g1_chain8 --nmf-groups 1 --tune 21=8
run g1_chain8_local --nmf-groups 1 --tune 21=8 --tune 24=1
run g1_chain9_local --nmf-groups 1 --tune 21=9 --tune 24=1
run g1_chain8_b --nmf-groups 1 --tune 21=8
run g1_chain8_local_b --nmf-groups 1 --tune 21=8 --tune 24=1
python - <<'PY'
import json, os, glob
for f in sorted(glob.glob(os.path.join(os.environ['OUT'], 'bench_*.json'))):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-22s value %.0f  ms/step %.2f  iter %.4f ms (%.3f of peak)  K3 %.4f ms  tdoa %s' % (os.path.basename(f)[6:-5], b['value'], b['ms_per_step'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['roofline']['avg_launch_ms'], b['tdoa_indexes_as_expected']))
    except Exception as e:
        print(f, 'failed', e)
PY
FILES="16 24 25 26 32 40 48 51 52 64 72 76 77 80 88 96 102 104" bash scripts/files_sweep.sh > $OUT/files_sweep_plain.txt 2>&1
FILES="16 24 25 26 32 40 48 51 52 64 72 76 77 80 88 96 102 104" TUNE="21=8" bash scripts/files_sweep.sh > $OUT/files_sweep_chain8.txt 2>&1
paste -d'\n' $OUT/files_sweep_plain.txt $OUT/files_sweep_chain8.txt | cut -c1-170
