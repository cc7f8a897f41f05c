#!/bin/bash
# Round 6, session p: the short-dictionary chain on resident workgroups with per-XCD tickets (key 24): parity, A/B, timeline
TAG=${1:-r06p}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 200 -k "short_dictionary_chained" > $OUT/pytest_short.log 2>&1; echo "short-dictionary tests exit $?"; tail -6 $OUT/pytest_short.log | cut -c1-250
GCCNMF_HIP_LIB=$EXP timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 200 -k "short_dictionary_chained" > $OUT/pytest_short_exp.log 2>&1; echo "lab build exit $?"; tail -3 $OUT/pytest_short_exp.log
run() { local name=$1; shift; timeout 300 python bench.py --steps 5 --warmup 2 --skip-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name exit $?"; }
for K in 128 64; do
  run K${K}_plain --dictionary-size $K --tune 21=0
  run K${K}_chain --dictionary-size $K
  run K${K}_resident --dictionary-size $K --tune 24=1
  run K${K}_chain_b --dictionary-size $K
  run K${K}_resident_b --dictionary-size $K --tune 24=1
done
run K128_hop128_chain --dictionary-size 128 --hop 128
run K128_hop128_resident --dictionary-size 128 --hop 128 --tune 24=1
python - <<'PY'
import json, os, glob
for f in sorted(glob.glob(os.path.join(os.environ['OUT'], 'bench_K*.json'))):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
        print('%-22s value %.0f  ms/step %.2f  iter %.4f ms (%.3f of peak)  tdoa %s' % (os.path.basename(f)[6:-5], b['value'], b['ms_per_step'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['tdoa_indexes_as_expected']))
    except Exception as e:
        print(f, 'failed', e, open(f[:-5] + '.err').read()[-300:])
PY
