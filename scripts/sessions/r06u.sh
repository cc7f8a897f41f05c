#!/bin/bash
# Round 6, session u: A/B of a cached first look at the ready counters (variant build -DGEMM_SYNC_CACHED_FIRST) against the product library
TAG=${1:-r06u}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
V9=$PWD/gcc_nmf_amd/libgccnmf_hip_v9.so
GCCNMF_HIP_LIB=$V9 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "chained_iteration_is_bitwise" > $OUT/pytest_v9.log 2>&1; echo "variant chain tests exit $?"; tail -2 $OUT/pytest_v9.log
run() { local name=$1; shift; timeout 300 python bench.py --steps 5 --warmup 2 --skip-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name exit $?"; }
run product_a
GCCNMF_HIP_LIB=$V9 run cached_a
run product_b
GCCNMF_HIP_LIB=$V9 run cached_b
run product_c
GCCNMF_HIP_LIB=$V9 run cached_c
python - <<'PY'
import json, os, glob
for f in sorted(glob.glob(os.path.join(os.environ['OUT'], 'bench_*.json'))):
    b = json.loads(open(f).read().strip().splitlines()[-1])
    print('%-12s value %.0f  ms/step %.2f  iter %.4f ms (%.3f of peak)  call %.2f ms' % (os.path.basename(f)[6:-5], b['value'], b['ms_per_step'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak'], b['roofline']['avg_launch_ms']))
PY
