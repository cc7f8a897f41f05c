#!/bin/bash
# K = 128 at batch scale: the one-pass W update on 32 atoms per workgroup (whole 128-byte lines) against 16.
TAG=${1:-r05l}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu -x -k "short_dictionary or fused or K128 or 128 or klnmf" --tb=short -p no:cacheprovider > $OUT/pytest_k128.log 2>&1; echo "K <= 128 tests exit $?"; tail -2 $OUT/pytest_k128.log
for t in "20=0" "20=1" "20=0" "20=1"; do
  for hop in 256 128; do
    GCCNMF_HIP_LIB=$EXP GCCNMF_TUNE=$t timeout 300 python bench.py --dictionary-size 128 --hop $hop --steps 5 --warmup 2 --skip-extras > $OUT/bench_${t}_$hop.json 2>/dev/null
    python - $OUT/bench_${t}_$hop.json $t $hop <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('tune %s hop %s: %.0f frames/s  iteration %.4f ms (%.3f of peak)' % (sys.argv[2], sys.argv[3], b['value'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak']))
PY
  done
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k128 -- python bench.py --dictionary-size 128 --steps 2 --warmup 1 --skip-extras --nmf-groups 1 > $OUT/prof.out 2> $OUT/prof.err
f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp $f $OUT/K128_g1_bench_kernel_stats.csv && head -5 $f | cut -c1-200; rm -rf $OUT/prof
timeout 300 python bench.py --dictionary-size 64 --steps 3 --warmup 1 --skip-extras > $OUT/bench_K64.json 2>/dev/null; python - $OUT/bench_K64.json <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('K 64: %.0f frames/s  iteration %.4f ms (%.3f of peak)' % (b['value'], b['nmf_iteration_one_stream']['ms'], b['nmf_iteration_one_stream']['frac_of_peak']))
PY
