#!/bin/bash
# Round 3, session B: stream groups in the shared-dictionary iteration; where the time-sharded step goes.
TAG=${1:-r03b}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -x -k "not bench_launches and not config4" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
timeout 600 python scripts/ts_profile.py 160 > $OUT/ts_profile_160.json 2> $OUT/ts_profile_160.err; echo "ts_profile exit $?"; cat $OUT/ts_profile_160.err | cut -c1-330
for g in 1 2 3; do
  timeout 300 python bench.py --mode shared-dictionary --steps 3 --warmup 1 --tune 8=$g > $OUT/shared_g$g.json 2> $OUT/shared_g$g.err; echo "shared g=$g exit $?"; python -c "import json;d=json.load(open('$OUT/shared_g$g.json'));print(d['value'],d['ms_per_step'])"
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ts -o ts -- python bench.py --mode time-sharded --steps 2 --warmup 1 > $OUT/prof_ts.json 2> $OUT/prof_ts.err
echo "rocprof ts exit $?"; find $OUT/prof_ts -name "*kernel_stats*.csv" | head -1 | xargs -r head -14 | cut -c1-200
find $OUT -name "*kernel_trace*.csv" -size +8M -delete
