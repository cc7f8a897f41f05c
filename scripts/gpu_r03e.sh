#!/bin/bash
TAG=${1:-r03e}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
FILES="25 26 40 48 52 64 77 80 96" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
for t in 0 1; do
timeout 300 python bench.py --gpus 1 --steps 5 --warmup 1 --skip-extras --tune 9=$t > $OUT/bench_tail$t.json 2> $OUT/bench_tail$t.err; python -c "import json;b=json.load(open('$OUT/bench_tail$t.json'));print('tail split $t', b['value'], b['ms_per_step'], b['roofline']['frac'])"
timeout 300 python bench.py --gpus 1 --steps 5 --warmup 1 --skip-extras --files 80 --tune 9=$t > $OUT/bench80_tail$t.json 2> $OUT/bench80_tail$t.err; python -c "import json;b=json.load(open('$OUT/bench80_tail$t.json'));print('80 files, tail split $t', b['value'], b['ms_per_step'], b['roofline']['frac'])"
done
