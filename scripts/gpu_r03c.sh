#!/bin/bash
TAG=${1:-r03c}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_distributed.py -q -m gpu --tb=short -p no:cacheprovider -x -k "time_sharded or column or one_call" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
timeout 900 python scripts/ts_profile2.py 160 40 20 > $OUT/ts_profile2.jsonl 2> $OUT/ts_profile2.err; echo "exit $?"; cat $OUT/ts_profile2.jsonl | cut -c1-120
for i in 1 2 3; do
timeout 300 python bench.py --mode time-sharded --steps 5 --warmup 2 > $OUT/ts160_$i.json 2> $OUT/ts160_$i.err; echo "ts160 exit $?"; python -c "import json;d=json.load(open('$OUT/ts160_$i.json'));print(d['value'],d['ms_per_step'])"
done
timeout 300 python bench.py --mode time-sharded --seconds 640 --steps 3 --warmup 1 > $OUT/ts640.json 2> $OUT/ts640.err; echo "ts640 exit $?"; python -c "import json;d=json.load(open('$OUT/ts640.json'));print(d['value'],d['ms_per_step'],d['column_blocks'])"
