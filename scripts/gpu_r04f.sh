#!/bin/bash
TAG=${1:-r04f}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python scripts/big_matrix.py 1024 20 20000 80000 > $OUT/big_matrix.jsonl 2> $OUT/big_matrix.err; echo "big_matrix exit $?"; cat $OUT/big_matrix.jsonl | cut -c1-600; tail -3 $OUT/big_matrix.err
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "big_matrix or batch_is_file or small_batches" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
for t in "9=1" "9=2" "9=0"; do
  echo "== K=128 tune $t"
  timeout 300 python bench.py --dictionary-size 128 --steps 3 --warmup 1 --skip-extras --nmf-groups 1 --tune $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.0f ms_per_step %.2f iter_ms %.4f k3_ms %.4f' % (d['value'], d['ms_per_step'], d['nmf_iteration_one_stream']['ms'], d['roofline']['avg_launch_ms']))"
done
