#!/bin/bash
# single-file latency path: parity tests of the small-tile kernels, tuning sweep, rocprofv3 kernel stats of one run
TAG=${1:-r02s}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | grep -v "^$" | tail -15
timeout 300 python scripts/single_file.py 2>&1 | grep -v amdgpu.ids | tee $OUT/single_file.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o single -- python scripts/single_file.py --profile > $OUT/prof.log 2>&1
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -r head -14 | cut -c1-200
find $OUT/prof -name "*kernel_trace*.csv" -size +10M -delete
