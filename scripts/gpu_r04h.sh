#!/bin/bash
TAG=${1:-r04h}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_realtime.py -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_rt.log 2>&1; echo "pytest exit $?"; tail -25 $OUT/pytest_rt.log
timeout 300 python bench.py --mode streaming > $OUT/streaming_bench.json 2> $OUT/streaming.err; echo "streaming exit $?"; cut -c1-400 $OUT/streaming_bench.json
