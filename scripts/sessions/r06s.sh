#!/bin/bash
# Round 6, session s: long calls as several chained launches (key 24), the staging / chain / ragged tests once more
TAG=${1:-r06s}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_named_functions.py tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "chained or ragged or long_call or named or page_locked or resident or magnitude or short_dictionary" > $OUT/pytest_sel.log 2>&1; echo "selected tests exit $?"; tail -4 $OUT/pytest_sel.log | cut -c1-200
