#!/bin/bash
# One call with the reference checkout staged (scripts/stage_reference.sh): the unmodified driver on the HIP functions, and the unmodified
# reference's CPU path timed on this box's host cores at the driver's own parameters (K = 128: BASELINE config 1 / runGCCNMF.py:41,60).
TAG=${1:-r04i}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s --tb=short -p no:cacheprovider -k unmodified_reference_driver > $OUT/dropin_driver.log 2>&1; echo "driver test exit $?"; grep -E "unmodified|passed|failed|skipped" $OUT/dropin_driver.log
timeout 900 python scripts/time_reference_cpu.py --dictionary-size 128 --hop 256 --reps 2 > $OUT/reference_cpu_K128_hop256.json 2> $OUT/ref1.err; echo "ref K128 hop256 exit $?"; cut -c1-300 $OUT/reference_cpu_K128_hop256.json
timeout 900 python scripts/time_reference_cpu.py --dictionary-size 128 --hop 128 --reps 2 > $OUT/reference_cpu_K128_hop128.json 2> $OUT/ref2.err; echo "ref K128 hop128 exit $?"; cut -c1-300 $OUT/reference_cpu_K128_hop128.json
