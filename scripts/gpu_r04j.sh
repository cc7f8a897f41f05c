#!/bin/bash
TAG=${1:-r04j}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "fft or stft or istft or signal_estimates" > $OUT/pytest_fft.log 2>&1; echo "pytest exit $?"; tail -8 $OUT/pytest_fft.log
for r in 0 1; do
  echo "== stage times, tuning 15=$r"
  GCCNMF_TUNE=15=$r timeout 300 python scripts/stage_times.py 2>&1 | tail -12
done
