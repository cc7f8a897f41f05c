#!/bin/bash
# Round 6, session x: the GPU parity suite under forced launch forms (GCCNMF_TUNE applies gccnmf_set_tuning pairs at library load)
TAG=${1:-r06x}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/tuning_matrix.txt
for t in "21=0" "21=8" "21=4" "21=8,23=0" "21=8,24=2" "9=0" "9=2" "2=1" "3=0" "7=0" "16=0,17=0"; do
  GCCNMF_TUNE=$t timeout 900 python -m pytest tests -q -m gpu --tb=line -p no:cacheprovider --timeout 300 > $OUT/pytest_tune_$t.log 2>&1
  echo "GCCNMF_TUNE=$t: exit $? $(grep -E 'passed|failed' $OUT/pytest_tune_$t.log | tail -1)" | tee -a $OUT/tuning_matrix.txt
  grep -E "^/root/repo.*(Error|assert)|^FAILED" $OUT/pytest_tune_$t.log | cut -c1-220 | head -8 | tee -a $OUT/tuning_matrix.txt
done
