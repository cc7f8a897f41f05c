#!/bin/bash
# quick GPU check: gemm/kernel tests + kbench variants.  usage: gpurun -- 'bash scripts/gpu_quick.sh tag'
TAG=${1:-q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest.log | cut -c1-300
for st in 0 1 3 8; do
  echo "== kbench stagger=$st"
  timeout 300 python scripts/kbench.py --reps 8 --stagger $st 2>&1 | grep -v "^{" | grep -v amdgpu.ids | tee $OUT/kbench_st$st.txt
done
