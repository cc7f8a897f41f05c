#!/bin/bash
OUT=gpurun_out/${1:-exp2}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -8
timeout 300 python scripts/single_file.py --fix 2>&1 | grep -v amdgpu.ids | tee $OUT/single_file_fix.txt
echo "== ktrace default"; timeout 120 python scripts/ktrace_single.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ktrace_default.txt
echo "== ktrace 8=3,9=3"; TUNE=8=3,9=3 timeout 120 python scripts/ktrace_single.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ktrace_fix3.txt
