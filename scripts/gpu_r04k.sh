#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04k
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider -k "debug_gemm" 2>&1 | tail -5
for d in 2 3 4; do
echo "== depth $d"
GCCNMF_TUNE=13=$d timeout 300 python scripts/kbench.py --reps 8 2>&1 | grep -E "K2|mfma-only" 
done
