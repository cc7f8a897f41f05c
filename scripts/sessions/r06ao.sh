#!/bin/bash
# Round 6: the K <= 128 slab kernel (K3 + K4a) with its prefetch requests behind the scratch reloads of the tile's first chunk -- kernel tests, then
# previous | new on one box at K = 128 (hop 256 / 128) and K = 64, 96.   usage: gpurun --timeout 2400 -- 'bash scripts/sessions/r06ao.sh [tag]'
TAG=${1:-r06ao}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
D=$PWD/gcc_nmf_amd
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_kernels.log 2>&1; echo "kernel tests exit $? $(grep -E 'passed|failed' $OUT/pytest_kernels.log | tail -1)"
GCCNMF_HIP_LIB=$D/libgccnmf_hip_exp.so timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_kernels_exp.log 2>&1; echo "kernel tests (lab build) exit $? $(grep -E 'passed|failed' $OUT/pytest_kernels_exp.log | tail -1)"
one() {  # label, lib, bench args...
  local label=$1 lib=$2; shift 2
  GCCNMF_HIP_LIB=$D/$lib timeout 600 python bench.py --steps 8 --warmup 2 --skip-extras --skip-cpu-baseline --no-live-traffic "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s %-24s value %9.0f  step %8.3f ms  roofline kernel frac %.4f (%.4f ms)  iteration %.4f ms (%.4f)' % ('$label', '$lib', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['nmf_iteration_one_stream']['ms'], d['nmf_iteration_one_stream']['frac_of_peak']))"
}
for rep in 1 2; do
  for lib in libgccnmf_hip_prev.so libgccnmf_hip.so; do one "K=128 hop 256" $lib --dictionary-size 128; done
  for lib in libgccnmf_hip_prev.so libgccnmf_hip.so; do one "K=128 hop 128" $lib --dictionary-size 128 --hop 128; done
  for lib in libgccnmf_hip_prev.so libgccnmf_hip.so; do one "K=64" $lib --dictionary-size 64; done
  for lib in libgccnmf_hip_prev.so libgccnmf_hip.so; do one "K=96" $lib --dictionary-size 96; done
done | tee $OUT/ab.txt
