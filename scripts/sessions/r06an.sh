#!/bin/bash
# Round 6: the generic tile epilogue with its arithmetic in front of the predicated stores -- kernel tests, then previous | new on one box:
# the bench's timed steps and one-stream iteration at 64 files (K = 1024, 256, 512) and 52 files.   usage: gpurun --timeout 2400 -- 'bash scripts/sessions/r06an.sh [tag]'
TAG=${1:-r06an}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
D=$PWD/gcc_nmf_amd
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_kernels.log 2>&1; echo "kernel tests exit $? $(grep -E 'passed|failed' $OUT/pytest_kernels.log | tail -1)"
GCCNMF_HIP_LIB=$D/libgccnmf_hip_exp.so timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_kernels_exp.log 2>&1; echo "kernel tests (lab build) exit $? $(grep -E 'passed|failed' $OUT/pytest_kernels_exp.log | tail -1)"
one() {  # label, lib, bench args...
  local label=$1 lib=$2; shift 2
  GCCNMF_HIP_LIB=$D/$lib timeout 600 python bench.py --steps 6 --warmup 2 --skip-extras --skip-cpu-baseline --no-live-traffic "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %-24s value %8.0f  step %8.3f ms  chain/K3 frac %.4f  iteration %.4f ms (%.4f)' % ('$label', '$lib', d['value'], d['ms_per_step'], d['roofline']['frac'], d['nmf_iteration_one_stream']['ms'], d['nmf_iteration_one_stream']['frac_of_peak']))"
}
for rep in 1 2; do
  for lib in libgccnmf_hip_prev.so libgccnmf_hip.so; do one "K=1024 64 files" $lib; done
  for lib in libgccnmf_hip_prev.so libgccnmf_hip.so; do one "K=1024 52 files" $lib --files 52; done
  for lib in libgccnmf_hip_prev.so libgccnmf_hip.so; do one "K=256 64 files" $lib --dictionary-size 256; done
  for lib in libgccnmf_hip_prev.so libgccnmf_hip.so; do one "K=1024 64 files plain" $lib --tune 21=0 --nmf-groups 1; done
done | tee $OUT/ab.txt
