#!/bin/bash
# A/B of two library builds on one box: default vs gcc_nmf_amd/libgccnmf_hip_v$V.so (make -C gcc_nmf_amd/csrc variant V=.. X=..)
TAG=${1:-ab}; V=${2:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
for lib in default v$V; do
  if [ $lib = default ]; then unset GCCNMF_HIP_LIB; else export GCCNMF_HIP_LIB=$PWD/gcc_nmf_amd/libgccnmf_hip_$lib.so; fi
  python bench.py --steps 4 --warmup 1 --skip-extras 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: e2e %.0f  K3 %.4f ms  frac %.3f  one-stream iteration %.3f ms' % (b['value'], b['roofline']['avg_launch_ms'], b['roofline']['frac'], b['nmf_iteration_one_stream']['ms']))"
  python bench.py --mode shared-dictionary --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep: shared %.0f  %.2f ms' % (b['value'], b['ms_per_step']))"
done
done 2>&1 | tee $OUT/ab.txt
