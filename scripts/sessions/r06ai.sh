#!/bin/bash
# Round 6: the fused inverse STFT with its fetches, its accumulator shift and its overlap-add batched (all reads of a phase in flight together):
# tests, stage times against the previous build on the same box, phase clocks.   usage: gpurun --timeout 1500 -- 'bash scripts/sessions/r06ai.sh [tag]'
TAG=${1:-r06ai}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
D=$PWD/gcc_nmf_amd
echo "== tests (stft / istft kernels, pipeline, named functions, realtime)"
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_named_functions.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "stft or istft or pipeline or named or waveform or separate or mixture or fft" > $OUT/pytest_istft.log 2>&1; echo "exit $? $(grep -E 'passed|failed' $OUT/pytest_istft.log | tail -1)"
echo "== stage times: previous | new, three times"
for rep in 1 2 3; do for lib in libgccnmf_hip_prev.so libgccnmf_hip.so; do [ -f $D/$lib ] || continue; echo -n "$lib: "; GCCNMF_HIP_LIB=$D/$lib timeout 300 python scripts/stage_times.py 2>&1 | tail -1; done; done | tee $OUT/stage_times_ab.txt
echo "== phase clocks (lab build)"
GCCNMF_HIP_LIB=$D/libgccnmf_hip_exp.so timeout 300 python scripts/ktrace_istft.py 2>&1 | tee $OUT/ktrace_istft.txt
