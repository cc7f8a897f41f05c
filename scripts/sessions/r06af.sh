#!/bin/bash
# Round 6: the fused inverse STFT capped at 168 VGPRs (three workgroups per CU instead of two; 172 bytes of scratch) against the product build.
# usage: gpurun --timeout 900 -- 'bash scripts/sessions/r06af.sh [tag]'
TAG=${1:-r06af}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
D=$PWD/gcc_nmf_amd
for rep in 1 2 3; do for lib in libgccnmf_hip.so libgccnmf_hip_lb3.so; do echo -n "$lib: "; GCCNMF_HIP_LIB=$D/$lib timeout 300 python scripts/stage_times.py 2>&1 | tail -1; done; done | tee $OUT/stage_times_lb3.txt
