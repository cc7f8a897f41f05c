#!/bin/bash
# Round 6: the fused inverse STFT at three workgroups per CU (168 VGPRs + 96 B of scratch, window back in global memory so that three 53.5 KB
# workgroups fit the LDS) against the product build.   usage: gpurun --timeout 900 -- 'bash scripts/sessions/r06aq.sh [tag]'
TAG=${1:-r06aq}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
D=$PWD/gcc_nmf_amd
for rep in 1 2 3; do for lib in libgccnmf_hip.so libgccnmf_hip_o3.so; do echo -n "$lib: "; GCCNMF_HIP_LIB=$D/$lib timeout 300 python scripts/stage_times.py 2>&1 | tail -n 1; done; done | tee $OUT/stage_times_o3.txt
