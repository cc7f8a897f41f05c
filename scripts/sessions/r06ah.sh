#!/bin/bash
# Round 6: per-phase shader clocks of the fused inverse STFT's workgroups (lab build).   usage: gpurun --timeout 600 -- 'bash scripts/sessions/r06ah.sh [tag]'
TAG=${1:-r06ah}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
GCCNMF_HIP_LIB=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so timeout 300 python scripts/ktrace_istft.py 2>&1 | tee $OUT/ktrace_istft.txt
