#!/bin/bash
# Round 6, session d: timeline of one iteration, four launches against one chained launch (experiment build)
TAG=${1:-r06d}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace_chain.py --files 64 > $OUT/ktrace_chain_64.txt 2>&1; echo "ktrace_chain exit $?"; cut -c1-400 $OUT/ktrace_chain_64.txt
