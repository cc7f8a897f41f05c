#!/bin/bash
# Round 6, session o: timeline of the short-dictionary chained call
TAG=${1:-r06o}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace_short.py --K 128 > $OUT/ktrace_short_K128.txt 2>&1; echo "exit $?"; cut -c1-400 $OUT/ktrace_short_K128.txt
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace_short.py --K 64 > $OUT/ktrace_short_K64.txt 2>&1; echo "exit $?"; head -8 $OUT/ktrace_short_K64.txt | cut -c1-300
