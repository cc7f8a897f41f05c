#!/bin/bash
# Round 6, session l: ragged + chained parity after removing the XCD-local counters
TAG=${1:-r06l}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "ragged" > $OUT/pytest_ragged.log 2>&1; echo "ragged tests exit $?"; tail -12 $OUT/pytest_ragged.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "chained" > $OUT/pytest_chain.log 2>&1; echo "chain tests exit $?"; tail -3 $OUT/pytest_chain.log
GCCNMF_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 200 -k "chained" > $OUT/pytest_chain_exp.log 2>&1; echo "chain tests (experiment build) exit $?"; tail -3 $OUT/pytest_chain_exp.log
