#!/bin/bash
# Round 6: the failure path of the chained launch (lab build, fault injection key 25) + the neighbouring tests on both builds
TAG=${1:-r06aa}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
GCCNMF_HIP_LIB=$EXP timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "failed_chained or ragged or file_groups" > $OUT/pytest_fault_exp.log 2>&1; echo "lab build exit $?"; tail -12 $OUT/pytest_fault_exp.log | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "failed_chained or ragged or file_groups" > $OUT/pytest_fault.log 2>&1; echo "product exit $?"; tail -3 $OUT/pytest_fault.log | cut -c1-250
