#!/bin/bash
TAG=${1:-r04g}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
echo "== distributed gpu tests"
timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_realtime.py -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_dist.log 2>&1; echo "pytest exit $?"; tail -6 $OUT/pytest_dist.log
echo "== two library-RCCL ranks on one device"
NCCL_DEBUG=WARN timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 scripts/rccl_one_device.py > $OUT/rccl_one_device.log 2>&1; echo "exit $?"; grep -E "^\{|WARN|Duplicate|error" $OUT/rccl_one_device.log | head -12
echo "== scale.sh on this box"
timeout 600 bash scripts/scale.sh $OUT/scale > /dev/null 2>&1; cat $OUT/scale/scale.jsonl
