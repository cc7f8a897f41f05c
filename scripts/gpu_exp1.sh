#!/bin/bash
# experiments: per-workgroup timeline of the single-file launches, hardware-queue count vs the stream layout of the batch path
OUT=gpurun_out/${1:-exp1}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_pipeline.py -q -m gpu --tb=short -p no:cacheprovider -k "split_k or in_launch" 2>&1 | grep -v amdgpu.ids | tail -4
echo "== ktrace default"; timeout 120 python scripts/ktrace_single.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ktrace_default.txt
echo "== ktrace 8=2,9=3"; TUNE=8=2,9=3 timeout 120 python scripts/ktrace_single.py 2>&1 | grep -v amdgpu.ids | tee $OUT/ktrace_fix.txt
B="python bench.py --gpus 1 --steps 5 --warmup 1 --skip-cpu-baseline --skip-roofline"
pick() { python -c "import json,sys; b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'value %.0f  ms/step %.2f  h2h %.0f  h2h pipelined %.0f' % (b['value'], b['ms_per_step'], b['host_to_host_frames_per_s'], b['host_to_host_pipelined_frames_per_s']))" $1 "$2"; }
timeout 200 $B > $OUT/b_default.json 2>/dev/null; pick $OUT/b_default.json "default queues, 2 groups:"
GPU_MAX_HW_QUEUES=8 timeout 200 $B > $OUT/b_q8.json 2>/dev/null; pick $OUT/b_q8.json "8 queues, 2 groups:"
GPU_MAX_HW_QUEUES=8 timeout 200 $B --nmf-groups 4 > $OUT/b_q8_g4.json 2>/dev/null; pick $OUT/b_q8_g4.json "8 queues, 4 groups:"
GPU_MAX_HW_QUEUES=2 timeout 200 $B > $OUT/b_q2.json 2>/dev/null; pick $OUT/b_q2.json "2 queues, 2 groups:"
