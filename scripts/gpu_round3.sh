#!/bin/bash
# Round 2, second session, first GPU call: parity tests of HEAD (incl. the in-launch split-K combines), smoke, the default bench line,
# the single-file variants, the unmodified reference timed on this box's host cores (staged checkout), rocprofv3 of the variants.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round3.sh [tag]'
TAG=${1:-r02j}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 800 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -v amdgpu.ids $OUT/pytest_gpu.log | tail -6
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== single-file variants"
timeout 300 python scripts/single_file.py --fix 2>&1 | grep -v amdgpu.ids | tee $OUT/single_file_fix.txt
echo "== bench"
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== reference on the host cores"
if [ -d oracle/_ref/reference_checkout ]; then
  timeout 300 python scripts/time_reference_cpu.py > $OUT/reference_cpu.json 2> $OUT/reference_cpu.err; echo "reference exit $?"; cut -c1-1200 $OUT/reference_cpu.json
fi
echo "== rocprofv3 single file"
for t in "8=0,9=1" "8=1,9=1" "8=1,9=3" "8=2,9=3"; do
  n=$(echo $t | tr '=,' '__')
  TUNE=$t timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_single_$n -o single -- python scripts/single_file.py --profile > $OUT/prof_single_$n.log 2>&1
  echo "tune $t:"; grep default $OUT/prof_single_$n.log
  find $OUT/prof_single_$n -name "*kernel_stats*.csv" | head -1 | xargs -r head -8 | cut -c1-150
done
find $OUT -name "*kernel_trace*.csv" -size +4M -delete
find $OUT -name "*agent_info*" -delete
