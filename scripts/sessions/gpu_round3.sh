#!/bin/bash
# One GPU-box session of round 3: parity tests, smoke, every bench mode, the 200-iteration line (BASELINE config 3 as written), rocprofv3
# kernel stats (default command, --nmf-groups 1, 200 iterations, single file, collective modes), batch-size sweep, HBM-traffic PMC passes.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_round3.sh [tag]'      everything lands in gpurun_out/<tag>/
TAG=${1:-r03}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 900 python bench.py --iterations 200 --steps 3 --warmup 1 --skip-cpu-baseline > $OUT/bench_it200.json 2> $OUT/bench_it200.err; echo "bench it200 exit $?"; cut -c1-300 $OUT/bench_it200.json
for mode in shared-dictionary streaming time-sharded; do
  timeout 300 python bench.py --mode $mode --steps 3 --warmup 1 > $OUT/${mode}_bench.json 2> $OUT/${mode}_bench.err; echo "$mode exit $?"; cut -c1-400 $OUT/${mode}_bench.json
done
timeout 300 python bench.py --mode time-sharded --seconds 640 --steps 3 --warmup 1 > $OUT/time-sharded_640s_bench.json 2> $OUT/ts640.err; echo "ts640 exit $?"; cut -c1-300 $OUT/time-sharded_640s_bench.json
for i in 1 2 3 4 5; do timeout 300 python bench.py --mode time-sharded --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('time-sharded 160 s, process $i: %.0f frames/s, %.2f ms per step' % (d['value'], d['ms_per_step']))"; done > $OUT/time-sharded_repeats.txt; cat $OUT/time-sharded_repeats.txt
echo "== batch-size sweep"
bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
echo "== rocprofv3 kernel stats"
prof() {  # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- python bench.py "$@" > $OUT/prof_$name.json 2> $OUT/prof_$name.err
  echo "rocprof $name exit $?"; find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1 | xargs -r head -5 | cut -c1-170
}
prof bench --gpus 1 --steps 2 --warmup 1 --skip-extras
prof g1_bench --gpus 1 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
prof it200 --gpus 1 --steps 1 --warmup 1 --skip-extras --nmf-groups 1 --iterations 200
prof shared-dictionary --mode shared-dictionary --steps 2 --warmup 1
prof time-sharded --mode time-sharded --steps 2 --warmup 1
prof streaming --mode streaming
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_single -o single -- python scripts/single_file.py --profile > $OUT/prof_single.log 2>&1
grep default $OUT/prof_single.log
find $OUT -name "*kernel_trace*.csv" -size +8M -delete
if [ -z "$SKIP_PMC" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- python bench.py --gpus 1 --steps 1 --warmup 0 --skip-extras --skip-roofline --nmf-groups 1 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
  echo "pmc $c exit $?"
  find $OUT/pmc_$c -name "*kernel_trace*" -delete
done
python - <<'PY'
import csv, collections, glob, json, os
out = os.environ['OUT']
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob(os.path.join(out, 'pmc_' + c, '*counter_collection.csv'))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r['Counter_Name'] == c and 'gccnmf_gemm' in r['Kernel_Name']:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    res[c] = {k: {'launches': len(v), 'mean_KB': sum(v) / len(v)} for k, v in agg.items()}
json.dump(res, open(os.path.join(out, 'pmc_traffic_raw.json'), 'w'), indent=1)
print(json.dumps(res, indent=1)[:1500])
PY
fi
