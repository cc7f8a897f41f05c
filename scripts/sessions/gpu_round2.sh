#!/bin/bash
# One GPU-box session of round 2: parity tests, smoke, the three bench modes, rocprofv3 kernel stats (default command, --nmf-groups 1,
# single file), HBM-traffic PMC passes.  Everything lands in gpurun_out/<tag>/.
# usage: gpurun --timeout 1800 -- 'bash scripts/gpu_round2.sh [tag]'
TAG=${1:-r02}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
for mode in shared-dictionary streaming; do
  timeout 300 python bench.py --mode $mode --steps 3 --warmup 1 > $OUT/${mode}_bench.json 2> $OUT/${mode}_bench.err; echo "$mode exit $?"; cut -c1-600 $OUT/${mode}_bench.json
done
echo "== rocprofv3 kernel stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --gpus 1 --steps 2 --warmup 1 --skip-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof exit $?"; find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -r head -6 | cut -c1-160
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_g1 -o bench -- python bench.py --gpus 1 --steps 2 --warmup 1 --skip-cpu-baseline --nmf-groups 1 > $OUT/prof_g1_bench.json 2> $OUT/prof_g1.err
echo "rocprof (--nmf-groups 1) exit $?"; find $OUT/prof_g1 -name "*kernel_stats*.csv" | head -1 | xargs -r head -6 | cut -c1-160
for mode in shared-dictionary streaming; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$mode -o bench -- python bench.py --mode $mode --steps 2 --warmup 1 > $OUT/prof_${mode}.json 2> $OUT/prof_${mode}.err
  echo "$mode rocprof exit $?"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_single -o single -- python scripts/single_file.py --profile > $OUT/prof_single.log 2>&1
grep default $OUT/prof_single.log
if [ -d oracle/_ref/reference_checkout ]; then     # staged for this one call (scripts/stage_reference.sh): the unmodified reference on this box's host cores
  timeout 300 python scripts/time_reference_cpu.py > $OUT/reference_cpu.json 2> $OUT/reference_cpu.err; echo "reference exit $?"; cut -c1-400 $OUT/reference_cpu.json
fi
find $OUT -name "*kernel_trace*.csv" -size +8M -delete
if [ -z "$SKIP_PMC" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- python bench.py --gpus 1 --steps 1 --warmup 0 --skip-cpu-baseline --skip-roofline --nmf-groups 1 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
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
print(json.dumps(res, indent=1)[:1200])
PY
fi
