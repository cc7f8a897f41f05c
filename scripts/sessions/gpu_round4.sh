#!/bin/bash
# One GPU-box session of round 4: parity tests, smoke, every bench mode, K = 128 lines (BASELINE config 1 / the reference driver's own
# parameters), rocprofv3 kernel stats (default command, --nmf-groups 1, K = 128, one mixture alone), batch-size sweep, HBM-traffic PMC passes.
# usage: gpurun --timeout 3000 -- 'bash scripts/gpu_round4.sh [tag]'      everything lands in gpurun_out/<tag>/
TAG=${1:-r04}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 python bench.py --dictionary-size 128 --steps 5 --warmup 1 > $OUT/bench_K128.json 2> $OUT/bench_K128.err; echo "bench K128 exit $?"; cut -c1-200 $OUT/bench_K128.json
timeout 600 python bench.py --dictionary-size 128 --hop 128 --steps 5 --warmup 1 > $OUT/bench_K128_hop128.json 2> $OUT/bench_K128_hop128.err; echo "bench K128 hop128 exit $?"; cut -c1-200 $OUT/bench_K128_hop128.json
timeout 900 python bench.py --iterations 200 --steps 3 --warmup 1 --skip-cpu-baseline > $OUT/bench_it200.json 2> $OUT/bench_it200.err; echo "bench it200 exit $?"; cut -c1-200 $OUT/bench_it200.json
for mode in shared-dictionary streaming time-sharded; do
  timeout 300 python bench.py --mode $mode --steps 3 --warmup 1 > $OUT/${mode}_bench.json 2> $OUT/${mode}_bench.err; echo "$mode exit $?"; cut -c1-300 $OUT/${mode}_bench.json
done
timeout 300 python bench.py --mode time-sharded --seconds 640 --steps 3 --warmup 1 > $OUT/time-sharded_640s_bench.json 2> $OUT/ts640.err; echo "ts640 exit $?"; cut -c1-200 $OUT/time-sharded_640s_bench.json
timeout 300 python scripts/big_matrix.py 1024 20 20000 80000 > $OUT/big_matrix.jsonl 2> $OUT/big_matrix.err; echo "big_matrix exit $?"; cut -c1-200 $OUT/big_matrix.jsonl
for cfg in "1024 256 1" "128 128 1" "128 256 1"; do
  timeout 300 python scripts/direct_bench.py $cfg > $OUT/direct_bench_$(echo $cfg | tr ' ' '_').txt 2>&1; echo "direct_bench $cfg exit $?"
  grep -E "^split|^direct|engine.run" $OUT/direct_bench_$(echo $cfg | tr ' ' '_').txt
done
echo "== stage times"
timeout 300 python scripts/stage_times.py > $OUT/stage_times.txt 2>&1; tail -1 $OUT/stage_times.txt
echo "== batch-size sweep"
bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
echo "== rocprofv3 kernel stats"
prof() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/prof_$name.out 2> $OUT/prof_$name.err
  echo "rocprof $name exit $?"
  f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv && head -6 $f | cut -c1-170
  rm -rf $OUT/prof_$name
}
prof bench python bench.py --gpus 1 --steps 2 --warmup 1 --skip-extras
prof g1_bench python bench.py --gpus 1 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
prof K128_g1_bench python bench.py --dictionary-size 128 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
prof K128_hop128_g1_bench python bench.py --dictionary-size 128 --hop 128 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
prof single_file python scripts/single_file.py --profile
K=128 HOP=128 prof single_file_K128_hop128 python scripts/single_file.py --profile
prof streaming python bench.py --mode streaming
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
    files = glob.glob(os.path.join(out, 'pmc_' + c, '**', '*counter_collection.csv'), recursive=True)
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
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
fi
