#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats.  Everything lands in gpurun_out/.
# usage: gpurun --timeout 2400 -- 'bash scripts/gpu_round.sh [tag]'
TAG=${1:-r01}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
rocm-smi --showproductname --showmeminfo vram > $OUT/rocm-smi.txt 2>&1
echo "== pytest -m gpu" 
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -3 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
echo "== rocprofv3 kernel stats"
# (a) the default command: KL-NMF as two file groups on two streams -- the groups' launches overlap in time, so the per-kernel
#     averages are those of 32-file launches sharing the chip
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --gpus 1 --steps 2 --warmup 1 --skip-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof exit $?"; ls -R $OUT/prof | head -20
find $OUT/prof -name "*kernel_stats*.csv" | head -1 | xargs -r head -8
# (b) one launch per stage over the whole batch: the configuration bench.py times its roofline kernel in
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_g1 -o bench -- python bench.py --gpus 1 --steps 2 --warmup 1 --skip-cpu-baseline --nmf-groups 1 > $OUT/prof_g1_bench.json 2> $OUT/prof_g1.err
echo "rocprof (--nmf-groups 1) exit $?"
find $OUT/prof_g1 -name "*kernel_stats*.csv" | head -1 | xargs -r head -8
find $OUT/prof_g1 -name "*kernel_trace*.csv" -size +20M -delete
# keep the merged-back payload small: drop the raw per-dispatch trace if it is huge
find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete
# HBM traffic of the GEMM kernels: separate PMC passes (never combined with the trace domains above)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o bench -- python bench.py --gpus 1 --steps 1 --warmup 0 --skip-cpu-baseline --skip-roofline --nmf-groups 1 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
  echo "pmc $c exit $?"
  find $OUT/pmc_$c -name "*kernel_trace*" -delete
done
python - <<'PY'
import csv, collections, glob, json, os, sys
out = os.environ.get('OUT', sys.argv[1] if len(sys.argv) > 1 else '.')
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
