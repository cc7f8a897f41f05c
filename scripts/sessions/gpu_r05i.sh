#!/bin/bash
# Round 5, after the ragged narrow tile moved to its natural place in the lists: tests, the driver's bench command, kernel stats, HBM traffic.
TAG=${1:-r05i}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
OLD=$PWD/gcc_nmf_amd/libgccnmf_hip_r04.so
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
PROBE=$PWD/gcc_nmf_amd/libgccnmf_hip_vprobe.so
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest_gpu.log | head
GCCNMF_HIP_LIB=$EXP timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu_exp.log 2>&1; echo "pytest exp exit $?"; tail -2 $OUT/pytest_gpu_exp.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-260 $OUT/bench.json; tail -2 $OUT/bench.err
b() { local name=$1; shift; env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --skip-extras ${EXTRA} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name exit $?"; python - $OUT/bench_$name.json <<'PY'
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=b.get('roofline',{}); print('  value %.0f  ms %.2f  k3 %.4f ms frac %.3f  iter %.3f ms' % (b['value'], b['ms_per_step'], r.get('avg_launch_ms',0), r.get('frac',0), b.get('nmf_iteration_one_stream',{}).get('ms',0)))
except Exception as e: print('  parse failed', e)
PY
}
b r04 GCCNMF_HIP_LIB=$OLD
b new GCCNMF_TUNE=
b new_wide GCCNMF_TUNE=9=0
b r04_2 GCCNMF_HIP_LIB=$OLD
b new_2 GCCNMF_TUNE=
EXTRA="--nmf-groups 1" b new_g1 GCCNMF_TUNE=
timeout 300 python scripts/kbench.py --reps 8 > $OUT/kbench.txt 2> $OUT/kbench.err; grep -E "^K[1-4]a? fused" $OUT/kbench.txt | cut -c1-110
GCCNMF_HIP_LIB=$OLD timeout 300 python scripts/kbench.py --reps 8 > $OUT/kbench_r04.txt 2> $OUT/kbench_r04.err; grep -E "^K[1-4]a? fused" $OUT/kbench_r04.txt | cut -c1-110
FILES="16 24 25 26 32 40 48 51 52 64 72 76 77 80 88 96 102 104" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
prof() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/prof_$name.out 2> $OUT/prof_$name.err
  echo "rocprof $name exit $?"
  f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv && head -5 $f | cut -c1-170
  rm -rf $OUT/prof_$name
}
prof bench python bench.py --gpus 1 --steps 2 --warmup 1 --skip-extras
prof g1_bench python bench.py --gpus 1 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
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
for c, d in res.items():
    for k, v in d.items():
        if v['launches'] >= 100: print(c, k[:70], v)
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
GCCNMF_HIP_LIB=$PROBE timeout 300 python scripts/ktrace.py --files 64 --stage 3 --probe > $OUT/ktrace_probe_classic.txt 2>&1; head -30 $OUT/ktrace_probe_classic.txt | cut -c1-160
GCCNMF_HIP_LIB=$PROBE timeout 300 python scripts/ktrace.py --files 64 --stage 3 --probe --resident > $OUT/ktrace_probe_resident.txt 2>&1; head -14 $OUT/ktrace_probe_resident.txt | cut -c1-160
GCCNMF_HIP_LIB=$PROBE timeout 300 python scripts/ktrace.py --files 64 --stage 2 --probe --resident > $OUT/ktrace_probe_resident_K2.txt 2>&1; head -12 $OUT/ktrace_probe_resident_K2.txt | cut -c1-160
