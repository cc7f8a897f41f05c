#!/bin/bash
# Round 6, final-code session on ONE box: GPU parity suite (product + the lab build's chain tests), smoke, the driver's bench command with every
# sub-record, rocprofv3 kernel stats of the same command (default and --tune 21=0), SQ counters of the chained kernel, the batch-size sweep
# (default rule | plain launches), the drop-in sequence, timelines.
# usage: gpurun --timeout 3000 -- 'bash scripts/sessions/r06_final.sh [tag]'      everything lands in gpurun_out/<tag>/
TAG=${1:-r06m}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
echo "== pytest -m gpu"
timeout 1700 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log; grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
GCCNMF_HIP_LIB=$EXP timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_gpu_exp.log 2>&1; echo "kernel tests on the lab build: exit $? $(grep -E 'passed|failed' $OUT/pytest_gpu_exp.log | tail -1)"
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
echo "== bench (the driver's command)"
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
tw() { local name=$1; shift; timeout 600 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name exit $?"; cut -c1-200 $OUT/bench_$name.json; }
tw plain --tune 21=0 --steps 10 --warmup 3 --skip-extras
tw plain_g1 --tune 21=0 --nmf-groups 1 --steps 10 --warmup 3 --skip-extras
tw K256 --dictionary-size 256 --steps 3 --warmup 1 --skip-extras
tw K512 --dictionary-size 512 --steps 3 --warmup 1 --skip-extras
tw it200 --iterations 200 --steps 3 --warmup 1 --skip-extras
echo "== rocprofv3 kernel stats"
prof() {  # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o $name -- "$@" > $OUT/prof_$name.out 2> $OUT/prof_$name.err
  echo "rocprof $name exit $?"
  f=$(find $OUT/prof_$name -name "*kernel_stats*.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${name}_kernel_stats.csv && head -7 $f | cut -c1-200
  rm -rf $OUT/prof_$name
}
prof bench python bench.py --gpus 1 --steps 2 --warmup 1 --skip-extras
prof plain_g1_bench python bench.py --gpus 1 --steps 2 --warmup 1 --skip-extras --nmf-groups 1 --tune 21=0
prof mixed python bench.py --gpus 1 --steps 1 --warmup 0 --skip-roofline --skip-cpu-baseline --skip-config-lines --no-live-traffic
echo "== SQ counters of the chained kernel"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc_sq -o bench -- python bench.py --gpus 1 --steps 1 --warmup 0 --skip-extras --skip-roofline > $OUT/pmc_sq.log 2>&1
echo "pmc sq exit $?"
find $OUT/pmc_sq -name "*kernel_trace*" -delete
python - "$OUT" <<'PY'
import csv, collections, glob, json, os, sys
out = sys.argv[1]
files = glob.glob(os.path.join(out, 'pmc_sq', '**', '*counter_collection.csv'), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(files[0])):
    if 'gccnmf_gemm' in r['Kernel_Name']:
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} | {'launches': len(next(iter(d.values())))} for k, d in agg.items()}
json.dump(res, open(os.path.join(out, 'pmc_sq.json'), 'w'), indent=1)
for k, d in res.items():
    busy = d.get('SQ_BUSY_CYCLES', 0) or 1
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1
    print(k[:100]); print('   launches %d  MFMA busy / (4 SIMDs x busy cycles) %.3f  MFMA busy / (GRBM active / 8 x 1024 SIMDs) %.3f  wait_any %.1f%% of wave cycles' % (
        d['launches'], d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4.0 * busy), d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (d.get('GRBM_GUI_ACTIVE', 1) / 8.0 * 1024), 100 * d.get('SQ_WAIT_ANY', 0) / wc))
PY
rm -rf $OUT/pmc_sq
echo "== batch-size sweep: the default rule, and plain launches"
FILES="16 24 25 26 32 40 48 51 52 64 72 76 77 80 88 96 102 104" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt | cut -c1-170
echo "== drop-in sequence"
timeout 600 python scripts/dropin_times.py > $OUT/dropin_times.json 2> $OUT/dropin_times.err; echo "dropin_times exit $?"
echo "== timelines (lab build)"
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace_chain.py --files 64 --chain 8 > $OUT/ktrace_chain8_64.txt 2>&1; echo "ktrace exit $?"; grep -E "^==|^  K[1-4]:|mean workgroups|^span" $OUT/ktrace_chain8_64.txt | cut -c1-260
