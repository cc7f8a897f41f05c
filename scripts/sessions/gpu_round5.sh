#!/bin/bash
# One GPU-box session of round 5 on the FINAL code: parity tests (product library), smoke, the default bench line with its live sub-records
# and their stand-alone twins, every mode, rocprofv3 kernel stats (default, --nmf-groups 1, K = 128 / 256 / 512, one mixture), the batch-size
# sweep, HBM-traffic and SQ PMC passes, the per-phase cycle tables of the throughput tile (classic grid and the resident-workgroup experiment).
# usage: gpurun --timeout 3000 -- 'bash scripts/gpu_round5.sh [tag]'      everything lands in gpurun_out/<tag>/
TAG=${1:-r05}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
OLD=$PWD/gcc_nmf_amd/libgccnmf_hip_r04.so
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
PROBE=$PWD/gcc_nmf_amd/libgccnmf_hip_vprobe.so
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -2 $OUT/smoke.log
echo "== bench (the driver's command, with every sub-record)"
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== stand-alone twins of the sub-records"
tw() { local name=$1; shift; timeout 600 python bench.py "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name exit $?"; cut -c1-220 $OUT/bench_$name.json; }
tw K128 --dictionary-size 128 --steps 5 --warmup 1 --skip-cpu-baseline
tw K128_hop128 --dictionary-size 128 --hop 128 --steps 5 --warmup 1 --skip-cpu-baseline
tw K64 --dictionary-size 64 --steps 3 --warmup 1 --skip-extras
tw K256 --dictionary-size 256 --steps 3 --warmup 1 --skip-extras
tw K512 --dictionary-size 512 --steps 3 --warmup 1 --skip-extras
tw it200 --iterations 200 --steps 3 --warmup 1 --skip-extras
tw g1 --nmf-groups 1 --steps 5 --warmup 2 --skip-extras
tw rcp --tune 7=0 --steps 5 --warmup 2 --skip-extras
for mode in shared-dictionary streaming time-sharded; do
  timeout 300 python bench.py --mode $mode --steps 3 --warmup 1 > $OUT/${mode}_bench.json 2> $OUT/${mode}_bench.err; echo "$mode exit $?"; cut -c1-250 $OUT/${mode}_bench.json
done
timeout 300 python bench.py --mode time-sharded --seconds 640 --steps 3 --warmup 1 > $OUT/time-sharded_640s_bench.json 2> $OUT/ts640.err; echo "ts640 exit $?"; cut -c1-200 $OUT/time-sharded_640s_bench.json
timeout 300 python scripts/big_matrix.py 1024 20 20000 80000 > $OUT/big_matrix.jsonl 2> $OUT/big_matrix.err; echo "big_matrix exit $?"; cut -c1-200 $OUT/big_matrix.jsonl
GCCNMF_HIP_LIB=$OLD timeout 600 python bench.py --steps 5 --warmup 2 --skip-extras > $OUT/bench_r04lib.json 2> $OUT/bench_r04lib.err; echo "r04 library, same box: exit $?"; cut -c1-160 $OUT/bench_r04lib.json
echo "== stage times"
timeout 300 python scripts/stage_times.py > $OUT/stage_times.txt 2>&1; tail -1 $OUT/stage_times.txt
echo "== batch-size sweep"
FILES="16 24 25 26 32 40 48 51 52 64 72 76 77 80 88 96 102 104" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
echo "== kbench (product library; K = 1024 and 256)"
timeout 300 python scripts/kbench.py --reps 8 > $OUT/kbench.txt 2> $OUT/kbench.err; grep -E "^K[1-4]" $OUT/kbench.txt | cut -c1-110
timeout 300 python scripts/kbench.py --reps 8 --K 256 > $OUT/kbench_K256.txt 2> $OUT/kbench_K256.err; grep -E "^K[1-4]a? fused|^K4b" $OUT/kbench_K256.txt | cut -c1-110
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
prof K256_g1_bench python bench.py --dictionary-size 256 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
prof K512_g1_bench python bench.py --dictionary-size 512 --steps 2 --warmup 1 --skip-extras --nmf-groups 1
prof single_file python scripts/single_file.py --profile
echo "== PMC passes (each counter set in its own run, --kernel-trace only)"
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
print(json.dumps(res, indent=1)[:1500])
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc_sq -o kbench -- python scripts/kbench.py --reps 3 > $OUT/pmc_sq.log 2>&1
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
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1
    busy = d.get('SQ_BUSY_CYCLES', 0) or 1
    print(k[:110])
    print('   launches %d  MFMA busy / (4 SIMDs x busy cycles) %.3f   of wave cycles: wait_any %.1f%%  wait_inst_any %.1f%%  active_inst %.1f%%' % (
        d['launches'], d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (4.0 * busy), 100 * d.get('SQ_WAIT_ANY', 0) / wc, 100 * d.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * d.get('SQ_ACTIVE_INST_ANY', 0) / wc))
PY
rm -rf $OUT/pmc_sq
echo "== per-phase cycle tables of the throughput tile (probe build of the experiment library)"
GCCNMF_HIP_LIB=$PROBE timeout 300 python scripts/ktrace.py --files 64 --stage 3 --probe > $OUT/ktrace_probe_classic.txt 2>&1; head -32 $OUT/ktrace_probe_classic.txt | cut -c1-200
GCCNMF_HIP_LIB=$PROBE timeout 300 python scripts/ktrace.py --files 64 --stage 3 --probe --resident > $OUT/ktrace_probe_resident.txt 2>&1; head -16 $OUT/ktrace_probe_resident.txt | cut -c1-200
GCCNMF_HIP_LIB=$PROBE timeout 300 python scripts/ktrace.py --files 64 --stage 2 --probe --resident > $OUT/ktrace_probe_resident_K2.txt 2>&1; head -12 $OUT/ktrace_probe_resident_K2.txt | cut -c1-200
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace.py --files 64 --stage 3 --repeat 3 > $OUT/ktrace_timeline.txt 2>&1; sed -n 2,6p $OUT/ktrace_timeline.txt
echo "== experiment library: kernel tests"
GCCNMF_HIP_LIB=$EXP timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -p no:cacheprovider > $OUT/pytest_gpu_exp.log 2>&1; echo "pytest exp exit $?"; tail -2 $OUT/pytest_gpu_exp.log
