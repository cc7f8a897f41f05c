#!/bin/bash
# Round 6, session v: wall time of the driver's default command; rocprofv3 kernel stats of a run that holds only 100-iteration chained calls
TAG=${1:-r06v}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
/usr/bin/time -v python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench exit $?"; grep -E "Elapsed|Maximum resident" $OUT/bench_default.err
python - <<'PY'
import json, os
b = json.loads(open(os.path.join(os.environ['OUT'], 'bench_default.json')).read().strip().splitlines()[-1])
print('value', b['value'], 'steps', b['steps'], 'warmup', b['warmup'], 'roofline', b['roofline']['frac'], b['roofline']['avg_launch_ms'], 'traffic', b['roofline']['traffic'], 'cfg s', b.get('config_lines_seconds'))
print(sorted(b.keys()))
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_chain -o chain -- python bench.py --gpus 1 --steps 5 --warmup 2 --skip-extras --skip-roofline > $OUT/prof_chain.out 2> $OUT/prof_chain.err
f=$(find $OUT/prof_chain -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp $f $OUT/chain_only_kernel_stats.csv && head -4 $f | cut -c1-220
rm -rf $OUT/prof_chain
EXP=$PWD/gcc_nmf_amd/libgccnmf_hip_exp.so
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace_short.py --K 128 > $OUT/ktrace_short_K128.txt 2>&1; echo "ktrace short exit $?"; head -8 $OUT/ktrace_short_K128.txt | cut -c1-330; sed -n 9,12p $OUT/ktrace_short_K128.txt | cut -c1-600
GCCNMF_HIP_LIB=$EXP timeout 300 python scripts/ktrace_chain.py --files 64 --chain 8 > $OUT/ktrace_chain8_64.txt 2>&1; echo "ktrace chain exit $?"; grep -A7 "one chained" $OUT/ktrace_chain8_64.txt | cut -c1-300 | grep -v "t\[us\]"
