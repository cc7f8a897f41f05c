#!/bin/bash
# SQ-side PMC pass over the four KL-NMF GEMM launches (kbench shapes = bench.py's): where do the wave cycles go?
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_pmc_sq.sh [tag]'   (PMC only with --kernel-trace, never with other trace domains)
TAG=${1:-r01}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc_sq -o kbench -- python scripts/kbench.py --reps 3 > $OUT/pmc_sq.log 2>&1
echo "pmc exit $?"
find $OUT/pmc_sq -name "*kernel_trace*" -delete
python - "$OUT" <<'PY'
import csv, collections, glob, json, os, sys
out = sys.argv[1]
files = glob.glob(os.path.join(out, 'pmc_sq', '*counter_collection.csv'))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(files[0])):
    if 'gccnmf_gemm' in r['Kernel_Name']:
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} | {'launches': len(next(iter(d.values())))} for k, d in agg.items()}
json.dump(res, open(os.path.join(out, 'pmc_sq.json'), 'w'), indent=1)
for k, d in res.items():
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1
    print(k)
    print('   ' + '  '.join('%s=%.4g' % (c, v) for c, v in sorted(d.items())))
    print('   of wave cycles: wait_any %.1f%%  wait_inst_any %.1f%%  active_inst %.1f%%  wait_inst_lds %.1f%%' % tuple(
        100 * d.get(c, 0) / wc for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS')))
PY
