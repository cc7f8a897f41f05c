#!/bin/bash
# SQ-side PMC pass over the latency path (one mixture alone): MFMA busy of the ring GEMM launches.
# usage: gpurun --timeout 600 -- 'bash scripts/gpu_pmc_single.sh [tag]'   (PMC only with --kernel-trace, never with other trace domains)
TAG=${1:-r02}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc_sq_single -o single -- python scripts/single_file.py --profile > $OUT/pmc_sq_single.log 2>&1
echo "pmc exit $?"
find $OUT/pmc_sq_single -name "*kernel_trace*" -delete
python - "$OUT" <<'PY'
import csv, collections, glob, json, os, sys
out = sys.argv[1]
files = glob.glob(os.path.join(out, 'pmc_sq_single', '**', '*counter_collection.csv'), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(files[0])):
    if 'gccnmf_gemm_ring' in r['Kernel_Name'] or 'gccnmf_direct' in r['Kernel_Name'] or 'onepass' in r['Kernel_Name'] or 'div_partials' in r['Kernel_Name']:
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} | {'launches': len(next(iter(d.values())))} for k, d in agg.items()}
json.dump(res, open(os.path.join(out, 'pmc_sq_single.json'), 'w'), indent=1)
for k, d in res.items():
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs; GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (488 k 'cycles' for a 21 us
    # launch), so launch cycles = GRBM / 8 and the chip offers 1024 SIMDs x that
    simd_cycles = 1024.0 * d.get('GRBM_GUI_ACTIVE', 0) / 8.0
    print(k[:90])
    print('   launches %d  GRBM_GUI_ACTIVE %.0f  MFMA busy %.4g = %.1f %% of 1024 SIMDs x launch cycles;  wave cycles: wait_any %.1f%%  active_inst %.1f%%' % (
        d['launches'], d.get('GRBM_GUI_ACTIVE', 0), d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0),
        100 * d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(simd_cycles, 1), 100 * d.get('SQ_WAIT_ANY', 0) / wc, 100 * d.get('SQ_ACTIVE_INST_ANY', 0) / wc))
PY
