#!/bin/bash
# Round 6: what bounds the two FFT kernels of a step (review item 5c asks for the limiting resource from counters): HBM bytes and SQ counters of
# istft_fused_kernel and stft_stereo_kernel at the bench shape.   usage: gpurun --timeout 1200 -- 'bash scripts/sessions/r06ad.sh [tag]'
TAG=${1:-r06ad}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --gpus 1 --steps 1 --warmup 0 --skip-extras --skip-roofline"
pass() {  # name, counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- $CMD > $OUT/pmc_$name.log 2>&1
  echo "pmc $name exit $?"
  find $OUT/pmc_$name -name "*kernel_trace*" -delete
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
python - "$OUT" <<'PY'
import csv, collections, glob, json, os, sys
out = sys.argv[1]
res = collections.defaultdict(dict)
for name in ('fetch', 'write', 'sq1', 'sq2'):
    files = glob.glob(os.path.join(out, 'pmc_' + name, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k = r['Kernel_Name']
        if 'stft' in k:
            agg[k.split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        for c, v in d.items():
            res[k][c] = sum(v) / len(v)
        res[k]['launches'] = len(next(iter(d.values())))
json.dump(res, open(os.path.join(out, 'pmc_fft.json'), 'w'), indent=1)
for k, d in res.items():
    print(k)
    if 'FETCH_SIZE' in d:
        print('   HBM: fetch %.1f MB (raw KB x 2, gfx950) + write %.1f MB per launch' % (2 * d['FETCH_SIZE'] * 1024 / 1e6, d.get('WRITE_SIZE', 0) * 1024 / 1e6))
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1
    busy = d.get('SQ_BUSY_CYCLES', 0) or 1
    print('   of wave cycles: wait_any %.1f%%  wait_inst_any %.1f%%  active VALU %.1f%%  active LDS %.1f%%  active VMEM %.1f%%  wait_inst_LDS %.1f%%' % tuple(
        100 * d.get(c, 0) / wc for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_WAIT_INST_LDS')))
    print('   insts: VALU %.3g  LDS %.3g  VMEM rd %.3g wr %.3g; LDS bank conflict cycles / LDS active cycles %.3f; waves x cycles / busy cycles (occupancy) %.2f' % (
        d.get('SQ_INSTS_VALU', 0), d.get('SQ_INSTS_LDS', 0), d.get('SQ_INSTS_VMEM_RD', 0), d.get('SQ_INSTS_VMEM_WR', 0),
        d.get('SQ_LDS_BANK_CONFLICT', 0) / (d.get('SQ_LDS_IDX_ACTIVE', 0) or 1), wc / busy))
PY
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2
