#!/bin/bash
# Round 6: the fused inverse STFT fetching its spectrograms 8 frames at a time (tuning key 26) -- its tests, stage times with and without, HBM bytes.
# usage: gpurun --timeout 1500 -- 'bash scripts/sessions/r06ae.sh [tag]'
TAG=${1:-r06ae}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests (stft / istft kernels, pipeline, named functions)"
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_named_functions.py -q -m gpu --tb=short -p no:cacheprovider --timeout 300 -k "stft or istft or pipeline or named or waveform or separate or mixture" > $OUT/pytest_istft.log 2>&1; echo "exit $? $(grep -E 'passed|failed' $OUT/pytest_istft.log | tail -1)"
echo "== stage times: key 26 = 0 | 1, twice"
for rep in 1 2; do for v in 0 1; do echo -n "26=$v: "; timeout 300 python scripts/stage_times.py 26=$v 2>&1 | tail -1; done; done | tee $OUT/stage_times_ab.txt
echo "== HBM bytes of the inverse STFT"
for v in 0 1; do
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$v -o p -- python scripts/stage_times.py 26=$v > $OUT/pmc_$v.log 2>&1
  python - "$OUT/pmc_$v" $v <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], '**', '*counter_collection.csv'), recursive=True)
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f[0])) if 'istft_fused' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
print('key 26 = %s: istft_fused_kernel FETCH_SIZE %.1f MB raw, %.1f MB doubled (gfx950), %d launches' % (sys.argv[2], sum(v) / len(v) * 1024 / 1e6, 2 * sum(v) / len(v) * 1024 / 1e6, len(v)))
PY
  rm -rf $OUT/pmc_$v
done | tee $OUT/istft_fetch.txt
echo "== K = 128 batch step"
for v in 0 1; do timeout 600 python bench.py --dictionary-size 128 --steps 5 --warmup 2 --skip-extras --skip-roofline --skip-cpu-baseline --tune 26=$v 2>/dev/null | cut -c1-200; done | tee $OUT/bench_K128_ab.txt
