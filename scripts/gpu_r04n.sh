#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04n
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s --tb=short -p no:cacheprovider -k "alone_on_the_direct" 2>&1 | grep -E "alone|passed|failed|Error" | tail -16
bash scripts/gpu_pmc_single.sh r04n 2>&1 | tail -14
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r04n/pmc_$c -o single -- python scripts/single_file.py --profile > /dev/null 2>&1
  f=$(find gpurun_out/r04n/pmc_$c -name "*counter_collection.csv" | head -1)
  python - "$f" $c <<'PY'
import csv, collections, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == sys.argv[2] and ('gccnmf_direct' in r['Kernel_Name'] or 'onepass' in r['Kernel_Name']):
        agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k, v in agg.items():
    print('%s %s mean %.1f KB over %d launches' % (sys.argv[2], k, sum(v) / len(v), len(v)))
PY
  rm -rf gpurun_out/r04n/pmc_$c
done
