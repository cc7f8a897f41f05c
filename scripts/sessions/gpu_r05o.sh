#!/bin/bash
# Gaps between the dependent launches of a KL-NMF iteration on ONE stream (kernel trace: start of launch n+1 minus end of launch n)
OUT=gpurun_out/r05o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python bench.py --gpus 1 --steps 1 --warmup 0 --skip-extras --skip-roofline --skip-config-lines --skip-cpu-baseline --no-live-traffic --nmf-groups 1 > $OUT/bench.json 2> $OUT/bench.err
echo "exit $?"
python - <<'PY'
import csv, glob, json, statistics as st
f = glob.glob('gpurun_out/r05o/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
k = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
gaps = {}
for (n0, s0, e0), (n1, s1, e1) in zip(k, k[1:]):
    if 'gccnmf_gemm_dma_kernel' in n0 and 'gccnmf_gemm_dma_kernel' in n1:
        key = n0.split('(')[0][-30:] + ' -> ' + n1.split('(')[0][-30:]
        gaps.setdefault(key, []).append((s1 - e0) / 1e3)
out = {}
for key, v in gaps.items():
    if len(v) >= 20:
        out[key] = {'n': len(v), 'median_us': st.median(v), 'p10_us': sorted(v)[len(v) // 10], 'p90_us': sorted(v)[9 * len(v) // 10]}
        print('%-70s n %4d  median %.1f us  p10 %.1f  p90 %.1f' % (key, len(v), out[key]['median_us'], out[key]['p10_us'], out[key]['p90_us']))
dur = {}
for n, s, e in k:
    if 'gccnmf_gemm_dma_kernel' in n:
        dur.setdefault(n.split('(')[0][-30:], []).append((e - s) / 1e3)
for n, v in dur.items():
    print('%-40s n %4d  median duration %.1f us' % (n, len(v), st.median(v)))
json.dump({'gaps': out, 'durations_median_us': {n: st.median(v) for n, v in dur.items()}}, open('gpurun_out/r05o/launch_gaps.json', 'w'), indent=1)
PY
rm -rf $OUT/trace
