#!/bin/bash
# The whole multi-GPU picture with one command, for whoever has an 8-GPU MI355X node (these build sessions lease ONE GPU):
#     bash scripts/scale.sh [OUTDIR]        -> OUTDIR/scale.jsonl, one JSON line per (mode, N), each with n_gpus, ranks_seen, value, collective
#                                           -> OUTDIR/SCALE_<mode>_N<n>.json, the complete bench.py line of that run (the shape of the driver's
#                                              BENCH / SCALE records: metric, value, unit, n_gpus, steps, warmup, ms_per_step, scaling, config, ...)
#                                           -> OUTDIR/SCALE_<mode>.json, {"mode", "runs": [that (mode, N) summary for N = 1, 2, 4, 8]}
# Modes (with the scaling to expect):
#   separate           independent files per rank, no data-path collective: near-linear (weak scaling; the driver's SCALE run)
#   shared-dictionary  BASELINE config 4: one all-reduce of F*K+K floats (2.2 MB at K = 1024) per KL-NMF iteration over the library's own
#                      RCCL communicator, against ~2.6 ms of kernels per iteration: near-linear (weak scaling)
#   time-sharded       ONE 160 s mixture cut into frame windows: strong scaling with a floor -- each of 8 ranks still runs 100 dependent
#                      iterations of short launches (14.5 ms) plus 100 all-reduces, against 83 ms on one GPU: expect ~4x at 8, not 8x
# bench.py launches its own ranks (torch.distributed.run on 127.0.0.1); GCCNMF_COLLECTIVE=torch routes the all-reduce through
# torch.distributed instead of the library's communicator if RCCL set-up through the library misbehaves on the node.
OUT=${1:-gpurun_out/scale}
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
: > $OUT/scale.jsonl
for mode in separate shared-dictionary time-sharded; do
  for n in 1 2 4 8; do
    [ "$n" -gt "$NGPU" ] && continue
    line=$(timeout 1200 python bench.py --gpus $n --mode $mode --steps 3 --warmup 1 --skip-extras 2> $OUT/${mode}_${n}.err | grep '^{' | tail -1)
    if [ -z "$line" ]; then echo "{\"mode\": \"$mode\", \"n_gpus\": $n, \"error\": \"see ${mode}_${n}.err\"}" | tee -a $OUT/scale.jsonl; continue; fi
    echo "$line" > $OUT/SCALE_${mode}_N${n}.json
    echo "$line" | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print(json.dumps({'mode': '$mode', 'n_gpus': d['n_gpus'], 'ranks_seen': d.get('ranks_seen'), 'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'],
                  'scaling': d['scaling'], 'collective': d.get('collective'), 'collective_backend': d.get('collective_backend')}))" | tee -a $OUT/scale.jsonl
  done
done
python - $OUT <<'PY'
import json, os, sys
out = sys.argv[1]
rows = [json.loads(l) for l in open(os.path.join(out, 'scale.jsonl')) if l.strip()]
for mode in sorted({r['mode'] for r in rows}):
    json.dump({'mode': mode, 'runs': [r for r in rows if r['mode'] == mode]}, open(os.path.join(out, 'SCALE_%s.json' % mode), 'w'), indent=1)
PY
