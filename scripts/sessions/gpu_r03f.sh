#!/bin/bash
TAG=${1:-r03f}
OUT=gpurun_out/$TAG; export OUT
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_distributed.py -q -m gpu --tb=short -p no:cacheprovider -x -k "not bench_launches and not config4" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
FILES="16 20 26 40 45 52 64 77 80 96" bash scripts/files_sweep.sh > $OUT/files_sweep.txt 2>&1; cat $OUT/files_sweep.txt
python - <<'PY' > $OUT/ts_groups.jsonl 2> $OUT/ts_groups.err
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from gcc_nmf_amd import _hip
from gcc_nmf_amd.distributed import HipTimeShard, train_shared_dictionary
from gcc_nmf_amd.synthetic import synthetic_mixture
lib = _hip.lib()
for seconds in (160.0, 80.0, 320.0):
    x = synthetic_mixture(7, numSamples=int(seconds * 16000))
    for split in (1, 0):
        for groups in (1, 2, 3):
            lib.gccnmf_set_tuning(8, groups); lib.gccnmf_set_tuning(9, split)
            local = HipTimeShard(x, 0, 1, dictionarySize=1024)
            local.stft()
            train_shared_dictionary(local.nmf, 100); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                train_shared_dictionary(local.nmf, 100)
            torch.cuda.synchronize()
            print(json.dumps({'seconds': seconds, 'tail_split': split, 'groups': groups, 'nmf100_ms': 1e3 * (time.perf_counter() - t0) / 3}), flush=True)
            del local
PY
cat $OUT/ts_groups.jsonl
