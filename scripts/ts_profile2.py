"""NMF phase of the time-sharded mode: stream groups (key 8) x GEMM tile policy (key 2) x mixture length."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gcc_nmf_amd import _hip
from gcc_nmf_amd.distributed import HipTimeShard, train_shared_dictionary
from gcc_nmf_amd.synthetic import synthetic_mixture
lib = _hip.lib()
for seconds in [float(v) for v in sys.argv[1:]] or [160.0]:
    x = synthetic_mixture(7, numSamples=int(seconds * 16000))
    for policy in (0, 1, 2):
        for groups in (1, 2, 3, 4):
            lib.gccnmf_set_tuning(8, groups); lib.gccnmf_set_tuning(2, policy)
            local = HipTimeShard(x, 0, 1, dictionarySize=1024)
            local.stft()
            train_shared_dictionary(local.nmf, 100); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                train_shared_dictionary(local.nmf, 100)
            torch.cuda.synchronize()
            print(json.dumps({'seconds': seconds, 'policy': policy, 'groups': groups, 'nmf100_ms': 1e3 * (time.perf_counter() - t0) / 3, 'blocks': [list(b) for b in local.nmf.blocks]}), flush=True)
            del local
lib.gccnmf_set_tuning(2, 0)
