import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcc_nmf_amd import _hip
from gcc_nmf_amd.distributed import HipTimeShard, train_shared_dictionary
from gcc_nmf_amd.synthetic import synthetic_mixture
for seconds in (10.0, 20.0, 40.0, 160.0):
    x = synthetic_mixture(7, numSamples=int(seconds * 16000))
    local = HipTimeShard(x, 0, 1, dictionarySize=1024)
    local.stft()
    train_shared_dictionary(local.nmf, 100); torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_shared_dictionary(local.nmf, 100)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({'seconds': seconds, 'frames': local.T_total, 'host_enqueue_ms': 1e3*(t1-t0), 'total_ms': 1e3*(t2-t0), 'blocks': [list(b) for b in local.nmf.blocks]}), flush=True)
