#!/usr/bin/env python
"""Is the KL-NMF loop of ONE mixture launch-bound?  gccnmf_klnmf enqueues 5 launches per iteration from C; here the same call captured
once into a HIP graph (torch.cuda.CUDAGraph) and replayed.   python scripts/graph_klnmf.py [K hop]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import torch                # noqa: E402
from gcc_nmf_amd.engine import GCCNMFEngine          # noqa: E402
from gcc_nmf_amd.synthetic import synthetic_batch    # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
hop = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x = synthetic_batch(0, 1)[0]
e = GCCNMFEngine(160000, dictionarySize=K, numIterations=100, batch=1, hopSize=hop)
e.upload(x)
e.stft()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


direct = timed(e.klnmf)
W_direct = e.W.clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    e.klnmf()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    e.klnmf()
torch.cuda.synchronize()
capture_ms = (time.perf_counter() - t0) * 1e3
graph = timed(g.replay)
same = bool(torch.equal(W_direct, e.W))
print('K = %d hop %d (N = %d): klnmf direct launches %.3f ms, graph replay %.3f ms (capture + instantiate %.1f ms), identical W: %s' % (
    K, hop, e.g.N, direct, graph, capture_ms, same))
