#!/usr/bin/env python
"""One mixture alone on the direct path: KL-NMF time per iteration and a checksum of the factors, for A/B runs of two builds of the library
(GCCNMF_HIP_LIB=...).  env: K (1024), HOP (256), FILES (1)"""
import hashlib
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                            # noqa: E402
from gcc_nmf_amd import _hip                            # noqa: E402
from gcc_nmf_amd.engine import GCCNMFEngine             # noqa: E402
from gcc_nmf_amd.synthetic import synthetic_batch       # noqa: E402

lib = _hip.lib()
K = int(os.environ.get('K', '1024'))
HOP = int(os.environ.get('HOP', '256'))
FILES = int(os.environ.get('FILES', '1'))
xs = synthetic_batch(0, FILES)
e = GCCNMFEngine(160000, dictionarySize=K, numIterations=100, batch=FILES, hopSize=HOP)
e.upload(xs if FILES > 1 else xs[0])
e.run()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    e.klnmf()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) * 1e3)
t0 = time.perf_counter()
for _ in range(3):
    e.run()
torch.cuda.synchronize()
whole = (time.perf_counter() - t0) / 3 * 1e3
h = hashlib.sha256(e.W.cpu().numpy().tobytes() + e.H.cpu().numpy().tobytes()).hexdigest()[:16]
print('%-28s K=%d hop=%d files=%d plan=%d: KL-NMF %.3f ms = %.2f us per iteration; whole run %.3f ms per batch; sha(W,H) %s' % (
    os.path.basename(os.environ.get('GCCNMF_HIP_LIB', 'libgccnmf_hip.so')), K, HOP, FILES, getattr(e, "plan", -1), best, best * 10, whole, h), flush=True)
