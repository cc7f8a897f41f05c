#!/usr/bin/env python
"""One mixture alone (BASELINE config 2's shape: 10 s, K = 1024, 100 iterations): wall time per file for the tuning variants of the
latency path.  usage (GPU box): python scripts/single_file.py [--profile]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                            # noqa: E402
from gcc_nmf_amd import _hip                            # noqa: E402
from gcc_nmf_amd.engine import GCCNMFEngine             # noqa: E402
from gcc_nmf_amd.synthetic import synthetic_batch       # noqa: E402

lib = _hip.lib()
xs = synthetic_batch(0, 1)
K = int(os.environ.get('K', '1024'))
HOP = int(os.environ.get('HOP', '256'))             # K=128 HOP=128: the reference driver's own call (runGCCNMF.py:41,60)
e = GCCNMFEngine(160000, dictionarySize=K, numIterations=100, batch=1, hopSize=HOP)
e.upload(xs[0])


def run(label, reps=3):
    e.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    e.klnmf()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.klnmf()
    torch.cuda.synchronize()
    nmf = (time.perf_counter() - t0) / reps * 1e3
    print('%-40s %7.2f ms per file  (KL-NMF alone %7.2f ms = %.1f us per iteration), tdoa %s' % (label, ms, nmf, nmf * 10, e.get_tdoa_indexes()[0].tolist()), flush=True)
    return e.y.clone()


for kv in filter(None, os.environ.get('TUNE', '').split(',')):       # e.g. TUNE=5=2,6=4
    key, val = [int(v) for v in kv.split('=')]
    assert lib.gccnmf_set_tuning(key, val) == 0, kv
if '--profile' in sys.argv:
    run('default', 2)
    sys.exit(0)
lib.gccnmf_set_tuning(10, 1)
y_new = run('direct path (round 4)')
lib.gccnmf_set_tuning(10, 0)
lib.gccnmf_set_tuning(4, 0)
y_old = run('register-staged (round 1)')
print('    waveform rms direct vs register-staged: %.2e' % float(((y_new - y_old) ** 2).mean().sqrt()))
lib.gccnmf_set_tuning(4, 1)
for wh in (2, 3, 4):
    for rht in (4,):
        lib.gccnmf_set_tuning(5, wh)
        lib.gccnmf_set_tuning(6, rht)
        y = run('ring, W.H splits %d, R.H^T splits %d' % (wh, rht))
        print('    waveform rms vs register-staged: %.2e' % float(((y - y_old) ** 2).mean().sqrt()))
