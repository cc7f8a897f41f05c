#!/usr/bin/env python
"""performKLNMF on ONE big matrix (the dictionary pre-training shape, gccNMF/realtime/gccNMFPretraining.py:79-80): the reference-named
function routes N > 4096 columns through in-place column blocks on the batched throughput kernels (gccnmf_klnmf_shared_run, ld > 0).
Prints one JSON line per N: device-resident rate of the training itself (HIP events) and the host-array call's wall time.
usage (GPU box): python scripts/big_matrix.py [K iterations N...]"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                                    # noqa: E402
from gcc_nmf_amd import gccNMFFunctions as G                    # noqa: E402
from gcc_nmf_amd.distributed import HipSharedColumns            # noqa: E402
from gcc_nmf_amd.engine import Geometry, padded                 # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
Ns = [int(v) for v in sys.argv[3:]] or [20000, 80000]
F = 513
for N in Ns:
    rng = np.random.RandomState(N)
    V = (np.abs(rng.standard_normal((F, N))) + 0.01).astype(np.float32)
    g = Geometry(F, 1, K)
    ld = -(-N // 64) * 64
    Vd = padded(V, (g.Fp, ld), 'cuda')
    W0 = (rng.rand(F, K) + 1e-16).astype(np.float32)
    H0 = (rng.rand(K, N) + 1e-16).astype(np.float32)
    Wd, Hd = padded(W0, (g.Fp, g.Kp), 'cuda'), padded(H0, (g.Kp, ld), 'cuda')
    run = HipSharedColumns(Vd, Hd, Wd, F, N, K)
    run.run(2, collective=False)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    run.run(iters, collective=False)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    flop = 4 * 2.0 * F * K * N * iters
    t0 = time.perf_counter()
    W, H = G.performKLNMF(V, K, iters, 0)
    wall = time.perf_counter() - t0
    print(json.dumps({'what': 'performKLNMF(V (%d, %d), %d, %d, 0): column blocks of one matrix on the batched throughput kernels' % (F, N, K, iters),
                      'N': N, 'K': K, 'iterations': iters, 'blocks': [list(x) for x in run.blocks],
                      'device_ms': ms, 'ms_per_iteration': ms / iters, 'tflops': flop / ms / 1e9, 'frac_of_f32_mfma_peak': flop / ms / 1e9 / 157.3,
                      'host_call_s': wall, 'host_call_note': 'includes the MT19937 draws of W0, H0 (F*K + K*N doubles, the reference\'s own initialisation) and both PCIe copies',
                      'W_unit_norm': bool(np.allclose(np.linalg.norm(W, axis=0), 1.0, atol=1e-4))}), flush=True)
