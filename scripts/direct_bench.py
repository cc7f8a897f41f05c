#!/usr/bin/env python
"""Latency path (one mixture alone): per-stage kernel times of one KL-NMF iteration for the direct kernels (csrc/direct.hip) against
the round-3 split-K path, per GEMM tile, and the 100-iteration wall time.  usage (GPU box): python scripts/direct_bench.py [K hop [batch]]"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                            # noqa: E402
from gcc_nmf_amd import _hip                            # noqa: E402
from gcc_nmf_amd.engine import GCCNMFEngine, _ptr, _stream             # noqa: E402
from gcc_nmf_amd.synthetic import synthetic_batch       # noqa: E402

lib = _hip.lib()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hop = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
xs = synthetic_batch(0, B)
lib.gccnmf_set_tuning(12, max(B, 1))
e = GCCNMFEngine(160000, dictionarySize=K, numIterations=100, batch=B, hopSize=hop)
e.upload(xs if B > 1 else xs[0])
e.stft()
g = e.g
TILES = ['auto', '1x1', '1x2', '1x5', '2x2', '2x4', '2x5', '4x4', '4x5']


def stage(s):
    _hip.check(lib.gccnmf_klnmf_stage(_ptr(e.V), _ptr(e.W), _ptr(e.H), _ptr(e.ws_nmf), g.F, g.N, g.K, e.batch, e.alpha, e.eps,
                                      e.klnmf_flags, s, _stream()), 'stage')


def reset():
    e.W.copy_(e.W0.unsqueeze(0).expand_as(e.W))
    e.H.copy_(e.H0.unsqueeze(0).expand_as(e.H))
    stage(0)
    for s in range(1, 6):
        stage(s)


def time_stage(s, reps=40):
    reset()
    for q in range(1, s):          # a consistent state in front of stage s
        stage(q)
    stage(s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        stage(s)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def time_iteration(reps=30):
    reset()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        for s in range(1, 6):
            stage(s)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def time_klnmf(reps=3):
    e.klnmf()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.klnmf()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {'K': K, 'hop': hop, 'batch': B, 'F': g.F, 'N': g.N}
for direct in (0, 1):
    lib.gccnmf_set_tuning(10, direct)
    lib.gccnmf_set_tuning(11, 0)
    name = 'direct' if direct else 'split-K (round 3)'
    st = {('K1', 'K2', 'K3', 'K4a', 'K4b')[s - 1]: round(time_stage(s), 2) for s in range(1, 6)}
    it = time_iteration()
    ms = time_klnmf()
    wf = e.W.clone()
    out[name] = {'stage_us_back_to_back': st, 'iteration_us': round(it, 2), 'klnmf_100_ms': round(ms, 3)}
    print('%-20s stages (us, same launch repeated) %s | iteration %.1f us | 100 iterations %.3f ms' % (name, st, it, ms), flush=True)
    if direct:
        print('   W rel diff direct vs split-K after 100 iterations: %.2e' % float(((wf - w_old).norm() / w_old.norm()).item()))
    else:
        w_old = wf
# every tile on every GEMM stage of the direct path
lib.gccnmf_set_tuning(10, 1)
sweep = {}
for t in range(1, 9):
    lib.gccnmf_set_tuning(11, t)
    sweep[TILES[t]] = {('K1', 'K2', 'K3', 'K4a')[s - 1]: round(time_stage(s, 20), 2) for s in range(1, 5)}
    print('tile %-4s %s' % (TILES[t], sweep[TILES[t]]), flush=True)
lib.gccnmf_set_tuning(11, 0)
out['tile_sweep_us'] = sweep
# register sets of the operand pipeline (tuning key 13), tile by the cost model
depth = {}
for dpt in (2, 3, 4):
    lib.gccnmf_set_tuning(13, dpt)
    depth[dpt] = {('K1', 'K2', 'K3', 'K4a')[s - 1]: round(time_stage(s, 20), 2) for s in range(1, 5)}
    depth[dpt]['iteration'] = round(time_iteration(), 2)
    print('pipeline depth %d: %s' % (dpt, depth[dpt]), flush=True)
lib.gccnmf_set_tuning(13, 0)
out['depth_sweep_us'] = depth
# the whole mixture through the engine
e.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    e.run()
torch.cuda.synchronize()
out['engine_run_ms_per_step'] = (time.perf_counter() - t0) / 3 * 1e3
print('engine.run: %.3f ms per step (%d file(s)), tdoa %s' % (out['engine_run_ms_per_step'], B, e.get_tdoa_indexes()[0].tolist()))
print(json.dumps(out))
