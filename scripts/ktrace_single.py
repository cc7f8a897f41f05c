#!/usr/bin/env python
"""Per-workgroup timeline of the single-file (latency path) KL-NMF launches: when each workgroup started, entered its main loop,
left it and finished.  python scripts/ktrace_single.py [--stage 1..4]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--K', type=int, default=1024)
    ap.add_argument('--stages', default='1,2,3,4')
    ap.add_argument('--ablate', type=int, default=0)
    a = ap.parse_args()
    import torch
    from gcc_nmf_amd import _hip
    from gcc_nmf_amd.engine import Geometry, _ptr, _stream
    lib = _hip.lib()
    if not hasattr(lib, 'gccnmf_debug_set_trace'):
        sys.exit('needs the experiment build: make -C gcc_nmf_amd/csrc EXPERIMENTS=1; GCCNMF_HIP_LIB=gcc_nmf_amd/libgccnmf_hip_exp.so')
    for kv in filter(None, os.environ.get('TUNE', '').split(',')):       # e.g. TUNE=8=2,9=3
        key, val = [int(v) for v in kv.split('=')]
        assert lib.gccnmf_set_tuning(key, val) == 0, kv
    F, T, K, B = 513, 622, a.K, 1
    g = Geometry(F, T, K)
    N = g.N
    dev = 'cuda'
    gen = torch.Generator(device=dev).manual_seed(0)
    V = torch.zeros((B, g.Fp, g.Np), device=dev)
    W = torch.zeros((B, g.Fp, g.Kp), device=dev)
    H = torch.zeros((B, g.Kp, g.Np), device=dev)
    V[:, :F, :N] = torch.rand((B, F, N), device=dev, generator=gen) + 0.01
    W[:, :F, :K] = torch.rand((B, F, K), device=dev, generator=gen) + 0.01
    H[:, :K, :N] = torch.rand((B, K, N), device=dev, generator=gen) + 0.01
    ws = torch.zeros(lib.gccnmf_klnmf_workspace_floats(F, N, K, B), device=dev)

    def stage(s):
        _hip.check(lib.gccnmf_klnmf_stage(_ptr(V), _ptr(W), _ptr(H), _ptr(ws), F, N, K, B, 0.0, 1e-16, 0, s, _stream()), 'stage')

    stage(0)
    for it in range(3):
        for s in (1, 2, 3, 4, 5):
            stage(s)
    torch.cuda.synchronize()
    nblk = 4096
    lib.gccnmf_set_tuning(1, a.ablate)
    for st in [int(x) for x in a.stages.split(',')]:
        trace = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
        for s in (1, 2, 3, 4, 5):          # a warm iteration, then the traced stage in sequence
            if s == st:
                lib.gccnmf_debug_set_trace(_ptr(trace), nblk)
            stage(s)
            if s == st:
                lib.gccnmf_debug_set_trace(None, 0)
        torch.cuda.synchronize()
        t = trace.cpu().numpy()
        t = t[t[:, 0] > 0]
        t0 = t[:, 0].min()
        us = (t[:, :4] - t0) / 100.0
        cu = t[:, 4] >> 8
        per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
        print('stage %d: %d workgroups on %d CUs (max %d per CU); launch span %.1f us' % (st, len(t), len(per_cu), per_cu.max(), us[:, 3].max()))
        cyc = (t[:, 6] - t[:, 5]).astype(float)
        print('   main loop: median %.0f shader-clock ticks = %.3f ticks per 10 ns (s_memtime)' % (np.median(cyc), np.median(cyc / np.maximum(t[:, 2] - t[:, 1], 1))))
        for name, v in [('start          ', us[:, 0]), ('prologue  t1-t0', us[:, 1] - us[:, 0]), ('main loop t2-t1', us[:, 2] - us[:, 1]),
                        ('epilogue  t3-t2', us[:, 3] - us[:, 2]), ('end            ', us[:, 3])]:
            print('   %s  min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us' % ((name,) + tuple(np.percentile(v, [0, 10, 50, 90, 100]))))


if __name__ == '__main__':
    main()
