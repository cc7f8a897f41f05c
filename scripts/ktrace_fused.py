#!/usr/bin/env python
"""Per-workgroup timeline of the fused short-dictionary launches (csrc/direct.hip): K1 main loop | tail bin | divide | K2 + finishing.
python scripts/ktrace_fused.py [--K 128] [--files 64] [--stage 1]       (GCCNMF_TUNE=16=1 must select the fused path)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--K', type=int, default=128)
    ap.add_argument('--files', type=int, default=64)
    ap.add_argument('--stage', type=int, default=1)
    ap.add_argument('--slab', action='store_true', help='slab-style timeline slots (start | loop done | end) for the stage')
    a = ap.parse_args()
    import torch
    from gcc_nmf_amd import _hip
    from gcc_nmf_amd.engine import Geometry, _ptr, _stream
    lib = _hip.lib()
    if not hasattr(lib, 'gccnmf_debug_set_trace'):
        sys.exit('needs the experiment build: make -C gcc_nmf_amd/csrc EXPERIMENTS=1; GCCNMF_HIP_LIB=gcc_nmf_amd/libgccnmf_hip_exp.so')
    F, T, K, B = 513, 622, a.K, a.files
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
    nblk = 8192
    trace = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
    for s in (1, 2, 3, 4, 5):
        if s == a.stage:
            lib.gccnmf_debug_set_trace(_ptr(trace), nblk)
        stage(s)
        if s == a.stage:
            lib.gccnmf_debug_set_trace(None, 0)
    torch.cuda.synchronize()
    t = trace.cpu().numpy()
    t = t[t[:, 0] > 0]
    if not len(t):
        print('no trace rows: the fused path was not taken')
        return
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    cu = t[:, 4] >> 8
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    print('stage %d: %d workgroups on %d CUs (max %d per CU); launch span %.1f us' % (a.stage, len(t), len(per_cu), per_cu.max(), us[:, 3].max()))
    if a.stage == 3 or a.slab:                                            # slab kernel: start | tile loop done | end
        for name, v in [('start            ', us[:, 0]), ('tile loop   t2-t0', us[:, 2] - us[:, 0]), ('finish      t3-t2', us[:, 3] - us[:, 2]), ('end              ', us[:, 3])]:
            print('   %s  min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us' % ((name,) + tuple(np.percentile(v, [0, 10, 50, 90, 100]))))
        xcd = t[:, 4] >> 16
        loop = us[:, 2] - us[:, 0]
        print('   tile loop by XCD  : ' + '  '.join('%d: %.0f' % (x, np.median(loop[xcd == x])) for x in np.unique(xcd)))
        fast = loop < np.median(loop)
        for nm, sel in (('faster half', fast), ('slower half', ~fast)):
            a1, a5, a6, a7 = us[sel, 1], us[sel, 5], us[sel, 6], us[sel, 7]
            print('   tile 8, %s: first product %.2f  divide+second product %.2f  store+barrier %.2f  whole %.2f us (medians, wave 0)' % (
                nm, np.median(a5 - a1), np.median(a6 - a5), np.median(a7 - a6), np.median(a7 - a1)))
        cu = t[:, 4] >> 8
        pair = {}
        for i, cid in enumerate(cu):
            pair.setdefault(int(cid), []).append(loop[i])
        d = np.array([abs(v[0] - v[1]) for v in pair.values() if len(v) == 2])
        print('   CUs with two workgroups: %d of %d; |difference| of the pair median %.1f us' % (len(d), len(pair), np.median(d) if len(d) else -1))
        return
    for name, v in [('start               ', us[:, 0]), ('main loop      t1-t0', us[:, 1] - us[:, 0]), ('tail bin       t5-t1', us[:, 5] - us[:, 1]),
                    ('divide         t2-t5', us[:, 2] - us[:, 5]), ('second product t3-t2', us[:, 3] - us[:, 2]), ('whole          t3-t0', us[:, 3] - us[:, 0]),
                    ('end                 ', us[:, 3])]:
        print('   %s  min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us' % ((name,) + tuple(np.percentile(v, [0, 10, 50, 90, 100]))))


if __name__ == '__main__':
    main()
