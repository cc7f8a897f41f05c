#!/usr/bin/env python
"""Per-workgroup timeline of ONE KL-NMF iteration at the headline shape, as four launches (tuning key 21 = 0) and as one chained launch
(21 = 4), from gccnmf_debug_set_trace (experiment build): per stage when its workgroups started, how long they waited for a producer,
their main loops and epilogues; per CU how many workgroups were in a main loop over time (the matrix pipe is 0.93 busy with two, 0.80
with one, LABBOOK R5.1).

    GCCNMF_HIP_LIB=gcc_nmf_amd/libgccnmf_hip_exp.so python scripts/ktrace_chain.py [--files 64] [--chain 4]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--files', type=int, default=64)
    ap.add_argument('--K', type=int, default=1024)
    ap.add_argument('--chain', type=int, default=4)
    ap.add_argument('--cus', type=int, default=4, help='per-CU timelines printed')
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
    tn = g.Np // 64
    per_list = [B // 8 * tn, B // 8 * 2 * tn, B // 8 * tn, B // 8 * (K // 64)]          # items per XCD list of K1, K2, K3, K4
    names = ['K1', 'K2', 'K3', 'K4']
    nblk = 8 * sum(per_list) + 64

    def klnmf(n):
        _hip.check(lib.gccnmf_klnmf(_ptr(V), _ptr(W), _ptr(H), _ptr(ws), F, N, K, B, n, 0.0, 1e-16, 0, _stream()), 'klnmf')

    def stage(s):
        _hip.check(lib.gccnmf_klnmf_stage(_ptr(V), _ptr(W), _ptr(H), _ptr(ws), F, N, K, B, 0.0, 1e-16, 0, s, _stream()), 'stage')

    def report(title, rows, stage_of):
        ok = rows[:, 0] > 0
        rows, stage_of = rows[ok], stage_of[ok]
        t0 = rows[:, 0].min()
        us = (rows[:, :4] - t0) / 100.0
        cu = rows[:, 4] >> 8
        print('== %s: %d workgroups on %d CUs, span %.1f us' % (title, len(rows), len(np.unique(cu)), us[:, 3].max()))
        for s in range(4):
            m = stage_of == s
            if not m.any():
                continue
            u = us[m]
            wait = u[:, 1] - u[:, 0]
            print('  %s: %4d items  first start %6.1f  last start %6.1f  last end %6.1f us | waited for a producer: median %.1f p90 %.1f max %.1f us, sum %.0f us '
                  '| main loop median %.1f | epilogue median %.1f' % (names[s], m.sum(), u[:, 0].min(), u[:, 0].max(), u[:, 3].max(), np.median(wait), np.percentile(wait, 90),
                                                                  wait.max(), wait.sum(), np.median(u[:, 2] - u[:, 1]), np.median(u[:, 3] - u[:, 2])))
        # occupancy: per 5 us bin, workgroups resident (start..end) and in a main loop, averaged over CUs
        end = us[:, 3].max()
        edges = np.arange(0, end, 5.0)
        ncu = len(np.unique(cu))
        res = np.array([np.sum((us[:, 0] <= x) & (us[:, 3] > x)) for x in edges]) / float(ncu)
        loop = np.array([np.sum((us[:, 1] <= x) & (us[:, 2] > x)) for x in edges]) / float(ncu)
        waiting = np.array([np.sum((us[:, 0] <= x) & (us[:, 1] > x)) for x in edges]) / float(ncu)
        print('  mean workgroups per CU: resident %.3f, in a main loop %.3f, waiting %.3f (over the span)' % (res.mean(), loop.mean(), waiting.mean()))
        step = max(1, len(edges) // 60)
        print('  t[us]:resident/in-loop per CU  ' + ' '.join('%.0f:%.2f/%.2f' % (edges[i], res[i], loop[i]) for i in range(0, len(edges), step)))
        by_cu = {}
        for i in np.argsort(us[:, 0]):
            by_cu.setdefault(int(cu[i]), []).append(i)
        for c in sorted(by_cu)[:a.cus]:
            print('  CU %05x: ' % c + ' '.join('%s[%.0f +%.0f %.0f %.0f]' % (names[stage_of[i]], us[i, 0], us[i, 1] - us[i, 0], us[i, 2], us[i, 3]) for i in by_cu[c]))
        return us[:, 3].max()

    # ---- four launches, each traced into its own buffer (absolute timestamps share one axis)
    assert lib.gccnmf_set_tuning(21, 0) == 0
    klnmf(2)
    torch.cuda.synchronize()
    bufs = [torch.zeros((8 * per_list[s] + 64, 8), dtype=torch.int64, device=dev) for s in range(4)]
    stage(1); stage(2); stage(3); stage(4)
    torch.cuda.synchronize()
    for s in range(4):
        lib.gccnmf_debug_set_trace(_ptr(bufs[s]), bufs[s].shape[0])
        stage(s + 1)
    torch.cuda.synchronize()
    lib.gccnmf_debug_set_trace(None, 0)
    rows = np.concatenate([b.cpu().numpy() for b in bufs])
    stage_of = np.concatenate([np.full(b.shape[0], s) for s, b in enumerate(bufs)])
    span4 = report('four launches back to back on one stream', rows, stage_of)

    # ---- one chained launch
    assert lib.gccnmf_set_tuning(21, a.chain) == 0
    klnmf(2)
    torch.cuda.synchronize()
    trace = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
    lib.gccnmf_debug_set_trace(_ptr(trace), nblk)
    klnmf(8)                              # per-iteration launches overwrite the rows: the LAST iteration stays; the whole-call launch traces its third-from-last
    torch.cuda.synchronize()
    lib.gccnmf_debug_set_trace(None, 0)
    lib.gccnmf_set_tuning(21, 0)
    rows = trace.cpu().numpy()
    first = np.cumsum([0] + per_list)
    stage_of = np.zeros(nblk, dtype=np.int64)
    for s in range(4):
        stage_of[8 * first[s]:8 * first[s + 1]] = s
    spanc = report('one chained launch (key 21 = %d)%s' % (a.chain, '' if a.chain == 4 else ' followed by the remaining launches'), rows, stage_of)
    print('span: four launches %.1f us, chained %.1f us' % (span4, spanc))


if __name__ == '__main__':
    main()
