#!/usr/bin/env python
"""Timeline of one iteration of the short-dictionary chained call (K <= 128; lab build: gccnmf_debug_set_trace): per stage (column tiles |
bin slabs | W update) when its workgroups arrived, how long they waited for their file's producer stage, how long they ran; workgroups
resident per CU over time; a few per-CU timelines.

    GCCNMF_HIP_LIB=gcc_nmf_amd/libgccnmf_hip_exp.so python scripts/ktrace_short.py [--K 128] [--files 64]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--files', type=int, default=64)
    ap.add_argument('--K', type=int, default=128)
    ap.add_argument('--cus', type=int, default=4)
    a = ap.parse_args()
    import torch
    from gcc_nmf_amd import _hip
    from gcc_nmf_amd.engine import Geometry, _ptr, _stream
    lib = _hip.lib()
    if not hasattr(lib, 'gccnmf_debug_set_trace'):
        sys.exit('needs the experiment build')
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
    assert lib.gccnmf_set_tuning(21, 8) == 0 and lib.gccnmf_klnmf_plan(F, N, K, B, 0) & 8
    group = 32 if B * (g.Kp // 32) >= 256 else 16
    per = [g.Np // 64, (F - 1) // 64, g.Kp // group]
    longest = -(-B // 8)
    first = np.cumsum([0] + [longest * p for p in per])
    nblk = 8 * int(first[3]) + 64

    def klnmf(n):
        _hip.check(lib.gccnmf_klnmf(_ptr(V), _ptr(W), _ptr(H), _ptr(ws), F, N, K, B, n, 0.0, 1e-16, 0, _stream()), 'klnmf')
    klnmf(3)
    torch.cuda.synchronize()
    trace = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
    lib.gccnmf_debug_set_trace(_ptr(trace), nblk)
    klnmf(8)                                  # the launch traces its third-from-last iteration: steady state
    torch.cuda.synchronize()
    lib.gccnmf_debug_set_trace(None, 0)
    lib.gccnmf_set_tuning(21, 1)
    rows = trace.cpu().numpy()
    stage_of = np.zeros(nblk, dtype=np.int64)
    for s in range(3):
        stage_of[8 * first[s]:8 * first[s + 1]] = s
    ok = rows[:, 3] > 0
    rows, stage_of = rows[ok], stage_of[ok]
    t0 = rows[:, 7].min()
    arrive, start, end = (rows[:, 7] - t0) / 100.0, (rows[:, 0] - t0) / 100.0, (rows[:, 3] - t0) / 100.0
    cu = rows[:, 4] >> 8
    names = ['K1+K2 column tiles', 'K3+K4a bin slabs', 'W update']
    print('short-dictionary chain, K = %d, %d files: a steady-state iteration (6th of 8), %d workgroups on %d CUs, span %.1f us' % (K, B, len(rows), len(np.unique(cu)), end.max()))
    for s in range(3):
        m = stage_of == s
        print('  %-20s %5d items  arrive %6.1f .. %6.1f  end %6.1f us | waited: median %.1f p90 %.1f max %.1f, sum %.0f us | ran: median %.1f p90 %.1f us' % (
            names[s], m.sum(), arrive[m].min(), arrive[m].max(), end[m].max(), np.median(start[m] - arrive[m]), np.percentile(start[m] - arrive[m], 90),
            (start[m] - arrive[m]).max(), (start[m] - arrive[m]).sum(), np.median(end[m] - start[m]), np.percentile(end[m] - start[m], 90)))
    edges = np.arange(0, end.max(), 5.0)
    ncu = float(len(np.unique(cu)))
    res = np.array([np.sum((arrive <= x) & (end > x)) for x in edges]) / ncu
    run = np.array([np.sum((start <= x) & (end > x)) for x in edges]) / ncu
    per_stage = [np.array([np.sum((start <= x) & (end > x) & (stage_of == s)) for x in edges]) / ncu for s in range(3)]
    print('  mean workgroups per CU: resident %.3f, running %.3f (tiles %.3f, slabs %.3f, W update %.3f)' % (res.mean(), run.mean(), per_stage[0].mean(), per_stage[1].mean(), per_stage[2].mean()))
    step = max(1, len(edges) // 50)
    print('  t[us]: running per CU tiles/slabs/W   ' + ' '.join('%.0f:%.2f/%.2f/%.2f' % (edges[i], per_stage[0][i], per_stage[1][i], per_stage[2][i]) for i in range(0, len(edges), step)))
    by_cu = {}
    for i in np.argsort(arrive):
        by_cu.setdefault(int(cu[i]), []).append(i)
    tag = ['T', 'S', 'W']
    for c in sorted(by_cu)[:a.cus]:
        print('  CU %05x: ' % c + ' '.join('%s[%.0f +%.0f %.0f]' % (tag[stage_of[i]], arrive[i], start[i] - arrive[i], end[i]) for i in by_cu[c]))


if __name__ == '__main__':
    main()
