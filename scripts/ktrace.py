#!/usr/bin/env python
"""Per-workgroup timeline of one KL-NMF GEMM launch at the headline shape (gccnmf_debug_set_trace): when each workgroup
started, left its main loop and finished its epilogue, and on which CU.

    python scripts/ktrace.py [--stage 1] [--files 64] [--probe] > gpurun_out/trace.json
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--files', type=int, default=64)
    ap.add_argument('--K', type=int, default=1024)
    ap.add_argument('--stage', type=int, default=1)
    ap.add_argument('--dump', default='')
    ap.add_argument('--resident', action='store_true', help='experiment build: 512 resident workgroups pulling tiles by ticket (tuning key 18); with --probe the per-phase cycles are per RESIDENT workgroup, summed over its tiles')
    ap.add_argument('--no-prefetch', action='store_true', help='with --resident: tuning key 19 = 0')
    ap.add_argument('--repeat', type=int, default=0, help='launch the stage this many times right before the traced launch (back-to-back launches of one kernel)')
    ap.add_argument('--probe', action='store_true', help='library built with -DGEMM_DMA_PROBE: per-phase cycles of the k-tile')
    a = ap.parse_args()
    import torch
    from gcc_nmf_amd import _hip
    from gcc_nmf_amd.engine import Geometry, _ptr, _stream
    lib = _hip.lib()
    if not hasattr(lib, 'gccnmf_debug_set_trace'):
        sys.exit('needs the experiment build: make -C gcc_nmf_amd/csrc EXPERIMENTS=1; GCCNMF_HIP_LIB=gcc_nmf_amd/libgccnmf_hip_exp.so')
    if a.resident:
        assert lib.gccnmf_set_tuning(18, 1) == 0 and lib.gccnmf_set_tuning(19, 0 if a.no_prefetch else 1) == 0, 'needs the experiment build'
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

    for s in (1, 2, 3, 4):
        stage(s)
    torch.cuda.synchronize()
    nblk = 16384
    trace = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
    for _ in range(a.repeat):
        stage(a.stage)
    lib.gccnmf_debug_set_trace(_ptr(trace), nblk)
    stage(a.stage)
    torch.cuda.synchronize()
    lib.gccnmf_debug_set_trace(None, 0)
    t_all = trace.cpu().numpy()
    tiles = {1: 1 * (g.Np // 64), 3: 1 * (g.Np // 64), 2: -(-K // 512) * (g.Np // 64), 4: -(-K // 64)}[a.stage]
    grid = 8 * (-(-B // 8)) * tiles
    t = t_all[:max(grid, int(np.nonzero(t_all[:, 0])[0].max()) + 1)] if not a.probe else t_all[:grid]
    if a.probe and a.resident:
        # the resident grid: rows [classic grid + 4 * workgroup + wave] = cycles per phase summed over ALL tiles of that workgroup, [7] = its tiles
        import ctypes
        plan = (ctypes.c_int * 8)()
        Mo, No = {1: (512, N), 3: (512, N), 2: (K, N), 4: (512, K)}[a.stage]
        lib.gccnmf_debug_gemm_plan(Mo, No, B, 1, 0, 1 if a.stage != 4 else 0, plan, None, 0)
        cg = plan[7]
        pr = t_all[cg:cg + 4 * 512].reshape(512, 4, 8).astype(np.float64)
        items = pr[:, 0, 7]
        nkt = {1: K // 16, 3: K // 16, 2: (F - 1) // 16, 4: g.Np // 16}[a.stage]
        names = ['G1 issue + 32 MFMA + DMA', 'wait G1', '8 MFMA', 'wait DMA, arrive, 16 MFMA', 'split-barrier wait', 'G0 issue + 8 MFMA', 'wait G0']
        print('resident grid, stage %d, %d files: classic grid %d items on 512 workgroups; tiles per workgroup min %d median %d max %d' % (
            a.stage, B, cg, items.min(), np.median(items), items.max()))
        per = pr[:, :, :7] / np.maximum(items, 1)[:, None, None] / nkt          # cycles per k-tile (wide-tile equivalents are not separated from narrow ones)
        m = per.mean(axis=(0, 1))
        print('  cycles per k-tile, mean over workgroups and waves: %.0f (solo MFMA time of a wide tile: 4096)' % m.sum())
        for n, v, w in zip(names, m, per.mean(axis=0).T):
            print('     %-28s %7.0f   per wave: %s' % (n, v, '  '.join('%6.0f' % x for x in w)))
    elif a.probe:
        pr = t_all[grid:5 * grid, :7].reshape(grid, 4, 7).astype(np.float64)
        nkt = {1: K // 16, 3: K // 16, 2: (F - 1) // 16, 4: g.Np // 16}[a.stage]
        names = ['G1 issue + 32 MFMA + DMA', 'wait G1', '8 MFMA', 'wait DMA, arrive, 16 MFMA', 'split-barrier wait', 'G0 issue + 8 MFMA', 'wait G0']
        start = (t[:, 0] - t[:, 0].min()) / 100.0
        loop = (t[:, 2] - t[:, 1]) / 100.0
        first = start < 5
        groups = {'round 1, faster half (older on its CU)': first & (loop <= np.median(loop[first])),
                  'round 1, slower half (younger)': first & (loop > np.median(loop[first])),
                  'last round (alone on its CU most of the time)': start > np.percentile(start, 85)}
        for name, sel in groups.items():
            m = pr[sel].mean(axis=(0, 1)) / nkt
            print('%s: %d workgroups, main loop %.0f us, cycles per k-tile %.0f (solo MFMA time 4096)' % (name, sel.sum(), loop[sel].mean(), m.sum()))
            mw = pr[sel].mean(axis=0) / nkt                      # per wave
            for i, (n, v) in enumerate(zip(names, m)):
                print('     %-28s %7.0f   per wave: %s' % (n, v, '  '.join('%6.0f' % x for x in mw[:, i])))
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    us = (t[:, :4] - t0) / 100.0                     # 100 MHz -> microseconds
    cu = t[:, 4]
    order = np.argsort(us[:, 0])
    print('stage %d: %d workgroups on %d distinct CUs, launch span %.1f us' % (a.stage, len(t), len(np.unique(cu >> 8)), us[:, 3].max()))
    print('  main loop          (t2-t1): median %.1f  p10 %.1f  p90 %.1f us' % tuple(np.percentile(us[:, 2] - us[:, 1], [50, 10, 90])))
    print('  epilogue           (t3-t2): median %.1f  p10 %.1f  p90 %.1f us' % tuple(np.percentile(us[:, 3] - us[:, 2], [50, 10, 90])))
    if t[:, 5].max() > 0:
        e5, e6 = (t[:, 5] - t0) / 100.0, (t[:, 6] - t0) / 100.0
        print('    lean epilogue, wave 0: first pair done after %.1f us, all stores issued after %.1f us, stores drained after %.1f us (medians)' % (
            np.median(e5 - us[:, 2]), np.median(e6 - us[:, 2]), np.median(us[:, 3] - us[:, 2])))
    # how many workgroups are in their epilogue / main loop over time
    edges = np.arange(0, us[:, 3].max() + 10, 10.0)
    in_epi = [(np.sum((us[:, 2] <= x) & (us[:, 3] > x)), np.sum((us[:, 1] <= x) & (us[:, 2] > x))) for x in edges]
    print('  t[us]: workgroups in epilogue / in main loop')
    print('   ' + '  '.join('%.0f:%d/%d' % (x, e, m) for x, (e, m) in zip(edges, in_epi)))
    # co-resident pairs: per CU, sorted starts
    cuid = cu >> 8
    first = {}
    for i in order:
        first.setdefault(int(cuid[i]), []).append(i)
    lens = np.array([len(v) for v in first.values()])
    print('  workgroups per CU: min %d median %d max %d' % (lens.min(), np.median(lens), lens.max()))
    d = [us[v[1], 1] - us[v[0], 1] for v in first.values() if len(v) > 1]
    print('  main-loop start of the 2nd minus the 1st workgroup on a CU: median %.1f  p10 %.1f  p90 %.1f us' % tuple(np.percentile(d, [50, 10, 90])))
    # per-CU timelines of a few CUs
    for c in sorted(first)[:6]:
        print('  CU %05x: ' % c + '  '.join('[%.0f %.0f %.0f %.0f]' % tuple(us[i]) for i in first[c]))
    if a.dump:
        json.dump({'us': us.tolist(), 'cu': cu.tolist()}, open(a.dump, 'w'))


if __name__ == '__main__':
    main()
