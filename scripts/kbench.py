#!/usr/bin/env python
"""Per-kernel micro-benchmark at the headline shape (64 files, F=513, N=1244, K=1024): every per-iteration launch of
the KL-NMF loop through gccnmf_klnmf_stage, and the same three GEMM shapes with a plain-store epilogue through
gccnmf_debug_gemm (main loop without the fused element-wise work).  HIP events on the launch stream, interleaved rounds.

    python scripts/kbench.py [--files 64] [--K 1024] [--reps 10] [--flags 0]
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
    ap.add_argument('--T', type=int, default=622)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--flags', type=int, default=0)
    ap.add_argument('--ablate', type=int, default=0)
    ap.add_argument('--dma', type=int, default=1)
    a = ap.parse_args()
    import torch
    from gcc_nmf_amd import _hip
    from gcc_nmf_amd.engine import Geometry, _ptr, _stream
    lib = _hip.lib()
    lib.gccnmf_set_tuning(1, a.ablate)
    lib.gccnmf_set_tuning(3, a.dma)
    F, T, K, B = 513, a.T, a.K, a.files
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
    U = torch.zeros((B, g.Fp, g.Kp), device=dev)
    G2 = torch.zeros((B, max(g.Kp, g.Fp), g.Np), device=dev)          # also the target of F-row outputs (the 'other output buffer' cases)
    R = ws[:B * g.Fp * g.Np].view(B, g.Fp, g.Np)

    def stage(s):
        _hip.check(lib.gccnmf_klnmf_stage(_ptr(V), _ptr(W), _ptr(H), _ptr(ws), F, N, K, B, 0.0, 1e-16, a.flags, s, _stream()), 'stage')

    def dbg(A, Bm, C, M, Nn, Kd, lda, ldb, ldc, ac, bc, layout, sA, sB, sC):
        _hip.check(lib.gccnmf_debug_gemm(_ptr(A), _ptr(Bm), _ptr(C), M, Nn, Kd, lda, ldb, ldc, ac, bc, layout, B, sA, sB, sC, 0, 0,
                                         _stream()), 'debug_gemm')

    cases = {
        'K1 fused  R=V/(W.(s*H))': lambda: stage(1),
        'K2 fused  H update': lambda: stage(2),
        'K3 fused  R=V/(W.H)': lambda: stage(3),
        'K4a fused U=R.H^T+rowsum': lambda: stage(4),
        'K4b W update+normalise': lambda: stage(5),
        'K1 shape, store only (A_KC, tail)': lambda: dbg(W, H, R, F, N, K, g.Kp, g.Np, g.Np, g.Fp - 1, g.Np - 4, 1 | 4, g.Fp * g.Kp, g.Kp * g.Np, g.Fp * g.Np),
        'K2 shape, store only (W^T.R)': lambda: dbg(W, R, G2, K, N, F, g.Kp, g.Np, g.Np, g.Kp - 4, g.Np - 4, 0, g.Fp * g.Kp, g.Fp * g.Np, g.Kp * g.Np),
        'K2 shape, store only, LDS-free stream tile (direct.hip)': lambda: dbg(W, R, G2, K, N, F - 1, g.Kp, g.Np, g.Np, g.Kp - 4, g.Np - 4, 32, g.Fp * g.Kp, g.Fp * g.Np, g.Kp * g.Np),
        'K2 shape, Kd=512, store only (dma tile)': lambda: dbg(W, R, G2, K, N, F - 1, g.Kp, g.Np, g.Np, g.Kp - 4, g.Np - 4, 0, g.Fp * g.Kp, g.Fp * g.Np, g.Kp * g.Np),
        'K4a shape, store only (KC,KC, tail)': lambda: dbg(R, H, U, F, K, N, g.Np, g.Np, g.Kp, g.Fp - 1, g.Kp - 1, 3 | 4, g.Fp * g.Np, g.Kp * g.Np, g.Fp * g.Kp),
        'K1 shape, store only, ONE H for all files (B L2-resident)': lambda: dbg(W, H, R, F, N, K, g.Kp, g.Np, g.Np, g.Fp - 1, g.Np - 4, 1 | 4, g.Fp * g.Kp, 0, g.Fp * g.Np),
        'K1 shape, store only, ONE W,H, one output tile set': lambda: dbg(W, H, R, F, N, K, g.Kp, g.Np, g.Np, g.Fp - 1, g.Np - 4, 1 | 4, 0, 0, 0),
        'K1 shape -> other output buffer': lambda: dbg(W, H, G2, F, N, K, g.Kp, g.Np, g.Np, g.Fp - 1, g.Np - 4, 1 | 4, g.Fp * g.Kp, g.Kp * g.Np, max(g.Kp, g.Fp) * g.Np),
        'K1 shape, N=1280 (full last tile)': lambda: dbg(W, H, G2, F, g.Np, K, g.Kp, g.Np, g.Np, g.Fp - 1, g.Np - 4, 1 | 4, g.Fp * g.Kp, g.Kp * g.Np, max(g.Kp, g.Fp) * g.Np),
        'K1 shape, no tail row (M=512)': lambda: dbg(W, H, G2, 512, N, K, g.Kp, g.Np, g.Np, g.Fp - 1, g.Np - 4, 1, g.Fp * g.Kp, g.Kp * g.Np, max(g.Kp, g.Fp) * g.Np),
        'K4a shape -> R buffer': lambda: dbg(V, H, R, F, K, N, g.Np, g.Np, g.Np, g.Fp - 1, g.Kp - 1, 3 | 4, g.Fp * g.Np, g.Kp * g.Np, g.Fp * g.Np),
    }
    if not hasattr(lib, 'gccnmf_debug_mfma_peak'):          # the product library: no LDS-free stream tile (debug_gemm layout bit 32 is an experiment build's)
        cases = {k: v for k, v in cases.items() if 'stream tile' not in k}
    # pure-MFMA probe: 512 blocks x 4 waves x 8 x 4096 MFMAs
    scratch = torch.zeros(16, device=dev)
    for blocks in ((256, 512, 1024) if hasattr(lib, 'gccnmf_debug_mfma_peak') else ()):      # the probe lives in the experiment build
        lib.gccnmf_debug_mfma_peak(_ptr(scratch), blocks, 4096, _stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.gccnmf_debug_mfma_peak(_ptr(scratch), blocks, 4096, _stream())
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        print('mfma-only probe %4d blocks: %.3f ms  %.1f TF/s' % (blocks, ms, blocks * 4 * 8 * 4096 * 4096.0 / (ms * 1e-3) / 1e12))
    stage(0)
    for _ in range(2):
        for s in range(1, 6):
            stage(s)
    for fn in cases.values():
        fn()
    torch.cuda.synchronize()
    times = {k: [] for k in cases}
    for r in range(a.reps):
        for name, fn in cases.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            times[name].append(e0.elapsed_time(e1))
    flop = 2.0 * F * K * N * B
    out = {}
    for name, ts in times.items():
        med = float(np.median(ts))
        out[name] = {'median_ms': med, 'min_ms': float(np.min(ts))}
        if 'K4b' not in name:
            out[name]['tflops_median'] = flop / (med * 1e-3) / 1e12
        print('%-40s median %.3f ms  min %.3f ms  %s' % (name, med, np.min(ts),
              '' if 'K4b' in name else '%.1f TF/s (%.0f %% of 157.3)' % (out[name]['tflops_median'], 100 * out[name]['tflops_median'] / 157.3)))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
