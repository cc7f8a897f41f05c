#!/usr/bin/env python
"""Short dictionaries: one KL-NMF iteration (stages 1-5 back to back, HIP events around 20 iterations) per batch size, with the launch
forms chosen by the library's cost models (tuning keys 16 / 17 = 1) against the four-launch form (= 0).  Checks that the automatic choice
never loses.   python scripts/k128_sweep.py [K [T]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from gcc_nmf_amd import _hip                            # noqa: E402
from gcc_nmf_amd.engine import Geometry, _ptr, _stream  # noqa: E402

lib = _hip.lib()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T = int(sys.argv[2]) if len(sys.argv) > 2 else 622
F = 513
g = Geometry(F, T, K)
N = g.N
dev = 'cuda'


def iteration_us(B, reps=20):
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
    for _ in range(3):
        for s in range(1, 6):
            stage(s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        for s in range(1, 6):
            stage(s)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


print('K = %d, N = %d: one KL-NMF iteration (us), launches by the cost models vs the four-launch form' % (K, N))
worst = 0.0
for B in (5, 8, 12, 16, 20, 24, 25, 26, 28, 32, 40, 48, 50, 56, 64, 72, 80, 96, 104, 128):
    row = {}
    for name, v in (('four', 0), ('auto', 1)):
        lib.gccnmf_set_tuning(16, v)
        lib.gccnmf_set_tuning(17, v)
        row[name] = iteration_us(B)
        row[name + '_plan'] = lib.gccnmf_klnmf_plan(F, N, K, B, 0)
    lib.gccnmf_set_tuning(16, 1)
    lib.gccnmf_set_tuning(17, 1)
    ratio = row['auto'] / row['four']
    worst = max(worst, ratio)
    print('files %3d: four launches %7.1f   auto %7.1f (plan %d)   auto / four %.3f%s' % (B, row['four'], row['auto'], row['auto_plan'], ratio,
                                                                                         '   <-- loses' if ratio > 1.03 else ''))
print('worst auto / four: %.3f' % worst)
