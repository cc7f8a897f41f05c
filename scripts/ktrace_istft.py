#!/usr/bin/env python
"""Where a workgroup of the fused inverse STFT spends its shader clocks (lab build): hand-out / shift of the sliding accumulator, spectrogram
fetch + bit-reversed placement, butterflies, windowed overlap-add -- at the bench shape (64 files, 1024-pt, hop 256).
GCCNMF_HIP_LIB=gcc_nmf_amd/libgccnmf_hip_exp.so python scripts/ktrace_istft.py [--files 64]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--files', type=int, default=64)
    ap.add_argument('--dictionary-size', type=int, default=128)
    a = ap.parse_args()
    import torch
    from gcc_nmf_amd import _hip
    from gcc_nmf_amd.engine import GCCNMFEngine, _ptr
    from gcc_nmf_amd.synthetic import synthetic_batch
    lib = _hip.lib()
    if not hasattr(lib, 'gccnmf_debug_set_trace'):
        sys.exit('needs the experiment build: make -C gcc_nmf_amd/csrc EXPERIMENTS=1; GCCNMF_HIP_LIB=gcc_nmf_amd/libgccnmf_hip_exp.so')
    e = GCCNMFEngine(160000, dictionarySize=a.dictionary_size, numIterations=10, batch=a.files)
    e.upload(synthetic_batch(0, a.files))
    e.run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record(); e.istft(); ev[1].record(); torch.cuda.synchronize()
    print('istft stage, untraced: %.3f ms' % ev[0].elapsed_time(ev[1]))
    rows = 8192
    trace = torch.zeros((rows, 8), dtype=torch.int64, device='cuda')
    lib.gccnmf_debug_set_trace(_ptr(trace), rows)
    ev[0].record(); e.istft(); ev[1].record(); torch.cuda.synchronize()
    lib.gccnmf_debug_set_trace(None, 0)
    t = trace.cpu().numpy()
    t = t[t[:, 0] > 0]
    names = ['hand-out / shift (+ its barriers)', 'spectrogram fetch + placement', 'butterflies (3 passes)', 'windowed overlap-add']
    cyc = t[:, 1:5].astype(float)
    tot = cyc.sum(axis=1)
    span_us = (t[:, 5] - t[:, 0]) / 100.0
    print('traced: %.3f ms; %d workgroups; a workgroup lives %.1f us (median), %.0f shader clocks in the four phases (%.2f clocks per 10 ns)' % (
        ev[0].elapsed_time(ev[1]), len(t), np.median(span_us), np.median(tot), np.median(tot / np.maximum(t[:, 5] - t[:, 0], 1))))
    for i, n in enumerate(names):
        print('   %-36s median %8.0f clocks = %4.1f %% of the workgroup   (p10 %8.0f  p90 %8.0f)' % (n, np.median(cyc[:, i]), 100 * np.median(cyc[:, i] / tot), *np.percentile(cyc[:, i], [10, 90])))
    start = (t[:, 0] - t[:, 0].min()) / 100.0
    print('   launch span %.1f us; workgroups in flight at the median start: %d' % ((t[:, 5].max() - t[:, 0].min()) / 100.0, int(((t[:, 0] <= np.median(t[:, 0])) & (t[:, 5] > np.median(t[:, 0]))).sum())))


if __name__ == '__main__':
    main()
