#!/usr/bin/env python
"""What the f32 matrix pipe sustains on this box as a function of how long it is kept busy: the pure-MFMA probe
(gccnmf_debug_mfma_peak: 8 independent v_mfma_f32_32x32x2_f32 chains per wave, two waves per SIMD, nothing else) for growing
durations, single launches and back-to-back trains.  Context for the roofline fractions: the 157.3 TFLOP/s peak assumes 2.4 GHz."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from gcc_nmf_amd import _hip                            # noqa: E402
from gcc_nmf_amd.engine import _ptr, _stream            # noqa: E402

lib = _hip.lib()
if not hasattr(lib, 'gccnmf_debug_mfma_peak'):
    sys.exit('needs the experiment build: make -C gcc_nmf_amd/csrc EXPERIMENTS=1; GCCNMF_HIP_LIB=gcc_nmf_amd/libgccnmf_hip_exp.so')
scratch = torch.zeros(16, device='cuda')


def run(blocks, iters, launches):
    lib.gccnmf_debug_mfma_peak(_ptr(scratch), blocks, iters, _stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        lib.gccnmf_debug_mfma_peak(_ptr(scratch), blocks, iters, _stream())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    flop = float(launches) * blocks * 4 * iters * 8 * 2 * 32 * 32 * 2
    return ms, flop / (ms * 1e-3) / 1e12


for blocks in (512,):
    for iters, launches in [(256, 1), (1024, 1), (4096, 1), (16384, 1), (65536, 1), (262144, 1), (1024, 16), (1024, 256), (4096, 64)]:
        ms, tf = run(blocks, iters, launches)
        print('blocks %4d  iters %6d  launches %3d : %9.3f ms  %6.1f TFLOP/s  (%.0f %% of 157.3)' % (blocks, iters, launches, ms, tf, 100 * tf / 157.3), flush=True)
for blocks in (256, 1024, 1536, 2048):
    ms, tf = run(blocks, 4096, 1)
    print('blocks %4d  iters %6d  launches %3d : %9.3f ms  %6.1f TFLOP/s  (%.0f %% of 157.3)' % (blocks, 4096, 1, ms, tf, 100 * tf / 157.3), flush=True)
