"""Two file groups of 32 on two streams: does starting the second group a fraction of a launch LATER (so that one group's tail meets the
middle of the other's launch instead of its tail) beat the simultaneous start?   python scripts/group_phase.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gcc_nmf_amd import _hip
from gcc_nmf_amd.engine import GCCNMFEngine, _ptr
from gcc_nmf_amd.synthetic import synthetic_batch
lib = _hip.lib()
B = 64
e = GCCNMFEngine(160000, dictionarySize=1024, numIterations=100, batch=B)
e.upload(synthetic_batch(0, B)); e.stft(); torch.cuda.synchronize()
g = e.g
streams = [torch.cuda.Stream() for _ in range(2)]
ws = [torch.zeros(lib.gccnmf_klnmf_workspace_floats(g.F, g.N, g.K, 32), dtype=torch.float32, device='cuda') for _ in range(2)]
for cycles in (0, 100000, 200000, 400000, 600000, 800000, 1200000, 0):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); torch.cuda._sleep(max(cycles, 1)); e1.record(); torch.cuda.synchronize()
    delay_us = 1e3 * e0.elapsed_time(e1)
    def run():
        e.W.copy_(e.W0.unsqueeze(0).expand_as(e.W)); e.H.copy_(e.H0.unsqueeze(0).expand_as(e.H))
        ready = torch.cuda.Event(); ready.record()
        for i in range(2):
            st = streams[i]; st.wait_event(ready)
            if i == 1 and cycles:
                with torch.cuda.stream(st):
                    torch.cuda._sleep(cycles)
            _hip.check(lib.gccnmf_klnmf(_ptr(e.V[32 * i]), _ptr(e.W[32 * i]), _ptr(e.H[32 * i]), _ptr(ws[i]), g.F, g.N, g.K, 32, 100, 0.0, 1e-16, 4 | 2 << 8, st.cuda_stream), 'klnmf')
            d = torch.cuda.Event(); d.record(st); torch.cuda.current_stream().wait_event(d)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 3
    print(json.dumps({'second_group_delay_cycles': cycles, 'delay_us': delay_us, 'nmf100_ms': ms, 'minus_delay_ms': ms - delay_us * 1e-3}), flush=True)
