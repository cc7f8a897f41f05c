"""KL-NMF of the 64-file bench batch as file groups of UNEQUAL sizes on separate streams (the engine uses equal halves): does a split
along whole rounds of 512 workgroups (25.6 files each) beat 32 + 32?   python scripts/group_split.py"""
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
streams = [torch.cuda.Stream() for _ in range(4)]
for sizes in ([64], [32, 32], [25, 25, 14], [26, 26, 12], [51, 13], [38, 26], [25, 39], [22, 21, 21], [16, 16, 16, 16], [48, 16], [40, 24], [32, 32], [64]):
    ws = [torch.zeros(lib.gccnmf_klnmf_workspace_floats(g.F, g.N, g.K, n), dtype=torch.float32, device='cuda') for n in sizes]
    def run():
        e.W.copy_(e.W0.unsqueeze(0).expand_as(e.W)); e.H.copy_(e.H0.unsqueeze(0).expand_as(e.H))
        ready = torch.cuda.Event(); ready.record()
        b0 = 0
        for i, n in enumerate(sizes):
            st = streams[i]; st.wait_event(ready)
            _hip.check(lib.gccnmf_klnmf(_ptr(e.V[b0]), _ptr(e.W[b0]), _ptr(e.H[b0]), _ptr(ws[i]), g.F, g.N, g.K, n, 100, 0.0, 1e-16, (4 | len(sizes) << 8) if len(sizes) > 1 else 0, st.cuda_stream), 'klnmf')
            d = torch.cuda.Event(); d.record(st); torch.cuda.current_stream().wait_event(d)
            b0 += n
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 3
    print(json.dumps({'groups': sizes, 'nmf100_ms': ms, 'frames_per_s_nmf_only': B * g.T / (ms * 1e-3)}), flush=True)
