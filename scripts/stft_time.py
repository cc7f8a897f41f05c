import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
from gcc_nmf_amd.engine import GCCNMFEngine
from gcc_nmf_amd.synthetic import synthetic_batch
e = GCCNMFEngine(160000, dictionarySize=1024, numIterations=1, batch=64)
e.upload(synthetic_batch(0, 64)); e.stft(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): e.stft()
b.record(); torch.cuda.synchronize()
print(os.environ.get('GCCNMF_HIP_LIB', 'default'), 'stft %.3f ms' % (a.elapsed_time(b) / 20))
