#!/usr/bin/env python
"""Per-stage device time of the headline step (64 files, K=1024): HIP events around each stage call of GCCNMFEngine.run().
usage: python scripts/stage_times.py [--tune KEY=VALUE ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                            # noqa: E402
from gcc_nmf_amd import _hip                            # noqa: E402
from gcc_nmf_amd.engine import GCCNMFEngine             # noqa: E402
from gcc_nmf_amd.synthetic import synthetic_batch       # noqa: E402

for kv in sys.argv[1:]:
    if '=' in kv:
        k, v = [int(x) for x in kv.split('=')]
        _hip.check(_hip.lib().gccnmf_set_tuning(k, v), 'tune')
B = 64
xs = synthetic_batch(0, B)
e = GCCNMFEngine(160000, dictionarySize=1024, numIterations=100, batch=B)
e.upload(xs)
e.run()
torch.cuda.synchronize()
names = ['stft', 'klnmf', 'localize', 'masks', 'reconstruct', 'istft']
tot = {n: 0.0 for n in names}
reps = 3
for _ in range(reps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    ev[0].record()
    for i, n in enumerate(names):
        getattr(e, n)()
        ev[i + 1].record()
    torch.cuda.synchronize()
    for i, n in enumerate(names):
        tot[n] += ev[i].elapsed_time(ev[i + 1]) / reps
print('  '.join('%s %.3f ms' % (n, tot[n]) for n in names), ' | non-NMF %.3f ms' % sum(v for n, v in tot.items() if n != 'klnmf'), 'tdoa ok', bool((e.get_tdoa_indexes() == torch.tensor([27, 59, 91]).numpy()).all()))
