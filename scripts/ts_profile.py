"""Where a time-sharded step spends its time on one GPU (phases separated by device syncs), and the effect of the stream-group
count of the shared-dictionary iteration (tuning key 8).   python scripts/ts_profile.py [seconds] """
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                                   # noqa: E402
from gcc_nmf_amd import _hip                                   # noqa: E402
from gcc_nmf_amd.distributed import HipTimeShard, train_shared_dictionary   # noqa: E402
from gcc_nmf_amd.synthetic import synthetic_mixture            # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 160.0
x = synthetic_mixture(7, numSamples=int(seconds * 16000))
out = {'seconds': seconds}
for groups in (1, 2, 3, 4):
    _hip.check(_hip.lib().gccnmf_set_tuning(8, groups), 'tune')
    for block in (None, 1024, 2560):
        local = HipTimeShard(x, 0, 1, dictionarySize=1024, block=block)
        sync = torch.cuda.synchronize

        def phase(fn, reps=3):
            fn(); sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            sync()
            return 1e3 * (time.perf_counter() - t0) / reps
        res = {'blocks': [list(b) for b in local.nmf.blocks]}
        res['stft_ms'] = phase(local.stft)
        res['nmf100_ms'] = phase(lambda: train_shared_dictionary(local.nmf, 100))
        if groups == 2 and block is None:
            s = local.angular_sum()
            res['angular_ms'] = phase(local.angular_sum)
            local.set_angular_mean(s / float(local.T_total))
            res['masks_spec_istft_ms'] = phase(local.masks_and_spectrograms)
            res['tail_ms'] = phase(local.tail_frames)
            res['overlap_add_to_host_ms'] = phase(lambda: local.overlap_add(None))
        out['groups%d_block%s' % (groups, block)] = res
        print('groups', groups, 'block', block, json.dumps(res), file=sys.stderr, flush=True)
        del local
print(json.dumps(out, indent=1))
