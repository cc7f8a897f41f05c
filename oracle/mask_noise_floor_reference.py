"""Noise floor of the coefficient masks measured on the UNMODIFIED reference itself (seanwood/gcc-nmf, /root/reference): its own functions
in runGCCNMF.py order on its six mixtures at K = 1024, once with 1 and once with 8 BLAS threads -- nothing differs but the sgemm
summation order inside numpy.dot.  Companion of oracle/mask_noise_floor.py, which does the same with the oracle PORT: DESIGN.md section 4
quotes both and says which is which.  Test infrastructure, build container only (the reference checkout does not travel):

    python oracle/mask_noise_floor_reference.py   ->  tests/golden/mask_noise_floor_reference.json   (~7 min on 8 cores)
"""
import json
import os
import sys

import numpy as np

REF_ROOT = os.environ.get('GCCNMF_REFERENCE_ROOT', '/root/reference')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF_ROOT)
sys.path.insert(0, REPO)
from threadpoolctl import threadpool_limits                      # noqa: E402
import gccNMF.gccNMFFunctions as R                               # noqa: E402  (the reference, unmodified)
from make_golden import reference_pipeline, WAVS                 # noqa: E402  (runGCCNMF.py:36-52 on the reference module)

res = {}
for name in WAVS:
    x, sr = R.wavread(os.path.join(REF_ROOT, 'data', name + '_mix.wav'))
    runs = []
    for th in (1, 8):
        with threadpool_limits(limits=th):
            runs.append(reference_pipeline(x, sr, 1024, 256, 128, 1.0, 3, 1024, 100))
    a, b = runs
    G = np.asarray(a['G'], np.float64)                            # (S, K, T): the scores the arg-max is taken over
    srt = np.sort(G, axis=0)
    gap = (srt[-1] - srt[-2]) / np.abs(srt[-1])
    flips = np.argmax(a['G'], axis=0) != np.argmax(b['G'], axis=0)
    res[name + '_mix.wav'] = dict(
        W_rel=float(np.linalg.norm(a['W'] - b['W']) / np.linalg.norm(a['W'])), H_rel=float(np.linalg.norm(a['H'] - b['H']) / np.linalg.norm(a['H'])),
        flips=int(flips.sum()), coeffs=int(flips.size), largest_flipped_gap=float(gap[flips].max()) if flips.any() else 0.0,
        idx_equal=bool((a['idx'] == b['idx']).all()), waveform_rms=float(np.sqrt(np.mean((np.asarray(a['y'], np.float64) - b['y']) ** 2))))
    print(name, res[name + '_mix.wav'], flush=True)
res['_what'] = 'unmodified reference functions (gccNMF/gccNMFFunctions.py), 1 vs 8 OpenBLAS threads, K = 1024, 100 iterations, hop 256'
json.dump(res, open(os.path.join(REPO, 'tests', 'golden', 'mask_noise_floor_reference.json'), 'w'), indent=1, sort_keys=True)
