"""Mask parity experiment: coefficient-mask flips against the reference goldens on the six reference mixtures at
K = 1024 -- count and largest flipped top-2 gap -- with the throughput tile's V / (W.H) as rcp + Newton (default) and as the IEEE
quotient (tuning key 7), through the small-launch kernels (automatic tile policy: these already divide exactly) and through the
throughput tile (policy 1).  Also the 64 bench files: flips BETWEEN the two division modes.
    python scripts/mask_flips.py > gpurun_out/<tag>/mask_flips.json"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from conftest import golden, golden_wav, mask_flips          # noqa: E402
from gcc_nmf_amd import _hip                                  # noqa: E402
from gcc_nmf_amd.engine import GCCNMFEngine                   # noqa: E402
from gcc_nmf_amd.synthetic import synthetic_batch             # noqa: E402

NAMES = ['dev1_female3_liverec_130ms_1m', 'dev_A_1_2_3_4', 'dev_B_1_8_9_16', 'dev_C_2_7_10_15', 'dev_D_13_14_15_16', 'dev_Sq1_Co_A']
lib = _hip.lib()
out = {'six_wavs_K1024': {}, 'bench_files': {}}
xs = np.stack([golden_wav(w)[0] for w in NAMES])
gw = golden('wh_sub_K1024')
rel = lambda a, b: float(np.linalg.norm((a.astype(np.float64) - b).ravel()) / np.linalg.norm(b.astype(np.float64).ravel()))
for policy in (0, 1):
    for exact in (0, 1):
        lib.gccnmf_set_tuning(2, policy)
        lib.gccnmf_set_tuning(7, exact)
        e = GCCNMFEngine(xs.shape[2], dictionarySize=1024, numIterations=100, batch=len(NAMES))
        e.separate(xs)
        am = e.get_argmax()
        W, H = e.get_WH()
        rows = {}
        for i, w in enumerate(NAMES):
            g = golden('%s_hop256_K1024' % w) if i else golden('dev1_hop256_K1024')
            flips, worst = mask_flips(am[i], g)
            rows[w] = {'flips': flips, 'largest_flipped_gap': worst, 'W_rel': rel(W[i][:, ::16], gw[w + '_W']), 'H_rel': rel(H[i][::16, ::2], gw[w + '_H'])}
        out['six_wavs_K1024']['tile_policy_%d_exact_div_%d' % (policy, exact)] = rows
        del e
lib.gccnmf_set_tuning(2, 0)
xb = synthetic_batch(0, 64)
am = {}
for exact in (0, 1):
    lib.gccnmf_set_tuning(7, exact)
    e = GCCNMFEngine(160000, dictionarySize=1024, numIterations=100, batch=64)
    e.separate(xb)
    am[exact] = e.get_argmax()
    W, H = e.get_WH()
    am[('W', exact)] = W
    del e
lib.gccnmf_set_tuning(7, 1)          # back to the default (IEEE division since round 5)
d = am[0] != am[1]
out['bench_files'] = {'coefficients': int(d.size), 'flips_between_division_modes': int(d.sum()), 'files_with_flips': int(d.any(axis=(1, 2)).sum()),
                      'W_rel_between_modes': rel(am[('W', 0)], am[('W', 1)])}
print(json.dumps(out, indent=1))
