import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def golden_wav(prefix):
    """float32 (2, n) samples + sample rate of a committed reference mixture; same
    conversion as the reference's wavread (gccNMF/wavfile.py:34-37)."""
    from scipy.io import wavfile
    sr, pcm = wavfile.read(os.path.join(GOLDEN, 'data', prefix + '_mix.wav'))
    return ((pcm.astype('float32') - 0) / 32768).T.copy(), sr


@pytest.fixture(scope='session')
def dev1():
    return golden_wav('dev1_female3_liverec_130ms_1m')


def mask_flips(argmax, g):
    """(number of (k, t) coefficients whose target assignment differs from the reference golden `g`, largest relative top-2
    gap of the REFERENCE's scores among them).  SURVEY 8(c): masks must be exact except at genuine near-ties; goldens list
    every position with a gap below `tie_below` (oracle/make_golden.py: near_ties), anything unlisted counts as gap = inf."""
    flipped = np.flatnonzero(np.asarray(argmax).ravel() != g['argmax'].ravel())
    gap = np.full(g['argmax'].size, np.inf)
    gap[g['tie_pos']] = g['tie_gap']
    return len(flipped), (float(gap[flipped].max()) if len(flipped) else 0.0)
