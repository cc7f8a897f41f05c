"""CPU: oracle/rt_oracle.py (the NumPy restatement the HIP streaming path is tested against) versus goldens produced by the
UNMODIFIED reference GCCNMFProcessor / OverlapAddProcessor / SharedMemoryCircularBuffer running on oracle/theano_stub
(oracle/make_rt_golden.py).  This is what pins the streaming oracle to a reference run."""
import json
import os
import warnings

import numpy as np
import pytest

from oracle import rt_oracle as R

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
FRAME_CASES = ['a', 'b', 'c', 'd', 'e', 'f']                  # e, f: window sizes 400 and 1000 (round 4)
STREAM_CASES = ['default', 'lowlatency', 'dev1', 'ws400', 'ws1000', 'bigblock']


def load(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def test_manifest_lists_every_fixture():
    m = json.load(open(os.path.join(GOLD, 'RT_MANIFEST.json')))
    assert sorted(m['files']) == sorted(['rt_frames_%s.npz' % c for c in FRAME_CASES] + ['rt_stream_%s.npz' % c for c in STREAM_CASES])
    assert 'unmodified' in m['theano']


@pytest.mark.parametrize('case', FRAME_CASES)
def test_process_frames_equals_reference_run(case):
    g = load('rt_frames_' + case)
    ws, K, D, Tc, seed = [int(v) for v in g['params']]
    W = R.make_rt_dictionary(seed, ws // 2 + 1, K)
    assert abs(W.astype(np.float64).sum() - float(g['W_sum'])) < 1e-9 * float(g['W_sum'])
    for i in range(3):
        p = R.GCCNMFProcessorOracle(16000, ws, Tc, W, float(g['d']), D, localizationEnabled=False, targetMode=int(g['mode%d' % i]))
        p.setTargetTDOARange(*g['target%d' % i])
        y, im = p.processFrames(g['frames%d' % i], return_intermediates=True)
        assert np.array_equal(im['X'], g['X%d' % i])                                   # same rfft, same cast
        assert np.array_equal(im['C'], g['C%d' % i], equal_nan=True)
        assert np.array_equal(im['argmaxTDOA'], g['argmax%d' % i])
        assert np.abs(im['HMask'] - g['HMask%d' % i]).max() < 1e-12
        # the reference's boxcar graph carries float16 0/1 constants into a float32 dot, its window graph a float32 row sum that
        # accumulates in float64 (theano_stub mirrors both); the oracle computes these two in float64 / NumPy pairwise float32
        assert np.abs(im['tfMask'] - g['tfMask%d' % i]).max() < 2e-6
        assert np.abs(im['gccPHAT'] - g['gccPHAT%d' % i]).max() < 1e-12
        assert np.abs(g['y%d' % i]).max() > 1e-5 and np.abs(y - g['y%d' % i]).max() < 2e-6 * np.abs(g['y%d' % i]).max()


@pytest.mark.parametrize('case', STREAM_CASES)
def test_stream_equals_reference_run(case):
    g = load('rt_stream_' + case)
    ws, hop, B, K, D, numBlocks, L, seed = [int(v) for v in g['params']]
    W = R.make_rt_dictionary(seed, ws // 2 + 1, K)
    p = R.GCCNMFProcessorOracle(16000, ws, B // hop, W, float(g['d']), D, localizationEnabled=True, localizationWindowSize=L)
    p.setTargetTDOARange(9.6, 5.0, 2.0, 0.0)
    ola = R.OverlapAddOracle(2, ws, hop, B, B // hop)
    x = g['x']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for b in range(numBlocks):
            y = ola.processFrames(x[:, b * B:(b + 1) * B], p.processFrames)
            assert float(p.targetTDOAIndex) == g['tdoa'][b], b                          # tracked TDOA, block by block
            # the reference hands float32 buffer contents to a float64 shared array (utils.py:116)
            assert np.abs(y - g['y'][:, b * B:(b + 1) * B]).max() <= 1e-6 * max(np.abs(g['y']).max(), 1e-12), b
    assert np.allclose(p.gccPHATHistory.getUnraveledArray(), g['gccPHATHistory'], atol=1e-12, equal_nan=True)
