"""Generate tests/golden/rt_*.npz by running the UNMODIFIED reference real-time processor
(``gccNMF/realtime/gccNMFProcessor.py: GCCNMFProcessor``, ``gccNMF/realtime/utils.py: OverlapAddProcessor`` and
``SharedMemoryCircularBuffer``) on top of ``oracle/theano_stub`` -- a NumPy-evaluated stand-in for the few Theano names the
processor uses (Theano itself cannot be installed here).  Build container only (/root/reference is not on the GPU box).

    python oracle/make_rt_golden.py           # rewrites tests/golden/rt_*.npz and tests/golden/RT_MANIFEST.json

Two kinds of fixtures:
  * ``rt_frames_<case>``: one ``processFrames`` call per target setting (two window-function settings, one boxcar), with every
    intermediate the reference's compiled functions expose (spectrogram, coherence, GCC-NMF scores -> arg-max and top-2 gap,
    HMask, tfMask, gccPHAT column, output frames);
  * ``rt_stream_<case>``: a block-by-block run through the reference's OverlapAddProcessor with online localisation on:
    output blocks, the tracked TDOA after every block, the final gccPHAT history.
"""
import json
import os
import sys
import time
import warnings

import numpy as np

REF_ROOT = os.environ.get('GCCNMF_REFERENCE_ROOT', '/root/reference')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, os.path.join(REPO, 'oracle', 'theano_stub'))
sys.path.insert(0, REF_ROOT)
sys.path.insert(0, REPO)

import theano                                                                   # noqa: E402  (the stub)
from gccNMF.realtime.gccNMFProcessor import GCCNMFProcessor, TARGET_MODE_BOXCAR, TARGET_MODE_WINDOW_FUNCTION   # noqa: E402
from gccNMF.realtime.utils import OverlapAddProcessor, SharedMemoryCircularBuffer                              # noqa: E402
import gccNMF.gccNMFFunctions as RF                                             # noqa: E402
from oracle import gccnmf_oracle as O                                           # noqa: E402  (synthetic-signal recipe only)
from oracle.rt_oracle import make_rt_dictionary                                 # noqa: E402  (seeded W recipe shared with the tests)

assert theano.__version__.endswith('numpy-stub')

FRAME_CASES = {   # name: (windowSize, K, D, Tc, d, seed)
    'a': (1024, 64, 64, 1, 0.1, 0),
    'b': (512, 200, 40, 4, 0.1, 1),
    'c': (1024, 1024, 64, 2, 0.1, 2),
    'd': (256, 96, 33, 8, 0.25, 3),
    'e': (400, 64, 48, 2, 0.1, 4),            # round 4: window sizes off the powers of two (the reference takes any, :202,:231)
    'f': (1000, 128, 64, 1, 0.1, 5),
}
TARGETS = [(TARGET_MODE_WINDOW_FUNCTION, (9.6, 5.0, 2.0, 0.0)), (TARGET_MODE_WINDOW_FUNCTION, (20.0, 3.0, 1.0, 0.2)),
           (TARGET_MODE_BOXCAR, (20.0, 12.0, 1.0, 0.0))]
STREAM_CASES = {  # name: (windowSize, hop, block, K, D, d, numBlocks, signal, localizationWindowSize)
    'default': (1024, 512, 512, 64, 64, 0.1, 60, 'synthetic', 6),             # realtime/config.py:50-73 defaults
    'lowlatency': (512, 64, 64, 256, 64, 0.1, 300, 'synthetic', 6),           # BASELINE config 5's window / hop
    'dev1': (512, 128, 256, 128, 48, 1.0, 80, 'dev1', 4),                      # two windows per block, a reference wav
    'ws400': (400, 100, 200, 64, 48, 0.1, 80, 'synthetic', 6),                 # round 4: non-power-of-two window, two windows per block
    'ws1000': (1000, 500, 500, 96, 64, 0.1, 40, 'synthetic', 6),
    'bigblock': (1024, 512, 1024, 64, 64, 0.1, 30, 'synthetic', 6),            # blocks of more than 512 samples
}


def build(ws, Tc, W, D, d, loc, L, Lh=128, with_histories=True):
    K = W.shape[1]
    hist = dict(gccPHATHistory=SharedMemoryCircularBuffer((D, Lh)), tdoaHistory=SharedMemoryCircularBuffer((1, Lh))) if with_histories else {}
    p = GCCNMFProcessor(16000, ws, Tc, {'Pretrained': {K: W}}, 'Pretrained', K, 0, d, loc, L, **hist)
    p.numTDOAs = D                   # arrives through the parameter queue in the app (gccNMFProcessor.py:139-146) before reset()
    return p


def frames_case(name, ws, K, D, Tc, d, seed):
    rng = np.random.RandomState(1000 + seed)
    W = make_rt_dictionary(seed, ws // 2 + 1, K)
    out = dict(params=np.array([ws, K, D, Tc, seed], np.int64), d=np.float64(d), W_sum=np.float64(W.astype(np.float64).sum()))
    for i, (mode, target) in enumerate(TARGETS):
        p = build(ws, Tc, W, D, d, False, 6)
        p.targetMode = mode
        p.reset()                                                                # buildTheanoFunctions (:233-270)
        p.setTargetTDOARange(*target)
        frames = (rng.standard_normal((2, ws, Tc)) * 0.1).astype(np.float32)
        y = p.processFrames(frames)
        realGCC = p.getComplexGCC()[0].real
        scores = p.getGCCNMF(realGCC)[0]                                         # (D, Tc, K)
        tfMask, HMask = p.getTFMask(realGCC)
        srt = np.sort(scores, axis=0)
        gap = ((srt[-1] - srt[-2]) / np.abs(srt[-1])).T                          # (K, Tc) relative top-2 gap of the reference scores
        C = theano.function([], [p.coherenceV])()[0]
        out.update({'frames%d' % i: frames, 'X%d' % i: p.complexMixtureSpectrogram.copy(), 'C%d' % i: C,
                    'argmax%d' % i: np.argmax(scores, axis=0).T.astype(np.int32), 'gap%d' % i: gap.astype(np.float32),
                    'HMask%d' % i: np.asarray(HMask, np.float64), 'tfMask%d' % i: np.asarray(tfMask, np.float64),
                    'gccPHAT%d' % i: p.gccPHATHistory.values[:, :Tc].copy(),
                    'y%d' % i: np.asarray(y, np.float64), 'mode%d' % i: np.int64(mode), 'target%d' % i: np.array(target, np.float64)})
    return out


def stream_case(name, ws, hop, B, K, D, d, numBlocks, signal, L):
    if signal == 'synthetic':
        x = O.synthetic_mixture(5, numSamples=numBlocks * B, delays=(-3, 1, 4))
    else:
        x, sr = RF.loadMixtureSignal(os.path.join(REF_ROOT, 'data', 'dev1_female3_liverec_130ms_1m_mix.wav'))
        x = np.ascontiguousarray(x[:, 16000:16000 + numBlocks * B])
    Tc = B // hop
    W = make_rt_dictionary(10 + len(name), ws // 2 + 1, K)
    p = build(ws, Tc, W, D, d, True, L)
    p.reset()
    p.setTargetTDOARange(9.6, 5.0, 2.0, 0.0)
    inputFrames, outputFrames = np.zeros((2, B)), np.zeros((2, B))              # float64 like runRealtimeGCCNMF.py:69-72
    ola = OverlapAddProcessor(2, ws, hop, B, Tc, inputFrames, outputFrames)
    y = np.zeros((2, numBlocks * B))
    tdoa, loc_gap = np.zeros(numBlocks), np.zeros(numBlocks)
    for b in range(numBlocks):
        inputFrames[:] = x[:, b * B:(b + 1) * B]
        ola.processFrames(p.processFrames)                                       # utils.py:99-116
        y[:, b * B:(b + 1) * B] = outputFrames
        tdoa[b] = p.targetTDOAIndex.get_value()
        m = np.sort(np.nanmean(p.gccPHATHistory.getUnraveledArray()[:, -L:], axis=-1))      # what :219 took the arg-max of
        loc_gap[b] = (m[-1] - m[-2]) / abs(m[-1]) if np.isfinite(m[-2:]).all() and m[-1] != 0 else 0.0
    return dict(params=np.array([ws, hop, B, K, D, numBlocks, L, 10 + len(name)], np.int64), d=np.float64(d), x=x.astype(np.float32),
                y=y, tdoa=tdoa, loc_gap=loc_gap, gccPHATHistory=p.gccPHATHistory.getUnraveledArray(), tdoaHistory=p.tdoaHistory.getUnraveledArray(),
                W_sum=np.float64(W.astype(np.float64).sum()))


def main():
    manifest = {'numpy': np.__version__, 'reference_root': REF_ROOT, 'generated_unix': int(time.time()),
                'theano': 'oracle/theano_stub (NumPy-evaluated stand-in), reference classes imported unmodified', 'files': []}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                                          # nanmean of all-NaN start-up columns
        only = set(sys.argv[1:])                   # names to (re)write; nothing given = everything.  The manifest always lists every case.
        for name, c in FRAME_CASES.items():
            path = os.path.join(OUT, 'rt_frames_%s.npz' % name)
            if only and name not in only and os.path.exists(path):
                manifest['files'].append(os.path.basename(path))
                continue
            np.savez_compressed(path, **frames_case(name, *c))
            manifest['files'].append(os.path.basename(path))
            print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))
        for name, c in STREAM_CASES.items():
            path = os.path.join(OUT, 'rt_stream_%s.npz' % name)
            if only and name not in only and os.path.exists(path):
                manifest['files'].append(os.path.basename(path))
                continue
            r = stream_case(name, *c)
            np.savez_compressed(path, **r)
            manifest['files'].append(os.path.basename(path))
            print('wrote %s (%.1f KB); tdoa track tail %s' % (path, os.path.getsize(path) / 1024.0, r['tdoa'][-5:]))
    with open(os.path.join(OUT, 'RT_MANIFEST.json'), 'w') as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
