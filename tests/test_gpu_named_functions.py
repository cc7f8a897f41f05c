"""-m gpu: the reference-named functions (gcc_nmf_amd/gccNMFFunctions.py), called with HOST arrays in the order of
gccNMF/runGCCNMF.py:36-52, at the reference driver's own parameters against the committed goldens of the unmodified reference --
no reference checkout needed, so nothing here can skip on the GPU box -- plus BASELINE config 3 as written (200 iterations)."""
import numpy as np
import pytest

from conftest import golden, mask_flips
from oracle import gccnmf_oracle as O
from test_gpu_pipeline import TIE_LIMIT, live_gaps, rel

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def run_named_sequence(stereoSamples, sampleRate, windowSize, hopSize, numTDOAs, microphoneSeparationInMetres, numTargets,
                       dictionarySize, numIterations, sparsityAlpha):
    """gccNMF/runGCCNMF.py:36-52 on the replacement module's names, statement by statement."""
    from gcc_nmf_amd.gccNMFFunctions import (computeComplexMixtureSpectrogram, performKLNMF, getAngularSpectrogram,
                                            estimateTargetTDOAIndexesFromAngularSpectrum, getTargetTDOAGCCNMFs,
                                            getTargetCoefficientMasks, getTargetSpectrogramEstimates, getTargetSignalEstimates,
                                            hanning, linspace, concatenate, array, hsplit, mean)
    windowFunction = hanning
    complexMixtureSpectrogram = computeComplexMixtureSpectrogram(stereoSamples, windowSize, hopSize, windowFunction)
    numChannels, numFrequencies, numTime = complexMixtureSpectrogram.shape
    frequenciesInHz = linspace(0, sampleRate / 2.0, numFrequencies)
    V = concatenate(abs(complexMixtureSpectrogram), axis=-1)
    W, H = performKLNMF(V, dictionarySize, numIterations, sparsityAlpha)
    stereoH = array(hsplit(H, numChannels))
    spectralCoherenceV = complexMixtureSpectrogram[0] * complexMixtureSpectrogram[1].conj() \
        / abs(complexMixtureSpectrogram[0]) / abs(complexMixtureSpectrogram[1])
    angularSpectrogram = getAngularSpectrogram(spectralCoherenceV, frequenciesInHz, microphoneSeparationInMetres, numTDOAs)
    meanAngularSpectrum = mean(angularSpectrogram, axis=-1)
    targetTDOAIndexes = estimateTargetTDOAIndexesFromAngularSpectrum(meanAngularSpectrum, microphoneSeparationInMetres, numTDOAs,
                                                                     numTargets)
    targetTDOAGCCNMFs = getTargetTDOAGCCNMFs(spectralCoherenceV, microphoneSeparationInMetres, numTDOAs, frequenciesInHz,
                                             targetTDOAIndexes, W, stereoH)
    targetCoefficientMasks = getTargetCoefficientMasks(targetTDOAGCCNMFs, numTargets)
    targetSpectrogramEstimates = getTargetSpectrogramEstimates(targetCoefficientMasks, complexMixtureSpectrogram, W, stereoH)
    targetSignalEstimates = getTargetSignalEstimates(targetSpectrogramEstimates, windowSize, hopSize, windowFunction)
    return dict(X=complexMixtureSpectrogram, V=V, W=W, H=H, C=spectralCoherenceV, A=angularSpectrogram, meanA=meanAngularSpectrum,
                idx=targetTDOAIndexes, G=targetTDOAGCCNMFs, M=targetCoefficientMasks, S=targetSpectrogramEstimates,
                y=targetSignalEstimates)


@pytest.fixture(params=['copying', 'resident'])
def dropin_mode(request):
    """Default mode (every function uploads its arguments and returns writable arrays) and the opt-in resident mode
    (device copies of returned arrays are reused when the same object comes back as an argument)."""
    from gcc_nmf_amd import gccNMFFunctions as G
    G.set_resident(request.param == 'resident')
    yield request.param
    G.set_resident(False)


@pytest.mark.parametrize('hop,K', [(128, 128), (256, 1024)], ids=['driver-defaults-hop128-K128', 'config2-hop256-K1024'])
def test_named_functions_at_the_reference_driver_parameters(dev1, dropin_mode, hop, K):
    """runGCCNMF.py's own parameters (:56-77: window 1024, hop 128, 128 TDOAs, d = 1 m, 3 targets, K = 128, 100 iterations,
    alpha 0) and BASELINE config 2 (hop 256, K = 1024) on the committed dev1 mixture, host arrays in and out of every named
    function, against the UNMODIFIED reference's outputs: TDOA indexes exact, mean angular spectrum, masks exact outside the
    reference's own near-ties, waveforms <= 1e-5 RMS (bar 1e-4), and at K = 1024 the factors themselves."""
    x, sr = dev1
    g = golden('dev1_female3_liverec_130ms_1m_hop128_K128' if hop == 128 else 'dev1_hop256_K1024')
    r = run_named_sequence(x, sr, 1024, hop, 128, 1.0, 3, K, 100, 0)
    assert r['idx'] == list(g['idx']) and isinstance(r['idx'], list)
    assert np.abs(r['meanA'] - g['meanA']).max() < 1e-3
    assert r['M'].shape == (3, K, r['X'].shape[2]) and r['M'].dtype == np.float32
    assert np.array_equal(r['M'].sum(axis=0), np.ones_like(r['M'][0]))            # one-hot over targets
    flips, worst = mask_flips(np.argmax(r['M'], axis=0), g)
    assert worst < TIE_LIMIT, (flips, worst)
    y = r['y']
    assert y.dtype == np.float32 and y.shape == (3, 2, hop * (r['X'].shape[2] - 1))
    ref = g['y'][:, :, ::8] if 'y' in g.files else g['y_sub']
    rms = np.sqrt(np.mean((y[:, :, ::8].astype(np.float64) - ref) ** 2))
    assert rms < 1e-5, rms
    wh = ''
    if K == 1024:
        sub = int(g['sub'])
        rw, rh = rel(r['W'][:, ::sub], g['W_sub']), rel(r['H'][::sub, :], g['H_sub'])
        assert rw < 1e-4 and rh < 1e-4, (rw, rh)
        assert np.abs(r['X'][:, ::4, ::7] - g['X_sub']).max() < 1e-5 * np.abs(g['X_sub']).max()
        wh = ' W rel %.2e H rel %.2e' % (rw, rh)
    print('named functions (%s), hop %d K=%d: mask flips %d (largest reference gap %.1e) waveform rms %.2e%s'
          % (dropin_mode, hop, K, flips, worst, rms, wh))


def test_named_functions_with_non_default_parameters(dropin_mode):
    """numTDOAs = 64, d = 0.1 m, two targets, sparsityAlpha = 0.2, K = 96 (not a multiple of any tile), 25 iterations through the
    named functions against the live oracle: every intermediate of runGCCNMF.py:36-52."""
    from gcc_nmf_amd.synthetic import synthetic_mixture
    x = synthetic_mixture(11, numSamples=48000, delays=(-3, 2))                    # |tau| <= 4.7 samples at d = 0.1 m
    kw = dict(dictionarySize=96, numIterations=25, sparsityAlpha=0.2)
    o = O.runGCCNMF(x, 16000, 1024, 256, 64, 0.1, 2, return_intermediates=True, **kw)
    assert o['idx'] == [18, 52]
    r = run_named_sequence(x, 16000, 1024, 256, 64, 0.1, 2, 96, 25, 0.2)
    assert r['idx'] == o['idx']
    assert np.abs(r['X'] - o['X']).max() < 1e-5 * np.abs(o['X']).max()
    assert rel(r['W'], o['W']) < 1e-4 and rel(r['H'], o['H']) < 1e-4
    assert r['A'].shape == (64, r['X'].shape[2]) and np.abs(r['meanA'] - o['meanA']).max() < 5e-3
    assert r['G'].shape == o['G'].shape == (2, 96, r['X'].shape[2])
    assert np.abs(r['G'] - o['G']).max() < 1e-4 * np.abs(o['G']).max()
    flipped = np.argmax(r['M'], 0) != np.argmax(o['M'], 0)
    assert (live_gaps(o['G'])[flipped] < TIE_LIMIT).all(), int(flipped.sum())
    if not flipped.any():
        assert np.abs(r['S'] - o['S']).max() < 1e-4 * np.abs(o['S']).max()
    assert r['y'].shape == o['y'].shape == (2, 2, 256 * (r['X'].shape[2] - 1))
    rms = np.sqrt(np.mean((r['y'].astype(np.float64) - o['y']) ** 2))
    assert rms < 1e-6, rms


def test_benchmark_batch_at_200_iterations():
    """BASELINE config 3 AS WRITTEN: 64 synthetic 10 s files, K = 1024, 200 iterations (the drift against the reference's
    arithmetic grows with the iteration count, gccNMFFunctions.py:75-81).  TDOA indexes of all 64 files against the oracle's
    localisation; files 0 and 63 through the oracle's whole pipeline: W / H <= 1e-4, masks exact up to near-ties, waveforms <= 1e-5."""
    from gcc_nmf_amd.engine import GCCNMFEngine
    from gcc_nmf_amd.synthetic import synthetic_batch
    xs = synthetic_batch(0, 64)
    e = GCCNMFEngine(160000, dictionarySize=1024, numIterations=200, batch=64)
    y = e.separate(xs)
    idx = e.get_tdoa_indexes()
    freqs = np.linspace(0, 8000.0, 513)
    for b in range(64):
        X = O.computeComplexMixtureSpectrogram(xs[b], 1024, 256, np.hanning)
        meanA = np.mean(O.getAngularSpectrogram(O.spectralCoherence(X), freqs, 1.0, 128), axis=-1)
        assert idx[b].tolist() == [int(i) for i in O.estimateTargetTDOAIndexesFromAngularSpectrum(meanA, 1.0, 128, 3)], b
    am = e.get_argmax()
    W, H = e.get_WH()
    for b in (0, 63):
        r = O.runGCCNMF(xs[b], 16000, 1024, 256, 128, 1.0, 3, dictionarySize=1024, numIterations=200, return_intermediates=True)
        rw, rh = rel(W[b], r['W']), rel(H[b], r['H'])
        assert rw < 1e-4 and rh < 1e-4, (b, rw, rh)
        flipped = am[b] != np.argmax(r['M'], 0)
        gaps = live_gaps(r['G'])[flipped]
        assert (gaps < TIE_LIMIT).all(), (b, int(flipped.sum()), gaps.max())
        rms = np.sqrt(np.mean((y[b].astype(np.float64) - r['y']) ** 2))
        assert rms < 1e-5, (b, rms)
        print('bench file %d at 200 iterations: W rel %.2e H rel %.2e mask flips %d (largest oracle gap %.1e) waveform rms %.2e'
              % (b, rw, rh, int(flipped.sum()), gaps.max() if len(gaps) else 0, rms))


def test_returned_arrays_own_their_page_locked_blocks(dropin_mode):
    """The named functions return arrays BASED on pooled page-locked blocks (gcc_nmf_amd/_staging.py): a block goes back to the pool only when
    the last array (or view) over it is gone -- so an earlier result must stay intact through any number of later calls, in both modes, and
    a view keeps its block alive after the array it was taken from is dropped."""
    import gc
    from gcc_nmf_amd import gccNMFFunctions as G
    from gcc_nmf_amd.synthetic import synthetic_mixture
    x1, x2 = synthetic_mixture(31, numSamples=30000), synthetic_mixture(32, numSamples=30000)
    X1 = G.computeComplexMixtureSpectrogram(x1, 1024, 256, np.hanning)
    keep = X1.copy()
    row = X1[1, 100]                                   # a view: the only reference to the block once X1 is dropped
    row_keep = row.copy()
    assert X1.flags.writeable == (dropin_mode == 'copying')
    for _ in range(4):
        X2 = G.computeComplexMixtureSpectrogram(x2, 1024, 256, np.hanning)
        assert np.array_equal(X1, keep) and not np.array_equal(X2, keep)
    del X1
    gc.collect()
    for _ in range(4):
        X2 = G.computeComplexMixtureSpectrogram(x2, 1024, 256, np.hanning)
    assert np.array_equal(row, row_keep)
    # the same spectrogram again: same bits whether the block is fresh or recycled
    assert np.array_equal(G.computeComplexMixtureSpectrogram(x1, 1024, 256, np.hanning), keep)


def test_resident_mode_recognises_only_untouched_objects():
    """dropin.install(resident=True): the device image behind a returned array is reused only for THAT object while it is read-only; a copy,
    a view or an array made writable again is uploaded like any other argument -- same results on every route."""
    from gcc_nmf_amd import gccNMFFunctions as G, _staging
    from gcc_nmf_amd.synthetic import synthetic_mixture
    x = synthetic_mixture(33, numSamples=40000)
    try:
        G.set_resident(True)
        X = G.computeComplexMixtureSpectrogram(x, 1024, 256, np.hanning)
        V = np.concatenate(abs(X), axis=-1)
        W, H = G.performKLNMF(V, 48, 8, 0)
        stereoH = np.array(np.hsplit(H, 2))
        assert not X.flags.writeable and not W.flags.writeable
        dev = G._device()
        assert _staging.lookup(X, 'X', dev) is not None and _staging.lookup(W, 'W', dev) is not None
        assert _staging.lookup(X.copy(), 'X', dev) is None and _staging.lookup(X[:], 'X', dev) is None and _staging.lookup(W, 'X', dev) is None
        M = (np.random.RandomState(0).rand(3, 48, X.shape[2]) > 0.5).astype(np.float32)
        S_res = G.getTargetSpectrogramEstimates(M, X, W, stereoH)                       # X and W from their device images
        S_up = G.getTargetSpectrogramEstimates(M, X.copy(), W.copy(), stereoH)          # everything uploaded
        # (the resident route takes |X| from the STFT epilogue, the uploaded one from gccnmf_magnitude: the same hypotf)
        assert np.array_equal(S_res, S_up)
        Wc = W.copy()
        Wc[:, 0] = 0                                                                      # a modified copy is just another array
        S_mod = G.getTargetSpectrogramEstimates(M, X, Wc, stereoH)
        assert not np.array_equal(S_mod, S_res)
        y_res = G.getTargetSignalEstimates(S_res, 1024, 256, np.hanning)
        y_up = G.getTargetSignalEstimates(S_res.copy(), 1024, 256, np.hanning)
        assert np.array_equal(y_res, y_up)
        G.set_resident(False)
        assert _staging.lookup(X, 'X', dev) is None and not _staging._RESIDENT
        assert G.computeComplexMixtureSpectrogram(x, 1024, 256, np.hanning).flags.writeable
    finally:
        G.set_resident(False)


def test_magnitude_entry_point():
    """gccnmf_magnitude: V = concatenate(abs(X), axis=-1) (runGCCNMF.py:40) from a spectrogram already on the device, the same hypotf as the STFT epilogue."""
    from gcc_nmf_amd import _hip
    from gcc_nmf_amd.engine import GCCNMFEngine
    from gcc_nmf_amd.synthetic import synthetic_batch
    e = GCCNMFEngine(30000, dictionarySize=16, numIterations=1, batch=3)
    e.upload(synthetic_batch(50, 3, numSamples=30000))
    e.stft()
    V = torch.zeros_like(e.V)
    assert _hip.lib().gccnmf_magnitude(e.X.data_ptr(), e.g.F, e.g.T, 3, V.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(V, e.V)
    X = e.get_X()
    assert np.abs(V[:, :e.g.F, :e.g.N].cpu().numpy() - np.concatenate([np.abs(X[:, 0]), np.abs(X[:, 1])], axis=-1)).max() < 1e-6 * np.abs(X).max()
