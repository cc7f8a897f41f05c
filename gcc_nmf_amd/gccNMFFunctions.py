"""Drop-in replacement for the reference's ``gccNMF/gccNMFFunctions.py`` on MI355X.

Same function names, positional order, defaults, return shapes and dtypes as the
reference (each docstring cites the reference lines it replaces); NumPy arrays in,
NumPy arrays out.  The arithmetic of every function on the hot path runs in
libgccnmf_hip.so (hand-written gfx950 kernels, C ABI in include/gccnmf_hip.h);
there is no CPU fallback -- without the library or a GPU the functions raise
``HipLibraryError``.  Each call uploads its arguments and downloads its result;
use ``gcc_nmf_amd.engine.GCCNMFEngine`` to keep a whole batch resident in HBM.

The module-level NumPy names below are part of the interface: the reference's
driver does ``from gccNMFFunctions import *`` and then uses ``hanning``,
``linspace``, ``float32``, ``concatenate``, ``array``, ``hsplit``, ``mean`` ...
without importing them (gccNMF/runGCCNMF.py:27-46).
"""
import logging
from os.path import basename, join

import numpy as np
import torch
from numpy import hanning, array, squeeze, arange, concatenate, sqrt, sum, dot, newaxis, linspace, \
    exp, outer, pi, einsum, argsort, mean, hsplit, zeros, empty, min, max, isnan, all, nanargmax, empty_like, \
    where, zeros_like, angle, arctan2, int16, float32, complex64, argmax, take
from numpy.random import random, seed
from scipy.signal import argrelmax

from . import _hip, _staging
from .engine import Geometry, padded, fft_twiddles, steering_tables, _ptr, _stream
from .librosaSTFT import stft, istft, ParameterError, _window_vector, _istft_device
from .wavfile import wavread, wavwrite

SPEED_OF_SOUND_IN_METRES_PER_SECOND = 340.29


def _device():
    if not torch.cuda.is_available():
        raise _hip.HipLibraryError('no ROCm device visible: gcc_nmf_amd has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


# ---- pass-throughs (gccNMF/gccNMFFunctions.py:40-59) ------------------------------------------------
def getMixtureFileName(mixtureFileNamePrefix):
    return mixtureFileNamePrefix + '_mix.wav'


def getSourceEstimateFileName(mixtureFileNamePrefix, targetIndex):
    return mixtureFileNamePrefix + '_sim_%d.wav' % (targetIndex + 1)


def loadMixtureSignal(mixtureFileName):
    return wavread(mixtureFileName)


def getMaxTDOA(microphoneSeparationInMetres):
    return microphoneSeparationInMetres / SPEED_OF_SOUND_IN_METRES_PER_SECOND


def getTDOAsInSeconds(microphoneSeparationInMetres, numTDOAs):
    maxTDOA = getMaxTDOA(microphoneSeparationInMetres)
    return linspace(-maxTDOA, maxTDOA, numTDOAs)


def getFrequenciesInHz(sampleRate, numFrequencies):
    return linspace(0, sampleRate / 2, numFrequencies)


# ---- hot path ------------------------------------------------------------------------------------------
def set_resident(on):
    """Opt-in resident mode (``dropin.install(resident=True)``): every array a named function returns comes back READ-ONLY and the
    device image behind it is kept while the array lives; when the same object is passed to a later named function (X, W, the scores,
    the masks, the spectrogram estimates in runGCCNMF.py:36-52) its re-upload is skipped.  Default (off): writable outputs, every
    argument uploaded.  Same kernels, same results either way."""
    return _staging.set_resident(on)


def _trig_table(frequenciesInHz, microphoneSeparationInMetres, numTDOAs, g, dev):
    f = np.ascontiguousarray(frequenciesInHz, dtype=np.float64)
    key = ('trig', f.tobytes(), float(microphoneSeparationInMetres), int(numTDOAs), g.Fp, g.Dp)
    return _staging.constant(key, lambda: steering_tables(f, getTDOAsInSeconds(microphoneSeparationInMetres, numTDOAs), g.Fp, g.Dp), dev)


def computeComplexMixtureSpectrogram(stereoSamples, windowSize, hopSize, windowFunction, fftSize=None):
    """gccNMF/gccNMFFunctions.py:61-67.  Like the reference, ``windowFunction`` is ignored
    (numpy.hanning is hard-coded at :65), ``windowSize`` is the FFT length and ``fftSize`` the
    window length.  Returns (2, F, T) complex64.  Both channels share one packed complex FFT."""
    if fftSize is None:
        fftSize = windowSize
    from .librosaSTFT import _stft_device
    chans = [np.ascontiguousarray(np.squeeze(stereoSamples[c])) for c in range(2)]       # (:64 copies each channel too)
    return _stft_device(chans[0], chans[1], windowSize, hopSize, fftSize, hanning, center=False, remember=True)


def performKLNMF(V, dictionarySize, numIterations, sparsityAlpha, epsilon=1e-16, seedValue=0):
    """gccNMF/gccNMFFunctions.py:69-83.  The initial W, H come from NumPy's GLOBAL legacy
    MT19937 exactly as in the reference (seed(seedValue); W first; :70-73) -- including the
    side effect on the global RNG state; the iteration loop (:75-81) runs on the GPU."""
    V = np.asarray(V)
    F, N = V.shape
    K = int(dictionarySize)
    lib, dev = _hip.lib(), _device()
    if N > LARGE_N_COLUMNS:
        return _performKLNMF_column_blocks(V, K, int(numIterations), float(sparsityAlpha), float(epsilon), seedValue, dev)
    g = Geometry(F, 1, K)
    Np = -(-N // 64) * 64
    # The initial factors depend on (seedValue, F, K, N, epsilon) only, and so does the state the reference leaves the GLOBAL generator
    # in (seeded, advanced by F*K + K*N draws).  Drawn once per shape: later calls copy the device images and put the generator into
    # that same state (1.8 M MT19937 draws and 7 MB of upload are 5 ms of a 15 ms call at K = 1024).
    init_key = (F, N, K, repr(seedValue), float(epsilon), dev.index)
    init = _KLNMF_INIT.get(init_key) if seedValue is not None else None      # seed(None) draws fresh entropy: never cached
    if init is None:
        seed(seedValue)
        W = random((F, K)).astype(float32) + epsilon
        H = random((K, N)).astype(float32) + epsilon
        init = dict(W=torch.from_numpy(W.astype(float32)).to(dev), H=torch.from_numpy(H.astype(float32)).to(dev), state=np.random.get_state())
        if seedValue is not None:
            while len(_KLNMF_INIT) >= 4:
                _KLNMF_INIT.pop(next(iter(_KLNMF_INIT)))
            _KLNMF_INIT[init_key] = init
    else:
        np.random.set_state(init['state'])
    with _staging.Scope(dev) as sc:
        # pooled padded buffers: the padding is zero and stays zero (uploads write the corner, the kernels never write beyond it)
        dV = sc.dev('V', (g.Fp, Np), corner=(F, N))
        dW = sc.dev('W', (g.Fp, g.Kp), corner=(F, K))
        dH = sc.dev('H', (g.Kp, Np), corner=(K, N))
        ws = sc.dev('ws_klnmf', (lib.gccnmf_klnmf_workspace_floats(F, N, K, 1),))
        dV[:F, :N].copy_(sc.upload(V, 'V', float32))
        dW[:F, :K].copy_(init['W'])
        dH[:K, :N].copy_(init['H'])
        _hip.check(lib.gccnmf_klnmf(_ptr(dV), _ptr(dW), _ptr(dH), _ptr(ws), F, N, K, 1, int(numIterations),
                                    float(sparsityAlpha), float(epsilon), 0, _stream()), 'gccnmf_klnmf')
        W = sc.remember(sc.download(dW[:F, :K]), 'W', dict(W=dW), dict(F=F, K=K))
        H = sc.download(dH[:K, :N])
    return W, H


_KLNMF_INIT = {}             # (F, N, K, seed, epsilon, device) -> device images of the initial factors + the generator state, a few shapes

# One BIG matrix (the dictionary pre-training set, gccNMF/realtime/gccNMFPretraining.py:79-80: performKLNMF on thousands of frames):
# beyond this many columns the launch fills the chip by itself, and the columns are handed to the batched throughput kernels IN PLACE
# as column blocks of one matrix (gccnmf_klnmf_shared_run with ld > 0 -- the machinery of the time-sharded mode, one rank, no
# collective) instead of the one-mixture latency path.  Same update (gccNMFFunctions.py:75-81): W's numerator sums over all blocks.
LARGE_N_COLUMNS = 4096


def _performKLNMF_column_blocks(V, K, numIterations, sparsityAlpha, epsilon, seedValue, dev):
    from .distributed import HipSharedColumns
    F, N = V.shape
    g = Geometry(F, 1, K)
    ld = -(-N // 64) * 64
    seed(seedValue)                                     # the reference's draws, W before H, and its side effect on the global generator
    W0 = random((F, K)).astype(float32) + epsilon
    H0 = random((K, N)).astype(float32) + epsilon
    with torch.cuda.device(dev):
        Vd = padded(np.ascontiguousarray(V, dtype=float32), (g.Fp, ld), dev)
        Wd = padded(W0.astype(float32), (g.Fp, g.Kp), dev)
        Hd = padded(H0.astype(float32), (g.Kp, ld), dev)
        run = HipSharedColumns(Vd, Hd, Wd, F, N, K, sparsityAlpha, epsilon)
        run.run(numIterations, collective=False)        # this call's columns only, whatever process group the caller may have set up
        return Wd[:F, :K].cpu().numpy(), Hd[:K, :N].cpu().numpy()


def _upload_coherence(sc, C, g):
    """(F, T) complex coherence -> the padded [2][Fp][Tp] Re / Im planes the angular and score GEMMs read (one contiguous upload, the
    de-interleave happens on the device)."""
    F, T = C.shape
    dC = sc.dev('CC', (2, g.Fp, g.Tp), corner=(F, T))
    up = sc.upload(C, 'C', complex64)
    dC[:, :F, :T].copy_(torch.view_as_real(up).permute(2, 0, 1))
    return dC


def _device_W(sc, W, g, dev):
    """Padded device image of a dictionary: the one behind the array if performKLNMF returned it (resident mode), else an upload."""
    rec = _staging.lookup(W, 'W', dev)
    if rec is not None and rec.meta == dict(F=g.F, K=g.K):
        return rec.tensors['W']
    dW = sc.dev('W', (g.Fp, g.Kp), corner=(g.F, g.K))
    dW[:g.F, :g.K].copy_(sc.upload(W, 'W', float32))
    return dW


def getAngularSpectrogram(spectralCoherenceV, frequenciesInHz, microphoneSeparationInMetres, numTDOAs):
    """gccNMF/gccNMFFunctions.py:85-92.  Returns (numTDOAs, T) float64 like the reference; the
    contraction itself is an f32 MFMA GEMM [cos;sin]^T.[Re C;Im C]."""
    C = np.asarray(spectralCoherenceV)
    F, T = C.shape
    lib, dev = _hip.lib(), _device()
    g = Geometry(F, T, 1, int(numTDOAs))
    trig = _trig_table(frequenciesInHz, microphoneSeparationInMetres, numTDOAs, g, dev)
    with _staging.Scope(dev) as sc:
        dC = _upload_coherence(sc, C, g)
        ang = sc.dev('ang', (g.Dp, g.Tp))
        _hip.check(lib.gccnmf_angular_spectrogram(_ptr(dC), _ptr(trig), F, T, g.D, 1, _ptr(ang), 0, _stream()),
                   'gccnmf_angular_spectrogram')
        out = sc.download(ang[:g.D, :T], dtype=np.float64)
    return out


def estimateTargetTDOAIndexesFromAngularSpectrum(angularSpectrum, microphoneSeparationInMetres, numTDOAs, numSources):
    """gccNMF/gccNMFFunctions.py:94-116: strict local maxima, top ``numSources`` by value, sorted
    ascending.  The reference's failure branches are NameErrors (:104 ``os``, :106 ``KMeans``);
    here they raise ValueError."""
    if not numSources:
        raise ValueError('numSources is required (the reference KMeans branch cannot run: gccNMFFunctions.py:106)')
    spectrum = np.ascontiguousarray(angularSpectrum, dtype=np.float64)
    D = spectrum.shape[0]
    S = int(numSources)
    lib, dev = _hip.lib(), _device()
    logging.info('numSources provided, taking first %d peaks' % numSources)
    with _staging.Scope(dev) as sc:
        dM = sc.upload(spectrum, 'meanA')
        res = sc.dev('peaks', (S + 1,), torch.int32)                  # [0..S): indexes, [S]: status
        _hip.check(lib.gccnmf_pick_tdoa_peaks(_ptr(dM), D, D, S, 1, res.data_ptr(), res.data_ptr() + 4 * S, _stream()),
                   'gccnmf_pick_tdoa_peaks')
        out = sc.download(res)
    if int(out[S]) != 0:
        raise ValueError("didn't find enough peaks in estimateTargetTDOAIndexesFromAngularSpectrum")
    sourcePeakIndexes = sorted(np.int64(i) for i in out[:S])
    logging.info('Found target TDOAs: %s' % str(sourcePeakIndexes))
    return sourcePeakIndexes


def getTargetTDOAGCCNMFs(coherenceV, microphoneSeparationInMetres, numTDOAs, frequenciesInHz, targetTDOAIndexes, W, stereoH):
    """gccNMF/gccNMFFunctions.py:118-135.  Returns (numTargets, K, T) float32."""
    numTargets = len(targetTDOAIndexes)
    C = np.asarray(coherenceV)
    F, T = C.shape
    numChannels, K, numTime = stereoH.shape
    lib, dev = _hip.lib(), _device()
    g = Geometry(F, T, K, int(numTDOAs), numTargets)
    trig = _trig_table(frequenciesInHz, microphoneSeparationInMetres, numTDOAs, g, dev)
    with _staging.Scope(dev) as sc:
        dC = _upload_coherence(sc, C, g)
        dW = _device_W(sc, W, g, dev)
        dIdx = sc.upload(np.asarray([int(i) for i in targetTDOAIndexes], dtype=np.int32), 'idx')
        ws = sc.dev('ws_scores', (lib.gccnmf_scores_workspace_floats(F, T, numTargets, 1),))
        scores = sc.dev('scores', (g.Kp, numTargets, g.Tp))
        _hip.check(lib.gccnmf_target_scores_masks(_ptr(dC), _ptr(trig), _ptr(dIdx), _ptr(dW), F, T, K, g.D, numTargets, 1,
                                                  _ptr(ws), _ptr(scores), 0, _stream()), 'gccnmf_target_scores_masks')
        out = sc.download(scores[:K, :, :T].permute(1, 0, 2), shape=(numTargets, K, T))
        sc.remember(out, 'G', dict(scores=scores), dict(S=numTargets, K=K, T=T))
    return out


def getTargetCoefficientMasks(targetTDOAGCCNMFs, numTargets):
    """gccNMF/gccNMFFunctions.py:137-143: one-hot of nanargmax over targets (first wins ties);
    only the first ``numTargets`` masks are filled, as in the reference loop."""
    G = np.asarray(targetTDOAGCCNMFs)
    S, K, T = G.shape
    lib, dev = _hip.lib(), _device()
    g = Geometry(2, T, K, 1, S)
    with _staging.Scope(dev) as sc:
        rec = _staging.lookup(G, 'G', dev)
        if rec is not None:
            scores = rec.tensors['scores']
        else:
            scores = sc.dev('scores', (g.Kp, S, g.Tp), corner=(K, S, T))
            scores[:K, :, :T].copy_(sc.upload(G, 'G', float32).permute(1, 0, 2))
        am = sc.dev('argmax', (g.Kp, g.Tp), torch.uint8)
        _hip.check(lib.gccnmf_argmax_targets(_ptr(scores), K, T, S, 1, _ptr(am), _stream()), 'gccnmf_argmax_targets')
        # numpy.nanargmax raises on a slice that is NaN for every target (:138); argument validation, checked where the data is
        allnan = torch.isnan(scores[:K, :, :T]).all(dim=1).any()
        # the reference's loop (:140-142): masks[i][argmax == i] = 1 for i < numTargets -- an exact 0 / 1 format expansion of the arg-max image
        targets = torch.arange(S, device=dev, dtype=torch.uint8)
        targets[int(numTargets):] = 255
        onehot = am[:K, :T].unsqueeze(0) == targets.view(S, 1, 1)
        masks = sc.download(onehot, shape=(S, K, T), dtype=G.dtype if G.dtype in (np.float32, np.float64) else np.float32)
        bad = sc.download(allnan.view(1).to(torch.uint8))
        if int(numTargets) >= S:
            sc.remember(masks, 'M', dict(argmax=am), dict(S=S, K=K, T=T))
    if bad[0]:
        raise ValueError('All-NaN slice encountered')          # numpy.nanargmax behaviour (:138)
    return masks


def getTargetSpectrogramEstimates(targetCoefficientMasks, complexMixtureSpectrogram, W, stereoH):
    """gccNMF/gccNMFFunctions.py:145-151.  Returns (numTargets, 2, F, T) complex64."""
    M = np.asarray(targetCoefficientMasks)
    X = np.asarray(complexMixtureSpectrogram)
    S, K, T = M.shape
    C, F, _ = X.shape
    if C != 2:
        raise ValueError('stereo spectrogram expected')
    lib, dev = _hip.lib(), _device()
    g = Geometry(F, T, K, 1, S)
    with _staging.Scope(dev) as sc:
        recM = _staging.lookup(M, 'M', dev)
        if recM is not None and recM.meta == dict(S=S, K=K, T=T):
            dA, dM = recM.tensors['argmax'], None                       # the arg-max image the masks were expanded from
        else:
            dA, dM = None, sc.dev('masks', (S, g.Kp, g.Tp), corner=(K, T))
            dM[:, :K, :T].copy_(sc.upload(M, 'M', float32))
        recX = _staging.lookup(X, 'X', dev)
        if recX is not None and recX.meta == dict(F=F, T=T):
            dX, dV = recX.tensors['X'], recX.tensors['V']
        else:
            dX = sc.dev('X', (2, g.Fp, g.Tp, 2), corner=(F, T))
            torch.view_as_complex(dX)[:, :F, :T].copy_(sc.upload(X, 'X', complex64))
            dV = sc.dev('V', (g.Fp, g.Np), corner=(F, g.N))
            _hip.check(lib.gccnmf_magnitude(_ptr(dX), F, T, 1, _ptr(dV), _stream()), 'gccnmf_magnitude')
        dW = _device_W(sc, W, g, dev)
        dH = sc.dev('H', (g.Kp, g.Np), corner=(K, g.N))
        dH[:K, :g.N].unflatten(1, (2, T)).copy_(sc.upload(np.asarray(stereoH), 'stereoH', float32).permute(1, 0, 2))   # (K, [L | R])
        ws = sc.dev('ws_rec', (lib.gccnmf_reconstruct_workspace_floats(T, K, S, 1),))
        spec = sc.dev('spec', (2 * S, g.Fp, g.Tp, 2))
        _hip.check(lib.gccnmf_reconstruct(_ptr(dW), _ptr(dH), _ptr(dA), _ptr(dM), _ptr(dX), _ptr(dV), F, T, K, S, 1, _ptr(ws),
                                          _ptr(spec), _stream()), 'gccnmf_reconstruct')
        out = sc.download(torch.view_as_complex(spec)[:, :F, :T], shape=(S, 2, F, T))
        sc.remember(out, 'S', dict(spec=spec), dict(nsig=2 * S, F=F, T=T))
    return out


def getTargetSignalEstimates(targetSpectrogramEstimates, windowSize, hopSize, windowFunction):
    """gccNMF/gccNMFFunctions.py:153-163: istft (center=True) of every (target, channel) times
    2*hop/ws.  Returns ndarray (numTargets, numChannels, hop*(T-1)) float32."""
    S4 = np.asarray(targetSpectrogramEstimates)
    numTargets, numChannels, numFreq, numTime = S4.shape
    stftGainFactor = hopSize / float(windowSize) * 2
    rec = _staging.lookup(S4, 'S', _device())
    if rec is not None and rec.meta != dict(nsig=numTargets * numChannels, F=numFreq, T=numTime):
        rec = None
    y = _istft_device(S4.reshape(numTargets * numChannels, numFreq, numTime), hopSize, windowSize, windowFunction,
                      center=True, gain=stftGainFactor, device_spec=None if rec is None else rec.tensors['spec'])
    return y.reshape(numTargets, numChannels, -1)


def saveTargetSignalEstimates(targetSignalEstimates, sampleRate, mixtureFileNamePrefix):
    """gccNMF/gccNMFFunctions.py:165-169."""
    numTargets = targetSignalEstimates.shape[0]
    for targetIndex in range(numTargets):
        wavwrite(targetSignalEstimates[targetIndex], getSourceEstimateFileName(mixtureFileNamePrefix, targetIndex), sampleRate)


# ---- named by BASELINE.json's north_star; not in the reference -----------------
def getTargetTDOAEstimates(complexMixtureSpectrogram, sampleRate, microphoneSeparationInMetres, numTDOAs, numSources):
    """Convenience wrapper over the reference's three-step TDOA estimation
    (runGCCNMF.py:44-47): coherence -> getAngularSpectrogram -> time mean ->
    estimateTargetTDOAIndexesFromAngularSpectrum.  Returns (targetTDOAIndexes, meanAngularSpectrum)."""
    X = np.asarray(complexMixtureSpectrogram).astype(complex64)
    C, F, T = X.shape
    lib, dev = _hip.lib(), _device()
    g = Geometry(F, T, 1, int(numTDOAs), int(numSources))
    frequenciesInHz = linspace(0, sampleRate / 2.0, F)
    trig = torch.from_numpy(steering_tables(frequenciesInHz, getTDOAsInSeconds(microphoneSeparationInMetres, numTDOAs),
                                            g.Fp, g.Dp)).to(dev)
    dX = padded(np.ascontiguousarray(X).view(float32).reshape(2, F, T, 2), (2, g.Fp, g.Tp, 2), dev)
    dC = torch.zeros((2, g.Fp, g.Tp), dtype=torch.float32, device=dev)
    ang = torch.zeros((g.Dp, g.Tp), dtype=torch.float32, device=dev)
    meanA = torch.zeros((g.Dp,), dtype=torch.float64, device=dev)
    idx = torch.zeros((g.S,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    _hip.check(lib.gccnmf_coherence(_ptr(dX), F, T, 1, _ptr(dC), _stream()), 'gccnmf_coherence')
    _hip.check(lib.gccnmf_angular_spectrogram(_ptr(dC), _ptr(trig), F, T, g.D, 1, _ptr(ang), _ptr(meanA), _stream()),
               'gccnmf_angular_spectrogram')
    _hip.check(lib.gccnmf_pick_tdoa_peaks(_ptr(meanA), g.D, g.Dp, g.S, 1, _ptr(idx), _ptr(status), _stream()),
               'gccnmf_pick_tdoa_peaks')
    if int(status.cpu()[0]) != 0:
        raise ValueError("didn't find enough peaks in getTargetTDOAEstimates")
    return sorted(np.int64(i) for i in idx.cpu().numpy()), meanA[:g.D].cpu().numpy()
