"""CPU ORACLE for the GCC-NMF hot path -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

This module is a plain-NumPy restatement of the reference algorithm
(seanwood/gcc-nmf: gccNMF/gccNMFFunctions.py + gccNMF/librosaSTFT.py and the
inline maths of gccNMF/runGCCNMF.py).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the *checker*.  The shipped path (``gcc_nmf_amd``) never
imports it and fails loudly when the HIP library is missing.

Parity pinning: the reference has no tests / golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned against OUTPUTS OF THE
REFERENCE ITSELF: ``oracle/make_golden.py`` imports the unmodified reference
from /root/reference inside the build container, runs it on the reference's
own ``data/*.wav`` fixtures and on synthetic inputs, and commits the results
under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function
here against those vectors (bit-exact for integer outputs; <=1e-6 relative for
floating point -- the differences are BLAS/einsum summation order only).

Every function cites the reference file:line it follows (paths relative to the
reference checkout root).
"""
import numpy as np
import scipy.fftpack
from scipy.signal import argrelmax

# gccNMF/gccNMFFunctions.py:38
SPEED_OF_SOUND_IN_METRES_PER_SECOND = 340.29


# --------------------------------------------------------------------------
# librosaSTFT.py
# --------------------------------------------------------------------------
class ParameterError(Exception):
    """gccNMF/librosaSTFT.py:288-295 (LibrosaError -> ParameterError)."""


def _pad_center(data, size):
    """gccNMF/librosaSTFT.py:354-368 -- centre ``data`` inside ``size`` zeros."""
    n = data.shape[-1]
    lpad = int((size - n) // 2)
    if lpad < 0:
        raise ParameterError('Target size ({:d}) must be at least input size ({:d})'.format(size, n))
    return np.pad(data, [(lpad, size - n - lpad)], mode='constant')


def _window_vector(window, win_length, n_fft):
    """gccNMF/librosaSTFT.py:133-151 / 251-270: callable -> window(win_length);
    vector -> must have n_fft entries; then centre-pad to n_fft."""
    if window is None:
        # reference default path calls scipy.signal.hann, which no longer
        # exists in SciPy >= 1.13; never reached by the hot path.
        raise ParameterError('window=None is not supported (reference default is dead code on modern SciPy)')
    if callable(window):
        w = window(win_length)
    else:
        w = np.asarray(window)
        if w.size != n_fft:
            raise ParameterError('Size mismatch between n_fft and len(window)')
    return _pad_center(w, n_fft)


def valid_audio(y):
    """gccNMF/librosaSTFT.py:476-491."""
    if not isinstance(y, np.ndarray):
        raise ParameterError('data must be of type numpy.ndarray')
    if y.ndim > 2:
        raise ParameterError('Invalid shape for audio: ndim={:d}, shape={}'.format(y.ndim, y.shape))
    if not np.isfinite(y).all():
        raise ParameterError('Audio buffer is not finite everywhere')
    return True


def num_frames(n_samples, frame_length, hop_length):
    """gccNMF/librosaSTFT.py:425 -- the tail that does not fill a frame is dropped."""
    return 1 + int((n_samples - frame_length) / hop_length)


def stft(y, n_fft, hop_length, win_length, window, center=False, dtype=np.complex64):
    """gccNMF/librosaSTFT.py:126-181.

    X[f,t] = conj( sum_n w[n] y[t*hop+n] exp(-2j*pi*f*n/n_fft) ), f=0..n_fft/2.
    Window multiply in float64, FFT in complex128, rounded to ``dtype``.
    """
    if hop_length < 1:
        raise ParameterError('Invalid hop_length: {:d}'.format(hop_length))
    if not y.flags['C_CONTIGUOUS']:
        raise ParameterError('Input buffer must be contiguous.')
    w = _window_vector(window, win_length, n_fft).astype(np.float64)
    if center:
        valid_audio(y)
        y = np.pad(y, int(n_fft // 2), mode='reflect')
    valid_audio(y)
    T = num_frames(len(y), n_fft, hop_length)
    if T < 1:
        raise ParameterError('Buffer is too short (n={:d}) for frame_length={:d}'.format(len(y), n_fft))
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(T)[None, :]
    frames = w[:, None] * y[idx]                      # float64 (n_fft, T)
    spec = scipy.fftpack.fft(frames, axis=0)[:1 + n_fft // 2].conj()
    return np.asfortranarray(spec.astype(dtype))


def istft(stft_matrix, hop_length, win_length, window, center=True, dtype=np.float32):
    """gccNMF/librosaSTFT.py:241-286.

    Per frame: full = [conj(S), S[-2:0:-1]]; ytmp = w * ifft(full).real (ifft in
    the precision of S, window product in float64); overlap-add into a ``dtype``
    buffer in ascending frame order; centre trim of n_fft//2 at both ends.
    """
    n_fft = 2 * (stft_matrix.shape[0] - 1)
    w = _window_vector(window, win_length, n_fft)
    T = stft_matrix.shape[1]
    y = np.zeros(n_fft + hop_length * (T - 1), dtype=dtype)
    full = np.concatenate((stft_matrix.conj(), stft_matrix[-2:0:-1]), axis=0)
    frames = scipy.fftpack.ifft(full, axis=0).real          # keeps single precision for complex64 input
    for t in range(T):
        s = t * hop_length
        y[s:s + n_fft] = y[s:s + n_fft] + w * frames[:, t]
    if center:
        y = y[int(n_fft // 2):-int(n_fft // 2)]
    return y


# --------------------------------------------------------------------------
# gccNMFFunctions.py
# --------------------------------------------------------------------------
def getMaxTDOA(microphoneSeparationInMetres):
    """gccNMF/gccNMFFunctions.py:50-51."""
    return microphoneSeparationInMetres / SPEED_OF_SOUND_IN_METRES_PER_SECOND


def getTDOAsInSeconds(microphoneSeparationInMetres, numTDOAs):
    """gccNMF/gccNMFFunctions.py:53-56."""
    maxTDOA = getMaxTDOA(microphoneSeparationInMetres)
    return np.linspace(-maxTDOA, maxTDOA, numTDOAs)


def getFrequenciesInHz(sampleRate, numFrequencies):
    """gccNMF/gccNMFFunctions.py:58-59."""
    return np.linspace(0, sampleRate / 2, numFrequencies)


def computeComplexMixtureSpectrogram(stereoSamples, windowSize, hopSize, windowFunction, fftSize=None):
    """gccNMF/gccNMFFunctions.py:61-67.  NB the reference ignores ``windowFunction``
    and hard-codes numpy.hanning; ``windowSize`` goes to n_fft and ``fftSize`` to
    win_length (:65)."""
    if fftSize is None:
        fftSize = windowSize
    return np.array([stft(np.squeeze(stereoSamples[c]).copy(), windowSize, hopSize, fftSize, np.hanning, center=False)
                     for c in range(2)])


def magnitudeSpectrogramV(complexMixtureSpectrogram):
    """gccNMF/runGCCNMF.py:40 -- V = concatenate(abs(X), axis=-1): (F, 2T) float32."""
    return np.concatenate(np.abs(complexMixtureSpectrogram), axis=-1)


def initKLNMF(numFrequencies, numColumns, dictionarySize, epsilon=1e-16, seedValue=0):
    """gccNMF/gccNMFFunctions.py:70-73 -- legacy MT19937 global seed, W drawn before H,
    float64 -> float32, then ``+ epsilon`` (a no-op in float32 unless the draw is 0)."""
    rs = np.random.RandomState(seedValue)       # == numpy.random.seed(seedValue) + numpy.random.random
    W = rs.random_sample((numFrequencies, dictionarySize)).astype(np.float32) + epsilon
    H = rs.random_sample((dictionarySize, numColumns)).astype(np.float32) + epsilon
    return W, H


def performKLNMF(V, dictionarySize, numIterations, sparsityAlpha, epsilon=1e-16, seedValue=0):
    """gccNMF/gccNMFFunctions.py:69-83 -- KL-divergence multiplicative updates, float32
    throughout; H update uses the old W, the W update uses the NEW H (:76-77);
    unit-L2 atom normalisation with compensating H rescale (:79-81)."""
    W, H = initKLNMF(V.shape[0], V.shape[1], dictionarySize, epsilon, seedValue)
    for _ in range(numIterations):
        H *= np.dot(W.T, V / np.dot(W, H)) / (np.sum(W, axis=0)[:, np.newaxis] + sparsityAlpha + epsilon)
        W *= np.dot(V / np.dot(W, H), H.T) / np.sum(H, axis=1)
        dictionaryAtomNorms = np.sqrt(np.sum(W ** 2, 0))
        W /= dictionaryAtomNorms
        H *= dictionaryAtomNorms[:, np.newaxis]
    return W, H


def spectralCoherence(complexMixtureSpectrogram):
    """gccNMF/runGCCNMF.py:44 -- PHAT-normalised cross spectrum in complex64."""
    X = complexMixtureSpectrogram
    return X[0] * X[1].conj() / np.abs(X[0]) / np.abs(X[1])


def getAngularSpectrogram(spectralCoherenceV, frequenciesInHz, microphoneSeparationInMetres, numTDOAs):
    """gccNMF/gccNMFFunctions.py:85-92 -- A[tau,t] = sum_f Re(C[f,t] exp(-2j pi f tau)).

    The reference materialises a (TDOA,FREQ,TIME) complex128 temporary through
    einsum and sums its real part over FREQ; this restatement contracts over f
    with one complex128 matmul (same products, different summation order;
    difference ~1e-13 relative, checked against the golden vectors)."""
    tdoasInSeconds = getTDOAsInSeconds(microphoneSeparationInMetres, numTDOAs)
    expJOmega = np.exp(np.outer(frequenciesInHz, -(2j * np.pi) * tdoasInSeconds))     # (F, D) c128
    return np.dot(expJOmega.T, spectralCoherenceV.astype(np.complex128)).real


def estimateTargetTDOAIndexesFromAngularSpectrum(angularSpectrum, microphoneSeparationInMetres, numTDOAs, numSources):
    """gccNMF/gccNMFFunctions.py:94-116 -- strict local maxima (scipy argrelmax, order 1,
    edges never qualify), keep the ``numSources`` largest, return them sorted
    ascending as a Python list.  The reference's failure branches are latent
    NameErrors (:104 os, :106 KMeans); here they raise ValueError."""
    peakIndexes = argrelmax(angularSpectrum)[0]
    if not numSources:
        raise ValueError('numSources must be given (the reference KMeans branch is a NameError)')
    sourcePeakIndexes = peakIndexes[np.argsort(angularSpectrum[peakIndexes])[-numSources:]]
    if len(sourcePeakIndexes) != numSources:
        raise ValueError('found %d peaks, need %d' % (len(sourcePeakIndexes), numSources))
    return sorted(sourcePeakIndexes)


def getTargetTDOAGCCNMFs(coherenceV, microphoneSeparationInMetres, numTDOAs, frequenciesInHz, targetTDOAIndexes, W, stereoH):
    """gccNMF/gccNMFFunctions.py:118-135 -- G_i[k,t] = Re sum_f W[f,k] C[f,t] exp(-2j pi f tau_i),
    computed in double precision then rounded to float32.  W is real, so the real
    part commutes with the contraction: G_i = W^T . Re(C * e_i)."""
    numTargets = len(targetTDOAIndexes)
    hypothesisTDOAs = getTDOAsInSeconds(microphoneSeparationInMetres, numTDOAs)
    numChannels, numAtom, numTime = stereoH.shape
    expJOmegaTau = np.exp(np.outer(frequenciesInHz, -(2j * np.pi) * hypothesisTDOAs))
    out = np.empty((numTargets, numAtom, numTime), np.float32)
    W64 = W.astype(np.float64)
    for i, tdoaIndex in enumerate(targetTDOAIndexes):
        gccChunk = coherenceV * expJOmegaTau[:, tdoaIndex][:, np.newaxis]          # (F, T) c128
        out[i] = np.dot(W64.T, gccChunk.real)
    return out


def getTargetCoefficientMasks(targetTDOAGCCNMFs, numTargets):
    """gccNMF/gccNMFFunctions.py:137-143 -- one-hot of nanargmax over the target axis
    (first index wins ties)."""
    nanArgMax = np.nanargmax(targetTDOAGCCNMFs, axis=0)
    masks = np.zeros_like(targetTDOAGCCNMFs)
    for i in range(numTargets):
        masks[i][np.where(nanArgMax == i)] = 1
    return masks


def getTargetSpectrogramEstimates(targetCoefficientMasks, complexMixtureSpectrogram, W, stereoH):
    """gccNMF/gccNMFFunctions.py:145-151 -- S[i,c] = (W . (H_c * M_i)) * exp(1j*angle(X_c))."""
    numTargets = targetCoefficientMasks.shape[0]
    est = np.zeros((numTargets,) + complexMixtureSpectrogram.shape, np.complex64)
    for i, mask in enumerate(targetCoefficientMasks):
        for c, coefficients in enumerate(stereoH):
            est[i, c] = np.dot(W, coefficients * mask)
    return est * np.exp(1j * np.angle(complexMixtureSpectrogram))


def getTargetSignalEstimates(targetSpectrogramEstimates, windowSize, hopSize, windowFunction):
    """gccNMF/gccNMFFunctions.py:153-163 -- istft (center=True default!) of every
    (target, channel), times the gain 2*hop/ws."""
    numTargets, numChannels, numFreq, numTime = targetSpectrogramEstimates.shape
    stftGainFactor = hopSize / float(windowSize) * 2
    out = [[istft(targetSpectrogramEstimates[i, c], hopSize, windowSize, windowFunction)
            for c in range(numChannels)] for i in range(numTargets)]
    return np.array(out) * stftGainFactor


# --------------------------------------------------------------------------
# wavfile.py (int16 <-> float32 conventions only; no file I/O here)
# --------------------------------------------------------------------------
def pcm2float(sig, dtype='float32'):
    """gccNMF/wavfile.py:57-89."""
    sig = np.asarray(sig)
    i = np.iinfo(sig.dtype)
    abs_max = 2 ** (i.bits - 1)
    offset = i.min + abs_max
    return (sig.astype(dtype) - offset) / abs_max


def float2pcm(sig, dtype='int16'):
    """gccNMF/wavfile.py:92-131 -- clip, truncate-cast."""
    sig = np.asarray(sig)
    i = np.iinfo(dtype)
    abs_max = 2 ** (i.bits - 1)
    offset = i.min + abs_max
    return (sig * abs_max + offset).clip(i.min, i.max).astype(dtype)


# --------------------------------------------------------------------------
# the canonical call sequence
# --------------------------------------------------------------------------
def runGCCNMF(stereoSamples, sampleRate, windowSize, hopSize, numTDOAs, microphoneSeparationInMetres,
              numTargets, dictionarySize=128, numIterations=100, sparsityAlpha=0, windowFunction=np.hanning,
              return_intermediates=False):
    """gccNMF/runGCCNMF.py:30-52 with the wav I/O stripped (float32 samples in,
    float32 waveforms out) and dictionarySize/numIterations/sparsityAlpha exposed
    (the reference hard-codes 128/100/0 at :41)."""
    X = computeComplexMixtureSpectrogram(stereoSamples, windowSize, hopSize, windowFunction)
    numChannels, numFrequencies, numTime = X.shape
    frequenciesInHz = np.linspace(0, sampleRate / 2.0, numFrequencies)
    V = magnitudeSpectrogramV(X)
    W, H = performKLNMF(V, dictionarySize, numIterations, sparsityAlpha)
    stereoH = np.array(np.hsplit(H, numChannels))
    C = spectralCoherence(X)
    A = getAngularSpectrogram(C, frequenciesInHz, microphoneSeparationInMetres, numTDOAs)
    meanA = np.mean(A, axis=-1)
    idx = estimateTargetTDOAIndexesFromAngularSpectrum(meanA, microphoneSeparationInMetres, numTDOAs, numTargets)
    G = getTargetTDOAGCCNMFs(C, microphoneSeparationInMetres, numTDOAs, frequenciesInHz, idx, W, stereoH)
    M = getTargetCoefficientMasks(G, numTargets)
    S = getTargetSpectrogramEstimates(M, X, W, stereoH)
    y = getTargetSignalEstimates(S, windowSize, hopSize, windowFunction)
    if return_intermediates:
        return dict(X=X, V=V, W=W, H=H, C=C, A=A, meanA=meanA, idx=[int(i) for i in idx], G=G, M=M, S=S, y=y)
    return y


# --------------------------------------------------------------------------
# synthetic mixtures (SURVEY.md 8d recipe): the generator lives with the product's input
# tooling; re-exported here so that tests / make_golden.py have one name for it
# --------------------------------------------------------------------------
from gcc_nmf_amd.synthetic import synthetic_mixture    # noqa: E402,F401
