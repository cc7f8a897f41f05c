"""CPU ORACLE (test infrastructure, not the product) for the real-time GCC-NMF frame processor:
a NumPy restatement of ``gccNMF/realtime/gccNMFProcessor.py:167-275`` (GCCNMFProcessor, whose Theano graph
cannot run here -- Theano is not installable) and ``gccNMF/realtime/utils.py:34-118`` (history ring buffer,
OverlapAddProcessor).

PINNED on a run of the unmodified reference: oracle/make_rt_golden.py imports the reference's GCCNMFProcessor,
OverlapAddProcessor and SharedMemoryCircularBuffer as they are, on top of oracle/theano_stub (a NumPy-evaluated stand-in for
the dozen Theano names the processor uses), and writes tests/golden/rt_*.npz; tests/test_rt_golden.py checks this
restatement against them (spectrogram / coherence / arg-max bit-equal, masks and frames to 2e-6, the tracked TDOA of
60-300 block streams block by block).  tests/test_rt_oracle.py adds an independent brute-force evaluation of the formulas.
The low-latency extensions further down (asymmetric windows, per-frame H inference) have no reference code in this
checkout and are marked "parity unpinned" where they are defined.
"""
import numpy as np
from numpy.fft import rfft, irfft

SPEED_OF_SOUND_IN_METRES_PER_SECOND = 340.29       # gccNMF/defs.py:40
TARGET_MODE_BOXCAR = 0                             # gccNMFProcessor.py:35-37
TARGET_MODE_WINDOW_FUNCTION = 2


def make_rt_dictionary(seed, numFrequencies, dictionarySize):
    """Seeded stand-in for a pre-trained dictionary (positive, unit-norm atoms); the recipe shared by oracle/make_rt_golden.py and
    the tests so that the goldens need not store W."""
    rng = np.random.RandomState(seed)
    W = rng.rand(numFrequencies, dictionarySize).astype(np.float32) + np.float32(0.02)
    return (W / np.linalg.norm(W, axis=0)).astype(np.float32)


def asymmetric_windows(windowSize, synthesisSize):
    """PARITY UNPINNED (no reference code: the low-latency notebook of README.md:74-78 is not in the checkout).  Asymmetric analysis /
    synthesis pair after Mauler & Martin (2007), the construction RT-GCC-NMF's low-latency variant cites: the analysis window rises
    over windowSize - M samples (half of a long periodic Hann, square-rooted) and falls over the last M = synthesisSize / 2 (half of a
    short one); the synthesis window lives on the last 2M samples only and is chosen so that analysis * synthesis is the periodic Hann
    of length 2M -- which overlap-adds to exactly 1 at hop M.  Algorithmic latency 2M samples instead of windowSize."""
    K, M = int(windowSize), int(synthesisSize) // 2
    n = np.arange(K, dtype=np.float64)
    hann = lambda L, m: 0.5 * (1.0 - np.cos(2.0 * np.pi * m / L))
    long_rise = np.sqrt(hann(2 * (K - M), n))
    short = hann(2 * M, n - (K - 2 * M))
    analysis = np.where(n < K - M, long_rise, np.sqrt(np.maximum(short, 0.0)))
    with np.errstate(divide='ignore', invalid='ignore'):
        synthesis = np.where(n < K - 2 * M, 0.0, np.where(n < K - M, short / long_rise, np.sqrt(np.maximum(short, 0.0))))
    return analysis.astype(np.float32), synthesis.astype(np.float32)


class CircularHistory(object):
    """gccNMF/realtime/utils.py:34-70 (SharedMemoryCircularBuffer without the shared memory): float64 ring,
    initialised to 0, ``set`` appends along the last axis, ``getUnraveledArray`` returns chronological order."""

    def __init__(self, shape, initValue=0):
        self.values = np.full(shape, float(initValue), np.float64)
        self.numValues = self.values.shape[-1]
        self.index = 0

    def set(self, newValues):
        n = newValues.shape[-1]
        if self.index + n < self.numValues:
            self.values[..., self.index:self.index + n] = newValues
            self.index += n
        else:
            numAtEnd = self.numValues - self.index
            numAtStart = n - numAtEnd
            self.values[..., self.index:] = newValues[..., :numAtEnd]
            self.values[..., :numAtStart] = newValues[..., numAtEnd:]
            self.index = numAtStart
        return self.index

    def getUnraveledArray(self):
        return np.concatenate([self.values[:, self.index:], self.values[:, :self.index]], axis=-1)


class GCCNMFProcessorOracle(object):
    """gccNMFProcessor.py:167-275.  ``W`` is the (F, K) float32 dictionary the reference picks from
    ``dictionariesW[type][size]`` (:241)."""

    def __init__(self, sampleRate, windowSize, numTimePerChunk, W, microphoneSeparationInMetres, numTDOAs,
                 localizationEnabled=True, localizationWindowSize=6, numTDOAHistory=128, targetMode=TARGET_MODE_WINDOW_FUNCTION,
                 numHUpdates=0, analysisWindow=None, synthesisWindow=None):
        self.sampleRate, self.windowSize, self.numTimePerChunk = sampleRate, windowSize, numTimePerChunk
        self.W = np.asarray(W, np.float32)
        self.numFrequencies, self.numAtom = self.W.shape
        self.numTDOAs = numTDOAs
        self.localizationEnabled, self.localizationWindowSize = localizationEnabled, localizationWindowSize
        self.targetMode = targetMode
        self.separationEnabled = True
        self.windowFunction = np.sqrt(np.hamming(windowSize).astype(np.float32))[:, np.newaxis]      # :186
        self.synthesisWindowFunction = self.windowFunction                                           # :187
        # low-latency extensions (parity unpinned: no reference code): separate windows, per-frame coefficient inference
        if analysisWindow is not None:
            self.windowFunction = np.asarray(analysisWindow, np.float32)[:, np.newaxis]
            self.synthesisWindowFunction = np.asarray(synthesisWindow if synthesisWindow is not None else analysisWindow, np.float32)[:, np.newaxis]
        self.numHUpdates = int(numHUpdates)
        # :195-198 (initial values; the app then calls setTargetTDOARange with the config's 5.0 / 2.0 / 0.0)
        self.targetTDOAIndex, self.targetTDOAEpsilon = np.float32(10.0), np.float32(2.0)
        self.targetTDOABeta, self.targetTDOANoiseFloor = np.float32(1.0), np.float32(0.0)
        # :245-249 -- float32 grids, complex64 steering table
        self.frequenciesInHz = np.linspace(0, sampleRate / 2, self.numFrequencies).astype(np.float32)
        self.maxTDOA = microphoneSeparationInMetres / SPEED_OF_SOUND_IN_METRES_PER_SECOND
        self.hypothesisTDOAs = np.linspace(-self.maxTDOA, self.maxTDOA, numTDOAs).astype(np.float32)
        self.expJOmegaTau = np.exp(np.outer(self.frequenciesInHz, -(2j * np.pi) * self.hypothesisTDOAs)).astype(np.complex64)
        self.gccPHATHistory = CircularHistory((numTDOAs, numTDOAHistory))                            # runRealtimeGCCNMF.py:75

    def setTargetTDOARange(self, targetTDOAIndex, targetTDOAEpsilon, targetTDOABeta, targetTDOANoiseFloor):
        """:272-275"""
        self.targetTDOAIndex, self.targetTDOAEpsilon = np.float32(targetTDOAIndex), np.float32(targetTDOAEpsilon)
        self.targetTDOABeta, self.targetTDOANoiseFloor = np.float32(targetTDOABeta), np.float32(targetTDOANoiseFloor)

    def coefficientMask(self, realGCC):
        """:259-265 -- GCC-NMF scores (TDOA, time, atom), arg-max over TDOA, soft or boxcar window around the target."""
        gccNMF = np.dot(realGCC.T, self.W)                               # (D, Tc, K) float32
        argmaxTDOA = np.argmax(gccNMF, axis=0).T                         # (K, Tc) int64, first maximum wins
        distance = np.abs(argmaxTDOA - self.targetTDOAIndex)             # int64 - float32 -> float64
        if self.targetMode == TARGET_MODE_BOXCAR:
            return np.where(distance < self.targetTDOAEpsilon, 1.0, 0.0), argmaxTDOA
        return np.exp(-(distance / self.targetTDOAEpsilon) ** self.targetTDOABeta) / (1 + self.targetTDOANoiseFloor) + self.targetTDOANoiseFloor, argmaxTDOA

    def processFrames(self, windowedSamples, return_intermediates=False):
        """:201-231.  windowedSamples (2, windowSize, Tc) -> processed frames (2, windowSize, Tc)."""
        X = rfft(windowedSamples * self.windowFunction, axis=1).astype(np.complex64)                # :202 (no conjugate)
        coherenceV = X[0] * X[1].conj() / np.abs(X[0]) / np.abs(X[1])                                # :253
        realGCC = (coherenceV[:, :, np.newaxis] * self.expJOmegaTau[:, np.newaxis]).real             # :254,206 (F, Tc, D)
        if self.separationEnabled:
            HMask, argmaxTDOA = self.coefficientMask(realGCC)                                        # (K, Tc)
            if self.numHUpdates == 0:
                recSource = np.dot(self.W, HMask)                                                    # :267
                recV = np.sum(self.W, axis=-1)                                                       # :268
                tfMask = (recSource.T / recV).T                                                      # :269
                outputSpectrogram = tfMask * X                                                       # :209
            else:
                # PARITY UNPINNED: the reference accepts numHUpdates and never uses it (:168).  Coefficient inference = the H update of
                # gccNMF/gccNMFFunctions.py:76 with W fixed and sparsityAlpha = 0, h0 = 1, per channel and frame on v = |X_c|; with
                # h = 1 the mask below IS the reference's tfMask, so this generalises it: m_c = W (h_c * HMask) / W h_c.
                W64 = self.W.astype(np.float64)
                Hc = np.ones((2, self.numAtom, X.shape[2]))
                for c in range(2):
                    v = np.abs(X[c]).astype(np.float64)
                    for _ in range(self.numHUpdates):
                        Hc[c] *= np.dot(W64.T, v / np.dot(W64, Hc[c])) / np.sum(W64, axis=0)[:, np.newaxis]
                tfMask = np.stack([np.dot(W64, Hc[c] * HMask) / np.dot(W64, Hc[c]) for c in range(2)])
                outputSpectrogram = tfMask * X
        else:
            HMask = argmaxTDOA = tfMask = None
            outputSpectrogram = X.copy()
        gccPHAT = np.nanmean(realGCC, axis=0).T                                                      # :214 (D, Tc)
        self.gccPHATHistory.set(gccPHAT)
        if self.localizationEnabled:                                                                 # :216-222 (takes effect next chunk)
            history = self.gccPHATHistory.getUnraveledArray()
            self.targetTDOAIndex = np.float32(np.argmax(np.nanmean(history[:, -self.localizationWindowSize:], axis=-1)))
        out = irfft(outputSpectrogram, axis=1) * self.synthesisWindowFunction                        # :231
        if return_intermediates:
            return out, dict(X=X, C=coherenceV, HMask=HMask, argmaxTDOA=argmaxTDOA, tfMask=tfMask, gccPHAT=gccPHAT,
                             targetTDOAIndex=float(self.targetTDOAIndex))
        return out


class OverlapAddOracle(object):
    """gccNMF/realtime/utils.py:72-118 with plain arrays instead of the shared-memory frames: 8-block input and output
    buffers, ``windowsPerBlock`` windows cut from the tail of the input buffer, processed frames overlap-added at the same
    positions, and the block that is two blocks old handed out (:116)."""

    def __init__(self, numChannels, windowSize, hopSize, blockSize, windowsPerBlock, outputDelayBlocks=2):
        self.outputDelayBlocks = int(outputDelayBlocks)      # 2 = the reference (utils.py:116); 1 for short synthesis windows (unpinned)
        self.numChannels, self.windowSize, self.hopSize = numChannels, windowSize, hopSize
        self.blockSize, self.windowsPerBlock = blockSize, windowsPerBlock
        self.numBlocksPerBuffer = 8
        self.inputBufferSize = self.outputBufferSize = blockSize * self.numBlocksPerBuffer
        self.inputBuffer = np.zeros((numChannels, self.inputBufferSize), np.float32)
        self.outputBuffer = np.zeros((numChannels, self.outputBufferSize), np.float32)
        self.windowedSamples = np.zeros((numChannels, windowSize, windowsPerBlock), np.float32)

    def processFrames(self, inputFrames, processFramesFunction):
        B = self.blockSize
        self.inputBuffer[:, :-B] = self.inputBuffer[:, B:]
        self.inputBuffer[:, -B:] = inputFrames
        self.outputBuffer[:, :-B] = self.outputBuffer[:, B:]
        self.outputBuffer[:, -B:] = 0
        windowIndexes = np.arange(self.inputBufferSize - self.windowSize - (self.windowsPerBlock - 1) * self.hopSize,
                                  self.inputBufferSize - self.windowSize + 1, self.hopSize)
        for i, windowIndex in enumerate(windowIndexes):
            self.windowedSamples[..., i] = self.inputBuffer[:, windowIndex:windowIndex + self.windowSize]
        processedFrames = processFramesFunction(self.windowedSamples)
        for i, windowIndex in enumerate(windowIndexes):
            self.outputBuffer[:, windowIndex:windowIndex + self.windowSize] += processedFrames[..., i]
        d = self.outputDelayBlocks
        return self.outputBuffer[:, -(d + 1) * B:(-d * B)].copy()


def run_stream(stereoSamples, processor, windowSize, hopSize, blockSize):
    """Feed a (2, n) signal block by block through OverlapAddOracle + processor; returns the (2, n_blocks*blockSize)
    output stream (delayed by two blocks, as in the reference)."""
    ola = OverlapAddOracle(2, windowSize, hopSize, blockSize, blockSize // hopSize)
    n_blocks = stereoSamples.shape[1] // blockSize
    out = np.zeros((2, n_blocks * blockSize), np.float32)
    for b in range(n_blocks):
        out[:, b * blockSize:(b + 1) * blockSize] = ola.processFrames(stereoSamples[:, b * blockSize:(b + 1) * blockSize], processor.processFrames)
    return out
