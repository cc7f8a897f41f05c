"""Streaming (real-time) GCC-NMF on MI355X -- the reference's ``gccNMF/realtime/gccNMFProcessor.py`` frame processor
(a Theano graph there) and ``gccNMF/realtime/utils.py`` overlap-add, as one C-ABI call per audio block
(``gccnmf_rt_process_block``, csrc/rt.hip).

``GCCNMFProcessor`` keeps the reference's constructor arguments, attributes and methods
(``processFrames(windowedSamples)``, ``setTargetTDOARange``, ``reset``); ``StreamingGCCNMF`` is the fused
block-in / block-out path (input ring, frames, mask, synthesis, overlap-add and TDOA tracking all on device).
No CPU fallback: without the library / a device the constructor raises ``HipLibraryError``.
"""
import numpy as np
import torch

from . import _hip
from .engine import padded, fft_twiddles, _ptr, _stream, _on_device

SPEED_OF_SOUND_IN_METRES_PER_SECOND = 340.29
TARGET_MODE_BOXCAR = 0                   # gccNMFProcessor.py:35-37
TARGET_MODE_MULTIPLE = 1
TARGET_MODE_WINDOW_FUNCTION = 2


def asymmetricWindows(windowSize, synthesisSize):
    """Low-latency analysis / synthesis window pair (README.md:74-78; construction after Mauler & Martin 2007): a long analysis window
    for spectral resolution, a synthesis window on the last ``synthesisSize`` samples only.  analysis * synthesis is the periodic Hann
    of length synthesisSize, which overlap-adds to 1 at hopSize = synthesisSize / 2: algorithmic latency synthesisSize samples."""
    K, M = int(windowSize), int(synthesisSize) // 2
    if synthesisSize % 2 or not 0 < 2 * M <= K:
        raise ValueError('synthesisSize must be even and at most windowSize')
    n = np.arange(K, dtype=np.float64)
    long_rise = np.sqrt(0.5 * (1.0 - np.cos(2.0 * np.pi * n / (2 * (K - M))))) if K > M else np.ones(K)
    short = 0.5 * (1.0 - np.cos(2.0 * np.pi * (n - (K - 2 * M)) / (2 * M)))
    analysis = np.where(n < K - M, long_rise, np.sqrt(np.maximum(short, 0.0)))
    with np.errstate(divide='ignore', invalid='ignore'):
        synthesis = np.where(n < K - 2 * M, 0.0, np.where(n < K - M, short / long_rise, np.sqrt(np.maximum(short, 0.0))))
    return analysis.astype(np.float32), synthesis.astype(np.float32)


def _device():
    if not torch.cuda.is_available():
        raise _hip.HipLibraryError('no ROCm device visible: gcc_nmf_amd has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


class GCCNMFProcessor(object):
    """gccNMF/realtime/gccNMFProcessor.py:167-275.  ``dictionariesW[dictionaryType][dictionarySize]`` is the (F, K) float32
    dictionary (:241).  ``numTDOAs`` is an attribute the reference receives through its parameter queue before ``reset()``;
    here it is also a constructor keyword."""

    def __init__(self, sampleRate, windowSize, numTimePerChunk, dictionariesW, dictionaryType, dictionarySize, numHUpdates,
                 microphoneSeparationInMetres, localizationEnabled, localizationWindowSize, gccPHATHistory=None, tdoaHistory=None,
                 inputSpectrogramHistory=None, outputSpectrogramHistory=None, coefficientMaskHistories=None, numTDOAs=64,
                 numTDOAHistory=128, analysisWindow=None, synthesisWindow=None):
        # any even size like the reference (numpy.fft.rfft / irfft, gccNMFProcessor.py:202,:231): powers of two from 64 up take the
        # radix-2 LDS transform, every other even size the direct-sum kernels of csrc/rt.hip
        if int(windowSize) != windowSize or int(windowSize) < 4 or int(windowSize) > 4096 or int(windowSize) % 2:
            raise ValueError('windowSize=%r is not supported by the HIP frame processor: an even size from 4 to 4096' % (windowSize,))
        self.lib = _hip.lib()
        self.device = _device()
        self.sampleRate, self.windowSize, self.numTimePerChunk = sampleRate, int(windowSize), int(numTimePerChunk)
        self.dictionariesW, self.dictionaryType, self.dictionarySize = dictionariesW, dictionaryType, dictionarySize
        # The reference accepts numHUpdates and never uses it (:168).  Here 0 (the reference's config default, config.py:73) IS the
        # reference's mask; n > 0 runs n KL-NMF coefficient updates per frame with W fixed (gccNMFFunctions.py:76) -- the "NMF
        # coefficients are inferred frame-by-frame" of README.md:74, for which the checkout holds no code (parity unpinned).
        self.numHUpdates = int(numHUpdates or 0)
        self.microphoneSeparationInMetres = microphoneSeparationInMetres
        self.localizationEnabled, self.localizationWindowSize = localizationEnabled, int(localizationWindowSize)
        self.numTDOAs, self.numTDOAHistory = int(numTDOAs), int(numTDOAHistory)
        # host mirrors for the GUI (gccNMFProcessor.py:180-184): any object with SharedMemoryCircularBuffer's set() (utils.py:34-70);
        # filled after every call from ONE small download of the device state (:211-229)
        self.gccPHATHistory, self.tdoaHistory = gccPHATHistory, tdoaHistory
        self.inputSpectrogramHistory, self.outputSpectrogramHistory = inputSpectrogramHistory, outputSpectrogramHistory
        self.coefficientMaskHistories = coefficientMaskHistories
        self.generation = 0
        self.separationEnabled = True
        self.targetMode = TARGET_MODE_WINDOW_FUNCTION
        self.windowFunction = np.sqrt(np.hamming(self.windowSize).astype(np.float32))[:, np.newaxis]      # :186
        self.synthesisWindowFunction = self.windowFunction
        if analysisWindow is not None:                    # low-latency extension: separate (asymmetric) windows, see asymmetricWindows()
            a = np.asarray(analysisWindow, np.float32)
            sy = np.asarray(synthesisWindow if synthesisWindow is not None else analysisWindow, np.float32)
            if a.shape != (self.windowSize,) or sy.shape != (self.windowSize,):
                raise ValueError('analysisWindow / synthesisWindow must have windowSize samples')
            self.windowFunction, self.synthesisWindowFunction = a[:, np.newaxis], sy[:, np.newaxis]
        self._target_host = np.array([10.0, 2.0, 1.0, 0.0], np.float32)                                   # :195-198
        self.reset()

    # ---- reference API ------------------------------------------------------------------------------------------
    @_on_device
    def reset(self):
        """:233-270 (buildTheanoFunctions): tables and buffers for the current dictionary / TDOA grid."""
        dev = self.device
        self.generation += 1              # every device buffer below is re-allocated: captured graphs that hold the old pointers are stale
        self.W = np.asarray(self.dictionariesW[self.dictionaryType][self.dictionarySize], np.float32)
        self.numFrequencies, self.numAtom = self.W.shape
        if self.numFrequencies != self.windowSize // 2 + 1:
            raise ValueError('dictionary has %d rows, window size %d needs %d' % (self.numFrequencies, self.windowSize, self.windowSize // 2 + 1))
        F, K, D, Tc = self.numFrequencies, self.numAtom, self.numTDOAs, self.numTimePerChunk
        self.Kp, self.Dp = -(-K // 64) * 64, -(-D // 32) * 32
        self.frequenciesInHz = np.linspace(0, self.sampleRate / 2, F).astype(np.float32)                 # :245
        self.maxTDOA = self.microphoneSeparationInMetres / SPEED_OF_SOUND_IN_METRES_PER_SECOND
        self.hypothesisTDOAs = np.linspace(-self.maxTDOA, self.maxTDOA, D).astype(np.float32)            # :247
        self.expJOmegaTau = np.exp(np.outer(self.frequenciesInHz, -(2j * np.pi) * self.hypothesisTDOAs)).astype(np.complex64)
        z = lambda *shape, **kw: torch.zeros(shape, dtype=kw.get('dtype', torch.float32), device=dev)
        self.dW = padded(self.W, (F, self.Kp), dev)
        self.dCos = padded(np.ascontiguousarray(self.expJOmegaTau.real), (F, self.Dp), dev)
        self.dSin = padded(np.ascontiguousarray(-self.expJOmegaTau.imag), (F, self.Dp), dev)
        self.dWindow = torch.from_numpy(np.ascontiguousarray(self.windowFunction[:, 0])).to(dev)
        self.dSynthWindow = torch.from_numpy(np.ascontiguousarray(self.synthesisWindowFunction[:, 0])).to(dev)
        self.dColsum = padded(self.W.sum(axis=0, dtype=np.float32), (self.Kp,), dev)          # sum_f W: denominator of the H update
        self.dHcoef, self.dRv = z(self.Kp, Tc, 2), z(F, Tc, 2)
        N = self.windowSize
        if N >= 64 and N & (N - 1) == 0:
            self.dTwiddle = torch.from_numpy(fft_twiddles(N)).to(dev)
        else:           # the full-circle table of the direct-sum kernels: (cos, sin)(2 pi k / N), float64 on the host like the twiddles
            ang = 2.0 * np.pi * np.arange(N, dtype=np.float64) / N
            self.dTwiddle = torch.from_numpy(np.ascontiguousarray(np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)).reshape(-1)).to(dev)
        # what the host mirrors are computed from lives in ONE block, so that they cost one download per call: X | Y | HMask | gccPHAT | target
        sizes = [2 * F * Tc * 2, 2 * F * Tc * 2, self.Kp * Tc, D * Tc, 4]
        offs = np.concatenate([[0], np.cumsum([-(-n // 4) * 4 for n in sizes])])
        self.dMirror = z(int(offs[-1]))
        part = lambda i, *shape: self.dMirror[int(offs[i]):int(offs[i]) + sizes[i]].view(*shape)
        self.dX, self.dY, self.dC = part(0, 2, F, Tc, 2), part(1, 2, F, Tc, 2), z(F, Tc, 2)
        self.dHMask, self.dArgmax = part(2, self.Kp, Tc), z(self.Kp, Tc, dtype=torch.int32)
        self.dTfMask, self.dGccPhat = z(2, F, Tc), part(3, D, Tc)     # tfMask: [F][Tc] used without coefficient inference, [2][F][Tc] with
        self.dHist, self.dHistPos = z(D, self.numTDOAHistory), z(1, dtype=torch.int32)
        self.dTarget = part(4, 4)
        self.dTarget.copy_(torch.from_numpy(self._target_host))
        self._mirror_offs, self._mirror_sizes, self._mirror_host = offs, sizes, None
        # _target_dirty: a device call ran with the online localisation on since the value was last known on the host
        self._calls, self._target_dirty, self._target_value, self._target_pin = 0, False, float(self._target_host[0]), None
        self.dFramesIn, self.dFramesOut = z(2, Tc, self.windowSize), z(2, Tc, self.windowSize)

    @_on_device
    def setTargetTDOARange(self, targetTDOAIndex, targetTDOAEpsilon, targetTDOABeta, targetTDOANoiseFloor):
        """:272-275"""
        self._target_host = np.array([targetTDOAIndex, targetTDOAEpsilon, targetTDOABeta, targetTDOANoiseFloor], np.float32)
        self.dTarget.copy_(torch.from_numpy(self._target_host))
        self._target_value, self._target_dirty = float(self._target_host[0]), False

    @property
    def targetTDOAIndex(self):
        """The tracked target TDOA index.  Only the online localisation (csrc/rt.hip) changes it on the device, and the masks always use the
        device value -- so after tracking has run and ``localizationEnabled`` is switched off (the reference toggles it at run time,
        gccNMFProcessor.py:112-116), the LAST TRACKED index stays the answer, not what setTargetTDOARange once set.  A device call that ran
        with the localisation on marks the host copy stale; the value is then fetched at most once (from the history mirror when that
        was downloaded anyway, else one 16-byte page-locked copy) -- repeated reads never touch the device."""
        if self._target_dirty:
            if self._target_pin is None:
                self._target_pin = torch.zeros(4, dtype=torch.float32).pin_memory()
            with torch.cuda.device(self.device):
                self._target_pin.copy_(self.dTarget, non_blocking=True)
                torch.cuda.current_stream(self.device).synchronize()
            self._target_value, self._target_dirty = float(self._target_pin[0]), False
        return self._target_value

    @_on_device
    def _call(self, block_in, block_out, in_ring, out_ring, hop, block, frames_mode, out_delay_blocks=2):
        self._calls += 1
        if self.localizationEnabled:
            self._target_dirty = True
        _hip.check(self.lib.gccnmf_rt_process_block_ll(
            _ptr(block_in), _ptr(block_out), _ptr(in_ring), _ptr(out_ring), _ptr(self.dX), _ptr(self.dY), _ptr(self.dC), _ptr(self.dHMask),
            _ptr(self.dArgmax), _ptr(self.dTfMask), _ptr(self.dHist), _ptr(self.dHistPos), _ptr(self.dTarget), _ptr(self.dGccPhat),
            _ptr(self.dW), _ptr(self.dCos), _ptr(self.dSin), _ptr(self.dWindow), _ptr(self.dSynthWindow), _ptr(self.dTwiddle),
            _ptr(self.dColsum), _ptr(self.dHcoef), _ptr(self.dRv), self.windowSize, hop, block,
            self.numAtom, self.Kp, self.numTDOAs, self.Dp, self.numTDOAHistory, int(self.targetMode), int(bool(self.separationEnabled)),
            int(bool(self.localizationEnabled)), self.localizationWindowSize, frames_mode, int(self.numHUpdates), int(out_delay_blocks),
            _stream()), 'gccnmf_rt_process_block_ll')

    @_on_device
    def processFrames(self, windowedSamples):
        """:201-231.  (2, windowSize, Tc) windowed-sample frames -> (2, windowSize, Tc) processed frames (float32)."""
        ws = np.asarray(windowedSamples, np.float32)
        Tc = self.numTimePerChunk
        if ws.shape != (2, self.windowSize, Tc):
            raise ValueError('expected windowedSamples of shape %s, got %s' % ((2, self.windowSize, Tc), ws.shape))
        self.dFramesIn.copy_(torch.from_numpy(np.ascontiguousarray(ws.transpose(0, 2, 1))))
        # hop = windowSize, block = Tc * windowSize describes Tc back-to-back frames to the kernels' start0/step arithmetic
        self._call(None, None, self.dFramesIn, self.dFramesOut, self.windowSize, Tc * self.windowSize, 1)
        out = self.dFramesOut.cpu().numpy().transpose(0, 2, 1)
        self.fill_histories()
        return out

    def wants_histories(self):
        return any(h is not None for h in (self.gccPHATHistory, self.tdoaHistory, self.inputSpectrogramHistory,
                                           self.outputSpectrogramHistory)) or bool(self.coefficientMaskHistories)

    @_on_device
    def fill_histories(self):
        """The reference's history updates (:211-229) from the state the last device call left: one download of dMirror, then its own
        NumPy expressions on X, Y, HMask and gccPHAT.  The TDOA tracking itself already ran on the device (history ring + arg-max,
        csrc/rt.hip); ``tdoaHistory`` receives the index it picked.  No-op when no history object was given."""
        if not self.wants_histories():
            return
        if self._mirror_host is None:
            self._mirror_host = torch.zeros(self.dMirror.shape, dtype=torch.float32).pin_memory()
        self._mirror_host.copy_(self.dMirror, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        m, o, n = self._mirror_host.numpy(), self._mirror_offs, self._mirror_sizes
        self._target_value, self._target_dirty = float(m[o[4]]), False                  # the tracked index came along
        F, Tc, K, D = self.numFrequencies, self.numTimePerChunk, self.numAtom, self.numTDOAs
        cplx = lambda i: m[o[i]:o[i] + n[i]].reshape(2, F, Tc, 2).copy().view(np.complex64)[..., 0]
        X = cplx(0)
        if self.separationEnabled and self.coefficientMaskHistories:
            self.coefficientMaskHistories[self.dictionarySize].set(1 - m[o[2]:o[2] + n[2]].reshape(self.Kp, Tc)[:K])          # :211-212
        if self.inputSpectrogramHistory is not None:
            self.inputSpectrogramHistory.set(-np.mean(np.abs(X), axis=0) ** (1 / 3.0))                                      # :216-217
        if self.gccPHATHistory is not None:
            self.gccPHATHistory.set(m[o[3]:o[3] + n[3]].reshape(D, Tc).copy())                                                 # :218-219
        if self.tdoaHistory is not None:
            self.tdoaHistory.set(np.array([[np.float32(m[o[4]])]]))                                                            # :220-227
        if self.outputSpectrogramHistory is not None:
            Y = cplx(1) if self.separationEnabled else X                                                                      # :213-214
            with np.errstate(invalid='ignore'):
                self.outputSpectrogramHistory.set(-np.nanmean(np.abs(Y), axis=0) ** (1 / 3.0))                                # :228-229

    # ---- device results of the last call, in the reference's shapes --------------------------------------------------
    @_on_device
    def intermediates(self):
        F, K, D = self.numFrequencies, self.numAtom, self.numTDOAs
        return dict(X=torch.view_as_complex(self.dX).cpu().numpy(), C=torch.view_as_complex(self.dC).cpu().numpy(),
                    HMask=self.dHMask[:K].cpu().numpy(), argmaxTDOA=self.dArgmax[:K].cpu().numpy(),
                    tfMask=(self.dTfMask.cpu().numpy() if self.numHUpdates else self.dTfMask.view(-1)[:F * self.numTimePerChunk].view(F, -1).cpu().numpy()),
                    Hcoef=self.dHcoef[:K].permute(2, 0, 1).contiguous().cpu().numpy(),
                    gccPHAT=self.dGccPhat.cpu().numpy(), targetTDOAIndex=self.targetTDOAIndex)


class StreamingGCCNMF(object):
    """``OverlapAddProcessor.processFrames(GCCNMFProcessor.processFrames)`` (utils.py:99-116) as one device call per block:
    ``process_block((2, blockSize)) -> (2, blockSize)``, output delayed by two blocks like the reference."""

    def __init__(self, processor, hopSize, blockSize, outputDelayBlocks=2, use_graph=True):
        self.use_graph = bool(use_graph)
        self.capture_error = None          # the exception of a failed HIP-graph capture (process_block then launches directly)
        # outputDelayBlocks: 2 = the reference's hand-out (utils.py:116); 1 is complete when the synthesis window spans two hops
        if outputDelayBlocks not in (1, 2, 3, 4, 5, 6, 7):
            raise ValueError('outputDelayBlocks must be 1..7')
        self.outputDelayBlocks = int(outputDelayBlocks)
        # The block handed out is complete only when no later frame adds into it: the synthesis window's non-zero support (first
        # non-zero sample to the end of the frame) must fit in outputDelayBlocks * blockSize + hopSize samples.  2 is the reference's
        # hand-out whatever the window (utils.py:116 -- with its own 512 / 64 low-latency setting it hands out partial sums, and
        # the goldens pin that); any other delay must be complete.
        if self.outputDelayBlocks != 2:
            nz = np.nonzero(np.asarray(processor.synthesisWindowFunction).reshape(-1))[0]
            support = processor.windowSize - int(nz[0]) if len(nz) else 0
            if support > self.outputDelayBlocks * int(blockSize) + int(hopSize):
                raise ValueError('outputDelayBlocks=%d hands a block out before it is complete: the synthesis window spans %d samples, '
                                 'at most %d fit (use asymmetricWindows, or the reference\'s delay of 2)'
                                 % (self.outputDelayBlocks, support, self.outputDelayBlocks * int(blockSize) + int(hopSize)))
        if blockSize % hopSize or blockSize // hopSize != processor.numTimePerChunk:
            raise ValueError('blockSize/hopSize must equal the processor\'s numTimePerChunk')
        if 8 * blockSize < processor.windowSize + (processor.numTimePerChunk - 1) * hopSize:
            raise ValueError('blockSize=%d is not supported: the 8-block buffers (utils.py:87-92) must cover one block\'s windows' % blockSize)
        self.p, self.hopSize, self.blockSize = processor, int(hopSize), int(blockSize)
        dev = processor.device
        self.in_ring = torch.zeros((2, 8 * blockSize), dtype=torch.float32, device=dev)
        self.out_ring = torch.zeros((2, 8 * blockSize), dtype=torch.float32, device=dev)
        self.block_in = torch.zeros((2, blockSize), dtype=torch.float32, device=dev)
        self.block_out = torch.zeros((2, blockSize), dtype=torch.float32, device=dev)

    def process_block_device(self, block_in, block_out):
        """Device tensors in/out, asynchronous on the current stream (what a capture/playback loop would call)."""
        self.p._call(block_in, block_out, self.in_ring, self.out_ring, self.hopSize, self.blockSize, 0, self.outputDelayBlocks)

    def process_block(self, block):
        """Host block in -> host block out.  The finished block is fetched (pinned buffer, its own event) BEFORE the tracking update
        of the next block's target has run: that kernel only writes state the next call reads, so it stays off the latency path.
        The fixed launch sequence (upload, kernels, download) is captured once into a HIP graph and replayed per block."""
        if getattr(self, '_pin_in', None) is None:
            self._pin_in = torch.zeros((2, self.blockSize), dtype=torch.float32).pin_memory()
            self._pin_out = torch.zeros((2, self.blockSize), dtype=torch.float32).pin_memory()
            self._ev_out = torch.cuda.Event()
            self._graph, self._graph_key = None, None
        p = self.p
        with torch.cuda.device(p.device):
            self._pin_in.copy_(torch.from_numpy(np.ascontiguousarray(block, dtype=np.float32)))
            # everything the captured launches depend on besides buffer contents: re-capture when one of them changes
            key = (int(p.targetMode), bool(p.separationEnabled), bool(p.localizationEnabled), int(p.localizationWindowSize),
                   int(p.numHUpdates), p.dW.data_ptr(), p.dTarget.data_ptr(), p.generation)      # reset() re-allocates every buffer
            if self.use_graph and self._graph_key != key:
                self._graph, self._graph_key = self._capture(), key
            if self._graph is not None:
                self._graph.replay()
            else:
                self._launch_front()
            self._ev_out.record()
            p._call(self.block_in, self.block_out, self.in_ring, self.out_ring, self.hopSize, self.blockSize, 4, self.outputDelayBlocks)
            self._ev_out.synchronize()
            out = self._pin_out.numpy().copy()
            p.fill_histories()                 # host mirrors (only when history objects were given): after the tracking update
        return out

    def _launch_front(self):
        self.block_in.copy_(self._pin_in, non_blocking=True)
        self.p._call(self.block_in, self.block_out, self.in_ring, self.out_ring, self.hopSize, self.blockSize, 2, self.outputDelayBlocks)
        self._pin_out.copy_(self.block_out, non_blocking=True)

    def _capture(self):
        """upload -> kernels (all but the tracking update) -> download as one HIP graph.  A failed capture is not silent: it is kept in
        ``capture_error`` and warned about once (the direct launches are correct but ~2x the p99 latency); use_graph=False opts out."""
        try:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            state = [t.clone() for t in (self.in_ring, self.out_ring, self.p.dHist, self.p.dHistPos, self.p.dTarget)]
            with torch.cuda.graph(g):
                self._launch_front()
            # capture does not execute on ROCm, but restore the state anyway in case a runtime runs the body once
            for t, s0 in zip((self.in_ring, self.out_ring, self.p.dHist, self.p.dHistPos, self.p.dTarget), state):
                t.copy_(s0)
            torch.cuda.synchronize()
            self.capture_error = None
            return g
        except Exception as e:
            import warnings
            if getattr(self, 'capture_error', None) is None:
                warnings.warn('StreamingGCCNMF: HIP graph capture failed (%s: %s); falling back to direct launches' % (type(e).__name__, e),
                              RuntimeWarning)
            self.capture_error = e
            return None

    def process_stream(self, stereoSamples):
        """(2, n) -> (2, n_blocks*blockSize); the whole signal is uploaded once, every block is one device call."""
        x = torch.from_numpy(np.ascontiguousarray(stereoSamples, dtype=np.float32)).to(self.p.device)
        B = self.blockSize
        n_blocks = x.shape[1] // B
        xb = x[:, :n_blocks * B].reshape(2, n_blocks, B).permute(1, 0, 2).contiguous()      # [block][2][B]
        out = torch.zeros((n_blocks, 2, B), dtype=torch.float32, device=self.p.device)
        for b in range(n_blocks):
            self.process_block_device(xb[b], out[b])
        return out.permute(1, 0, 2).reshape(2, n_blocks * B).cpu().numpy()
