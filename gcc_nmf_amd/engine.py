"""Device-resident GCC-NMF pipeline for a batch of equally shaped stereo mixtures.

Python here is plumbing only: PyTorch-ROCm owns the HBM allocations and the HIP
stream, every stage is one (or a few) calls into libgccnmf_hip.so through
``_hip`` (C ABI, include/gccnmf_hip.h).  Nothing in this module computes on the
host except the input-independent constant tables (window, FFT twiddles,
steering cos/sin, MT19937 initial W/H), which the reference also derives on the
host (gccNMF/gccNMFFunctions.py:53-59, :70-73, :87-89).

Stage order = gccNMF/runGCCNMF.py:36-52.
"""
import numpy as np
import torch

from . import _hip

SPEED_OF_SOUND_IN_METRES_PER_SECOND = 340.29          # gccNMF/gccNMFFunctions.py:38


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream(device=None):
    """Raw hipStream_t of torch's current stream on `device` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


def _on_device(method):
    """The C library launches on the CURRENT HIP device: make the object's device current for the duration of the call."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        with torch.cuda.device(self.device):
            return method(self, *args, **kwargs)
    return wrapper


def num_frames(n_samples, n_fft, hop):
    """gccNMF/librosaSTFT.py:425."""
    return 1 + int((n_samples - n_fft) / hop)


def klnmf_initial_factors(F, N, K, epsilon=1e-16, seedValue=0):
    """gccNMF/gccNMFFunctions.py:70-73: legacy MT19937, W before H, float64 -> float32, + epsilon."""
    rs = np.random.RandomState(seedValue)
    W = rs.random_sample((F, K)).astype(np.float32) + epsilon
    H = rs.random_sample((K, N)).astype(np.float32) + epsilon
    return W.astype(np.float32), H.astype(np.float32)


def fft_twiddles(n_fft):
    k = np.arange(n_fft // 2, dtype=np.float64)
    tw = np.exp(-2j * np.pi * k / n_fft).astype(np.complex64)
    return np.ascontiguousarray(tw).view(np.float32)


def steering_tables(frequenciesInHz, tdoasInSeconds, Fp, Dp):
    """cos / sin of 2*pi*f*tau, evaluated in float64 like the reference's
    exp(outer(f, -2j*pi*tau)) (gccNMF/gccNMFFunctions.py:89,127), zero padded to [2][Fp][Dp]."""
    E = np.exp(np.outer(np.asarray(frequenciesInHz, np.float64), -(2j * np.pi) * np.asarray(tdoasInSeconds, np.float64)))
    F, D = E.shape
    trig = np.zeros((2, Fp, Dp), np.float32)
    trig[0, :F, :D] = E.real
    trig[1, :F, :D] = -E.imag
    return trig


class Geometry(object):
    """Padded storage geometry; the pitches come from the library so host and kernels cannot disagree."""

    def __init__(self, F, T, K, D=1, S=1):
        import ctypes
        vals = [ctypes.c_int() for _ in range(4)]
        _hip.check(_hip.lib().gccnmf_pitches(F, T, K, *[ctypes.byref(v) for v in vals]), 'gccnmf_pitches')
        self.F, self.T, self.K, self.D, self.S = F, T, K, D, S
        self.N = 2 * T
        self.Fp, self.Kp, self.Np, self.Tp = [v.value for v in vals]
        self.Dp = -(-D // 64) * 64


def padded(host, shape, device, dtype=torch.float32):
    """Upload ``host`` (ndarray) into the top-left corner of a zero tensor of ``shape``."""
    t = torch.zeros(shape, dtype=dtype, device=device)
    src = torch.from_numpy(np.ascontiguousarray(host))
    idx = tuple(slice(0, n) for n in host.shape)
    t[idx] = src.to(device)
    return t


class GCCNMFEngine(object):
    """All buffers of one batch shape, allocated once; ``separate()`` runs the full path on device.

    ``GCCNMFEngine(lengths=[n_0, n_1, ...], ...)`` -- mixtures of DIFFERENT lengths -- returns a ``RaggedGCCNMFEngine``."""

    def __new__(cls, n_samples=None, *args, **kwargs):
        if cls is GCCNMFEngine and kwargs.get('lengths') is not None:
            kwargs = dict(kwargs)
            return RaggedGCCNMFEngine(kwargs.pop('lengths'), *args, **kwargs)
        return super(GCCNMFEngine, cls).__new__(cls)

    def __init__(self, n_samples, sampleRate=16000, windowSize=1024, hopSize=256, numTDOAs=128,
                 microphoneSeparationInMetres=1.0, numTargets=3, dictionarySize=128, numIterations=100,
                 sparsityAlpha=0, epsilon=1e-16, seedValue=0, batch=1, windowFunction=np.hanning,
                 device='cuda:0', klnmf_flags=0, nmf_groups=None):
        if not torch.cuda.is_available():
            raise _hip.HipLibraryError('no ROCm device visible: the GCC-NMF HIP path has no CPU fallback')
        self.lib = _hip.lib()
        self.device = torch.device(device)
        self.n_samples, self.sampleRate = int(n_samples), sampleRate
        self.n_fft, self.hop = int(windowSize), int(hopSize)
        self.iters, self.alpha, self.eps, self.seed = int(numIterations), float(sparsityAlpha), float(epsilon), seedValue
        self.batch = int(batch)
        self.d = microphoneSeparationInMetres
        self.klnmf_flags = klnmf_flags
        # KL-NMF runs per file group, each group on its own stream (the mixtures are independent): the end of one group's
        # launch -- when its last workgroups no longer fill both slots of every CU -- overlaps the start of another
        # group's.  A group must still fill the chip with throughput tiles by itself: >= 16 files.  Bitwise the same result.
        # Measured (64 files, K = 1024): 1 group 287.0 ms per step, 2 groups 278-280 ms; 4 groups 308 ms with the default 4
        # hardware queues (two groups end up sharing one and serialise) and 277 ms with GPU_MAX_HW_QUEUES=8 -- so 2.
        if nmf_groups is None:
            # two groups only when each half alone still makes the library's launch-size decisions exactly as the whole
            # batch would (>= 256 throughput tiles per GEMM launch, >= 256 atom tiles for the fused W update): the same
            # kernels run either way and the outputs are bit-identical
            half = self.batch // 2
            tiles_wh = -(-(2 * num_frames(self.n_samples, self.n_fft, self.hop)) // 64)
            atoms = -(-int(dictionarySize) // 64)
            nmf_groups = 2 if (self.batch % 2 == 0 and half >= 16 and half * tiles_wh >= 256 and half * atoms >= 256) else 1
            # Round 6: where the library runs the whole KL-NMF call as ONE chained launch (gccnmf_klnmf_plan bit 3: tiles handed over
            # between the GEMMs through ready counters, no launch boundaries left to overlap), one group is as fast as two plain ones at
            # 64 files and faster everywhere else (40 files: 134 k -> 156 k frames/s) -- and a second chained launch beside it only costs.
            Fq, Nq = int(windowSize) // 2 + 1, 2 * num_frames(self.n_samples, self.n_fft, self.hop)
            if nmf_groups > 1 and Nq > 0 and self.lib.gccnmf_klnmf_plan(Fq, Nq, int(dictionarySize), self.batch, klnmf_flags) & 8:
                nmf_groups = 1
        if nmf_groups < 1 or self.batch % nmf_groups:
            raise ValueError('nmf_groups must divide the batch')
        self.nmf_groups = int(nmf_groups)
        F = self.n_fft // 2 + 1
        T = num_frames(self.n_samples, self.n_fft, self.hop)
        if T < 2:
            raise ValueError('Buffer is too short (n=%d) for frame_length=%d' % (n_samples, self.n_fft))
        self.g = g = Geometry(F, T, int(dictionarySize), int(numTDOAs), int(numTargets))
        self.L = self.hop * (T - 1)
        dev, B = self.device, self.batch
        f32 = torch.float32

        with torch.cuda.device(dev):
            # constant tables
            self.window = torch.from_numpy(np.asarray(windowFunction(self.n_fft), np.float64).astype(np.float32)).to(dev)
            self.twiddle = torch.from_numpy(fft_twiddles(self.n_fft)).to(dev)
            maxTDOA = self.d / SPEED_OF_SOUND_IN_METRES_PER_SECOND
            self.tdoasInSeconds = np.linspace(-maxTDOA, maxTDOA, g.D)
            self.frequenciesInHz = np.linspace(0, sampleRate / 2.0, F)
            self.trig = torch.from_numpy(steering_tables(self.frequenciesInHz, self.tdoasInSeconds, g.Fp, g.Dp)).to(dev)
            W0, H0 = klnmf_initial_factors(F, g.N, g.K, self.eps, seedValue)
            self.W0 = padded(W0, (g.Fp, g.Kp), dev)
            self.H0 = padded(H0, (g.Kp, g.Np), dev)

            z = lambda *shape: torch.zeros(shape, dtype=f32, device=dev)
            self.x = z(B, 2, self.n_samples)
            self.X = z(B, 2, g.Fp, g.Tp, 2)
            self.V = z(B, g.Fp, g.Np)
            self.CC = z(B, 2, g.Fp, g.Tp)
            self.W = z(B, g.Fp, g.Kp)
            self.H = z(B, g.Kp, g.Np)
            # nmf_groups equal group workspaces (for one group == the whole-batch workspace)
            self.ws_nmf = z(self.nmf_groups * self.lib.gccnmf_klnmf_workspace_floats(F, g.N, g.K, B // self.nmf_groups))
            self.nmf_streams = [torch.cuda.Stream(device=dev) for _ in range(self.nmf_groups)] if self.nmf_groups > 1 else []
            # Copy streams of separate_batches, created ONCE and right behind the group streams.  The runtime maps streams onto a
            # few hardware queues in creation order (4 by default), and a queue is in-order: a 244 MB download that shares its queue
            # with a KL-NMF group holds that group's next kernels back for its 5 ms.  Fresh streams per call walked through torch's
            # pool and landed on the groups' queues every other call (measured: 264 ms per batch instead of 258).
            self.copy_streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
            self.ang = z(B, g.Dp, g.Tp)
            self.mean_ang = torch.zeros((B, g.Dp), dtype=torch.float64, device=dev)
            self.tdoa_idx = torch.zeros((B, g.S), dtype=torch.int32, device=dev)
            self.status = torch.zeros((B,), dtype=torch.int32, device=dev)
            self.ws_scores = z(self.lib.gccnmf_scores_workspace_floats(F, T, g.S, B))
            self.scores = z(B, g.Kp, g.S * g.Tp)
            self.argmax = torch.zeros((B, g.Kp, g.Tp), dtype=torch.uint8, device=dev)
            self.ws_rec = z(self.lib.gccnmf_reconstruct_workspace_floats(T, g.K, g.S, B))
            self.spec = z(B, 2 * g.S, g.Fp, g.Tp, 2)
            # windowed time frames [B][2S][T][n_fft]: only the two-kernel iSTFT needs them (allocated on first use); the default is the
            # fused inverse-transform + overlap-add pass, available while n_fft + 3 * hop <= 2048
            self.frames = None
            # (a fused workgroup transforms 35 frames in sequence: with fewer than ~256 of them the two-kernel form is the faster one --
            # one file: 181 us fused against 42 us)
            self.fused_istft = self.n_fft + 3 * self.hop <= 2048 and B * g.S * -(-T // 32) >= 256
            self.y = z(B, g.S, 2, self.L)
            self.pcm_in = None        # set by upload_pcm16(): the STFT then reads int16 frames directly
            self.pcm_out = None

    # ---- stages (each asynchronous on the current torch stream) ---------------------------------
    @_on_device
    def stft(self):
        g = self.g
        if self.pcm_in is not None:       # int16 interleaved frames straight from the wav data chunk
            _hip.check(self.lib.gccnmf_stft_stereo_pcm16(_ptr(self.pcm_in), self.n_samples, self.n_samples, self.n_fft, self.hop, g.T,
                                                         self.batch, _ptr(self.window), _ptr(self.twiddle), _ptr(self.X), _ptr(self.V),
                                                         _ptr(self.CC), _stream()), 'gccnmf_stft_stereo_pcm16')
            return
        _hip.check(self.lib.gccnmf_stft_stereo(_ptr(self.x), 2 * self.n_samples, self.n_samples, self.n_fft, self.hop, g.T,
                                               self.batch, _ptr(self.window), _ptr(self.twiddle), _ptr(self.X), _ptr(self.V),
                                               _ptr(self.CC), _stream()), 'gccnmf_stft_stereo')

    @_on_device
    def pack_pcm16(self):
        """y -> int16 interleaved [batch][S][L][2] with wavwrite's clip protection per target (wavfile.py:39-48)."""
        g = self.g
        if self.pcm_out is None:
            self.pcm_out = torch.zeros((self.batch, g.S, self.L, 2), dtype=torch.int16, device=self.device)
            self.pcm_peak = torch.zeros((self.batch * g.S,), dtype=torch.int32, device=self.device)
        _hip.check(self.lib.gccnmf_pack_pcm16(_ptr(self.y), self.batch * g.S, self.L, _ptr(self.pcm_peak), _ptr(self.pcm_out), _stream()),
                   'gccnmf_pack_pcm16')

    @_on_device
    def klnmf(self):
        g = self.g
        self.W.copy_(self.W0.unsqueeze(0).expand_as(self.W))
        self.H.copy_(self.H0.unsqueeze(0).expand_as(self.H))
        if self.nmf_groups == 1:
            _hip.check(self.lib.gccnmf_klnmf(_ptr(self.V), _ptr(self.W), _ptr(self.H), _ptr(self.ws_nmf), g.F, g.N, g.K, self.batch,
                                             self.iters, self.alpha, self.eps, self.klnmf_flags, _stream()), 'gccnmf_klnmf')
            return
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        per = self.batch // self.nmf_groups
        ws_per = self.ws_nmf.numel() // self.nmf_groups
        for i, st in enumerate(self.nmf_streams):
            st.wait_event(ready)
            b0 = i * per
            # GCCNMF_FLAG_GROUPS(n) = 4 | n << 8: the groups' launches share the chip, so launch forms are chosen for all groups together (a
            # file's bits do not depend on the split) and each keeps the throughput tile -- its
            # partial last round overlaps the other group's kernels (all-half-height tiles, which win for a 32-file launch ALONE,
            # lose here: 152.4 k vs 155.4 k frames/s)
            _hip.check(self.lib.gccnmf_klnmf(_ptr(self.V[b0]), _ptr(self.W[b0]), _ptr(self.H[b0]), _ptr(self.ws_nmf[i * ws_per:]), g.F, g.N,
                                             g.K, per, self.iters, self.alpha, self.eps, self.klnmf_flags | 4 | (self.nmf_groups << 8), st.cuda_stream), 'gccnmf_klnmf')
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)

    @_on_device
    def localize(self):
        g = self.g
        _hip.check(self.lib.gccnmf_angular_spectrogram(_ptr(self.CC), _ptr(self.trig), g.F, g.T, g.D, self.batch, _ptr(self.ang),
                                                       _ptr(self.mean_ang), _stream()), 'gccnmf_angular_spectrogram')
        _hip.check(self.lib.gccnmf_pick_tdoa_peaks(_ptr(self.mean_ang), g.D, g.Dp, g.S, self.batch, _ptr(self.tdoa_idx),
                                                   _ptr(self.status), _stream()), 'gccnmf_pick_tdoa_peaks')

    @_on_device
    def masks(self):
        g = self.g
        _hip.check(self.lib.gccnmf_target_scores_masks(_ptr(self.CC), _ptr(self.trig), _ptr(self.tdoa_idx), _ptr(self.W), g.F, g.T,
                                                       g.K, g.D, g.S, self.batch, _ptr(self.ws_scores), _ptr(self.scores),
                                                       _ptr(self.argmax), _stream()), 'gccnmf_target_scores_masks')

    @_on_device
    def reconstruct(self):
        g = self.g
        _hip.check(self.lib.gccnmf_reconstruct(_ptr(self.W), _ptr(self.H), _ptr(self.argmax), 0, _ptr(self.X), _ptr(self.V), g.F,
                                               g.T, g.K, g.S, self.batch, _ptr(self.ws_rec), _ptr(self.spec), _stream()),
                   'gccnmf_reconstruct')

    @_on_device
    def istft(self, keep_frames=False):
        """spec -> y.  keep_frames: the two-kernel form that also leaves the windowed time frames in ``self.frames``."""
        g = self.g
        gain = np.float32(self.hop / float(self.n_fft) * 2)           # gccNMFFunctions.py:155
        frames = None
        if keep_frames or not self.fused_istft:
            if self.frames is None:
                self.frames = torch.zeros((self.batch, 2 * g.S, g.T, self.n_fft), dtype=torch.float32, device=self.device)
            frames = self.frames
        _hip.check(self.lib.gccnmf_istft_ola(_ptr(self.spec), 2 * g.S, self.n_fft, self.hop, g.T, self.batch, _ptr(self.window),
                                             _ptr(self.twiddle), gain, 1, _ptr(frames), _ptr(self.y), _stream()),
                   'gccnmf_istft_ola')

    @_on_device
    def run(self, stft=True):
        """samples already in ``self.x`` -> separated waveforms in ``self.y`` (all on device, asynchronous)."""
        if stft:
            self.stft()
        self.klnmf()
        self.localize()
        self.masks()
        self.reconstruct()
        self.istft()

    # ---- host <-> device -------------------------------------------------------------------------
    @_on_device
    def upload(self, stereoSamples):
        x = np.asarray(stereoSamples, dtype=np.float32)
        if x.ndim == 2:
            x = x[None]
        if x.shape != (self.batch, 2, self.n_samples):
            raise ValueError('expected samples of shape %s, got %s' % ((self.batch, 2, self.n_samples), x.shape))
        if not np.isfinite(x).all():
            raise ValueError('Audio buffer is not finite everywhere')      # librosaSTFT.py:488-489
        self.pcm_in = None
        self.x.copy_(torch.from_numpy(np.ascontiguousarray(x)))

    @_on_device
    def upload_pcm16(self, pcm):
        """(batch, n, 2) int16 interleaved stereo frames, exactly as scipy.io.wavfile.read returns them (no host conversion)."""
        pcm = np.asarray(pcm)
        if pcm.ndim == 2:
            pcm = pcm[None]
        if pcm.dtype != np.int16 or pcm.shape != (self.batch, self.n_samples, 2):
            raise ValueError('expected int16 frames of shape %s, got %s %s' % ((self.batch, self.n_samples, 2), pcm.dtype, pcm.shape))
        if self.pcm_in is None:
            self.pcm_in = torch.zeros((self.batch, self.n_samples, 2), dtype=torch.int16, device=self.device)
        self.pcm_in.copy_(torch.from_numpy(np.ascontiguousarray(pcm)))

    @_on_device
    def separate_pcm16(self, pcm):
        """int16 frames in -> int16 frames out: (batch, n, 2) -> (batch, S, hop*(T-1), 2), i.e. loadMixtureSignal ...
        saveTargetSignalEstimates (runGCCNMF.py:35-54) minus the file system, with both wav conversions on the device."""
        self.upload_pcm16(pcm)
        self.run()
        self.pack_pcm16()
        out = self.pcm_out.cpu().numpy()
        self.check_status()
        self.check_pcm_finite()
        return out

    @_on_device
    def separate(self, stereoSamples):
        """(batch, 2, n) float32 host samples -> (batch, S, 2, hop*(T-1)) float32 host waveforms."""
        x = np.asarray(stereoSamples, dtype=np.float32)
        if x.ndim == 2:
            x = x[None]
        if x.shape != tuple(self.x.shape):
            raise ValueError('expected samples of shape %s, got %s' % (tuple(self.x.shape), x.shape))
        if not np.isfinite(x).all():
            raise ValueError('Audio buffer is not finite everywhere')      # librosaSTFT.py:488-489
        # one page-locked staging pair (allocated on first use: 82 MB + 244 MB of host memory for a 64-file batch, nothing extra on the
        # device): both copies move at PCIe speed instead of through pageable bounce buffers (313 -> 285 ms host to host for one
        # batch).  The double-buffered pipeline -- a second x / y pair in HBM, two more pinned pairs -- belongs to separate_batches.
        if getattr(self, '_pin', None) is None:
            self._pin = (torch.zeros(self.x.shape, dtype=torch.float32).pin_memory(), torch.zeros(self.y.shape, dtype=torch.float32).pin_memory())
        hx, hy = self._pin
        hx.copy_(torch.from_numpy(np.ascontiguousarray(x)))
        self.pcm_in = None
        self.x.copy_(hx, non_blocking=True)
        self.run()
        if self.chain_failed():
            # The chained KL-NMF launch did not hand over cleanly (a consumer timed out / a work list ran on more than one XCC): its factors are
            # NaN by construction.  Do not fail the batch: switch this process to the plain launches and run the stages behind the STFT again.
            import warnings
            warnings.warn('gcc_nmf_amd: a chained KL-NMF launch did not hand over cleanly; this process falls back to the plain launches '
                          '(gccnmf_set_tuning(21, 0)) and repeats the batch', RuntimeWarning)
            _hip.check(self.lib.gccnmf_set_tuning(21, 0), 'gccnmf_set_tuning')
            self.run(stft=False)
        hy.copy_(self.y, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        self.check_status()
        return hy.numpy().copy()

    def separate_batches(self, batches):
        """Generator over an iterable of (batch, 2, n) float32 host arrays -> one (batch, S, 2, hop*(T-1)) float32 array per
        input, in order.  Same results as ``separate`` per batch, but the PCIe transfers (pinned staging buffers, their own
        streams) of batch i+1 (up) and i-1 (down) run under the compute of batch i: sustained host-to-host throughput
        approaches the device rate instead of paying both copies per batch."""
        dev = self.device
        g = self.g
        with torch.cuda.device(dev):
            compute = torch.cuda.current_stream(dev)
            s_in, s_out = self.copy_streams
            if getattr(self, '_pipe', None) is None:       # second device buffers + pinned staging: allocated once (page-locking is slow)
                self._pipe = dict(x=torch.zeros_like(self.x), y=torch.zeros_like(self.y),
                                  hx=[torch.zeros(self.x.shape, dtype=torch.float32).pin_memory() for _ in range(2)],
                                  hy=[torch.zeros(self.y.shape, dtype=torch.float32).pin_memory() for _ in range(2)],
                                  hs=[torch.zeros(self.status.shape, dtype=self.status.dtype).pin_memory() for _ in range(2)],
                                  ds=[torch.zeros_like(self.status) for _ in range(2)])
            xs, ys = [self.x, self._pipe['x']], [self.y, self._pipe['y']]
            hx, hy, status = self._pipe['hx'], self._pipe['hy'], self._pipe['hs']
            ev_in = [torch.cuda.Event() for _ in range(2)]
            ev_done = [torch.cuda.Event() for _ in range(2)]
            ev_out = [torch.cuda.Event() for _ in range(2)]
            ev_stft = torch.cuda.Event()
            pending = []                       # slots whose results have not been yielded yet, oldest first
            download = None                    # the deferred download of the batch enqueued last

            def collect(slot):
                ev_out[slot].synchronize()
                st = status[slot].numpy()
                if st.any():
                    raise ValueError('fewer than %d angular-spectrum peaks in file(s) %s' % (g.S, np.nonzero(st)[0].tolist()))
                return hy[slot].numpy().copy()

            try:
                for i, batch in enumerate(batches):
                    slot = i & 1
                    if len(pending) == 2:                                # this slot's previous occupant must be handed out first
                        yield collect(pending.pop(0))
                    x = np.asarray(batch, dtype=np.float32)
                    if x.shape != tuple(self.x.shape):
                        raise ValueError('expected samples of shape %s, got %s' % (tuple(self.x.shape), x.shape))
                    if not np.isfinite(x).all():
                        raise ValueError('Audio buffer is not finite everywhere')      # librosaSTFT.py:488-489
                    hx[slot].copy_(torch.from_numpy(np.ascontiguousarray(x)))    # host memcpy into the pinned buffer
                    with torch.cuda.stream(s_in):
                        s_in.wait_event(ev_done[slot])                   # batch i-2 no longer reads this x buffer
                        xs[slot].copy_(hx[slot], non_blocking=True)
                        ev_in[slot].record(s_in)
                    compute.wait_event(ev_in[slot])
                    compute.wait_event(ev_out[slot])                     # batch i-2's waveforms have left this y buffer
                    self.x, self.y, self.pcm_in = xs[slot], ys[slot], None
                    self.stft()
                    if download is not None:                             # batch i-1 goes down now that this batch's STFT is past
                        ev_stft.record(compute)
                        download(ev_stft)
                    self.run(stft=False)
                    self._pipe['ds'][slot].copy_(self.status)            # per-slot snapshot ON the compute stream: batch i+1's
                    ev_done[slot].record(compute)                        # localize() rewrites self.status before s_out has copied it

                    # The download is a shader copy on this runtime (rocprofv3: __amd_rocclr_copyBuffer, 4.6 ms for 244 MB at PCIe
                    # speed); next to the HBM-bound STFT of the following batch it held that kernel up from 0.6 to 4.9 ms.  So it is
                    # enqueued behind that STFT (or at once for the last batch) and runs in the shadow of the MFMA-bound KL-NMF.
                    def download(after=None, slot=slot):
                        with torch.cuda.stream(s_out):
                            s_out.wait_event(ev_done[slot])
                            if after is not None:
                                s_out.wait_event(after)
                            hy[slot].copy_(ys[slot], non_blocking=True)
                            status[slot].copy_(self._pipe['ds'][slot], non_blocking=True)
                            ev_out[slot].record(s_out)
                    pending.append(slot)
                if download is not None:
                    download()
                while pending:
                    yield collect(pending.pop(0))
            finally:
                torch.cuda.synchronize(dev)
                self.x, self.y = xs[0], ys[0]

    @_on_device
    def check_pcm_finite(self):
        """After pack_pcm16(): a NaN / Inf sample in a waveform shows up in that group's peak image (csrc/fft.hip)."""
        bad = (self.pcm_peak.cpu().numpy().view(np.uint32) >= 0x7F800000).reshape(self.batch, self.g.S)
        if bad.any():
            raise ValueError('non-finite samples in the separated waveforms of file(s) %s' % np.nonzero(bad.any(axis=1))[0].tolist())

    @_on_device
    def check_status(self):
        self.check_chain_status()
        st = self.status.cpu().numpy()
        if st.any():
            raise ValueError('fewer than %d angular-spectrum peaks in file(s) %s' % (self.g.S, np.nonzero(st)[0].tolist()))

    @_on_device
    def chain_failed(self):
        """Status word of the last chained KL-NMF launch(es) of this engine (0 = clean or not chained); synchronises the stream."""
        import ctypes
        g, st, worst = self.g, ctypes.c_int(0), 0
        per = self.batch // self.nmf_groups
        ws_per = self.ws_nmf.numel() // self.nmf_groups
        torch.cuda.current_stream(self.device).synchronize()
        for i in range(self.nmf_groups):
            _hip.check(self.lib.gccnmf_klnmf_chain_status(_ptr(self.ws_nmf[i * ws_per:]), g.F, g.N, g.K, per, ctypes.byref(st)), 'gccnmf_klnmf_chain_status')
            worst |= st.value
        return worst

    @_on_device
    def check_chain_status(self):
        """A chained KL-NMF launch whose hand-over failed has turned W and H into NaN; say so instead of letting NaN travel on."""
        import ctypes
        g, st = self.g, ctypes.c_int(0)
        per = self.batch // self.nmf_groups
        ws_per = self.ws_nmf.numel() // self.nmf_groups
        torch.cuda.current_stream(self.device).synchronize()
        for i in range(self.nmf_groups):
            _hip.check(self.lib.gccnmf_klnmf_chain_status(_ptr(self.ws_nmf[i * ws_per:]), g.F, g.N, g.K, per, ctypes.byref(st)), 'gccnmf_klnmf_chain_status')
            if st.value:
                raise _hip.HipLibraryError('the chained KL-NMF launch did not hand over cleanly (status %d: %s): W and H of this batch are NaN.  '
                                           'GCCNMF_TUNE="21=0" runs the plain launches.'
                                           % (st.value, 'a consumer timed out' if st.value & 1 else 'a work list ran on more than one XCC'))

    # ---- views of device results in the reference's shapes ------------------------------------------
    def get_X(self):
        g = self.g
        return torch.view_as_complex(self.X)[:, :, :g.F, :g.T].cpu().numpy()

    def get_V(self):
        g = self.g
        return self.V[:, :g.F, :g.N].cpu().numpy()

    def get_C(self):
        g = self.g
        c = self.CC[:, :, :g.F, :g.T].cpu().numpy()
        return (c[:, 0] + 1j * c[:, 1]).astype(np.complex64)

    def get_WH(self):
        g = self.g
        return self.W[:, :g.F, :g.K].cpu().numpy(), self.H[:, :g.K, :g.N].cpu().numpy()

    def get_angular(self):
        g = self.g
        return self.ang[:, :g.D, :g.T].cpu().numpy(), self.mean_ang[:, :g.D].cpu().numpy()

    def get_tdoa_indexes(self):
        return self.tdoa_idx.cpu().numpy()

    def get_scores(self):
        g = self.g
        s = self.scores.view(self.batch, g.Kp, g.S, g.Tp)[:, :g.K, :, :g.T]
        return s.permute(0, 2, 1, 3).contiguous().cpu().numpy()

    def get_argmax(self):
        g = self.g
        return self.argmax[:, :g.K, :g.T].cpu().numpy()

    def get_spec(self):
        g = self.g
        s = torch.view_as_complex(self.spec)[:, :, :g.F, :g.T].cpu().numpy()
        return s.reshape(self.batch, g.S, 2, g.F, g.T)


class RaggedGCCNMFEngine(object):
    """A batch of mixtures of DIFFERENT lengths (the reference separates a file of any length per call, gccNMF/runGCCNMF.py:30-36; sharding
    "independent mixture files" over GPUs means files as they come).  ``lengths``: samples per file, in the caller's order.

    KL-NMF -- 98 % of the path -- runs over ALL files in ONE chained launch whose work lists hold each file's own column tiles
    (gccnmf_klnmf_ragged: padding to the 64-column tile only, the files dealt out to the XCDs by length).  The one-shot stages (STFT,
    localisation, masks, reconstruction, iSTFT) run per distinct length, on the ordinary engine of that length.  A file's results are bit
    for bit those it gets in an equal-length batch.  Where the library has no chained form for the shape (GCCNMF_ERR_UNSUPPORTED: short
    dictionaries, a handful of files) the files of each length run their KL-NMF as a batch of their own."""

    def __init__(self, lengths, sampleRate=16000, windowSize=1024, hopSize=256, numTDOAs=128, microphoneSeparationInMetres=1.0,
                 numTargets=3, dictionarySize=128, numIterations=100, sparsityAlpha=0, epsilon=1e-16, seedValue=0,
                 windowFunction=np.hanning, device='cuda:0', klnmf_flags=0):
        if not torch.cuda.is_available():
            raise _hip.HipLibraryError('no ROCm device visible: the GCC-NMF HIP path has no CPU fallback')
        self.lib = _hip.lib()
        self.device = torch.device(device)
        self.lengths = [int(n) for n in lengths]
        if not self.lengths:
            raise ValueError('no files')
        self.batch = len(self.lengths)
        self.iters, self.alpha, self.eps = int(numIterations), float(sparsityAlpha), float(epsilon)
        self.klnmf_flags = klnmf_flags
        kw = dict(sampleRate=sampleRate, windowSize=windowSize, hopSize=hopSize, numTDOAs=numTDOAs,
                  microphoneSeparationInMetres=microphoneSeparationInMetres, numTargets=numTargets, dictionarySize=dictionarySize,
                  numIterations=numIterations, sparsityAlpha=sparsityAlpha, epsilon=epsilon, seedValue=seedValue,
                  windowFunction=windowFunction, device=device, klnmf_flags=klnmf_flags)
        # one ordinary engine per distinct length: its files (caller's indexes, ascending) are its batch
        self.files_of = {}
        for i, n in enumerate(self.lengths):
            self.files_of.setdefault(n, []).append(i)
        self.sub = dict((n, GCCNMFEngine(n, batch=len(idx), nmf_groups=1, **kw)) for n, idx in sorted(self.files_of.items()))
        longest = self.sub[max(self.sub)]
        self.g = g = longest.g                                   # geometry of the longest file: the pitch of every file's V / H block
        self.N = [self.sub[n].g.N for n in self.lengths]       # columns per file
        self.frames = [self.sub[n].g.T for n in self.lengths]
        with torch.cuda.device(self.device):
            z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.device)
            self.ragged = None
            if len(self.sub) > 1:
                import ctypes
                ws = self.lib.gccnmf_klnmf_ragged_workspace_floats(g.F, g.N, g.K, self.batch)
                if ws > 0:
                    self.ragged = dict(V=z(self.batch, g.Fp, g.Np), W=z(self.batch, g.Fp, g.Kp), H=z(self.batch, g.Kp, g.Np), ws=z(ws),
                                       N=(ctypes.c_int * self.batch)(*self.N))
        self.ragged_klnmf_used = None                            # set by run(): True = one ragged launch, False = one call per length

    def klnmf(self):
        """KL-NMF of every file: one ragged chained launch, or -- where the library has none for this shape -- one call per length."""
        with torch.cuda.device(self.device):
            r = self.ragged
            if r is not None:
                g = self.g
                for n, e in self.sub.items():
                    idx = torch.as_tensor(self.files_of[n], device=self.device)
                    r['V'][idx, :, :e.g.Np] = e.V                # (columns beyond a file's own stay zero: never written)
                    r['W'][idx] = e.W0
                    r['H'][idx, :, :e.g.Np] = e.H0
                rc = self.lib.gccnmf_klnmf_ragged(_ptr(r['V']), _ptr(r['W']), _ptr(r['H']), _ptr(r['ws']), g.F, r['N'], g.N, g.K, self.batch,
                                                  self.iters, self.alpha, self.eps, self.klnmf_flags, _stream())
                if rc == 0:
                    self._ragged_ran = True
                    for n, e in self.sub.items():
                        idx = torch.as_tensor(self.files_of[n], device=self.device)
                        e.W.copy_(r['W'][idx])
                        e.H.copy_(r['H'][idx, :, :e.g.Np])
                    self.ragged_klnmf_used = True
                    return
                if rc != 3:                                      # GCCNMF_ERR_UNSUPPORTED: no chained form for this shape
                    _hip.check(rc, 'gccnmf_klnmf_ragged')
            self.ragged_klnmf_used = False
            for e in self.sub.values():
                e.klnmf()

    def run(self, stft=True):
        if stft:
            for e in self.sub.values():
                e.stft()
        self.klnmf()
        for e in self.sub.values():
            e.localize()
            e.masks()
            e.reconstruct()
            e.istft()

    def upload(self, mixtures):
        """mixtures[i]: (2, lengths[i]) float32 samples of file i."""
        if len(mixtures) != self.batch:
            raise ValueError('expected %d mixtures' % self.batch)
        for n, e in self.sub.items():
            e.upload(np.stack([np.asarray(mixtures[i], dtype=np.float32) for i in self.files_of[n]]))

    def separate(self, mixtures):
        """list of (2, lengths[i]) float32 host arrays -> list of (S, 2, hop * (T_i - 1)) float32 host waveforms, in the caller's order."""
        self.upload(mixtures)
        self.run()
        out = [None] * self.batch
        if self.ragged_klnmf_used:
            import ctypes
            st = ctypes.c_int(0)
            torch.cuda.current_stream(self.device).synchronize()
            _hip.check(self.lib.gccnmf_klnmf_chain_status(_ptr(self.ragged['ws']), self.g.F, self.g.N, self.g.K, self.batch, ctypes.byref(st)),
                       'gccnmf_klnmf_chain_status')
            if st.value:
                raise _hip.HipLibraryError('the ragged chained KL-NMF launch did not hand over cleanly (status %d): W and H are NaN' % st.value)
        for n, e in self.sub.items():
            y = e.y.cpu().numpy()
            e.check_status()
            for k, i in enumerate(self.files_of[n]):
                out[i] = y[k]
        return out

    def file(self, i):
        """(engine of file i's length, its index in that engine's batch): ``e, k = eng.file(i); e.get_WH()[0][k]``."""
        n = self.lengths[i]
        return self.sub[n], self.files_of[n].index(i)
