"""stft / istft with the call signatures of the reference's vendored librosa
(gccNMF/librosaSTFT.py:20-286), computed by the HIP kernels in csrc/fft.hip.

Argument validation and error types mirror the reference (ParameterError for
non-contiguous / non-finite / too-short audio, bad hop, window-size mismatch);
the transform itself has no CPU implementation here.
"""
import numpy as np
import torch

from . import _hip, _staging
from .engine import Geometry, padded, fft_twiddles, _ptr, _stream, num_frames


class LibrosaError(Exception):
    """gccNMF/librosaSTFT.py:288-290."""


class ParameterError(LibrosaError):
    """gccNMF/librosaSTFT.py:293-295."""


def pad_center(data, size, axis=-1, **kwargs):
    """gccNMF/librosaSTFT.py:297-368."""
    kwargs.setdefault('mode', 'constant')
    n = data.shape[axis]
    lpad = int((size - n) // 2)
    lengths = [(0, 0)] * data.ndim
    lengths[axis] = (lpad, size - n - lpad)
    if lpad < 0:
        raise ParameterError('Target size ({:d}) must be at least input size ({:d})'.format(size, n))
    return np.pad(data, lengths, **kwargs)


def valid_audio(y, mono=False):
    """gccNMF/librosaSTFT.py:437-491."""
    if not isinstance(y, np.ndarray):
        raise ParameterError('data must be of type numpy.ndarray')
    if mono and y.ndim != 1:
        raise ParameterError('Invalid shape for monophonic audio: ndim={:d}, shape={}'.format(y.ndim, y.shape))
    elif y.ndim > 2:
        raise ParameterError('Invalid shape for audio: ndim={:d}, shape={}'.format(y.ndim, y.shape))
    if not np.isfinite(y).all():
        raise ParameterError('Audio buffer is not finite everywhere')
    return True


def _window_vector(window, win_length, n_fft, inverse=False):
    """gccNMF/librosaSTFT.py:133-151 (stft) / :251-270 (istft)."""
    if window is None:
        # reference default: scipy.signal.hann(win_length, sym=False) [* 2/3 for istft] (:135, :254)
        w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(win_length) / win_length)
        if inverse:
            w = w * (2.0 / 3)
    elif callable(window):
        w = window(win_length)
    else:
        w = np.asarray(window)
        if w.size != n_fft:
            raise ParameterError('Size mismatch between n_fft and len(window)' if not inverse
                                 else 'Size mismatch between n_fft and window size')
    return pad_center(w, n_fft)


RADIX2_N_FFT = (64, 128, 256, 512, 1024, 2048, 4096)     # the LDS radix-2 kernels (csrc/fft.hip); every other size: DFT as a GEMM on the matrix cores
MAX_N_FFT = 8192


def _check_n_fft(n_fft):
    """Like the reference, any n_fft: powers of two from 64 to 4096 run the radix-2 kernels, every other size up to 8192 the
    DFT-as-GEMM path (gccnmf_stft_dft / gccnmf_istft_dft)."""
    if not 2 <= int(n_fft) <= MAX_N_FFT:
        raise ParameterError('n_fft={} is not supported by the HIP STFT/iSTFT: 2 .. {}'.format(n_fft, MAX_N_FFT))


def dft_basis(window, n_fft, Fp):
    """[round_up(n_fft,16)][2*Fp] float32: w[n] cos / sin(2 pi f n / n_fft), float64 on the host (librosaSTFT.py:176-179 transforms in
    float64 too)."""
    F = n_fft // 2 + 1
    ang = 2.0 * np.pi * np.outer(np.arange(n_fft, dtype=np.float64), np.arange(F, dtype=np.float64)) / n_fft
    w = np.asarray(window, np.float64)[:, None]
    b = np.zeros((-(-n_fft // 16) * 16, 2 * Fp), np.float32)
    b[:n_fft, :F] = w * np.cos(ang)
    b[:n_fft, Fp:Fp + F] = w * np.sin(ang)
    return b


def idft_basis(window, n_fft, Fp):
    """[2*Fp][round_up(n_fft,64)] float32: c_k w[n] cos / sin(2 pi k n / n_fft) / n_fft with c_0 = c_{n_fft/2} = 1, else 2."""
    F = n_fft // 2 + 1
    ang = 2.0 * np.pi * np.outer(np.arange(F, dtype=np.float64), np.arange(n_fft, dtype=np.float64)) / n_fft
    c = np.full((F, 1), 2.0)
    c[0] = c[F - 1] = 1.0
    w = np.asarray(window, np.float64)[None, :]
    b = np.zeros((2 * Fp, -(-n_fft // 64) * 64), np.float32)
    b[:F, :n_fft] = c * w * np.cos(ang) / n_fft
    b[Fp:Fp + F, :n_fft] = c * w * np.sin(ang) / n_fft
    return b


def _device():
    if not torch.cuda.is_available():
        raise _hip.HipLibraryError('no ROCm device visible: gcc_nmf_amd has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _check_frames(y, n_fft, hop_length):
    """gccNMF/librosaSTFT.py:416-430 (frame())."""
    if hop_length < 1:
        raise ParameterError('Invalid hop_length: {:d}'.format(hop_length))
    if not y.flags['C_CONTIGUOUS']:
        raise ParameterError('Input buffer must be contiguous.')
    valid_audio(y)
    T = num_frames(len(y), n_fft, hop_length)
    if T < 1:
        raise ParameterError('Buffer is too short (n={:d}) for frame_length={:d}'.format(len(y), n_fft))
    return T


def frame(y, frame_length=2048, hop_length=512):
    """gccNMF/librosaSTFT.py:370-435: (frame_length, n_frames) strided VIEW of ``y`` with ``y_frames[i, j] == y[j * hop_length + i]``
    (host utility, no copy; the device STFT reads the same overlapping frames straight from the sample buffer)."""
    if hop_length < 1:
        raise ParameterError('Invalid hop_length: {:d}'.format(hop_length))
    if not y.flags['C_CONTIGUOUS']:
        raise ParameterError('Input buffer must be contiguous.')
    valid_audio(y)
    n_frames = 1 + int((len(y) - frame_length) / hop_length)
    if n_frames < 1:
        raise ParameterError('Buffer is too short (n={:d}) for frame_length={:d}'.format(len(y), frame_length))
    return np.lib.stride_tricks.as_strided(y, shape=(frame_length, n_frames), strides=(y.itemsize, hop_length * y.itemsize))


def _fft_tables(w, n_fft, dev):
    wf = np.asarray(w, np.float64).astype(np.float32)
    dwin = _staging.constant(('window', n_fft, wf.tobytes()), lambda: wf, dev)
    dtw = _staging.constant(('twiddle', n_fft), lambda: fft_twiddles(n_fft), dev)
    return dwin, dtw


def _stft_device(y0, y1, n_fft, hop_length, win_length, window, center, remember=False):
    """One packed complex FFT per frame carries both real signals (y1 may be None).  `remember`: in resident mode the device
    spectrogram and its magnitude V (the STFT epilogue's) stay behind the returned array (_staging)."""
    _check_n_fft(n_fft)
    w = _window_vector(window, win_length, n_fft)
    chans = []
    for y in (y0, y1):
        if y is None:
            continue
        if center:
            valid_audio(y)
            y = np.pad(y, int(n_fft // 2), mode='reflect')
        chans.append(y)
    T = None
    for y in chans:
        Tc = _check_frames(y, n_fft, hop_length)
        if T is not None and Tc != T:
            raise ParameterError('channels of different length')
        T = Tc
    n = len(chans[0])
    lib, dev = _hip.lib(), _device()
    F = n_fft // 2 + 1
    g = Geometry(F, T, 1)
    if int(n_fft) not in RADIX2_N_FFT:
        x = torch.zeros((2, n), dtype=torch.float32, device=dev)
        for c, y in enumerate(chans):
            x[c] = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)).to(dev)
        basis = torch.from_numpy(dft_basis(w, n_fft, g.Fp)).to(dev)
        ws = torch.zeros(lib.gccnmf_dft_workspace_floats(n_fft, T, 2), dtype=torch.float32, device=dev)
        X = torch.zeros((2, g.Fp, g.Tp, 2), dtype=torch.float32, device=dev)
        _hip.check(lib.gccnmf_stft_dft(_ptr(x), n, n, n_fft, hop_length, T, 2, _ptr(basis), _ptr(ws), _ptr(X), _stream()), 'gccnmf_stft_dft')
        out = torch.view_as_complex(X)[:, :F, :T].cpu().numpy()
        return out if y1 is not None else out[0]
    dwin, dtw = _fft_tables(w, n_fft, dev)
    with _staging.Scope(dev) as sc:
        x = sc.dev('x', (2, n), corner=(len(chans), n))                 # one signal: the second row stays zero
        for c, y in enumerate(chans):
            x[c].copy_(sc.upload(y, 'x%d' % c, np.float32))
        X = sc.dev('X', (2, g.Fp, g.Tp, 2), corner=(F, T))
        V = sc.dev('V', (g.Fp, g.Np), corner=(F, g.N)) if remember and _staging.resident_mode() else None
        _hip.check(lib.gccnmf_stft_stereo(_ptr(x), 2 * n, n, n_fft, hop_length, T, 1, _ptr(dwin), _ptr(dtw), _ptr(X), _ptr(V), 0,
                                          _stream()), 'gccnmf_stft_stereo')
        nsig = 2 if y1 is not None else 1
        out = sc.download(torch.view_as_complex(X)[:nsig, :F, :T])
        if V is not None and nsig == 2:
            sc.remember(out, 'X', dict(X=X, V=V), dict(F=F, T=T))
    return out if y1 is not None else out[0]


def stft(y, n_fft=2048, hop_length=None, win_length=None, window=None, center=True, dtype=np.complex64):
    """gccNMF/librosaSTFT.py:20-181.  Returns (1 + n_fft/2, T), Fortran-ordered like the reference."""
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = int(win_length / 4)
    return np.asfortranarray(_stft_device(y, None, n_fft, hop_length, win_length, window, center).astype(dtype))


def _istft_device(specs, hop_length, win_length, window, center, gain=1.0, device_spec=None):
    """specs: (nsig, F, T) complex -> (nsig, L) float32.  `device_spec`: the padded device image [nsig][Fp][Tp] of `specs` when it is
    already in HBM (resident mode, nsig even)."""
    specs = np.asarray(specs)
    nsig, F, T = specs.shape
    n_fft = 2 * (F - 1)
    _check_n_fft(n_fft)
    w = _window_vector(window, win_length, n_fft, inverse=True)
    lib, dev = _hip.lib(), _device()
    g = Geometry(F, T, 1)
    npad = nsig + (nsig & 1)
    L = n_fft + hop_length * (T - 1) - (n_fft if center else 0)
    if int(n_fft) not in RADIX2_N_FFT:
        host = np.ascontiguousarray(specs.astype(np.complex64)).view(np.float32).reshape(nsig, F, T, 2)
        dS = padded(host, (npad, g.Fp, g.Tp, 2), dev)
        if L < 1:
            return np.zeros((nsig, 0), np.float32)
        ibasis = torch.from_numpy(idft_basis(w, n_fft, g.Fp)).to(dev)
        ws = torch.zeros(lib.gccnmf_dft_workspace_floats(n_fft, T, npad), dtype=torch.float32, device=dev)
        y = torch.zeros((npad, L), dtype=torch.float32, device=dev)
        _hip.check(lib.gccnmf_istft_dft(_ptr(dS), npad, n_fft, hop_length, T, _ptr(ibasis), np.float32(gain), 1 if center else 0, _ptr(ws),
                                        _ptr(y), _stream()), 'gccnmf_istft_dft')
        return y[:nsig].cpu().numpy()
    if L < 1:
        return np.zeros((nsig, 0), np.float32)
    dwin, dtw = _fft_tables(w, n_fft, dev)
    with _staging.Scope(dev) as sc:
        if device_spec is not None and npad == nsig:
            dS = device_spec
        else:
            dS = sc.dev('spec', (npad, g.Fp, g.Tp, 2), corner=(nsig, F, T))
            torch.view_as_complex(dS)[:nsig, :F, :T].copy_(sc.upload(specs, 'spec', np.complex64))
        frames = sc.dev('frames', (npad, T, n_fft))                      # scratch of the two-kernel form: every frame t < T is written
        y = sc.dev('y', (npad, L))
        _hip.check(lib.gccnmf_istft_ola(_ptr(dS), npad, n_fft, hop_length, T, 1, _ptr(dwin), _ptr(dtw), np.float32(gain),
                                        1 if center else 0, _ptr(frames), _ptr(y), _stream()), 'gccnmf_istft_ola')
        out = sc.download(y[:nsig])
    return out


def istft(stft_matrix, hop_length=None, win_length=None, window=None, center=True, dtype=np.float32):
    """gccNMF/librosaSTFT.py:183-286."""
    n_fft = 2 * (stft_matrix.shape[0] - 1)
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = int(win_length / 4)
    return _istft_device(np.asarray(stft_matrix)[None], hop_length, win_length, window, center)[0].astype(dtype)
