"""wav <-> float32 conventions of the reference (gccNMF/wavfile.py:34-131), host side.

File I/O stays on the host in Python as in the reference; the
int16 <-> float32 conversion rules are the part callers depend on:
  read : int16 -> float32 / 32768, transposed to (channels, n)          (wavfile.py:34-37, :57-89)
  write: rescale to 0.99 if max|x| >= 1, then x*32768 clipped, truncated to int16  (wavfile.py:39-48, :92-131)
"""
import logging

import numpy as np
from scipy.io import wavfile as _scipy_wavfile

CLIP_PROTECTION_MAX_SAMPLE_VALUE = 0.99


def pcm2float(sig, dtype='float32'):
    sig = np.asarray(sig)
    if sig.dtype.kind not in 'iu':
        raise TypeError("'sig' must be an array of integers")
    dtype = np.dtype(dtype)
    if dtype.kind != 'f':
        raise TypeError("'dtype' must be a floating point type")
    info = np.iinfo(sig.dtype)
    abs_max = 2 ** (info.bits - 1)
    offset = info.min + abs_max
    return (sig.astype(dtype) - offset) / abs_max


def float2pcm(sig, dtype='int16'):
    sig = np.asarray(sig)
    if sig.dtype.kind != 'f':
        raise TypeError("'sig' must be a float array")
    dtype = np.dtype(dtype)
    if dtype.kind not in 'iu':
        raise TypeError("'dtype' must be an integer type")
    info = np.iinfo(dtype)
    abs_max = 2 ** (info.bits - 1)
    offset = info.min + abs_max
    return (sig * abs_max + offset).clip(info.min, info.max).astype(dtype)


def wavread(filePath):
    sampleRate, samples_pcm = _scipy_wavfile.read(filePath)
    return pcm2float(samples_pcm).T, sampleRate


def wavwrite(samples_float32, filePath, sampleRate, clipProtection=True):
    maxAbsValue = np.max(np.abs(samples_float32))
    if maxAbsValue >= 1:
        if not clipProtection:
            raise ValueError('wavwrite: max abs signal value exceeds 1')
        logging.warning('wavwrite: max abs signal value exceeds 1, rescaling to %2f' % CLIP_PROTECTION_MAX_SAMPLE_VALUE)
        samples_float32 = samples_float32 / maxAbsValue * CLIP_PROTECTION_MAX_SAMPLE_VALUE
    _scipy_wavfile.write(filePath, sampleRate, float2pcm(samples_float32.astype(np.float32)).T)
