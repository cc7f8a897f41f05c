"""Seeded synthetic stereo mixtures of the benchmark shape.

Three low-passed, slowly amplitude-modulated noise sources; the right channel holds
integer-sample delayed copies (TDOA peak at tau = -delay/sampleRate); independent sensor
noise keeps |X| > 0 in every bin (no NaN coherence); samples are int16-representable
float32 so the data could have come from a wav file (wavfile.pcm2float convention).
Pure NumPy/SciPy input generation -- no part of the measured path.
"""
import numpy as np


def synthetic_mixture(fileIndex, numSamples=160000, sampleRate=16000, delays=(-20, 3, 27)):
    from scipy.signal import butter, lfilter
    rng = np.random.default_rng(20260925 + fileIndex)
    t = np.arange(numSamples) / float(sampleRate)
    b, a = butter(4, 4000.0 / (sampleRate / 2.0))
    left = np.zeros(numSamples)
    right = np.zeros(numSamples)
    for j, d in enumerate(delays):
        s = lfilter(b, a, rng.standard_normal(numSamples))
        phi = rng.uniform(0, 2 * np.pi)
        s = s * 0.5 * (1 + np.sin(2 * np.pi * (0.7 + 0.3 * j) * t + phi))
        left += s
        right += np.roll(s, d)
    left += rng.normal(0, 1e-3, numSamples)
    right += rng.normal(0, 1e-3, numSamples)
    x = np.stack([left, right])
    x = x / np.max(np.abs(x)) * 0.1
    pcm = np.round(x * 32768).astype(np.int16)
    return ((pcm.astype('float32') - 0) / 32768).astype(np.float32)


def synthetic_batch(firstIndex, count, numSamples=160000, sampleRate=16000):
    return np.stack([synthetic_mixture(firstIndex + i, numSamples, sampleRate) for i in range(count)])
