"""CPU ORACLE (test infrastructure, not the product) for the frame-sharded single-mixture mode: the per-rank arithmetic that
gcc_nmf_amd.distributed.HipTimeShard does on the GPU, in NumPy on top of oracle/gccnmf_oracle.py and the shared-dictionary oracle
shard.  With one rank it is exactly oracle.runGCCNMF -- i.e. the reference pipeline (gccNMF/runGCCNMF.py:36-52) -- which is how
it is pinned (tests/test_distributed_cpu.py)."""
import numpy as np
import scipy.fftpack
import torch

from oracle import gccnmf_oracle as O
from oracle.shared_nmf_oracle import NumpySharedNMF


def istft_frames(stft_matrix, n_fft, window):
    """The per-frame part of gccNMF/librosaSTFT.py:276-280: full spectrum [conj(S), S[-2:0:-1]], ifft in the precision of S, real part,
    times the float64 window -> (T, n_fft) float64 (what the reference adds into its float32 buffer frame by frame)."""
    full = np.concatenate((stft_matrix.conj(), stft_matrix[-2:0:-1]), axis=0)
    return (np.asarray(window, np.float64)[:, None] * scipy.fftpack.ifft(full, axis=0).real).T


class NumpyTimeShard(object):
    def __init__(self, stereoSamples, rank, world_size, sampleRate=16000, windowSize=1024, hopSize=256, numTDOAs=128,
                 microphoneSeparationInMetres=1.0, numTargets=3, dictionarySize=128):
        from gcc_nmf_amd.distributed import shard_frames, time_shard_initial_factors
        self.x = np.asarray(stereoSamples, np.float32)
        self.sr, self.n_fft, self.hop, self.D, self.d, self.S, self.K = sampleRate, windowSize, hopSize, numTDOAs, microphoneSeparationInMetres, numTargets, dictionarySize
        self.T_total = 1 + int((self.x.shape[1] - windowSize) / hopSize)
        self.halo = -(-windowSize // hopSize) - 1
        self.t0, self.t1 = shard_frames(self.T_total, world_size, rank)
        self.last = rank == world_size - 1
        self._factors = lambda F: time_shard_initial_factors(F, self.T_total, dictionarySize, self.t0, self.t1)
        self.nmf = None

    def stft(self):
        lo, hi = self.t0 * self.hop, (self.t1 - 1) * self.hop + self.n_fft
        self.X = O.computeComplexMixtureSpectrogram(self.x[:, lo:hi], self.n_fft, self.hop, np.hanning)
        self.F, self.Tr = self.X.shape[1], self.X.shape[2]
        assert self.Tr == self.t1 - self.t0
        self.freqs = np.linspace(0, self.sr / 2.0, self.F)
        W0, H0 = self._factors(self.F)
        self.nmf = NumpySharedNMF([O.magnitudeSpectrogramV(self.X)], W0, [H0])
        self.C = O.spectralCoherence(self.X)

    def nmf_done(self):
        self.W, self.stereoH = self.nmf.W(), np.array(np.hsplit(self.nmf.H()[0], 2))

    def angular_sum(self):
        return torch.from_numpy(np.sum(O.getAngularSpectrogram(self.C, self.freqs, self.d, self.D), axis=-1))

    def set_angular_mean(self, mean):
        self.idx = O.estimateTargetTDOAIndexesFromAngularSpectrum(mean.numpy(), self.d, self.D, self.S)

    def masks_and_spectrograms(self):
        G = O.getTargetTDOAGCCNMFs(self.C, self.d, self.D, self.freqs, self.idx, self.W, self.stereoH)
        M = O.getTargetCoefficientMasks(G, self.S)
        spec = O.getTargetSpectrogramEstimates(M, self.X, self.W, self.stereoH).reshape(2 * self.S, self.F, self.Tr)
        win = np.hanning(self.n_fft)
        # windowed time frames exactly as gccNMF/librosaSTFT.py:276-280 builds them before accumulating
        self.frames = np.stack([istft_frames(s, self.n_fft, win) for s in spec])                          # (nsig, Tr, n_fft) float64

    def tail_frames(self):
        return torch.from_numpy(np.ascontiguousarray(self.frames[:, self.Tr - self.halo:, :]))

    def overlap_add(self, previous):
        nsig = 2 * self.S
        prev = previous.numpy() if previous is not None else np.zeros((nsig, self.halo, self.n_fft), np.float64)
        frames = np.concatenate([prev, self.frames], axis=1)
        total = self.n_fft + self.hop * (frames.shape[1] - 1)
        y = np.zeros((nsig, total), np.float32)
        for t in range(frames.shape[1]):                                       # ascending frame order, float32 buffer (:273,:281)
            y[:, t * self.hop:t * self.hop + self.n_fft] = y[:, t * self.hop:t * self.hop + self.n_fft] + frames[:, t]
        L = (self.Tr - 1) * self.hop + self.n_fft if self.last else self.Tr * self.hop
        seg = y[:, self.halo * self.hop:self.halo * self.hop + L] * np.float32(self.hop / float(self.n_fft) * 2)
        return seg.reshape(self.S, 2, L), self.t0 * self.hop - self.n_fft // 2
