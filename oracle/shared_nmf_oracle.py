"""CPU ORACLE (test infrastructure, not the product) for the shared-dictionary KL-NMF shard that
gcc_nmf_amd.distributed.HipSharedNMF computes on the GPU: same begin / step_a / step_b / finish protocol,
plain NumPy float32.  With one shard holding every column and the 'concat' initialisation this is exactly
performKLNMF (gccNMF/gccNMFFunctions.py:69-83) on the concatenated V, which is how it is pinned
(tests/test_distributed_cpu.py)."""
import numpy as np
import torch


class NumpySharedNMF(object):
    def __init__(self, V_files, W0, H0_files, sparsityAlpha=0, epsilon=1e-16):
        self.V = [np.asarray(v, np.float32) for v in V_files]
        self.Wm = np.array(W0, np.float32)
        self.Hs = [np.array(h, np.float32) for h in H0_files]
        self.alpha, self.eps = sparsityAlpha, epsilon
        self.F, self.K = self.Wm.shape

    def begin(self):
        self.hscale = np.ones(self.K, np.float32)

    def step_a(self):
        W = self.Wm
        num = np.zeros((self.F, self.K), np.float32)
        den = np.zeros(self.K, np.float32)
        for i, V in enumerate(self.V):
            H = self.Hs[i] * self.hscale[:, None]                                     # :81, applied lazily
            H *= np.dot(W.T, V / np.dot(W, H)) / (np.sum(W, axis=0)[:, None] + self.alpha + self.eps)   # :76
            num += np.dot(V / np.dot(W, H), H.T)                                      # :77 numerator
            den += np.sum(H, axis=1)                                                  # :77 denominator
            self.Hs[i] = H
        self.hscale = np.ones(self.K, np.float32)
        return torch.from_numpy(np.concatenate([num.ravel(), den]))

    def step_b(self, partial):
        p = partial.numpy()
        num, den = p[:self.F * self.K].reshape(self.F, self.K), p[self.F * self.K:]
        self.Wm *= num / den
        norms = np.sqrt(np.sum(self.Wm ** 2, 0))                                      # :79
        self.Wm /= norms                                                              # :80
        self.hscale = norms.astype(np.float32)

    def finish(self):
        self.Hs = [h * self.hscale[:, None] for h in self.Hs]
        self.hscale = np.ones(self.K, np.float32)

    def W(self):
        return self.Wm

    def H(self):
        return self.Hs
