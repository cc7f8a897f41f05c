"""Multi-GPU modes of the GCC-NMF path: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm).

1. File-sharded separation (BASELINE configs 2-3): mixtures are independent units (own dictionary per
   file, gccNMF/runGCCNMF.py:41), so ranks take disjoint files and share NOTHING on the data path --
   ``shard_files`` + one ``GCCNMFEngine`` per rank.  No collective.

2. Shared-dictionary training (BASELINE config 4; the reference analogue is performKLNMF on one big
   training matrix, gccNMF/realtime/gccNMFPretraining.py:79-80): the column (file/time) axis of V and H is
   sharded over files and ranks, W is replicated.  The H update (gccNMFFunctions.py:76) is column-local; the W
   update (:77) needs  num = sum_cols (V/WH).H^T  (F x K)  and  den = sum_cols H  (K) over ALL columns: one
   all-reduce(sum, f32) of Fp*Kp + Kp floats per iteration, num||den fused in one buffer (latency-bound on
   xGMI: 2.1 MB at K=1024).  Every rank then applies W *= num/den, normalises atoms and rescales its H
   identically -- no second collective.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _hip
from .engine import Geometry, padded, _ptr, _stream, _on_device


def shard_files(num_files, world_size, rank):
    """Contiguous, balanced shard of file indexes for ``rank`` (first ``num_files % world_size`` ranks get one more)."""
    if not 0 <= rank < world_size:
        raise ValueError('rank %d outside world of %d' % (rank, world_size))
    q, r = divmod(num_files, world_size)
    start = rank * q + min(rank, r)
    return list(range(start, start + q + (1 if rank < r else 0)))


def shared_initial_factors(F, file_columns, K, file_indexes, epsilon=1e-16, seedValue=0, mode='per_file'):
    """Initial W (F,K) and this rank's H blocks.

    mode='concat'   : exactly performKLNMF on the concatenation of ALL files (gccNMFFunctions.py:70-73): one MT19937
                      stream, W first, then H (K, sum N) of which each file takes its column block.  Needs the whole
                      H0 on every rank -- for tests and small problems.
    mode='per_file' : W from RandomState(seed); file i's H from RandomState(seed + 1 + i).  World-size independent and
                      O(local) memory; this is what config 4 (512 files) uses."""
    file_columns = list(file_columns)
    if mode == 'concat':
        rs = np.random.RandomState(seedValue)
        W = rs.random_sample((F, K)).astype(np.float32) + epsilon
        H = rs.random_sample((K, int(np.sum(file_columns)))).astype(np.float32) + epsilon
        off = np.concatenate([[0], np.cumsum(file_columns)])
        Hs = [np.ascontiguousarray(H[:, off[i]:off[i + 1]]).astype(np.float32) for i in file_indexes]
    elif mode == 'per_file':
        W = np.random.RandomState(seedValue).random_sample((F, K)).astype(np.float32) + epsilon
        Hs = [(np.random.RandomState(seedValue + 1 + i).random_sample((K, file_columns[i])).astype(np.float32) + epsilon).astype(np.float32)
              for i in file_indexes]
    else:
        raise ValueError(mode)
    return W.astype(np.float32), Hs


class HipSharedNMF(object):
    """This rank's shard of a shared-dictionary KL-NMF on the GPU (csrc/nmf.hip, gccnmf_klnmf_shared_*).
    All local files must have the same number of columns N."""

    def __init__(self, V_files, W0, H0_files, sparsityAlpha=0, epsilon=1e-16, device=None):
        if not torch.cuda.is_available():
            raise _hip.HipLibraryError('no ROCm device visible: gcc_nmf_amd has no CPU fallback')
        self.lib = _hip.lib()
        self.device = torch.device(device if device is not None else ('cuda:%d' % torch.cuda.current_device()))
        V_files = [np.asarray(v, np.float32) for v in V_files]
        self.B = len(V_files)
        self.F, self.N = V_files[0].shape
        self.K = W0.shape[1]
        self.alpha, self.eps = float(sparsityAlpha), float(epsilon)
        g = self.g = Geometry(self.F, 1, self.K)
        self.Np = -(-self.N // 64) * 64
        dev = self.device
        self.V = padded(np.stack(V_files), (self.B, g.Fp, self.Np), dev)
        self.Wd = padded(np.asarray(W0, np.float32), (g.Fp, g.Kp), dev)
        self.Hd = padded(np.stack([np.asarray(h, np.float32) for h in H0_files]), (self.B, g.Kp, self.Np), dev)
        self.ws = torch.zeros(self.lib.gccnmf_klnmf_shared_workspace_floats(self.F, self.N, self.K, self.B), dtype=torch.float32, device=dev)
        self.partial = torch.zeros(self.lib.gccnmf_klnmf_shared_partial_floats(self.F, self.K), dtype=torch.float32, device=dev)
        self._W0d, self._H0d = self.Wd.clone(), self.Hd.clone()

    @classmethod
    def from_device(cls, V_dev, F, N, W0, H0_files, sparsityAlpha=0, epsilon=1e-16):
        """Shard whose V is already resident ([B][Fp][Np] padded device tensor, e.g. GCCNMFEngine.V after stft())."""
        self = cls.__new__(cls)
        self.lib = _hip.lib()
        self.device = V_dev.device
        self.B, self.F, self.N, self.K = V_dev.shape[0], int(F), int(N), W0.shape[1]
        self.alpha, self.eps = float(sparsityAlpha), float(epsilon)
        g = self.g = Geometry(self.F, 1, self.K)
        self.Np = -(-self.N // 64) * 64
        assert tuple(V_dev.shape) == (self.B, g.Fp, self.Np)
        self.V = V_dev
        self.Wd = torch.zeros((g.Fp, g.Kp), dtype=torch.float32, device=self.device)
        self.Hd = torch.zeros((self.B, g.Kp, self.Np), dtype=torch.float32, device=self.device)
        self.ws = torch.zeros(self.lib.gccnmf_klnmf_shared_workspace_floats(self.F, self.N, self.K, self.B), dtype=torch.float32,
                              device=self.device)
        self.partial = torch.zeros(self.lib.gccnmf_klnmf_shared_partial_floats(self.F, self.K), dtype=torch.float32, device=self.device)
        self.reset(W0, H0_files)
        return self

    @_on_device
    def reset(self, W0=None, H0_files=None):
        """(Re)load initial factors; the padding stays zero.  New arrays are uploaded (and remembered); ``reset()`` without
        arguments restores the factors of the last upload without touching the host."""
        g = self.g
        if W0 is not None:
            self._W0d = padded(np.asarray(W0, np.float32), (g.Fp, g.Kp), self.device)
        if H0_files is not None:
            self._H0d = padded(np.stack([np.asarray(h, np.float32) for h in H0_files]), (self.B, g.Kp, self.Np), self.device)
        if not hasattr(self, '_W0d') or not hasattr(self, '_H0d'):
            raise ValueError('reset() without arguments needs initial factors from an earlier reset(W0, H0_files)')
        self.Wd.copy_(self._W0d)
        self.Hd.copy_(self._H0d)

    @_on_device
    def begin(self):
        _hip.check(self.lib.gccnmf_klnmf_shared_begin(_ptr(self.Wd), _ptr(self.ws), self.F, self.N, self.K, self.B, _stream()),
                   'gccnmf_klnmf_shared_begin')

    @_on_device
    def step_a(self):
        _hip.check(self.lib.gccnmf_klnmf_shared_step_a(_ptr(self.V), _ptr(self.Wd), _ptr(self.Hd), _ptr(self.ws), _ptr(self.partial),
                                                       self.F, self.N, self.K, self.B, self.alpha, self.eps, _stream()),
                   'gccnmf_klnmf_shared_step_a')
        return self.partial

    @_on_device
    def step_b(self, partial):
        _hip.check(self.lib.gccnmf_klnmf_shared_step_b(_ptr(self.Wd), _ptr(self.ws), _ptr(partial), self.F, self.N, self.K, self.B,
                                                       _stream()), 'gccnmf_klnmf_shared_step_b')

    @_on_device
    def finish(self):
        _hip.check(self.lib.gccnmf_klnmf_shared_finish(_ptr(self.Hd), _ptr(self.ws), self.F, self.N, self.K, self.B, _stream()),
                   'gccnmf_klnmf_shared_finish')

    def W(self):
        return self.Wd[:self.F, :self.K].cpu().numpy()

    def H(self):
        return [h[:self.K, :self.N].cpu().numpy() for h in self.Hd]


def train_shared_dictionary(local, numIterations, group=None):
    """The shared-dictionary iteration, identical on every rank.  ``local`` owns this rank's columns and provides
    begin() / step_a() -> partial tensor [num (Fp*Kp) || den (Kp)] / step_b(partial) / finish().
    One all-reduce per iteration; with a single process (or no process group) it degenerates to plain KL-NMF."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    local.begin()
    for _ in range(int(numIterations)):
        partial = local.step_a()
        if distributed:
            dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)      # RCCL over xGMI when backend == "nccl"
        local.step_b(partial)
    local.finish()
    return local
