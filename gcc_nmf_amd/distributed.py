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


class CompositeSharedNMF(object):
    """Several shards of one rank behind the begin / step_a / step_b / finish protocol: their [num || den] partials are added (fixed
    order) before the all-reduce and every member applies the same reduced buffer.  Lets one rank hold column blocks of different
    widths (``HipSharedNMF`` needs equal N per object): e.g. the equal blocks of a long mixture plus its ragged remainder."""

    def __init__(self, members):
        self.members = list(members)
        self.partial = self.members[0].partial

    def begin(self):
        for m in self.members:
            m.begin()

    def step_a(self):
        parts = [m.step_a() for m in self.members]
        total = parts[0]
        for p in parts[1:]:
            total += p
        return total

    def step_b(self, partial):
        for m in self.members:
            m.step_b(partial)

    def finish(self):
        for m in self.members:
            m.finish()

    def W(self):
        return self.members[0].W()


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


# ---- 3. one LONG mixture, sharded over frame windows ------------------------------------------------------------------------
# north_star: "sharding ... frame windows with RCCL all-reduce over xGMI only for the shared-dictionary W update".  Ranks own contiguous
# frame ranges of ONE mixture.  Per rank: STFT of its own sample range; KL-NMF with W replicated (mode 2: one all-reduce of [num || den]
# per iteration); the time mean of the angular spectrum (runGCCNMF.py:46) is a second, tiny all-reduce (D float64, once) so that every
# rank picks the same target TDOAs; scores, masks and spectrogram estimates are frame-local; the overlap-add (librosaSTFT.py:275-284)
# of a rank's first n_fft - hop samples also needs the previous rank's last n_fft/hop - 1 FRAMES: they are exchanged as frames, not as
# partial sums, and added in ascending frame order, so the stitched waveform is bit-identical to the single-rank one wherever the
# factors are.
def shard_frames(num_frames, world_size, rank):
    """Contiguous, balanced frame range [t0, t1) of ``rank``."""
    if not 0 <= rank < world_size:
        raise ValueError('rank %d outside world of %d' % (rank, world_size))
    q, r = divmod(num_frames, world_size)
    t0 = rank * q + min(rank, r)
    return t0, t0 + q + (1 if rank < r else 0)


def time_shard_initial_factors(F, T, K, t0, t1, epsilon=1e-16, seedValue=0):
    """performKLNMF's initial factors (gccNMFFunctions.py:70-73: one MT19937 stream, W first, then H (K, 2T) over the columns
    [left frames | right frames]) restricted to the frames [t0, t1): W (F, K) and H (K, 2*(t1-t0)) in the same [left | right] order."""
    rs = np.random.RandomState(seedValue)
    W = rs.random_sample((F, K)).astype(np.float32) + epsilon
    H = rs.random_sample((K, 2 * T)).astype(np.float32) + epsilon
    Hl = np.concatenate([H[:, t0:t1], H[:, T + t0:T + t1]], axis=1)
    return W.astype(np.float32), np.ascontiguousarray(Hl).astype(np.float32)


def separate_time_sharded(local, numIterations, group=None):
    """The frame-sharded separation, identical on every rank.  ``local`` owns this rank's frames [t0, t1) of a T-frame mixture and
    provides:  stft();  nmf (begin/step_a/step_b/finish as in mode 2);  nmf_done();  angular_sum() -> float64 tensor (D,) = sum over
    own frames;  set_angular_mean(mean) (picks the target TDOAs);  masks_and_spectrograms();  tail_frames() -> tensor (nsig, halo,
    n_fft);  overlap_add(previous_tail or None) -> (segment (S, 2, len) ndarray, first global sample index of the TRIMMED output).
    Returns (segment, start).  Collectives: numIterations all-reduces of F*K + K floats, one of D doubles, one all-gather of the
    halo frames (nsig * halo * n_fft floats per rank)."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distributed else 0
    local.stft()
    train_shared_dictionary(local.nmf, numIterations, group)
    local.nmf_done()
    s = local.angular_sum()
    if distributed:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    local.set_angular_mean(s / float(local.T_total))
    local.masks_and_spectrograms()
    tail = local.tail_frames()
    previous = None
    if distributed:
        tails = [torch.empty_like(tail) for _ in range(dist.get_world_size(group))]
        dist.all_gather(tails, tail, group=group)
        if rank > 0:
            previous = tails[rank - 1]
    return local.overlap_add(previous)


def stitch_time_shards(segments, numTargets, total_frames, hop):
    """[(segment (S, 2, len), start)] of every rank -> (S, 2, hop*(T-1)) like getTargetSignalEstimates on the whole mixture."""
    L = hop * (total_frames - 1)
    y = np.zeros((numTargets, 2, L), np.float32)
    for seg, start in segments:
        lo, hi = max(start, 0), min(start + seg.shape[2], L)
        if hi > lo:
            y[:, :, lo:hi] = seg[:, :, lo - start:hi - start]
    return y


class HipTimeShard(object):
    """This rank's frames [t0, t1) of one long mixture on the GPU: a batch-1 ``GCCNMFEngine`` over the rank's sample range plus a
    ``HipSharedNMF`` on its magnitude spectrogram.  ``stereoSamples`` is the WHOLE (2, n) mixture (only the own range is uploaded)."""

    def __init__(self, stereoSamples, rank, world_size, sampleRate=16000, windowSize=1024, hopSize=256, numTDOAs=128,
                 microphoneSeparationInMetres=1.0, numTargets=3, dictionarySize=128, sparsityAlpha=0, epsilon=1e-16, seedValue=0,
                 device=None):
        from .engine import GCCNMFEngine, num_frames
        x = np.asarray(stereoSamples, np.float32)
        self.n_fft, self.hop, self.S = int(windowSize), int(hopSize), int(numTargets)
        self.T_total = num_frames(x.shape[1], self.n_fft, self.hop)
        self.halo = -(-self.n_fft // self.hop) - 1
        self.t0, self.t1 = shard_frames(self.T_total, world_size, rank)
        self.last = rank == world_size - 1
        Tr = self.t1 - self.t0
        if Tr < max(self.halo, 2):
            raise ValueError('time shard of %d frames is shorter than the overlap-add halo (%d frames)' % (Tr, self.halo))
        lo, hi = self.t0 * self.hop, (self.t1 - 1) * self.hop + self.n_fft
        dev = device if device is not None else 'cuda:%d' % torch.cuda.current_device()
        self.e = e = GCCNMFEngine(hi - lo, sampleRate=sampleRate, windowSize=windowSize, hopSize=hopSize, numTDOAs=numTDOAs,
                                  microphoneSeparationInMetres=microphoneSeparationInMetres, numTargets=numTargets,
                                  dictionarySize=dictionarySize, numIterations=0, sparsityAlpha=sparsityAlpha, epsilon=epsilon, batch=1,
                                  device=dev)
        assert e.g.T == Tr
        self.device = e.device
        e.upload(x[:, lo:hi])
        self._W0, self._H0 = time_shard_initial_factors(e.g.F, self.T_total, e.g.K, self.t0, self.t1, epsilon, seedValue)
        self._alpha, self._eps = sparsityAlpha, epsilon
        self.nmf = None

    # The rank's 2*T_r NMF columns are handed to the batched kernels as pseudo-files of BLOCK frames ([left | right] columns of those
    # frames, the layout of one 10 s file) plus one ragged remainder: the per-file launches then fill the chip (a single "file" of
    # 20 000 columns would leave R.H^T with 16 workgroups of 1 250 k-tiles), and the column partition does not change the update
    # (gccNMFFunctions.py:76 is column-local, :77 sums over all columns).
    BLOCK = 640

    def _blocks(self):
        Tr = self.e.g.T
        nb, rem = divmod(Tr, self.BLOCK)
        out = [(j * self.BLOCK, self.BLOCK) for j in range(nb)]
        if rem:
            out.append((nb * self.BLOCK, rem))
        return out                                               # [(first frame, frames)]

    def _gather(self, src, T0, Tb, Np_b, rows):
        """columns [T0, T0+Tb) of both channel halves of a [rows][Np] matrix -> one [rows][Np_b] block"""
        Tr = self.e.g.T
        blk = torch.zeros((rows, Np_b), dtype=torch.float32, device=self.device)
        blk[:, :Tb] = src[:, T0:T0 + Tb]
        blk[:, Tb:2 * Tb] = src[:, Tr + T0:Tr + T0 + Tb]
        return blk

    @_on_device
    def stft(self):
        e, g = self.e, self.e.g
        e.stft()
        H0 = torch.zeros((g.Kp, g.Np), dtype=torch.float32, device=self.device)
        H0[:g.K, :g.N] = torch.from_numpy(self._H0).to(self.device)
        members, self._groups = [], []
        blocks = self._blocks()
        for width in sorted(set(tb for _, tb in blocks), reverse=True):
            mine = [(t0, tb) for t0, tb in blocks if tb == width]
            Np_b = -(-2 * width // 64) * 64
            V = torch.stack([self._gather(e.V[0], t0, tb, Np_b, g.Fp) for t0, tb in mine])
            shard = HipSharedNMF.from_device(V, g.F, 2 * width, self._W0, [np.zeros((g.K, 2 * width), np.float32)] * len(mine), self._alpha, self._eps)
            shard.Hd.copy_(torch.stack([self._gather(H0, t0, tb, Np_b, g.Kp) for t0, tb in mine]))
            shard._H0d = shard.Hd.clone()
            members.append(shard)
            self._groups.append(mine)
        self.nmf = CompositeSharedNMF(members)

    @_on_device
    def nmf_done(self):
        e, Tr = self.e, self.e.g.T
        e.W[0].copy_(self.nmf.members[0].Wd)
        for shard, mine in zip(self.nmf.members, self._groups):
            for j, (t0, tb) in enumerate(mine):
                e.H[0][:, t0:t0 + tb] = shard.Hd[j][:, :tb]
                e.H[0][:, Tr + t0:Tr + t0 + tb] = shard.Hd[j][:, tb:2 * tb]

    @_on_device
    def angular_sum(self):
        e, g = self.e, self.e.g
        _hip.check(e.lib.gccnmf_angular_spectrogram(_ptr(e.CC), _ptr(e.trig), g.F, g.T, g.D, 1, _ptr(e.ang), _ptr(e.mean_ang), _stream()),
                   'gccnmf_angular_spectrogram')
        return e.mean_ang[0, :g.D] * float(g.T)                  # float64: mean over own frames * own frames

    @_on_device
    def set_angular_mean(self, mean):
        e, g = self.e, self.e.g
        e.mean_ang[0, :g.D].copy_(mean.to(e.mean_ang.device))
        _hip.check(e.lib.gccnmf_pick_tdoa_peaks(_ptr(e.mean_ang), g.D, g.Dp, g.S, 1, _ptr(e.tdoa_idx), _ptr(e.status), _stream()),
                   'gccnmf_pick_tdoa_peaks')
        e.check_status()

    @_on_device
    def masks_and_spectrograms(self):
        self.e.masks()
        self.e.reconstruct()
        self.e.istft(keep_frames=True)                           # windowed time frames of the own spectrogram estimates -> e.frames

    @_on_device
    def tail_frames(self):
        return self.e.frames[0, :, self.e.g.T - self.halo:, :].contiguous()

    @_on_device
    def overlap_add(self, previous):
        e, g = self.e, self.e.g
        nsig = 2 * g.S
        prev = previous.to(self.device) if previous is not None else torch.zeros((nsig, self.halo, self.n_fft), dtype=torch.float32, device=self.device)
        frames = torch.cat([prev, e.frames[0]], dim=1).contiguous()                      # [nsig][halo + T_r][n_fft], ascending frames
        L = (g.T - 1) * self.hop + self.n_fft if self.last else g.T * self.hop           # the samples this rank owns
        y = torch.zeros((nsig, L), dtype=torch.float32, device=self.device)
        gain = np.float32(self.hop / float(self.n_fft) * 2)                              # gccNMFFunctions.py:155
        _hip.check(e.lib.gccnmf_ola_frames(_ptr(frames), nsig, self.n_fft, self.hop, self.halo + g.T, 1, self.halo * self.hop, L, gain,
                                           _ptr(y), _stream()), 'gccnmf_ola_frames')
        return y.view(g.S, 2, L).cpu().numpy(), self.t0 * self.hop - self.n_fft // 2    # centre trim (librosaSTFT.py:283-284)

    def tdoa_indexes(self):
        return self.e.get_tdoa_indexes()[0]
