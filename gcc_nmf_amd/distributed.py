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


def shard_files(num_files, world_size, rank, frames=None):
    """File indexes of ``rank``.  Equal-length files (``frames`` None): a contiguous, balanced shard (the first ``num_files % world_size``
    ranks get one more).  Files of DIFFERENT lengths (``frames`` = the frame count, or any cost, of every file): balanced by frames, not by
    count -- longest file first, each to the rank with the fewest frames so far (ties: the lower rank; the same deterministic deal on every
    rank), returned in ascending file order.  With equal ``frames`` this deals round-robin: the counts per rank are those of the contiguous form."""
    if not 0 <= rank < world_size:
        raise ValueError('rank %d outside world of %d' % (rank, world_size))
    if frames is not None:
        frames = [int(f) for f in frames]
        if len(frames) != num_files:
            raise ValueError('frames must name every file')
        load = [0] * world_size
        mine = []
        for i in sorted(range(num_files), key=lambda i: (-frames[i], i)):
            r = min(range(world_size), key=lambda r: (load[r], r))
            load[r] += frames[i]
            if r == rank:
                mine.append(i)
        return sorted(mine)
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


# ---- how the [num || den] buffer is summed over the ranks inside gccnmf_klnmf_shared_run --------------------------------------------
# backend "nccl": the library's own RCCL communicator (csrc/collective.hip), created once per process group from a unique id that
# rank 0 broadcasts over torch.distributed -- the all-reduce is then enqueued from C between the two halves of an iteration, on the
# compute stream, with no host round trip.  Any other backend (gloo: CPU tests, several ranks rehearsing on one GPU), or
# GCCNMF_COLLECTIVE=torch: a host callback that runs dist.all_reduce on the same tensor (the C loop calls back once per iteration).
_rccl_comms = {}


def _default_group():
    """The process group object behind group=None (a re-initialised default group is a NEW object: its communicator is not reused)."""
    try:
        return dist.distributed_c10d._get_default_group()
    except Exception:
        return dist.group.WORLD


def _ipc_env_note():
    """'' when the environment allows RCCL between processes on this driver stack, else what is wrong with it."""
    import os
    v = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')
    if v == '0':
        import gcc_nmf_amd
        if gcc_nmf_amd.IPC_ENV_SET_BY_PACKAGE and gcc_nmf_amd.HIP_STARTED_BEFORE_IMPORT:
            return ('HSA_ENABLE_IPC_MODE_LEGACY=0 was set by gcc_nmf_amd AFTER the HIP runtime had started in this process (torch touched a '
                    'device before the import), so it has no effect: export it in the environment of the launcher instead')
        return ''
    return ('HSA_ENABLE_IPC_MODE_LEGACY is %s in this process (it must be 0 BEFORE the HIP runtime starts: the host driver only supports '
            'dmabuf IPC, RCCL otherwise fails with hipIpcGetMemHandle: invalid argument)' % ('unset' if v is None else repr(v)))


def _call_with_timeout(fn, seconds, what):
    """fn() on a helper thread; HipLibraryError if it has not returned after `seconds` (a collective whose peers never arrive
    blocks forever inside librccl: the caller gets an explanation instead of a silent hang; the stuck thread is left behind)."""
    import threading
    box = {}

    def work():
        try:
            box['value'] = fn()
        except BaseException as e:            # noqa: B902 -- re-raised on the caller's thread
            box['error'] = e
    t = threading.Thread(target=work, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        note = _ipc_env_note()
        raise _hip.HipLibraryError('%s did not return within %.0f s: a peer rank never arrived or RCCL cannot reach it over xGMI%s.  '
                                   'GCCNMF_COLLECTIVE=torch routes the all-reduce through torch.distributed instead; '
                                   'GCCNMF_RCCL_INIT_TIMEOUT sets this limit.' % (what, seconds, ('; ' + note) if note else ''))
    if 'error' in box:
        raise box['error']
    return box['value']


_rccl_given_up = [False]     # a communicator set-up timed out on some rank: the torch route from then on (never a second init beside an orphaned one)


def _try_rccl_init(init, seconds, what):
    """(ok, why): init() behind the time limit.  A timeout is NOT raised here: the caller still owes its peers the agreement all-reduce
    (raising first would leave them waiting in it until torch's own timeout), so it comes back as (False, explanation)."""
    try:
        return bool(_call_with_timeout(init, seconds, what)), None
    except _hip.HipLibraryError as e:
        return False, str(e)


def _rccl_comm(group, device):
    """The library-owned RCCL communicator of (group, device), or None when RCCL cannot be used on EVERY rank."""
    import ctypes
    import os
    gobj = group if group is not None else _default_group()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    key = (gobj, torch.device(device).index)           # the group object itself: no id() reuse after a destroy, no stale default group
    hit = _rccl_comms.get(key)
    if hit is not None and hit[1:] == (world, rank):
        return hit[0]
    lib = _hip.lib()
    if _rccl_given_up[0]:                             # an earlier set-up timed out somewhere: torch route for the rest of this process (every rank alike)
        return None
    src = dist.get_global_rank(group, 0) if group is not None else 0
    ident = torch.zeros(_hip.RCCL_UNIQUE_ID_BYTES + 1, dtype=torch.uint8, device=device)     # [id bytes | ok flag]
    if rank == 0 and lib.gccnmf_rccl_available():
        buf = ctypes.create_string_buffer(_hip.RCCL_UNIQUE_ID_BYTES)
        if lib.gccnmf_rccl_unique_id(buf) == 0:
            ident[:-1] = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(device)
            ident[-1] = 1
    dist.broadcast(ident, src=src, group=group)
    ident = ident.cpu().numpy()
    flag = torch.tensor([1 if (ident[-1] and lib.gccnmf_rccl_available()) else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    comm = None
    if int(flag.item()):
        handle = ctypes.c_void_p()
        torch.cuda.synchronize(device)
        timeout = float(os.environ.get('GCCNMF_RCCL_INIT_TIMEOUT', '120'))

        abandoned = []                                 # set once the caller has stopped waiting for this init

        def init():
            with torch.cuda.device(device):
                done = lib.gccnmf_rccl_comm_init(ident[:-1].tobytes(), world, rank, ctypes.byref(handle)) == 0
                if done and abandoned:                     # it came back after the time limit: nobody will ever use this communicator
                    lib.gccnmf_rccl_comm_destroy(handle)
                    return False
                return done
        ok, why = _try_rccl_init(init, timeout, 'gccnmf_rccl_comm_init (rank %d of %d, device %s)' % (rank, world, device))
        if why:
            abandoned.append(True)
        # one agreement for both questions: did EVERY rank get a communicator, and did NO rank give up on a timeout
        flag = torch.tensor([1 if ok else 0, 0 if why else 1], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        agreed, nobody_timed_out = int(flag[0].item()), int(flag[1].item())
        if agreed:
            comm = handle.value
        else:
            if ok:
                lib.gccnmf_rccl_comm_destroy(handle)
            import warnings
            warnings.warn('gcc_nmf_amd: the library RCCL communicator could not be set up on every rank (%s); the shared-dictionary all-reduce '
                          'goes through torch.distributed instead (GCCNMF_COLLECTIVE=torch semantics, same results)'
                          % (why or 'gccnmf_rccl_comm_init failed on this or another rank'), RuntimeWarning)
            if not nobody_timed_out:
                # A timeout somewhere.  The rank that gave up may still have a thread inside ncclCommInitRank (it destroys its communicator
                # itself should it ever return), and destroying or re-initialising communicators beside it can block: NO rank tries RCCL
                # again in this process -- the agreement all-reduce made that decision the same on every rank.
                _rccl_given_up[0] = True
                return None
    _rccl_comms[key] = (comm, world, rank)
    return comm


def destroy_rccl_communicators():
    """Free the library-owned communicators (call before dist.destroy_process_group())."""
    for comm, _, _ in _rccl_comms.values():
        if comm:
            _hip.lib().gccnmf_rccl_comm_destroy(comm)
    _rccl_comms.clear()


def collective_hook(partial, group=None, force=False):
    """(function pointer, context, keep-alive, description) for gccnmf_klnmf_shared_run's all-reduce of ``partial``.

    nccl backend: the library's own RCCL communicator, enqueued from C (GCCNMF_COLLECTIVE=torch: a host callback into
    torch.distributed instead; =rccl: RCCL or an error).  Any other backend: the host callback.  A single rank needs no exchange;
    ``force`` (or GCCNMF_COLLECTIVE_FORCE=1) builds the hook anyway, so that a one-GPU box drives the same call path."""
    import ctypes
    import os
    force = force or os.environ.get('GCCNMF_COLLECTIVE_FORCE', '') not in ('', '0')
    if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)):
        return None, None, None, 'single rank'
    want = os.environ.get('GCCNMF_COLLECTIVE', '')
    if want not in ('', 'rccl', 'torch'):
        raise ValueError('GCCNMF_COLLECTIVE must be rccl or torch, not %r' % want)
    if want != 'torch' and dist.get_backend(group) == 'nccl':
        comm = _rccl_comm(group, partial.device)
        if comm:
            return _hip.lib().gccnmf_rccl_allreduce_hook(), comm, None, 'rccl (library communicator, enqueued from C)'
        if want == 'rccl':
            raise _hip.HipLibraryError('GCCNMF_COLLECTIVE=rccl but librccl could not be bound / initialised on every rank.  ' + _ipc_env_note())
    failure = []

    def allreduce(ctx, buf, count, stream):
        try:
            if buf != partial.data_ptr() or count != partial.numel():
                raise RuntimeError('all-reduce hook called on an unexpected buffer')
            dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)        # ordered after the kernels on torch's current stream
            return 0
        except BaseException as e:                                             # never unwind through the C frame
            failure.append(e)
            return 1
    cb = _hip.ALLREDUCE_FN(allreduce)
    return ctypes.cast(cb, ctypes.c_void_p).value, None, (cb, failure), 'torch.distributed (%s) host callback' % dist.get_backend(group)


class _SharedRun(object):
    """run(): the whole training of this rank's columns in ONE library call (gccnmf_klnmf_shared_run).  Subclasses provide lib,
    device, F, K, alpha, eps, Wd, partial, vec and _shards() -> [(V, H, workspace, N, batch, ld)] (device tensors)."""

    @_on_device
    def run(self, numIterations, group=None, collective=True):
        """collective=False: this object's columns alone, no all-reduce even inside an initialised process group (performKLNMF on one
        big matrix)."""
        descs = self._shards()
        arr = (_hip.SharedShard * max(len(descs), 1))()
        for d, (V, H, ws, N, batch, ld) in zip(arr, descs):
            d.V, d.H, d.workspace, d.N, d.batch, d.ld = _ptr(V), _ptr(H), _ptr(ws), N, batch, ld
        fn, ctx, keep, self.collective = collective_hook(self.partial, group) if collective else (None, None, None, 'none (local columns only)')
        if fn is not None:
            torch.cuda.current_stream(self.device).synchronize()      # nothing of torch's own collectives is still in flight on the devices
        rc = self.lib.gccnmf_klnmf_shared_run(arr, len(descs), _ptr(self.Wd), _ptr(self.partial), _ptr(self.vec), self.F, self.K,
                                              int(numIterations), self.alpha, self.eps, fn, ctx, _stream())
        if keep is not None and keep[1]:
            raise keep[1][0]
        _hip.check(rc, 'gccnmf_klnmf_shared_run')
        if fn is not None:
            torch.cuda.current_stream(self.device).synchronize()
        return self

    def W(self):
        return self.Wd[:self.F, :self.K].cpu().numpy()


class HipSharedNMF(_SharedRun):
    """This rank's shard of a shared-dictionary KL-NMF on the GPU (csrc/nmf.hip, gccnmf_klnmf_shared_*).
    All local files must have the same number of columns N.  ``V_files`` may be EMPTY (a rank without files when there are fewer
    files than ranks): the rank then contributes zeros to every all-reduce and still applies every W update."""

    def __init__(self, V_files, W0, H0_files, sparsityAlpha=0, epsilon=1e-16, device=None):
        if not torch.cuda.is_available():
            raise _hip.HipLibraryError('no ROCm device visible: gcc_nmf_amd has no CPU fallback')
        self.lib = _hip.lib()
        self.device = torch.device(device if device is not None else ('cuda:%d' % torch.cuda.current_device()))
        V_files = [np.asarray(v, np.float32) for v in V_files]
        self.B = len(V_files)
        self.F, self.K = W0.shape
        self.N = V_files[0].shape[1] if self.B else 0
        if any(v.shape != (self.F, self.N) for v in V_files) or len(H0_files) != self.B:
            raise ValueError('every local file needs V of shape (%d, %d) and one H block' % (self.F, self.N))
        self.alpha, self.eps = float(sparsityAlpha), float(epsilon)
        self._geometry()
        g, dev = self.g, self.device
        self.V = padded(np.stack(V_files), (self.B, g.Fp, self.Np), dev) if self.B else None
        self.Wd = padded(np.asarray(W0, np.float32), (g.Fp, g.Kp), dev)
        self.Hd = padded(np.stack([np.asarray(h, np.float32) for h in H0_files]), (self.B, g.Kp, self.Np), dev) if self.B else None
        self._scratch()
        self._W0d, self._H0d = self.Wd.clone(), (self.Hd.clone() if self.B else None)

    def _geometry(self):
        self.g = Geometry(self.F, 1, self.K)
        self.Np = -(-self.N // 64) * 64
        self._gN, self._gB = (self.N, self.B) if self.B else (1, 1)      # the four-call protocol's geometry (a dummy 1 x 1 shard when empty)

    def _scratch(self):
        # legacy layout: [shard scratch (R | Upart | rowsum_part)] [colsumW | hscale]
        self.ws = torch.zeros(self.lib.gccnmf_klnmf_shared_workspace_floats(self.F, self._gN, self.K, self._gB), dtype=torch.float32,
                              device=self.device)
        self.vec = self.ws[self.ws.numel() - 2 * self.g.Kp:]
        self.partial = torch.zeros(self.lib.gccnmf_klnmf_shared_partial_floats(self.F, self.K), dtype=torch.float32, device=self.device)

    def _shards(self):
        return [(self.V, self.Hd, self.ws, self.N, self.B, 0)] if self.B else []

    @classmethod
    def from_device(cls, V_dev, F, N, W0, H0_files, sparsityAlpha=0, epsilon=1e-16):
        """Shard whose V is already resident ([B][Fp][Np] padded device tensor, e.g. GCCNMFEngine.V after stft())."""
        self = cls.__new__(cls)
        self.lib = _hip.lib()
        self.device = V_dev.device
        self.B, self.F, self.N, self.K = V_dev.shape[0], int(F), int(N), W0.shape[1]
        self.alpha, self.eps = float(sparsityAlpha), float(epsilon)
        self._geometry()
        g = self.g
        assert tuple(V_dev.shape) == (self.B, g.Fp, self.Np)
        self.V = V_dev
        self.Wd = torch.zeros((g.Fp, g.Kp), dtype=torch.float32, device=self.device)
        self.Hd = torch.zeros((self.B, g.Kp, self.Np), dtype=torch.float32, device=self.device)
        self._scratch()
        self.reset(W0, H0_files)
        return self

    @_on_device
    def reset(self, W0=None, H0_files=None):
        """(Re)load initial factors; the padding stays zero.  New arrays are uploaded (and remembered); ``reset()`` without
        arguments restores the factors of the last upload without touching the host."""
        g = self.g
        if W0 is not None:
            self._W0d = padded(np.asarray(W0, np.float32), (g.Fp, g.Kp), self.device)
        if H0_files is not None and self.B:
            self._H0d = padded(np.stack([np.asarray(h, np.float32) for h in H0_files]), (self.B, g.Kp, self.Np), self.device)
        if not hasattr(self, '_W0d') or (self.B and getattr(self, '_H0d', None) is None):
            raise ValueError('reset() without arguments needs initial factors from an earlier reset(W0, H0_files)')
        self.Wd.copy_(self._W0d)
        if self.B:
            self.Hd.copy_(self._H0d)

    # the four-call protocol (one all-reduce between step_a and step_b, driven by the host): kept for hosts that own the collective
    @_on_device
    def begin(self):
        _hip.check(self.lib.gccnmf_klnmf_shared_begin(_ptr(self.Wd), _ptr(self.ws), self.F, self._gN, self.K, self._gB, _stream()),
                   'gccnmf_klnmf_shared_begin')

    @_on_device
    def step_a(self):
        if not self.B:
            self.partial.zero_()
            return self.partial
        _hip.check(self.lib.gccnmf_klnmf_shared_step_a(_ptr(self.V), _ptr(self.Wd), _ptr(self.Hd), _ptr(self.ws), _ptr(self.partial),
                                                       self.F, self.N, self.K, self.B, self.alpha, self.eps, _stream()),
                   'gccnmf_klnmf_shared_step_a')
        return self.partial

    @_on_device
    def step_b(self, partial):
        _hip.check(self.lib.gccnmf_klnmf_shared_step_b(_ptr(self.Wd), _ptr(self.ws), _ptr(partial), self.F, self._gN, self.K, self._gB,
                                                       _stream()), 'gccnmf_klnmf_shared_step_b')

    @_on_device
    def finish(self):
        if self.B:
            _hip.check(self.lib.gccnmf_klnmf_shared_finish(_ptr(self.Hd), _ptr(self.ws), self.F, self.N, self.K, self.B, _stream()),
                       'gccnmf_klnmf_shared_finish')

    def H(self):
        return [h[:self.K, :self.N].cpu().numpy() for h in self.Hd] if self.B else []


class HipSharedColumns(_SharedRun):
    """The columns of ONE matrix pair V [Fp][ld] / H [Kp][ld] (device tensors, padding zero) as this rank's part of a shared-dictionary
    KL-NMF, without copying them: the N columns are handed to the batched kernels as column BLOCKS of ``block`` columns (a multiple of 64)
    plus one ragged remainder (gccnmf_shared_shard with ld > 0).  The partition does not change the update (gccNMFFunctions.py:76 is
    column-local, :77 sums over all columns); it only sets how many workgroups the R.H^T launch has (16 per block at K = 1024)."""

    def __init__(self, V, H, W, F, N, K, sparsityAlpha=0, epsilon=1e-16, block=None):
        self.lib = _hip.lib()
        self.device = V.device
        self.F, self.N, self.K = int(F), int(N), int(K)
        self.alpha, self.eps = float(sparsityAlpha), float(epsilon)
        g = self.g = Geometry(self.F, 1, self.K)
        self.ld = V.shape[1]
        assert tuple(V.shape) == (g.Fp, self.ld) and tuple(H.shape) == (g.Kp, self.ld) and tuple(W.shape) == (g.Fp, g.Kp)
        assert self.ld % 64 == 0 and self.ld >= self.N and V.is_contiguous() and H.is_contiguous() and W.is_contiguous()
        self.V, self.Hd, self.Wd = V, H, W
        if block is None:
            if self.ld <= 4096:
                # a short column range (at eight ranks a 160 s mixture leaves 20 s = 2494 columns each) is latency-bound like one
                # mixture alone: ONE block, which the library sends down that path's split-K launches (csrc/nmf.hip)
                block = self.ld
            else:
                # R.H^T has Kp/64 workgroups per block: aim at >= 512 (two per CU), blocks of at least 256 columns (16 k-tiles)
                want = max(1, -(-512 // (g.Kp // 64)))
                block = max(256, 64 * (self.N // (64 * want)))
        if block % 64:
            raise ValueError('column blocks must be multiples of 64 columns')
        self.block = int(block)
        nb, rem = divmod(self.N, self.block)
        self.blocks = ([(0, self.block, nb)] if nb else []) + ([(nb * self.block, rem, 1)] if rem else [])    # (first column, width, count)
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=self.device)
        self._ws = [z(self.lib.gccnmf_klnmf_shared_shard_workspace_floats(self.F, width, self.K, count, self.ld)) for _, width, count in self.blocks]
        self.partial = z(self.lib.gccnmf_klnmf_shared_partial_floats(self.F, self.K))
        self.vec = z(2 * g.Kp)

    def _shards(self):
        return [(self.V[:, c0:], self.Hd[:, c0:], ws, width, count, self.ld) for (c0, width, count), ws in zip(self.blocks, self._ws)]

    def H(self):
        return [self.Hd[:self.K, :self.N].cpu().numpy()]


class CompositeSharedNMF(_SharedRun):
    """Several ``HipSharedNMF`` members of one rank -- file groups of DIFFERENT widths (a member needs equal N per object) -- trained
    as one: run() hands every member's files to ONE gccnmf_klnmf_shared_run call as separate shards (their [num || den] partials are
    added in member order before the all-reduce), with the first member's W, partial and vector scratch; the other members' W are
    synchronised afterwards.  The begin / step_a / step_b / finish protocol is kept for hosts that own the collective."""

    def __init__(self, members):
        self.members = list(members)
        first = self.members[0]
        if any((m.F, m.K, m.alpha, m.eps, m.device) != (first.F, first.K, first.alpha, first.eps, first.device) for m in self.members):
            raise ValueError('members must share F, K, the regularisation and the device')
        self.lib, self.device, self.F, self.K, self.alpha, self.eps = first.lib, first.device, first.F, first.K, first.alpha, first.eps
        self.Wd, self.partial, self.vec = first.Wd, first.partial, first.vec

    def _shards(self):
        return sum([m._shards() for m in self.members], [])

    def run(self, numIterations, group=None):
        _SharedRun.run(self, numIterations, group)
        for m in self.members[1:]:
            m.Wd.copy_(self.Wd)
        return self

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


def train_shared_dictionary(local, numIterations, group=None):
    """The shared-dictionary iteration, identical on every rank: one all-reduce of [num (Fp*Kp) || den (Kp)] per iteration; with a
    single process (or no process group) it degenerates to plain KL-NMF.  A ``local`` with run() (the HIP shards) does the whole
    loop inside the library -- kernels and collective enqueued from C; otherwise ``local`` provides begin() / step_a() -> partial
    tensor / step_b(partial) / finish() and the loop is driven from here (the NumPy oracle shards of the CPU tests)."""
    if hasattr(local, 'run'):
        return local.run(numIterations, group)
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
    ``HipSharedColumns`` on its magnitude spectrogram (column blocks of the engine's own V / H: the NMF runs in place).
    ``stereoSamples`` is the WHOLE (2, n) mixture (only the own range is uploaded)."""

    def __init__(self, stereoSamples, rank, world_size, sampleRate=16000, windowSize=1024, hopSize=256, numTDOAs=128,
                 microphoneSeparationInMetres=1.0, numTargets=3, dictionarySize=128, sparsityAlpha=0, epsilon=1e-16, seedValue=0,
                 device=None, block=None):
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
        W0, H0 = time_shard_initial_factors(e.g.F, self.T_total, e.g.K, self.t0, self.t1, epsilon, seedValue)
        g = e.g
        with torch.cuda.device(self.device):
            self._W0d = padded(W0, (g.Fp, g.Kp), self.device)
            self._H0d = padded(H0, (g.Kp, g.Np), self.device)          # [left frames | right frames], the layout of e.V / e.H
            # The rank's 2*T_r NMF columns stay where the STFT put them (e.V[0]) and where the mask stages read them (e.W[0], e.H[0]):
            # the shared-dictionary kernels take them as column blocks of those matrices (no gather / scatter, no second copy).
            self.nmf = HipSharedColumns(e.V[0], e.H[0], e.W[0], g.F, g.N, g.K, sparsityAlpha, epsilon, block=block)

    @_on_device
    def stft(self):
        self.e.stft()
        self.e.W[0].copy_(self._W0d)
        self.e.H[0].copy_(self._H0d)

    def nmf_done(self):
        pass                                                         # W and H were updated in place in the engine's buffers

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
        """-> (segment (S, 2, L) float32, first global sample index of the trimmed output).  The segment is a view of a page-locked
        buffer that the next call overwrites (61 MB per 160 s of audio: copy it if it must outlive the next step)."""
        e, g = self.e, self.e.g
        nsig = 2 * g.S
        if previous is not None:
            previous = previous.to(self.device).contiguous()
        L = (g.T - 1) * self.hop + self.n_fft if self.last else g.T * self.hop           # the samples this rank owns
        if getattr(self, '_y', None) is None:
            self._y = torch.zeros((nsig, L), dtype=torch.float32, device=self.device)
            self._y_host = torch.zeros((nsig, L), dtype=torch.float32).pin_memory()
        gain = np.float32(self.hop / float(self.n_fft) * 2)                              # gccNMFFunctions.py:155
        # frames = the previous rank's last `halo` frames, then the own ones, in ascending order -- read from where they are
        halo = self.halo if previous is not None else 0
        first = self.halo * self.hop - (self.halo - halo) * self.hop                      # rank 0 has no predecessor: its stream starts at its own frame 0
        _hip.check(e.lib.gccnmf_ola_frames_halo(_ptr(previous), halo, _ptr(e.frames[0]), nsig, self.n_fft, self.hop, g.T, first, L, gain,
                                                _ptr(self._y), _stream()), 'gccnmf_ola_frames_halo')
        self._y_host.copy_(self._y, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return self._y_host.numpy().reshape(g.S, 2, L), self.t0 * self.hop - self.n_fft // 2    # centre trim (librosaSTFT.py:283-284)

    def tdoa_indexes(self):
        return self.e.get_tdoa_indexes()[0]
