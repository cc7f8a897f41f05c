"""-m gpu: the shared-dictionary shard on the device (HipSharedNMF) against the oracle, including the
two-shards-one-exchange structure of the multi-GPU mode emulated on a single device, and a 1-rank RCCL
process group driving the real all-reduce call."""
import os
import socket

import numpy as np
import pytest

from oracle import gccnmf_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def _problem(F, K, cols, seed=1):
    rng = np.random.RandomState(seed)
    return [(np.abs(rng.standard_normal((F, n))) + 0.05).astype(np.float32) for n in cols]


def _protocol(shards, iters):
    """The four-call protocol driven from the host; the all-reduce over ranks is the explicit sum of the shards' partial buffers."""
    for s in shards:
        s.begin()
    for _ in range(iters):
        parts = [s.step_a() for s in shards]
        total = parts[0].clone()
        for p in parts[1:]:
            total += p
        for s in shards:
            s.step_b(total)
    for s in shards:
        s.finish()


@pytest.mark.parametrize('F,K,N,B,alpha', [(513, 128, 100, 4, 0), (257, 64, 70, 3, 0.2), (513, 1024, 130, 9, 0)])
def test_shared_dictionary_matches_performKLNMF_on_concatenation(F, K, N, B, alpha):
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors, train_shared_dictionary
    V = _problem(F, K, [N] * B)
    W0, H0 = shared_initial_factors(F, [N] * B, K, range(B), mode='concat')
    local = train_shared_dictionary(HipSharedNMF(V, W0, H0, sparsityAlpha=alpha), 8)
    Wr, Hr = O.performKLNMF(np.concatenate(V, axis=1), K, 8, alpha)
    W, H = local.W(), np.concatenate(local.H(), axis=1)
    assert np.linalg.norm(W - Wr) < 1e-4 * np.linalg.norm(Wr)
    assert np.linalg.norm(H - Hr) < 1e-4 * np.linalg.norm(Hr)


def test_two_shards_one_exchange():
    """Ranks' shards as two objects on one device; the all-reduce is the explicit sum of their partial buffers."""
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors, shard_files
    F, K, N, B = 513, 96, 64, 6
    V = _problem(F, K, [N] * B, seed=3)
    shards = []
    for rank in range(2):
        mine = shard_files(B, 2, rank)
        W0, H0 = shared_initial_factors(F, [N] * B, K, mine, mode='concat')
        shards.append(HipSharedNMF([V[i] for i in mine], W0, H0))
    _protocol(shards, 6)                             # the sum of the two partials == dist.all_reduce(SUM) over two ranks
    assert np.array_equal(shards[0].W(), shards[1].W())
    Wr, Hr = O.performKLNMF(np.concatenate(V, axis=1), K, 6, 0)
    H = np.concatenate(shards[0].H() + shards[1].H(), axis=1)
    assert np.linalg.norm(shards[0].W() - Wr) < 1e-4 * np.linalg.norm(Wr)
    assert np.linalg.norm(H - Hr) < 1e-4 * np.linalg.norm(Hr)


def test_one_call_run_is_bitwise_the_four_call_protocol():
    """gccnmf_klnmf_shared_run (the loop inside the library) against begin / step_a / step_b / finish driven from the host."""
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors
    F, K, N, B = 513, 128, 100, 4
    V = _problem(F, K, [N] * B, seed=7)
    W0, H0 = shared_initial_factors(F, [N] * B, K, range(B), mode='concat')
    a, b = HipSharedNMF(V, W0, H0, sparsityAlpha=0.1), HipSharedNMF(V, W0, H0, sparsityAlpha=0.1)
    _protocol([a], 7)
    b.run(7)
    assert b.collective == 'single rank'
    assert np.array_equal(a.W(), b.W())
    assert all(np.array_equal(x, y) for x, y in zip(a.H(), b.H()))


def test_rank_without_files_contributes_zeros():
    """3 files over 4 ranks (shard_files leaves rank 3 empty): the empty shard adds a zero partial and follows every W update."""
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors, shard_files
    F, K, N, B, world = 257, 64, 70, 3, 4
    V = _problem(F, K, [N] * B, seed=9)
    shards = []
    for rank in range(world):
        mine = shard_files(B, world, rank)
        W0, H0 = shared_initial_factors(F, [N] * B, K, mine, mode='concat')
        shards.append(HipSharedNMF([V[i] for i in mine], W0, H0))
    assert shards[3].B == 0 and shards[3].H() == []
    _protocol(shards, 6)
    assert all(np.array_equal(shards[0].W(), s.W()) for s in shards[1:])
    Wr, Hr = O.performKLNMF(np.concatenate(V, axis=1), K, 6, 0)
    assert np.linalg.norm(shards[3].W() - Wr) < 1e-4 * np.linalg.norm(Wr)
    H = np.concatenate(sum([s.H() for s in shards], []), axis=1)
    assert np.linalg.norm(H - Hr) < 1e-4 * np.linalg.norm(Hr)


@pytest.mark.parametrize('N,block', [(1000, 256), (1244, None), (640, 320), (200, 256)])
def test_column_blocks_of_one_matrix(N, block):
    """HipSharedColumns: the columns of ONE matrix as column blocks + ragged remainder, in place (ld > 0 shards) == performKLNMF."""
    from gcc_nmf_amd.distributed import HipSharedColumns
    from gcc_nmf_amd.engine import Geometry, klnmf_initial_factors, padded
    F, K, iters = 513, 128, 8
    V = _problem(F, K, [N], seed=N)[0]
    W0, H0 = klnmf_initial_factors(F, N, K)
    g = Geometry(F, 1, K)
    ld = -(-N // 64) * 64
    Vd, Hd, Wd = padded(V, (g.Fp, ld), 'cuda'), padded(H0, (g.Kp, ld), 'cuda'), padded(W0, (g.Fp, g.Kp), 'cuda')
    nmf = HipSharedColumns(Vd, Hd, Wd, F, N, K, block=block)
    assert sum(w * c for _, w, c in nmf.blocks) == N
    nmf.run(iters)
    Wr, Hr = O.performKLNMF(V, K, iters, 0)
    assert np.linalg.norm(nmf.W() - Wr) < 1e-4 * np.linalg.norm(Wr)
    assert np.linalg.norm(nmf.H()[0] - Hr) < 1e-4 * np.linalg.norm(Hr)
    assert ld == N or float(Hd[:, N:].abs().max()) == 0.0                       # the padding columns stay zero


def test_file_groups_of_different_widths_as_one_training():
    """CompositeSharedNMF: two groups of files with different N in ONE gccnmf_klnmf_shared_run call (two shards) == performKLNMF on
    the concatenation, and bitwise the host-driven protocol over the same members."""
    from gcc_nmf_amd.distributed import CompositeSharedNMF, HipSharedNMF, shared_initial_factors, train_shared_dictionary
    F, K, cols = 257, 96, [70, 70, 70, 130, 130]
    V = _problem(F, K, cols, seed=21)
    W0, H0 = shared_initial_factors(F, cols, K, range(5), mode='concat')

    def build():
        return CompositeSharedNMF([HipSharedNMF(V[:3], W0, H0[:3]), HipSharedNMF(V[3:], W0, H0[3:])])
    a, b = build(), build()
    train_shared_dictionary(a, 7)                         # one C call, two shards
    _protocol([b], 7)
    assert np.array_equal(a.W(), b.W()) and np.array_equal(a.members[1].W(), a.W())
    Ha = np.concatenate(a.members[0].H() + a.members[1].H(), axis=1)
    assert np.array_equal(Ha, np.concatenate(b.members[0].H() + b.members[1].H(), axis=1))
    Wr, Hr = O.performKLNMF(np.concatenate(V, axis=1), K, 7, 0)
    assert np.linalg.norm(a.W() - Wr) < 1e-4 * np.linalg.norm(Wr) and np.linalg.norm(Ha - Hr) < 1e-4 * np.linalg.norm(Hr)


def test_short_column_range_takes_the_latency_path():
    """A rank's whole, short column range (one 'file' in the plain layout, K and N >= 512) runs the split-K launches of the
    one-mixture-alone path inside gccnmf_klnmf_shared_run: == performKLNMF, == the column-block form to summation order, and
    repeatable bit for bit."""
    from gcc_nmf_amd.distributed import HipSharedColumns
    from gcc_nmf_amd.engine import Geometry, klnmf_initial_factors, padded
    F, N, K, iters = 513, 1244, 512, 8
    V = _problem(F, K, [N], seed=4)[0]
    W0, H0 = klnmf_initial_factors(F, N, K)
    g = Geometry(F, 1, K)
    ld = -(-N // 64) * 64
    outs = []
    for block in (None, None, 256):
        Vd, Hd, Wd = padded(V, (g.Fp, ld), 'cuda'), padded(H0, (g.Kp, ld), 'cuda'), padded(W0, (g.Fp, g.Kp), 'cuda')
        nmf = HipSharedColumns(Vd, Hd, Wd, F, N, K, block=block)
        nmf.run(iters)
        outs.append((nmf.W(), nmf.H()[0], nmf.blocks))
    assert outs[0][2] == [(0, N, 1)] and len(outs[2][2]) == 2
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.linalg.norm(outs[0][0] - outs[2][0]) < 1e-5 * np.linalg.norm(outs[2][0])
    Wr, Hr = O.performKLNMF(V, K, iters, 0)
    assert np.linalg.norm(outs[0][0] - Wr) < 1e-4 * np.linalg.norm(Wr) and np.linalg.norm(outs[0][1] - Hr) < 1e-4 * np.linalg.norm(Hr)


def test_library_rccl_communicator_single_rank():
    """csrc/collective.hip on hardware: librccl bound by dlopen, a 1-rank communicator from a unique id, ncclAllReduce enqueued from C
    inside gccnmf_klnmf_shared_run -- same bits as the run without a collective."""
    import ctypes
    from gcc_nmf_amd import _hip
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors
    from gcc_nmf_amd.engine import _ptr, _stream
    lib = _hip.lib()
    assert lib.gccnmf_rccl_available() == 1
    ident = ctypes.create_string_buffer(_hip.RCCL_UNIQUE_ID_BYTES)
    _hip.check(lib.gccnmf_rccl_unique_id(ident), 'gccnmf_rccl_unique_id')
    comm = ctypes.c_void_p()
    torch.cuda.set_device(0)
    _hip.check(lib.gccnmf_rccl_comm_init(ident.raw, 1, 0, ctypes.byref(comm)), 'gccnmf_rccl_comm_init')
    try:
        t = torch.arange(4096, dtype=torch.float32, device='cuda')
        _hip.check(lib.gccnmf_rccl_allreduce(comm, _ptr(t), t.numel(), _stream()), 'gccnmf_rccl_allreduce')
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(4096, dtype=torch.float32))
        F, K, N, B = 129, 32, 40, 2
        V = _problem(F, K, [N] * B, seed=5)
        W0, H0 = shared_initial_factors(F, [N] * B, K, range(B), mode='concat')
        a, b = HipSharedNMF(V, W0, H0), HipSharedNMF(V, W0, H0)
        a.run(5)
        arr = (_hip.SharedShard * 1)()
        arr[0].V, arr[0].H, arr[0].workspace, arr[0].N, arr[0].batch, arr[0].ld = _ptr(b.V), _ptr(b.Hd), _ptr(b.ws), N, B, 0
        _hip.check(lib.gccnmf_klnmf_shared_run(arr, 1, _ptr(b.Wd), _ptr(b.partial), _ptr(b.vec), F, K, 5, 0.0, 1e-16,
                                               lib.gccnmf_rccl_allreduce_hook(), comm, _stream()), 'gccnmf_klnmf_shared_run')
        torch.cuda.synchronize()
        assert np.array_equal(a.W(), b.W())
    finally:
        _hip.check(lib.gccnmf_rccl_comm_destroy(comm), 'gccnmf_rccl_comm_destroy')


def test_rccl_single_rank_group():
    import torch.distributed as dist
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors, train_shared_dictionary
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        t = torch.ones(1024, device='cuda')
        dist.all_reduce(t)                                     # RCCL call path is alive
        assert float(t.sum()) == 1024.0
        F, K, N, B = 129, 32, 40, 2
        V = _problem(F, K, [N] * B, seed=5)
        W0, H0 = shared_initial_factors(F, [N] * B, K, range(B), mode='concat')
        local = train_shared_dictionary(HipSharedNMF(V, W0, H0), 5)
        Wr, Hr = O.performKLNMF(np.concatenate(V, axis=1), K, 5, 0)
        assert np.linalg.norm(local.W() - Wr) < 1e-4 * np.linalg.norm(Wr)
    finally:
        dist.destroy_process_group()


def test_forced_hook_on_a_one_rank_nccl_group_rccl_and_torch_paths():
    """One GPU is enough to drive BOTH all-reduce routes of the nccl backend end to end (GCCNMF_COLLECTIVE_FORCE=1 builds the hook for a
    single rank): the library's own RCCL communicator created through the process group (unique id broadcast, init agreed on by
    MIN all-reduces, ncclAllReduce enqueued from C), and GCCNMF_COLLECTIVE=torch, the host callback into torch.distributed."""
    import torch.distributed as dist
    from gcc_nmf_amd import distributed as D
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=torch.device('cuda', 0))
    saved = {k: os.environ.get(k) for k in ('GCCNMF_COLLECTIVE', 'GCCNMF_COLLECTIVE_FORCE')}
    try:
        F, K, N, B = 129, 32, 40, 2
        V = _problem(F, K, [N] * B, seed=5)
        W0, H0 = D.shared_initial_factors(F, [N] * B, K, range(B), mode='concat')
        os.environ['GCCNMF_COLLECTIVE_FORCE'] = '1'
        os.environ.pop('GCCNMF_COLLECTIVE', None)
        a = D.train_shared_dictionary(D.HipSharedNMF(V, W0, H0), 5)
        assert a.collective.startswith('rccl'), a.collective
        os.environ['GCCNMF_COLLECTIVE'] = 'torch'
        b = D.train_shared_dictionary(D.HipSharedNMF(V, W0, H0), 5)
        assert b.collective.startswith('torch.distributed (nccl)'), b.collective
        os.environ.pop('GCCNMF_COLLECTIVE_FORCE')
        c = D.train_shared_dictionary(D.HipSharedNMF(V, W0, H0), 5)
        assert c.collective == 'single rank'
        assert np.array_equal(a.W(), c.W()) and np.array_equal(b.W(), c.W())          # a sum over one rank changes nothing
        Wr, _ = O.performKLNMF(np.concatenate(V, axis=1), K, 5, 0)
        assert np.linalg.norm(c.W() - Wr) < 1e-4 * np.linalg.norm(Wr)
        # a re-initialised default group gets a NEW communicator (the cache is keyed by the group object)
        assert len(D._rccl_comms) == 1
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        D.destroy_rccl_communicators()
        dist.destroy_process_group()


def test_two_gpus_library_rccl_between_devices(tmp_path):
    """ncclAllReduce between two DEVICES through the library's communicator (skipped on a one-GPU box): W identical on both ranks and
    equal to the single-process result."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    import sys
    from conftest import REPO
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors, train_shared_dictionary
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from shared_rank_worker import problem
    F, K, N, B, iters = 513, 256, 128, 6, 8
    os.environ['GCCNMF_WORKER_BACKEND'] = 'nccl'
    try:
        _run_ranks(2, 'shared_rank_worker.py', [tmp_path, F, K, N, B, iters])
    finally:
        os.environ.pop('GCCNMF_WORKER_BACKEND')
    W = [np.load(tmp_path / ('W_rank%d.npy' % i)) for i in range(2)]
    assert np.array_equal(W[0], W[1])
    assert open(tmp_path / 'collective_rank1.txt').read().startswith('rccl')
    W0, H0 = shared_initial_factors(F, [N] * B, K, range(B), mode='concat')
    one = train_shared_dictionary(HipSharedNMF(problem(F, [N] * B, 11), W0, H0), iters)
    assert np.linalg.norm(W[0] - one.W()) < 1e-5 * np.linalg.norm(one.W())


def test_shared_run_rejects_a_second_call_on_the_same_device():
    """gccnmf_klnmf_shared_run owns per-device side streams: a call that arrives while another is inside (here: from the all-reduce
    callback of the first) is rejected with GCCNMF_ERR_UNSUPPORTED instead of racing."""
    import ctypes
    from gcc_nmf_amd import _hip
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors
    from gcc_nmf_amd.engine import _ptr, _stream
    lib = _hip.lib()
    F, K, N, B = 129, 32, 40, 2
    V = _problem(F, K, [N] * B, seed=5)
    W0, H0 = shared_initial_factors(F, [N] * B, K, range(B), mode='concat')
    local = HipSharedNMF(V, W0, H0)
    descs = local._shards()
    arr = (_hip.SharedShard * len(descs))()
    for d, (Vd, Hd, ws, n, batch, ld) in zip(arr, descs):
        d.V, d.H, d.workspace, d.N, d.batch, d.ld = _ptr(Vd), _ptr(Hd), _ptr(ws), n, batch, ld
    seen = []

    def hook(ctx, buf, count, stream):
        seen.append(lib.gccnmf_klnmf_shared_run(arr, len(descs), _ptr(local.Wd), _ptr(local.partial), _ptr(local.vec), F, K, 1, 0.0, 1e-16,
                                                None, None, _stream()))
        return 0
    cb = _hip.ALLREDUCE_FN(hook)
    rc = lib.gccnmf_klnmf_shared_run(arr, len(descs), _ptr(local.Wd), _ptr(local.partial), _ptr(local.vec), F, K, 2, 0.0, 1e-16,
                                     ctypes.cast(cb, ctypes.c_void_p).value, None, _stream())
    torch.cuda.synchronize()
    assert rc == 0 and seen == [3, 3]                      # GCCNMF_ERR_UNSUPPORTED, twice (once per iteration)
    assert lib.gccnmf_klnmf_shared_run(arr, len(descs), _ptr(local.Wd), _ptr(local.partial), _ptr(local.vec), F, K, 1, 0.0, 1e-16, None, None,
                                       _stream()) == 0        # and the guard is released afterwards


def test_pretraining_cache_roundtrip(tmp_path):
    """Reference-format dictionary cache (data/pretrainedW/W_<K>.npy, float32 (F,K)) trained on the GPU from the mixtures in
    DATA_DIR; second load hits the cache; dictionaries agree with the oracle trained on the same matrix."""
    import shutil
    from conftest import GOLDEN
    from gcc_nmf_amd import pretraining as P
    shutil.copy(os.path.join(GOLDEN, 'data', 'dev1_female3_liverec_130ms_1m_mix.wav'), tmp_path)
    P.configure(str(tmp_path))
    W = P.loadPretrainedW(32)
    path = tmp_path / 'pretrainedW' / 'W_32.npy'
    assert path.exists() and W.shape == (513, 32) and W.dtype == np.float32
    assert np.allclose(np.linalg.norm(W, axis=0), 1.0, atol=1e-5)
    stamp = path.stat().st_mtime_ns
    assert np.array_equal(P.loadPretrainedW(32), W) and path.stat().st_mtime_ns == stamp      # cached, not retrained
    V = P.buildTrainingSet(1024, 512)
    assert V.shape == (513, 2 * 311)
    Wr, _ = O.performKLNMF(V, 32, 100, 0)
    assert np.linalg.norm(W - Wr) < 1e-3 * np.linalg.norm(Wr)
    d = P.getDictionariesW(1024, [32], ordered=True)
    assert list(d) == ['Pretrained', 'Random'] and d['Pretrained'][32].shape == (513, 32)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_hip_shards_over_gloo_on_one_gpu(tmp_path):
    """Two PROCESSES, each with a HIP shard of the files, one real all-reduce per iteration (gloo; both ranks on GPU 0): every
    rank ends with the same W, equal to the single-process result over all files to summation order."""
    import subprocess
    import sys
    from conftest import REPO
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors, train_shared_dictionary
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from shared_rank_worker import problem
    F, K, N, B, iters = 513, 1024, 128, 8, 12
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(REPO, 'tests', 'shared_rank_worker.py'), str(tmp_path)] + [str(v) for v in (F, K, N, B, iters)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    W = [np.load(tmp_path / ('W_rank%d.npy' % i)) for i in range(2)]
    H = np.concatenate([np.load(tmp_path / ('H_rank%d.npy' % i)) for i in range(2)], axis=1)
    assert np.array_equal(W[0], W[1])
    assert 'gloo' in open(tmp_path / 'collective_rank1.txt').read()          # the C loop called back into torch.distributed
    V = problem(F, [N] * B, 11)
    W0, H0 = shared_initial_factors(F, [N] * B, K, range(B), mode='concat')
    one = train_shared_dictionary(HipSharedNMF(V, W0, H0), iters)
    assert np.linalg.norm(W[0] - one.W()) < 1e-5 * np.linalg.norm(one.W())
    assert np.linalg.norm(H - np.concatenate(one.H(), axis=1)) < 1e-5 * np.linalg.norm(H)


def _run_ranks(world, script, args, timeout=600):
    import subprocess
    import sys
    from conftest import REPO
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(REPO, 'tests', script)] + [str(a) for a in args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]


def test_eight_ranks_more_ranks_than_files(tmp_path):
    """8-rank rehearsal on one GPU (gloo): 3 files -> ranks 3..7 hold no columns; every rank ends with the same W = the 1-rank W."""
    import sys
    from conftest import REPO
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors, train_shared_dictionary
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from shared_rank_worker import problem
    F, K, N, B, iters = 513, 256, 128, 3, 6
    _run_ranks(8, 'shared_rank_worker.py', [tmp_path, F, K, N, B, iters])
    W = [np.load(tmp_path / ('W_rank%d.npy' % i)) for i in range(8)]
    assert all(np.array_equal(W[0], w) for w in W[1:])
    W0, H0 = shared_initial_factors(F, [N] * B, K, range(B), mode='concat')
    one = train_shared_dictionary(HipSharedNMF(problem(F, [N] * B, 11), W0, H0), iters)
    assert np.linalg.norm(W[0] - one.W()) < 1e-5 * np.linalg.norm(one.W())


def test_eight_ranks_at_config4_per_rank_shape(tmp_path):
    """BASELINE config 4 rehearsed on one GPU: 8 ranks x 64 files of (513, 1244), K = 1024, per-file initialisation, the all-reduce of
    541,696 floats per iteration over gloo; W identical on every rank and within 1e-5 (rel. Frobenius) of ONE rank holding all 512
    files ('W parity vs single-GPU <= 1e-5')."""
    import sys
    from conftest import REPO
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors, train_shared_dictionary
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from shared_rank_worker import problem
    F, K, N, B, iters = 513, 1024, 1244, 512, 3
    _run_ranks(8, 'shared_rank_worker.py', [tmp_path, F, K, N, B, iters, 'per_file'], timeout=900)
    W = [np.load(tmp_path / ('W_rank%d.npy' % i)) for i in range(8)]
    assert all(np.array_equal(W[0], w) for w in W[1:])
    W0, H0 = shared_initial_factors(F, [N] * B, K, range(B), mode='per_file')
    one = train_shared_dictionary(HipSharedNMF(problem(F, [N] * B, 11), W0, H0), iters)
    rel = np.linalg.norm(W[0] - one.W()) / np.linalg.norm(one.W())
    print('config-4 shape, 8 ranks vs 1 rank: W rel. Frobenius %.2e' % rel)
    assert rel < 1e-5


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment re-executes itself under torch.distributed.run, and the JSON line
    reports the ranks that answered an all-reduce (here over gloo, two ranks sharing the one GPU of the test box)."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['GCCNMF_BENCH_BACKEND'] = 'gloo'
    common = ['--files', '4', '--seconds', '2', '--iterations', '5', '--dictionary-size', '128', '--steps', '1', '--warmup', '0',
              '--skip-cpu-baseline', '--skip-roofline']
    # 2 ranks, and the 8-rank rehearsal of the driver's scaling run (every mode; no flags beyond --gpus / --mode needed there)
    for gpus, mode in ((2, 'separate'), (2, 'shared-dictionary'), (8, 'separate'), (8, 'shared-dictionary'), (8, 'time-sharded')):
        r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', str(gpus), '--mode', mode] + common, env=env,
                           capture_output=True, text=True, timeout=900, cwd=REPO)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
        assert line['n_gpus'] == gpus and line['ranks_seen'] == gpus, line
        assert np.isfinite(line['value']) and line['value'] > 0
    # a launcher environment that disagrees with --gpus is an error, not a warning
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2'] + common, env=env2, capture_output=True, text=True,
                       timeout=600, cwd=REPO)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_shared_reset_reloads_new_factors():
    """reset(W0, H0) with NEW arrays must use them; reset() restores the last upload."""
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors
    F, K, N, B = 129, 32, 40, 2
    V = _problem(F, K, [N] * B, seed=5)
    W0, H0 = shared_initial_factors(F, [N] * B, K, range(B), mode='concat')
    s = HipSharedNMF(V, W0, H0)
    assert np.array_equal(s.W(), W0)
    W1 = (W0 * 0.5).astype(np.float32)
    s.reset(W1, [h * 2 for h in H0])
    assert np.array_equal(s.W(), W1) and np.array_equal(s.H()[1], H0[1] * 2)
    s.begin(); s.step_b(s.step_a()); s.finish()
    assert not np.array_equal(s.W(), W1)
    s.reset()
    assert np.array_equal(s.W(), W1)


def test_time_sharded_single_rank_equals_the_engine_and_the_oracle():
    """Mode 3 with one rank: the frame-sharded path (shared-dictionary NMF kernels + frame-wise overlap-add) against the batch engine
    and the oracle pipeline on the same mixture."""
    from gcc_nmf_amd.distributed import HipTimeShard, separate_time_sharded, stitch_time_shards
    from gcc_nmf_amd.engine import GCCNMFEngine
    x = O.synthetic_mixture(11, numSamples=48000)
    local = HipTimeShard(x, 0, 1, dictionarySize=64)
    y = stitch_time_shards([separate_time_sharded(local, 20)], 3, local.T_total, 256)
    r = O.runGCCNMF(x, 16000, 1024, 256, 128, 1.0, 3, dictionarySize=64, numIterations=20, return_intermediates=True)
    assert local.tdoa_indexes().tolist() == r['idx']
    assert np.sqrt(np.mean((y.astype(np.float64) - r['y']) ** 2)) < 1e-5
    e = GCCNMFEngine(48000, dictionarySize=64, numIterations=20)
    ye = e.separate(x)[0]
    assert np.sqrt(np.mean((y - ye) ** 2)) < 1e-5


def test_time_sharded_two_ranks_hip_over_gloo(tmp_path):
    """Two PROCESSES with half of the frames of one mixture each (HIP shards on GPU 0, gloo collectives): same TDOAs on both ranks, the
    stitched waveform equals the single-rank one up to the all-reduce summation order, and the oracle's within the waveform bar."""
    import subprocess
    import sys
    from conftest import REPO
    from gcc_nmf_amd.distributed import HipTimeShard, separate_time_sharded, stitch_time_shards
    n, K, iters = 64000, 128, 30
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(REPO, 'tests', 'time_shard_worker.py'), str(tmp_path), str(n), str(K), str(iters)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    parts = [np.load(tmp_path / ('time_rank%d.npz' % k)) for k in range(2)]
    assert parts[0]['idx'].tolist() == parts[1]['idx'].tolist()
    T = int(parts[0]['T'])
    y2 = stitch_time_shards([(p['seg'], int(p['start'])) for p in parts], 3, T, 256)
    x = O.synthetic_mixture(11, numSamples=n)
    one = HipTimeShard(x, 0, 1, dictionarySize=K)
    y1 = stitch_time_shards([separate_time_sharded(one, iters)], 3, T, 256)
    assert one.tdoa_indexes().tolist() == parts[0]['idx'].tolist()
    assert np.sqrt(np.mean((y2 - y1) ** 2)) < 1e-5
    ref = O.runGCCNMF(x, 16000, 1024, 256, 128, 1.0, 3, dictionarySize=K, numIterations=iters)
    assert np.sqrt(np.mean((y2.astype(np.float64) - ref) ** 2)) < 1e-5        # bar: 1e-4
