"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: file sharding and the shared-dictionary
iteration with its one all-reduce per iteration.  The per-rank arithmetic is supplied by the NumPy oracle
shard (oracle/shared_nmf_oracle.py); on the GPU box the same protocol is driven by HipSharedNMF
(tests/test_gpu_distributed.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def test_shard_files():
    from gcc_nmf_amd.distributed import shard_files
    for n, w in [(64, 8), (512, 8), (10, 4), (3, 8), (0, 2), (7, 1)]:
        shards = [shard_files(n, w, r) for r in range(w)]
        assert sorted(sum(shards, [])) == list(range(n))                      # disjoint cover
        assert max(map(len, shards)) - min(map(len, shards)) <= 1             # balanced
        assert all(s == list(range(s[0], s[0] + len(s))) for s in shards if s)  # contiguous
    assert shard_files(64, 8, 3) == list(range(24, 32))
    with pytest.raises(ValueError):
        shard_files(4, 2, 2)
    # files of different lengths: balanced by frames, every file on exactly one rank, the same deal on every rank
    rng = np.random.RandomState(3)
    frames = [int(v) for v in rng.choice([310, 622, 935, 1243], size=61)]
    for w in (2, 4, 8):
        shards = [shard_files(61, w, r, frames=frames) for r in range(w)]
        assert sorted(sum(shards, [])) == list(range(61)) and all(s == sorted(s) for s in shards)
        loads = [sum(frames[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(frames)                     # greedy longest-first: within one file of each other
        assert max(loads) <= 1.05 * sum(frames) / w
    assert [len(shard_files(10, 4, r, frames=[5] * 10)) for r in range(4)] == [3, 3, 2, 2]
    with pytest.raises(ValueError):
        shard_files(3, 2, 0, frames=[1, 2])


def _problem():
    rng = np.random.RandomState(42)
    F, K, cols = 40, 6, [30, 30, 30, 30]
    V = [(np.abs(rng.standard_normal((F, n))) + 0.05).astype(np.float32) for n in cols]
    return F, K, cols, V


def test_shared_dictionary_single_process_is_performKLNMF():
    from gcc_nmf_amd.distributed import shared_initial_factors, train_shared_dictionary
    from oracle.shared_nmf_oracle import NumpySharedNMF
    from oracle import gccnmf_oracle as O
    F, K, cols, V = _problem()
    W0, H0 = shared_initial_factors(F, cols, K, range(4), mode='concat')
    for alpha in (0, 0.25):
        local = train_shared_dictionary(NumpySharedNMF(V, W0, H0, sparsityAlpha=alpha), 12)
        Wr, Hr = O.performKLNMF(np.concatenate(V, axis=1), K, 12, alpha)
        assert np.abs(local.W() - Wr).max() < 1e-5 * np.abs(Wr).max()
        assert np.abs(np.concatenate(local.H(), axis=1) - Hr).max() < 1e-5 * np.abs(Hr).max()


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from gcc_nmf_amd.distributed import shard_files, shared_initial_factors, train_shared_dictionary
    from oracle.shared_nmf_oracle import NumpySharedNMF
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    try:
        F, K, cols, V = _problem()
        mine = shard_files(len(V), world, rank)
        W0, H0 = shared_initial_factors(F, cols, K, mine, mode='concat')
        local = train_shared_dictionary(NumpySharedNMF([V[i] for i in mine], W0, H0), 12)
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), W=local.W(), H=np.concatenate(local.H(), axis=1), files=np.array(mine))
    finally:
        dist.destroy_process_group()


def test_shared_dictionary_two_ranks_gloo(tmp_path):
    from oracle import gccnmf_oracle as O
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    assert r0['files'].tolist() == [0, 1] and r1['files'].tolist() == [2, 3]
    assert np.array_equal(r0['W'], r1['W'])                                  # replicated dictionary stays identical
    F, K, cols, V = _problem()
    Wr, Hr = O.performKLNMF(np.concatenate(V, axis=1), K, 12, 0)
    assert np.abs(r0['W'] - Wr).max() < 1e-5 * np.abs(Wr).max()              # differs from 1 process by summation order only
    H = np.concatenate([r0['H'], r1['H']], axis=1)
    assert np.abs(H - Hr).max() < 1e-5 * np.abs(Hr).max()


def test_per_file_initialisation_is_world_size_independent():
    from gcc_nmf_amd.distributed import shard_files, shared_initial_factors
    F, K, cols = 20, 4, [10, 12, 9, 11, 10]
    Wa, Ha = shared_initial_factors(F, cols, K, range(5), mode='per_file')
    for world in (2, 3):
        for rank in range(world):
            mine = shard_files(5, world, rank)
            Wb, Hb = shared_initial_factors(F, cols, K, mine, mode='per_file')
            assert np.array_equal(Wa, Wb)
            for j, i in enumerate(mine):
                assert np.array_equal(Ha[i], Hb[j])


# ---- mode 3: one long mixture sharded over frame windows ---------------------------------------------------------------------
def _long_mixture():
    from oracle import gccnmf_oracle as O
    return O.synthetic_mixture(11, numSamples=24000)          # 90 frames at hop 256


def test_shard_frames_and_initial_factors():
    from gcc_nmf_amd.distributed import shard_frames, time_shard_initial_factors
    for T, w in [(90, 2), (622, 8), (7, 3)]:
        r = [shard_frames(T, w, k) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == T and all(r[k][1] == r[k + 1][0] for k in range(w - 1))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    rs = np.random.RandomState(0)
    W = rs.random_sample((5, 3)).astype(np.float32)
    H = rs.random_sample((3, 20)).astype(np.float32)
    W0, H0 = time_shard_initial_factors(5, 10, 3, 4, 7, epsilon=0)
    assert np.array_equal(W0, W) and np.array_equal(H0, np.concatenate([H[:, 4:7], H[:, 14:17]], axis=1))


def test_time_sharded_single_rank_is_the_reference_pipeline():
    """One shard holding every frame == oracle.runGCCNMF (gccNMF/runGCCNMF.py:36-52) on the whole mixture."""
    from gcc_nmf_amd.distributed import separate_time_sharded, stitch_time_shards
    from oracle.time_shard_oracle import NumpyTimeShard
    from oracle import gccnmf_oracle as O
    x = _long_mixture()
    local = NumpyTimeShard(x, 0, 1, dictionarySize=16)
    seg = separate_time_sharded(local, 10)
    y = stitch_time_shards([seg], 3, local.T_total, 256)
    r = O.runGCCNMF(x, 16000, 1024, 256, 128, 1.0, 3, dictionarySize=16, numIterations=10, return_intermediates=True)
    assert local.idx == r['idx'] and y.shape == r['y'].shape
    assert np.abs(y - r['y']).max() < 1e-5 * np.abs(r['y']).max()


def _time_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, REPO)
    from gcc_nmf_amd.distributed import separate_time_sharded
    from oracle.time_shard_oracle import NumpyTimeShard
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    try:
        local = NumpyTimeShard(_long_mixture(), rank, world, dictionarySize=16)
        seg, start = separate_time_sharded(local, 10)
        np.savez(os.path.join(out_dir, 'time_rank%d.npz' % rank), seg=seg, start=start, idx=np.array(local.idx), W=local.W, T=local.T_total)
    finally:
        dist.destroy_process_group()


def test_time_sharded_two_ranks_gloo(tmp_path):
    """Two processes, each with half of the frames: W all-reduce per iteration, one all-reduce of the angular spectrum, halo frames
    all-gathered; the stitched waveform equals the single-shard one up to the summation order of the two collectives."""
    from gcc_nmf_amd.distributed import separate_time_sharded, stitch_time_shards
    from oracle.time_shard_oracle import NumpyTimeShard
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_time_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / ('time_rank%d.npz' % k)) for k in range(2)]
    assert np.array_equal(r[0]['W'], r[1]['W']) and r[0]['idx'].tolist() == r[1]['idx'].tolist()
    T = int(r[0]['T'])
    y2 = stitch_time_shards([(r[k]['seg'], int(r[k]['start'])) for k in range(2)], 3, T, 256)
    one = NumpyTimeShard(_long_mixture(), 0, 1, dictionarySize=16)
    y1 = stitch_time_shards([separate_time_sharded(one, 10)], 3, T, 256)
    assert one.idx == r[0]['idx'].tolist()
    assert int(r[0]['start']) + r[0]['seg'].shape[2] == int(r[1]['start'])          # the segments tile the output
    assert np.abs(y2 - y1).max() < 1e-5 * np.abs(y1).max()
    assert np.sqrt(np.mean((y2 - y1) ** 2)) < 1e-6


# ---- the all-reduce hook of gccnmf_klnmf_shared_run over a non-RCCL backend ------------------------------------------------------
def _hook_worker(rank, world, port, out_dir):
    import ctypes
    import sys
    sys.path.insert(0, REPO)
    from gcc_nmf_amd import _hip
    from gcc_nmf_amd.distributed import collective_hook
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    try:
        partial = torch.full((1000,), float(rank + 1))
        fn, ctx, keep, what = collective_hook(partial)
        assert ctx is None and 'gloo' in what
        call = ctypes.cast(fn, _hip.ALLREDUCE_FN)                      # what the C loop does once per iteration
        assert call(None, partial.data_ptr(), partial.numel(), None) == 0
        assert call(None, partial.data_ptr() + 4, partial.numel(), None) == 1 and isinstance(keep[1][0], RuntimeError)   # carried out, not raised through C
        np.save(os.path.join(out_dir, 'hook_rank%d.npy' % rank), partial.numpy())
    finally:
        dist.destroy_process_group()


def test_allreduce_hook_is_a_host_callback_over_gloo(tmp_path):
    """distributed.collective_hook with a gloo group: the gccnmf_allreduce_fn handed to gccnmf_klnmf_shared_run is a ctypes callback
    that runs dist.all_reduce on the partial tensor and turns an exception into a status code."""
    from gcc_nmf_amd.distributed import collective_hook
    assert collective_hook(torch.zeros(4))[:2] == (None, None)               # no process group: single rank, no hook
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_hook_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / ('hook_rank%d.npy' % r)), np.full(1000, 3.0, np.float32))


def test_collective_timeout_and_ipc_environment_messages(monkeypatch):
    """A collective set-up that never returns surfaces as a HipLibraryError that says what to check (ADVICE r3 / VERDICT r3 #4c), and the
    package sets the dmabuf-IPC switch RCCL needs on this driver stack unless the user chose a value."""
    import time
    import gcc_nmf_amd                                   # noqa: F401  (import sets the default)
    from gcc_nmf_amd import _hip, distributed as D
    assert os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') is not None
    monkeypatch.setenv('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    assert D._ipc_env_note() == ''
    assert D._call_with_timeout(lambda: 41 + 1, 5, 'quick') == 42
    with pytest.raises(ValueError):
        D._call_with_timeout(lambda: (_ for _ in ()).throw(ValueError('boom')), 5, 'raises')
    monkeypatch.setenv('HSA_ENABLE_IPC_MODE_LEGACY', '1')
    assert 'HSA_ENABLE_IPC_MODE_LEGACY' in D._ipc_env_note()
    t0 = time.time()
    with pytest.raises(_hip.HipLibraryError) as e:
        D._call_with_timeout(lambda: time.sleep(30), 0.2, 'gccnmf_rccl_comm_init (rank 1 of 2, device cuda:1)')
    assert time.time() - t0 < 5
    msg = str(e.value)
    assert 'gccnmf_rccl_comm_init' in msg and 'GCCNMF_COLLECTIVE=torch' in msg and 'HSA_ENABLE_IPC_MODE_LEGACY' in msg
    # inside the communicator set-up the same timeout is a RESULT, not an exception: the rank still takes part in the agreement all-reduce
    # (its peers are waiting there) and every rank falls back to torch.distributed together (VERDICT r4 #8c)
    t0 = time.time()
    ok, why = D._try_rccl_init(lambda: time.sleep(30), 0.2, 'gccnmf_rccl_comm_init (rank 1 of 2, device cuda:1)')
    assert ok is False and time.time() - t0 < 5 and 'did not return within' in why and 'GCCNMF_COLLECTIVE=torch' in why
    assert D._try_rccl_init(lambda: True, 5, 'quick') == (True, None) and D._try_rccl_init(lambda: False, 5, 'refused') == (False, None)
    # a single process (no group) needs no exchange; an unknown route is an error only once a group exists
    hook = D.collective_hook(None)
    assert hook[0] is None and hook[3] == 'single rank'
