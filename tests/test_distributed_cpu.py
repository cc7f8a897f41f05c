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
