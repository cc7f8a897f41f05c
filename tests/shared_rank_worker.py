"""Worker for tests/test_gpu_distributed.py: one rank of a world-size-N shared-dictionary training whose shard is the HIP path
(HipSharedNMF, the whole loop inside gccnmf_klnmf_shared_run); the ranks share GPU 0 and exchange the [num || den] buffer through a
real torch.distributed all-reduce (gloo, because RCCL cannot put two ranks on one device).  Ranks beyond the number of files hold
no columns (HipSharedNMF with an empty file list)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def problem(F, cols, seed, files=None):
    """file i of the problem depends on (seed, i) only, so a rank can build just its own files"""
    out = []
    for i in (range(len(cols)) if files is None else files):
        rng = np.random.RandomState(seed * 100003 + i)
        out.append((np.abs(rng.standard_normal((F, cols[i]))) + 0.05).astype(np.float32))
    return out


def main(out_dir, F, K, N, B, iters, init='concat'):
    import torch
    import torch.distributed as dist
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors, shard_files, train_shared_dictionary
    import datetime
    backend = os.environ.get('GCCNMF_WORKER_BACKEND', 'gloo')                     # nccl: one rank per GPU, the library's RCCL communicator
    if backend == 'nccl':
        local_rank = int(os.environ['LOCAL_RANK'])
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', timeout=datetime.timedelta(seconds=180), device_id=torch.device('cuda', local_rank))
    else:
        dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=180))      # a failed rank must not park the others for half an hour
        torch.cuda.set_device(0)
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_files(B, world, rank)
    V = problem(F, [N] * B, 11, mine)
    W0, H0 = shared_initial_factors(F, [N] * B, K, mine, mode=init)
    local = train_shared_dictionary(HipSharedNMF(V, W0, H0), iters)
    np.save(os.path.join(out_dir, 'W_rank%d.npy' % rank), local.W())
    if mine and B * N * K < 1 << 26:
        np.save(os.path.join(out_dir, 'H_rank%d.npy' % rank), np.concatenate(local.H(), axis=1))
    with open(os.path.join(out_dir, 'collective_rank%d.txt' % rank), 'w') as f:
        f.write(local.collective)
    dist.barrier()
    from gcc_nmf_amd.distributed import destroy_rccl_communicators
    destroy_rccl_communicators()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1], *[int(v) for v in sys.argv[2:7]], *sys.argv[7:8])
