"""Worker for tests/test_gpu_distributed.py::test_two_ranks_hip_shards_over_gloo_on_one_gpu: one rank of a world-size-N
shared-dictionary training whose shard is the HIP path (HipSharedNMF); the ranks share GPU 0 and exchange the [num || den]
buffer through a real torch.distributed all-reduce (gloo, because RCCL cannot put two ranks on one device)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def problem(F, cols, seed):
    rng = np.random.RandomState(seed)
    return [(np.abs(rng.standard_normal((F, n))) + 0.05).astype(np.float32) for n in cols]


def main(out_dir, F, K, N, B, iters):
    import torch
    import torch.distributed as dist
    from gcc_nmf_amd.distributed import HipSharedNMF, shared_initial_factors, shard_files, train_shared_dictionary
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    V = problem(F, [N] * B, 11)
    mine = shard_files(B, world, rank)
    W0, H0 = shared_initial_factors(F, [N] * B, K, mine, mode='concat')
    local = train_shared_dictionary(HipSharedNMF([V[i] for i in mine], W0, H0), iters)
    np.save(os.path.join(out_dir, 'W_rank%d.npy' % rank), local.W())
    np.save(os.path.join(out_dir, 'H_rank%d.npy' % rank), np.concatenate(local.H(), axis=1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1], *[int(v) for v in sys.argv[2:7]])
