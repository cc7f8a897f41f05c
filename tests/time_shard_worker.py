"""Worker for tests/test_gpu_distributed.py::test_time_sharded_two_ranks_hip_over_gloo: one rank of a frame-sharded separation of ONE
mixture whose shard is the HIP path (HipTimeShard); the ranks share GPU 0 and run the real collectives over gloo."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main(out_dir, n, K, iters):
    import torch
    import torch.distributed as dist
    from gcc_nmf_amd.distributed import HipTimeShard, separate_time_sharded
    from gcc_nmf_amd.synthetic import synthetic_mixture
    import datetime
    dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=180))      # a failed rank must not park the others for half an hour
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    x = synthetic_mixture(11, numSamples=n)
    local = HipTimeShard(x, rank, world, dictionarySize=K)
    seg, start = separate_time_sharded(local, iters)
    np.savez(os.path.join(out_dir, 'time_rank%d.npz' % rank), seg=seg, start=start, idx=local.tdoa_indexes(), T=local.T_total)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1], *[int(v) for v in sys.argv[2:5]])
