#!/usr/bin/env python
"""Can TWO ranks of the library's own RCCL communicator (csrc/collective.hip) live on ONE device?  (VERDICT r3 #4a: record the outcome
either way.)  Two processes rendezvous over gloo, rank 0 broadcasts the unique id, both call gccnmf_rccl_comm_init on cuda:0 behind
the time limit of distributed._call_with_timeout, then -- if that worked -- one ncclAllReduce.  Prints one JSON line per rank.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 scripts/rccl_one_device.py"""
import ctypes
import datetime
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                            # noqa: E402
import torch.distributed as dist                        # noqa: E402
from gcc_nmf_amd import _hip                            # noqa: E402
from gcc_nmf_amd.distributed import _call_with_timeout  # noqa: E402

dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=120))
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
lib = _hip.lib()
ident = torch.zeros(_hip.RCCL_UNIQUE_ID_BYTES, dtype=torch.uint8)
if rank == 0:
    buf = ctypes.create_string_buffer(_hip.RCCL_UNIQUE_ID_BYTES)
    assert lib.gccnmf_rccl_unique_id(buf) == 0
    ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
dist.broadcast(ident, src=0)
out = {'rank': rank, 'world': world, 'device': 'cuda:0 for every rank', 'rccl_available': int(lib.gccnmf_rccl_available())}
handle = ctypes.c_void_p()
try:
    rc = _call_with_timeout(lambda: lib.gccnmf_rccl_comm_init(ident.numpy().tobytes(), world, rank, ctypes.byref(handle)), 60, 'gccnmf_rccl_comm_init')
    out['comm_init_status'] = int(rc)
    if rc == 0:
        t = torch.full((1024,), float(rank + 1), device='cuda')
        rc2 = _call_with_timeout(lambda: (lib.gccnmf_rccl_allreduce(handle, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream), torch.cuda.synchronize())[0],
                                 60, 'gccnmf_rccl_allreduce')
        out['allreduce_status'], out['allreduce_value'] = int(rc2), float(t[0].item())
except Exception as e:                                  # noqa: BLE001 -- the outcome is the point
    out['error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
print(json.dumps(out), flush=True)
os._exit(0)                                             # a communicator that half exists may not tear down cleanly
