"""Worker of tests/test_gpu_dist.py: one rank under torch.distributed.run on the 1-GPU box, backend nccl (= RCCL).  Runs bench.py's
timing protocol -- barrier, timed steps, barrier, all-reduce(MAX) of the elapsed time on a device tensor -- with the collectives
forced on (world passed as 2), around real launches of the forward.  What it cannot show on one GPU is a second rank; the
partition arithmetic and the max-over-ranks are covered by the world-size-2 gloo tests on CPU."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import flashattention_c_amd as fa  # noqa: E402

rank, world, local = bench.dist_env()
assert (rank, world, local) == (0, 1, 0), (rank, world, local)
torch.cuda.set_device(local)
device = torch.device("cuda", local)
dist.init_process_group(backend="nccl")
q, k, v = bench.make_inputs(4, 1024, 64, "bf16", device, seed=rank)
out = torch.empty_like(q)
calls = []


def step():
    calls.append(1)
    fa.forward(q, k, v, False, out=out)


dt = bench.timed_region(step, steps=5, warmup=2, sync_fn=torch.cuda.synchronize, world=2, dist=dist, device=device)
assert len(calls) == 7 and 0.0 < dt < 5.0, (len(calls), dt)
b0, b1 = fa.shard_range(1024, 8, 3)
assert (b0, b1) == (384, 512)
dist.barrier()
dist.destroy_process_group()
print(f"NCCL_ONE_RANK_OK dt={dt:.6f}", flush=True)
