"""Worker of the world_size-2 gloo test: each rank accumulates the gradients of its own frames into
the flat bucket, then one all-reduce must produce the mean over ALL frames on every rank."""
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import parallel  # noqa: E402

rank, world, _ = parallel.init_from_env('gloo')
torch.manual_seed(0)
params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
bucket = parallel.GradBucket(params)
frames_total = 6
mine = parallel.shard_frames(frames_total, rank, world)
bucket.zero()
for f in mine:                                   # "backward" of frame f: gradient = (f+1) * ones
    loss = sum(((f + 1.0) * p).sum() for p in params)
    loss.backward()
assert params[0].grad.data_ptr() == bucket.flat.data_ptr(), 'gradients must stay views of the flat bucket'
bucket.all_reduce_mean(frames_total)
expect = sum(f + 1.0 for f in range(frames_total)) / frames_total
assert torch.allclose(bucket.flat, torch.full_like(bucket.flat, expect)), (bucket.flat, expect)
dist.barrier()
dist.destroy_process_group()
print('DP_OK rank', rank)
