"""Worker of tests/test_parallel_gpu.py: ONE rank on the `nccl` backend (= RCCL on ROCm) all-reduces the real flat
gradient bucket of the whole model (7,378,480 fp32 = 29.5 MB) on cuda:0 -- the only way a one-GPU box can
put this code base's collective through RCCL (two ranks on one device are refused by RCCL; the 2-rank rehearsals use gloo)."""
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
sys.argv = sys.argv[:1]
from modules import parallel  # noqa: E402
from MVXNet import MVXNet  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
dist.init_process_group(backend='nccl', rank=0, world_size=1)
assert dist.get_backend() == 'nccl'
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
torch.manual_seed(0)
model = MVXNet().to(dev)
params = [p for p in model.parameters() if p.requires_grad]
# the layout of bench.py / train_like.py: the first fusion layer's weight gradient (the last kernel of a step) is exchanged in a
# second, small call after the large early part, which goes out on the communication stream
late = model.head.fusion.fcn1.fc.weight
bucket = parallel.GradBucket(params, late=[late])
bucket.timing = True
assert bucket.flat.numel() == 7378480 and bucket.flat.is_cuda and bucket.n_early == 7378480 - 768 * 768
g = torch.Generator(device='cpu').manual_seed(1)
pattern = torch.randn(bucket.flat.numel(), generator=g).to(dev)
bucket.flat.copy_(pattern)
bucket.all_reduce_mean(4)                           # SUM over the (one) rank through RCCL, then / frames
torch.cuda.synchronize()
assert torch.equal(bucket.flat, pattern * 0.25)
kinds = {k for k, _, _ in bucket.times}
assert kinds == {'early', 'late'}, kinds            # both collectives went through RCCL
bucket.flat.copy_(pattern)
bucket.all_reduce_mean(frames_local=4)              # the frame count travels in the bucket's count slot (no host read)
torch.cuda.synchronize()
assert torch.equal(bucket.flat, pattern / 4.0) and float(bucket._count[0]) == 4.0
assert parallel.global_count(3, dev) == 3           # the frame-count exchange of train_like.py on the same backend
t0 = time.perf_counter()
for _ in range(10):
    bucket.all_reduce_mean(1)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
parallel.assert_replicas_in_sync(params)
dist.barrier()
dist.destroy_process_group()
print('RCCL_OK world 1, bucket %d floats in two calls (%s ms), all_reduce_mean %.3f ms per call' % (bucket.flat.numel(), {k: round(v, 3) for k, v in bucket.collective_ms().items()}, ms))
