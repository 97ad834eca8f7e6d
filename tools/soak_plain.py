"""The hot step WITHOUT bench.py's per-kernel timers, many times: peak allocated / reserved device memory and step time over the
run (developer tool).  bench.py's timed region keeps, for every timed step, the small closures that count executed FLOPs after the
region -- they hold that step's tile-flag tensors (~0.35 MB per step) until they are evaluated, which is the "creep" tools/soak.py
sees; this loop is what a training run does.    usage: python tools/soak_plain.py [steps]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sys.argv = sys.argv[:1]
import torch  # noqa: E402
import bench  # noqa: E402
import modules.config as cfg  # noqa: E402
from modules import _hip, parallel  # noqa: E402
from modules import pipeline as pl  # noqa: E402
from MVXNet import MVXNet  # noqa: E402

dev = torch.device('cuda')
torch.manual_seed(0)
model = MVXNet().to(dev)
hot = [p for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
bucket = parallel.GradBucket(hot, late=[model.head.fusion.fcn1.fc.weight])
opt = torch.optim.AdamW(hot, lr=1e-3, eps=cfg.eps, fused=True)
batches = [bench.make_batch([4 * j + k for k in range(4)], dev, 20000, 'S2') for j in range(3)]      # three different frame sets
g = torch.Generator(device='cpu').manual_seed(77)
grad_mid = (torch.randn((1, 128, cfg.voxelshape[0], cfg.voxelshape[1]), generator=g) * 1e-3).to(dev)
imsize = [float(v) for v in cfg.imsize]
ready, statuses_all = None, []
marks = {}
t0 = None
for it in range(steps + 20):
    if it == 20:
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
    b, nb = batches[it % 3], batches[(it + 1) % 3]
    bucket.zero()
    nv, statuses, ready = pl.train_step_frame_set(model, b, grad_mid, imsize, ready=ready, prepare_next=nb)
    bucket.all_reduce_mean(4)
    opt.step()
    statuses_all.extend(statuses)
    if len(statuses_all) > 64:                             # one host read per ~20 steps, like a training loop's logging
        bad = 0
        for v in torch.stack([t.reshape(()) for t in statuses_all]).tolist():
            bad |= int(v)
        _hip.raise_on_status(bad)
        del statuses_all[:]
    if it - 20 + 1 in (50, 200, steps):
        torch.cuda.synchronize()
        marks[it - 20 + 1] = (round(torch.cuda.max_memory_allocated() / 2**20), round(torch.cuda.memory_reserved() / 2**20),
                              round((time.perf_counter() - t0) / (it - 20 + 1) * 1e3, 3))
for k, (a, r, ms) in marks.items():
    print('after %5d steps: peak allocated %d MB, reserved %d MB, %.3f ms per step' % (k, a, r, ms))
