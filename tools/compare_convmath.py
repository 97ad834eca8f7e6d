"""Full-size frame: bf16x3 vs exact-f32 convolution arithmetic -- forward map and parameter gradients
(developer tool; prints max-relative differences)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd')); sys.path.insert(0, REPO)
import modules.config as cfg
from MVXNet import MVXNet
from modules.pipeline import voxelize_batch
import bench
dev = torch.device('cuda')
torch.manual_seed(0)
model = MVXNet().to(dev)
batch = bench.make_batch([0], dev, 20000)
imsize = torch.tensor([370.0, 1224.0], device=dev)
G = (torch.randn((1, 128, 352, 400), generator=torch.Generator().manual_seed(77)) * 1e-3).to(dev)
res = {}
for mode in ('f32', 'bf16x3'):
    cfg.config['convmath'] = mode
    model.zero_grad()
    frames, _ = voxelize_batch(batch)
    vox, idx = frames[0]
    mid = model.middle(vox.clone(), batch.fpn_levels[0], idx, [None], imsize)
    mid.backward(G)
    res[mode] = (mid.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
def rms(a, b): return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
print('BEV map: max-rel %.2e  rms-rel %.2e' % (rel(res['bf16x3'][0], res['f32'][0]), rms(res['bf16x3'][0], res['f32'][0])))
for k in res['f32'][1]:
    a, b = res['bf16x3'][1][k], res['f32'][1][k]
    print('%-36s max-rel %.2e  rms-rel %.2e' % (k, rel(a, b), rms(a, b)))
