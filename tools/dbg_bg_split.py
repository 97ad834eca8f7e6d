"""Debug: the CML stack with convbackground on / off in every arithmetic (the body of
tests/test_voxelnet_gpu.py::test_full_size_background_rewrite_equals_dense_cml), all pairwise distances."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
import modules.config as cfg  # noqa: E402
from modules.layers import Blocks  # noqa: E402
from modules.voxelnet import VoxelNet  # noqa: E402
from modules.voxelnet.VoxelNet import BEVFunction  # noqa: E402

DEV = 'cuda'


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


gen = torch.Generator().manual_seed(21)
D, H, W = cfg.voxelshape[2], cfg.voxelshape[0], cfg.voxelshape[1]
V = 3000
ix = torch.randint(0, H, (V,), generator=gen)
iy = torch.randint(0, W, (V,), generator=gen)
iz = torch.randint(0, D, (V,), generator=gen)
ix[:2000] = (ix[:2000] % 60) + 100
iy[:2000] = (iy[:2000] % 80) + 40
key = (iz * H + ix) * W + iy
_, first = torch.unique(key, return_inverse=False, return_counts=False, sorted=True), None
seen, keep = set(), torch.zeros(V, dtype=torch.bool)
for v in range(V):
    k = int(key[v])
    if k not in seen:
        seen.add(k)
        keep[v] = True
ix, iy, iz = ix[keep], iy[keep], iz[keep]
V = int(keep.sum())
idx = torch.stack([torch.zeros(V, dtype=torch.long), ix, iy, iz], 1).to(DEV)
feat0 = torch.randn(V, 128, generator=gen).to(DEV)
G = (torch.randn(1, 128, H, W, generator=gen) * 1e-2).to(DEV)
torch.manual_seed(5)
net = VoxelNet().to(DEV)
Blocks.RESTRICTED_BACKWARD = True
res = {}
for math in ('f32', 'bf16x3', 'bf16x6'):
    cfg.config['convmath'] = math
    for mode in (True, False):
        cfg.config['convbackground'] = mode
        net.zero_grad()
        feat = feat0.clone().requires_grad_(True)
        x = net.cml.conv1.forward_voxels(feat, idx, (D, H, W))
        x = net.cml.conv3(net.cml.conv2(x))
        mid = BEVFunction.apply(x)
        (mid * G).sum().backward()
        torch.cuda.synchronize()
        res[(math, mode)] = (mid.detach().clone(), feat.grad.clone(), {k: p.grad.clone() for k, p in net.cml.named_parameters() if p.grad is not None})
ref = res[('f32', False)]
for key_, r in res.items():
    print(key_, 'mid %.2e dfeat %.2e' % (rel(r[0], ref[0]), rel(r[1], ref[1])), ' '.join('%s %.1e' % (k.split('.')[0] + k[-4:], rel(r[2][k], ref[2][k])) for k in ref[2]))
