"""Debug: the CML stack with convbackground on / off (the body of
tests/test_voxelnet_gpu.py::test_full_size_background_rewrite_equals_dense_cml): where do the voxel-row gradients differ?"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
math = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
import modules.config as cfg  # noqa: E402
from modules.layers import Blocks  # noqa: E402
from modules.voxelnet import VoxelNet  # noqa: E402
from modules.voxelnet.VoxelNet import BEVFunction  # noqa: E402

DEV = 'cuda'
from modules import Extension as X  # noqa: E402
if os.environ.get('SPLIT16'):
    X.check(X.lib.mvx_tuning_set(1, int(os.environ['SPLIT16'])), 'tune')
gen = torch.Generator().manual_seed(21)
D, H, W = cfg.voxelshape[2], cfg.voxelshape[0], cfg.voxelshape[1]
V = 3000
ix = torch.randint(0, H, (V,), generator=gen)
iy = torch.randint(0, W, (V,), generator=gen)
iz = torch.randint(0, D, (V,), generator=gen)
ix[:2000] = (ix[:2000] % 60) + 100
iy[:2000] = (iy[:2000] % 80) + 40
special = [(0, 0, 0), (0, W - 1, D - 1), (H - 1, 0, 0), (H - 1, W - 1, D - 1), (0, 200, 3), (H - 1, 17, 9), (123, 0, 5), (77, W - 1, 0)]
for k, (a, b, c) in enumerate(special):
    ix[2000 + k], iy[2000 + k], iz[2000 + k] = a, b, c
key = (iz * H + ix) * W + iy
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
cfg.config['convmath'] = math
res = {}
for mode in (True, False):
    cfg.config['convbackground'] = mode
    net.zero_grad()
    feat = feat0.clone().requires_grad_(True)
    x = net.cml.conv1.forward_voxels(feat, idx, (D, H, W))
    x = net.cml.conv3(net.cml.conv2(x))
    mid = BEVFunction.apply(x)
    (mid * G).sum().backward()
    torch.cuda.synchronize()
    res[mode] = (mid.detach().clone(), feat.grad.clone(), {k: p.grad.clone() for k, p in net.cml.named_parameters() if p.grad is not None})
d = (res[True][1] - res[False][1]).abs().max(1).values / res[False][1].abs().max()
print(math, 'mid', float((res[True][0] - res[False][0]).abs().max() / res[False][0].abs().max()), 'dfeat max', float(d.max()))
top = torch.argsort(d, descending=True)[:12].cpu()
for v in top:
    print('voxel %4d (ix %3d iy %3d iz %d)  rel diff %.2e' % (int(v), int(ix[v]), int(iy[v]), int(iz[v]), float(d[v])))
print('voxels above 1e-4:', int((d > 1e-4).sum()), 'of', V)
for k in res[False][2]:
    a, b = res[True][2][k], res[False][2][k]
    print(k, float((a - b).abs().max() / b.abs().max()))
