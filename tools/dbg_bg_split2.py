"""Debug: which kernel output differs between 16x16 and 8x16 split units in the CML chain (bg on / off)?"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
math = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
import modules.config as cfg  # noqa: E402
from modules import _hip  # noqa: E402
from modules import Extension as X  # noqa: E402
from modules.layers import Blocks  # noqa: E402
from modules.voxelnet import VoxelNet  # noqa: E402
from modules.voxelnet.VoxelNet import BEVFunction  # noqa: E402

DEV = 'cuda'
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
LOG = []


def wrap(name):
    fn = getattr(_hip, name)

    def f(*a, **k):
        r = fn(*a, **k)
        t = r[0] if isinstance(r, tuple) else r
        torch.cuda.synchronize()
        LOG.append((name, t.clone()))
        return r
    setattr(_hip, name, f)


for n in ('conv3d_forward_bg', 'conv3d_forward', 'conv3d_dgrad_tiles', 'conv3d_dgrad', 'bn_relu_backward_tiles', 'bn_relu_backward'):
    if hasattr(_hip, n):
        wrap(n)
res = {}
for units in (0, 1 << 60):
    X.check(X.lib.mvx_tuning_set(1, units), 'tune')
    for mode in (True, False):
        cfg.config['convbackground'] = mode
        del LOG[:]
        net.zero_grad()
        feat = feat0.clone().requires_grad_(True)
        x = net.cml.conv1.forward_voxels(feat, idx, (D, H, W))
        x = net.cml.conv3(net.cml.conv2(x))
        mid = BEVFunction.apply(x)
        (mid * G).sum().backward()
        torch.cuda.synchronize()
        res[(units, mode)] = (list(LOG), feat.grad.clone())
for mode in (True, False):
    a, b = res[(0, mode)], res[(1 << 60, mode)]
    print('bg' if mode else 'dense', 'dfeat 16x16 vs 8x16: %.2e' % float((a[1] - b[1]).abs().max() / b[1].abs().max()))
    for (n1, t1), (n2, t2) in zip(a[0], b[0]):
        d = (t1 - t2).abs()
        fin = torch.isfinite(d)
        # restricted tensors hold garbage off their tiles: compare where both are finite and report the largest finite difference
        dm = torch.where(fin, d, torch.zeros_like(d))
        i = int(dm.argmax())
        loc = []
        for s_ in reversed(t1.shape):
            loc.append(i % s_)
            i //= s_
        print('   %-24s shape %s  max diff %.3e (scale %.3e) at %s' % (n1, tuple(t1.shape), float(dm.max()), float(t2[torch.isfinite(t2)].abs().max()), loc[::-1]))
