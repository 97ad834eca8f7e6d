"""Debug: stage-wise comparison of the RPN backward (modules/rpn_frames.py) with float64 autograd: the upstream gradient of
every block layer (dL/d(layer output)), on a small map."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 96)
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
import mvx_oracle as O  # noqa: E402
from modules import _hip, parallel  # noqa: E402
from modules import rpn_frames as rf  # noqa: E402
from modules.voxelnet.Pipe import RPN  # noqa: E402

DEV = 'cuda'
gen = torch.Generator().manual_seed(31)
rpn = RPN().to(DEV)
P = {}
for k, p in rpn.state_dict().items():
    if k.endswith('weight'):
        fan = p.shape[1] * p.shape[2] * p.shape[3] if 'deconv' not in k else p.shape[0]
        v = torch.randn(p.shape, generator=gen) / np.sqrt(fan)
        if k.startswith(('cls', 'reg')):
            v = v * 0.3
    else:
        v = torch.zeros(p.shape) if k.startswith(('cls', 'reg')) else torch.full(p.shape, 0.5)
    P['rpn.' + k] = v
rpn.load_state_dict({k[4:]: v for k, v in P.items()})
mid = torch.randn((1, 128, H, W), generator=gen)
d_heads = (torch.randn(((H // 2) * (W // 2), 16), generator=gen) * 0.1)
params = list(rpn.named_parameters())
bucket = parallel.GradBucket([p for _, p in params])


def to_planes(m):
    F_, _, h, w = m.shape
    return m.view(F_, 64, 2, h, w).permute(0, 2, 3, 4, 1).reshape(F_ * 2, h, w, 64).contiguous()


with torch.no_grad():
    heads, S = rf.rpn_forward(rpn, to_planes(mid.to(DEV)), 1, 2, H, W, 64)
    S['capture'] = []
    bucket.zero()
    _hip.ASYNC_WGRAD = True
    g = rf.rpn_backward(rpn, S, d_heads.to(DEV))
    _hip.join_side_stream()
    torch.cuda.synchronize()
cap = {k: (a.cpu().double(), b.cpu().double()) for k, a, b in S['capture']}
# float64 with every block activation kept
P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
x = mid.double().requires_grad_(True)
acts = {}
cur = x
for bi, (name, n) in enumerate((('blk1', 4), ('blk2', 6), ('blk3', 6))):
    for i in range(n):
        cur = O.crb2d(cur, P64['rpn.%s.%d.conv.weight' % (name, i)], P64['rpn.%s.%d.conv.bias' % (name, i)], 2 if i == 0 else 1, 1)
        cur.retain_grad()
        acts[(bi, i)] = cur
x1, x2, x3 = acts[(0, 3)], acts[(1, 5)], acts[(2, 5)]
ups = [O.decrb2d(x1, P64['rpn.deconv1.deconv.weight'], P64['rpn.deconv1.deconv.bias'], 1, 1),
       O.decrb2d(x2, P64['rpn.deconv2.deconv.weight'], P64['rpn.deconv2.deconv.bias'], 2, 0),
       O.decrb2d(x3, P64['rpn.deconv3.deconv.weight'], P64['rpn.deconv3.deconv.bias'], 4, 0)]
up = torch.cat(ups, 1)
w_heads = torch.cat([P64['rpn.cls.weight'].view(2, 768), P64['rpn.reg.weight'].view(14, 768)])
b_heads = torch.cat([P64['rpn.cls.bias'], P64['rpn.reg.bias']])
hd = Fn.conv2d(up, w_heads.view(16, 768, 1, 1), b_heads)              # (1,16,h1,w1)
ref_heads = hd[0].permute(1, 2, 0).reshape(-1, 16)
print('heads fwd rel', float((heads.cpu().double() - ref_heads).abs().max() / ref_heads.abs().max()))
(ref_heads * d_heads.double()).sum().backward()
for key in sorted(cap):
    gu, gi = cap[key]
    ref = acts[key].grad[0].permute(1, 2, 0)                          # (h,w,C)
    gu = gu.reshape(ref.shape)
    print('layer %s: upstream gradient rel 2-norm %.3e   (norm hip %.4e ref %.4e ratio %.5f)' %
          (key, float((gu - ref).norm() / ref.norm()), float(gu.norm()), float(ref.norm()), float(gu.norm() / ref.norm())))
off = 0
for k, p in params:
    gr = p.grad.detach().cpu().double()
    r = P64['rpn.' + k].grad
    print('%-26s %.3e' % (k, float((gr - r).norm() / r.norm())))
