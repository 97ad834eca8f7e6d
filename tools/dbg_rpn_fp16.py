"""Developer tool: where does the input gradient of RPN block 1 / layer 1 differ from the float64 layer oracle (fp16x3)?"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, REPO)
import numpy as np
import torch
import modules.config as cfg
cfg.config['convmath'] = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
from modules import _hip, parallel
from modules import rpn_frames as rf
sys.path.insert(0, os.path.join(REPO, "oracle"))
import mvx_oracle as O
import test_rpn_gpu as T

def golden(name):
    return dict(np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz')))
F = 3
P = O.rpn_params(golden('rpn_shapes'))
rpn = T._load_rpn(P)
bucket = parallel.GradBucket(list(rpn.parameters()))
H, W = 64, 96
gen = torch.Generator().manual_seed(3)
mids = torch.randn((F, 128, H, W), generator=gen)
d_heads = torch.randn((F, H // 2, W // 2, 16), generator=gen) * 0.1
bucket.zero()
heads, S = rf.rpn_forward(rpn, T._to_planes(mids.to('cuda')), F, 2, H, W, 64)
S['capture'] = []
g_in = rf.rpn_backward(rpn, S, d_heads.reshape(-1, 16).to('cuda'))
_hip.join_side_stream(); torch.cuda.synchronize()
P64 = {k: v.double() for k, v in P.items()}
cap = {k: (g, dx) for k, g, dx in S['capture']}
for bi, name in enumerate(('blk1', 'blk2', 'blk3')):
    layers = S['blocks'][bi]['layers']
    for li, rec in enumerate(layers):
        w_ = P64['rpn.%s.%d.conv.weight' % (name, li)].clone().requires_grad_(True)
        b_ = P64['rpn.%s.%d.conv.bias' % (name, li)].clone().requires_grad_(True)
        xin = T._nchw_input_of(rec)
        g_up, g_dx = cap[(bi, li)]
        g_up = g_up.cpu().double().permute(0, 3, 1, 2)
        for f in range(F):
            xf = xin[f:f + 1].clone().requires_grad_(True)
            yh = O.crb2d(xf, w_, b_, 2 if li == 0 else 1, 1)
            gw, gb, gxf = torch.autograd.grad((yh * g_up[f:f + 1]).sum(), (w_, b_, xf))
            ours = T._nchw_grad_of(g_dx, rec, F)[f:f + 1]
            d = (ours - gxf).abs()
            mx = float(gxf.abs().max())
            n_big = int((d > 1e-4 * mx).sum())
            # pre-activation values closest to the kink in the oracle's conv output
            y64 = torch.nn.functional.conv2d(xf.detach(), w_.detach(), b_.detach(), stride=2 if li == 0 else 1, padding=1)
            print(name, li, 'frame', f, 'rel %.2e' % (float(d.max()) / mx), 'elements > 1e-4: %d of %d' % (n_big, d.numel()),
                  'min |y| %.2e' % float(y64.abs().min()), 'amax g_up %.2e' % float(g_up[f].abs().max()), flush=True)
