"""Per-parameter distance of the HIP RPN backward from a float64 oracle run (developer tool)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
for p in (os.path.join(REPO, 'mvxnet-makise_amd'), os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import mvx_oracle as O
from test_rpn_gpu import _to_planes, _from_planes, _load_rpn, rel
from modules import _hip, parallel
from modules import rpn_frames as rf
g = np.load(os.path.join(REPO, 'tests/golden/rpn_shapes.npz'))
P = O.rpn_params(g)
rpn = _load_rpn(P)
bucket = parallel.GradBucket(list(rpn.parameters()))
F, H, W = 1, 64, 96
gen = torch.Generator().manual_seed(3)
mids = torch.randn((F, 128, H, W), generator=gen)
d_heads = torch.randn((F, H // 2, W // 2, 16), generator=gen) * 0.1
bucket.zero()
heads, S = rf.rpn_forward(rpn, _to_planes(mids.cuda()), F, 2, H, W, 64)
g_in = rf.rpn_backward(rpn, S, d_heads.reshape(-1, 16).cuda())
_hip.join_side_stream(); torch.cuda.synchronize()
Pd = {k: v.double().requires_grad_(True) for k, v in P.items()}
x = mids.double().requires_grad_(True)
# oracle with intermediate capture
import torch.nn.functional as Fn
score, reg = O.rpn(x, Pd)
logits = torch.log(score / (1 - score))
out = torch.cat([logits, reg], dim=1)[0].permute(1, 2, 0)
(out * d_heads[0].double()).sum().backward()
print('heads fwd rel', rel(heads.view(H // 2, W // 2, 16).cpu().double(), out.detach()))
print('input grad', rel(_from_planes(g_in, F).cpu().double(), x.grad))
for k, p in rpn.named_parameters():
    print('%-28s %.2e' % (k, rel(p.grad.cpu().double(), Pd['rpn.' + k].grad)))

# ---- deconv3 in isolation: forward pieces and backward pieces against float64 torch
print('--- deconv3 pieces')
def blk(x, name, n):
    for i in range(n):
        x = O.crb2d(x, Pd['rpn.%s.%d.conv.weight' % (name, i)], Pd['rpn.%s.%d.conv.bias' % (name, i)], 2 if i == 0 else 1, 1)
    return x
with torch.no_grad():
    x1 = blk(x, 'blk1', 4); x2 = blk(x1, 'blk2', 6); x3 = blk(x2, 'blk3', 6)
ours_x3 = S['blocks'][2]['out'].cpu().double().permute(0, 3, 1, 2)
print('x3 rel', rel(ours_x3, x3))
rec = S['dk'][1]
w, b = Pd['rpn.deconv3.deconv.weight'].detach(), Pd['rpn.deconv3.deconv.bias'].detach()
x3l = x3.detach().requires_grad_(True)
wl = w.clone().requires_grad_(True); bl = b.clone().requires_grad_(True)
y = Fn.relu(Fn.conv_transpose2d(x3l, wl, bl, 4, 0))
t_ours = rec['t'].cpu().double().view(1, 8, 12, 4, 4, 256).permute(0, 5, 1, 3, 2, 4).reshape(1, 256, 32, 48)
print('t (relu output) rel', rel(t_ours, y.detach()))
mean = y.mean(dim=(0, 2, 3)); var = y.var(dim=(0, 2, 3), unbiased=False)
print('mean rel', rel(rec['mi'][0, 0].cpu().double(), mean.detach()), 'inv rel', rel(rec['mi'][0, 1].cpu().double(), (1 / torch.sqrt(var + 1e-6)).detach()))
print('min var', float(var.min()), 'max inv', float((1 / torch.sqrt(var + 1e-6)).max()))

# ---- deconv3 backward in isolation, fed with OUR x3 and the exact upstream gradient
W_heads = torch.cat([Pd['rpn.cls.weight'].detach().view(2, 768), Pd['rpn.reg.weight'].detach().view(14, 768)])
g_up = (d_heads.reshape(-1, 16).double() @ W_heads).view(1, 32, 48, 768).permute(0, 3, 1, 2)     # (1,768,32,48)
for name, sl, s_, xin in (('deconv2', slice(256, 512), 2, S['blocks'][1]['out']), ('deconv3', slice(512, 768), 4, S['blocks'][2]['out'])):
    xo = xin.cpu().double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    wl = Pd['rpn.%s.deconv.weight' % name].detach().clone().requires_grad_(True)
    bl = Pd['rpn.%s.deconv.bias' % name].detach().clone().requires_grad_(True)
    yh = O.decrb2d(xo, wl, bl, s_, 0)
    (yh * g_up[:, sl]).sum().backward()
    m = getattr(rpn, name)
    print(name, 'isolated: dW %.2e  db %.2e' % (rel(m.deconv.weight.grad.cpu().double(), wl.grad), rel(m.deconv.bias.grad.cpu().double(), bl.grad)))

# ---- per-layer forward error: each HIP layer output against the float64 layer applied to OUR input of that layer
print('--- per-layer forward error (HIP layer vs float64 layer on the same input), and accumulated error vs the float64 chain')
xo = x.detach()
for bi, name in enumerate(('blk1', 'blk2', 'blk3')):
    for li, recl in enumerate(S['blocks'][bi]['layers']):
        w_, b_ = Pd['rpn.%s.%d.conv.weight' % (name, li)].detach(), Pd['rpn.%s.%d.conv.bias' % (name, li)].detach()
        nxt = S['blocks'][bi]['layers'][li + 1]['x'] if li + 1 < len(S['blocks'][bi]['layers']) else S['blocks'][bi]['out']
        ours_out = nxt.cpu().double()
        if li == 0:
            # our input of a stride-2 layer is the space-to-depth image: rebuild the full-resolution NCHW input from it
            xs = recl['x'].cpu().double()
            Fh, hh, ww, cc = xs.shape
            pl = recl['planes']; Cf = recl['cfull']
            full = xs.view(Fh, hh, ww, 2, 2, pl, Cf).permute(0, 5, 6, 1, 3, 2, 4).reshape(Fh, pl, Cf, hh * 2, ww * 2)
            ours_in = full.permute(0, 2, 1, 3, 4).reshape(Fh, Cf * pl, hh * 2, ww * 2)      # channel = c*planes + d
        else:
            ours_in = recl['x'].cpu().double().permute(0, 3, 1, 2)
        ref_same = O.crb2d(ours_in, w_, b_, 2 if li == 0 else 1, 1)
        xo = O.crb2d(xo, w_, b_, 2 if li == 0 else 1, 1)
        o = ours_out.permute(0, 3, 1, 2)
        print('%s.%d  layer %.2e   chain %.2e' % (name, li, rel(o, ref_same), rel(o, xo)))
