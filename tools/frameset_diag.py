"""Where do a two-frame frame set and the same frames run one at a time differ?  (developer tool, full size)"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
for p in (os.path.join(REPO, 'mvxnet-makise_amd'), os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p)
import mvx_oracle as O
from test_fullsize_gpu import _make_batch
import modules.pipeline as pl
from modules import frames as fr, parallel, _hip
from MVXNet import MVXNet

dev = torch.device('cuda')
torch.manual_seed(0)
model = MVXNet().to(dev)
hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
bucket = parallel.GradBucket([p for _, p in hot])
G = (torch.randn((1, 128, 352, 400), generator=torch.Generator().manual_seed(77)) * 1e-3).to(dev)
imsize = [370.0, 1224.0]


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def run(batch):
    cap = {}
    orig = fr.bn_relu_backward
    calls = []

    def spy(*a, **k):
        out = orig(*a, **k)
        calls.append(out.clone())
        return out
    fr.bn_relu_backward = spy
    bucket.zero()
    fs, live, counts, status = pl.prepare_frame_set(batch)
    _hip.GRAD_SINK = True
    _hip.arena_begin(dev, doubles=1 << 21)
    with torch.no_grad():
        mid, S = fr.middle_forward(model, fs, [batch.fpn_levels[f] for f in live], imsize, [])
        fr.middle_backward(model, S, G)
    _hip.arena_end(); _hip.join_side_stream(); _hip.GRAD_SINK = False
    torch.cuda.synchronize()
    fr.bn_relu_backward = orig
    cap['mid'] = mid.clone()
    cap['y'] = [r['y'].clone() for r in S.convs] + [S.conv1['y'].clone()]
    cap['mi'] = [r['mi'].clone() for r in S.convs] + [S.conv1['mi'].clone()]
    cap['dz3'] = calls[0]
    cap['grads'] = {k: p.grad.clone() for k, p in hot}
    return cap


both = run(_make_batch((0, 1), dev)[0])
again = run(_make_batch((0, 1), dev)[0])
s0 = run(_make_batch((0,), dev)[0])
s1 = run(_make_batch((1,), dev)[0])
print('determinism (same set twice): mid %.2e  worst grad %.2e' % (rel(again['mid'], both['mid']), max(rel(again['grads'][k], both['grads'][k]) for k, _ in hot)))
for f, s in enumerate((s0, s1)):
    print('frame %d: mid %.2e' % (f, rel(both['mid'][f:f + 1], s['mid'])))
    for li, name in enumerate(('conv2', 'conv3', 'conv1')):
        D = s['y'][li].shape[0]
        print('   %s y %.2e  mean %.2e  inv %.2e' % (name, rel(both['y'][li][f * D:(f + 1) * D], s['y'][li]),
              rel(both['mi'][li][f, 0], s['mi'][li][0, 0]), rel(both['mi'][li][f, 1], s['mi'][li][0, 1])))
    D = s['dz3'].shape[0]
    print('   dz3 %.2e   (max |dz3| %.3e)' % (rel(both['dz3'][f * D:(f + 1) * D], s['dz3']), float(s['dz3'].abs().max())))
for k, _ in hot:
    print('%-40s %.2e' % (k, rel(both['grads'][k], s0['grads'][k] + s1['grads'][k])))
