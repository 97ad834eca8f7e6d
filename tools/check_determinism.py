"""Repeats one training step and compares gradients across repeats, background on/off, tape/autograd -- developer tool."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
import bench
import modules.config as cfg
from modules import parallel, _hip
import modules.pipeline as pl
from modules.layers import Blocks
from MVXNet import MVXNet
dev = torch.device('cuda'); torch.manual_seed(4)
model = MVXNet().to(dev)
hot = [p for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
names = [k for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
bucket = parallel.GradBucket(hot)
grad_mid = torch.randn((1, 128, cfg.voxelshape[0], cfg.voxelshape[1]), device=dev) * 1e-3
imsize = [float(v) for v in cfg.imsize]
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
def run(batch, **kw):
    old = {}
    for k, v in kw.items():
        if k == 'bg': old[k] = cfg.config.get('convbackground', True); cfg.config['convbackground'] = v
        if k == 'tape': old[k] = pl.TAPE; pl.TAPE = v
        if k == 'lanes': old[k] = pl.LANES; pl.LANES = v
    bucket.zero()
    pl.train_step_frames(model, batch, grad_mid, imsize)
    torch.cuda.synchronize()
    for k, v in old.items():
        if k == 'bg': cfg.config['convbackground'] = v
        if k == 'tape': pl.TAPE = v
        if k == 'lanes': pl.LANES = v
    return bucket.flat.clone()
pts = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
one = bench.make_batch([0], dev, pts)
a = run(one); b = run(one)
print('same call twice          ', rel(a, b))
c = run(one, bg=False)
print('background off           ', rel(a, c))
d = run(one, tape=False)
print('autograd path (dense bwd)', rel(d, c), ' vs tape:', rel(a, d))
two = bench.make_batch([0, 1], dev, pts); two.n_points[1] = 0
e = run(two)
print('two-frame batch, 2nd empty', rel(e, a))
f = run(two, bg=False)
print('  same with background off', rel(f, c))
# per-parameter breakdown of the worst
off = 0
worst = []
for n, p in zip(names, hot):
    k = p.numel(); worst.append((rel(e[off:off+k], a[off:off+k]) if a[off:off+k].abs().max() > 0 else 0, n)); off += k
print(sorted(worst, reverse=True)[:6])
four = bench.make_batch([0, 1, 2, 3], dev, pts)
g0 = run(four)
worst = 0.0
for _ in range(5):
    worst = max(worst, rel(run(four), g0))
print('4 frames / 2 lanes, 5 repeats: worst deviation', worst)
print('   vs background off', rel(g0, run(four, bg=False)), ' vs one lane', rel(g0, run(four, lanes=1)))
cfg.config['convmath'] = 'bf16x3'
h0 = run(four)
worst = 0.0
for _ in range(3):
    worst = max(worst, rel(run(four), h0))
print('bf16x3: repeats', worst, ' vs f32', rel(h0, g0), ' vs bf16x3 background off', rel(h0, run(four, bg=False)))
