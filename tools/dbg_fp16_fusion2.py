"""Developer tool: fusion-MLP gradients of a small frame set: fp16x3 (side-stream weight gradients / inline) vs bf16x6."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
import numpy as np
import torch
import modules.config as cfg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg.config['voxelshape'] = [16, 24, 10]
cfg.config['velorange'] = [0.0, -2.4, -3.0, 3.2, 2.4, 1.0]
cfg.config['voxelsize'] = [0.2, 0.2, 0.4]
from modules import _hip, parallel
import test_frames_gpu as T

def golden(name):
    return dict(np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz')))
from MVXNet import MVXNet
from modules.pipeline import train_step_frame_set, train_step_frames
torch.manual_seed(3)
model = MVXNet().to('cuda')
batch, G = T._small_batch(golden, B, False)
for f in range(B):
    nlive = int(batch.n_points[f])
    batch.perms[f, :nlive] = torch.randperm(nlive, generator=torch.Generator().manual_seed(f)).to('cuda')
hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
bucket = parallel.GradBucket([p for _, p in hot])
res = {}
scope = _hip._wgrad_scope
from modules import pipeline
for name, math, inline, fn, lanes in (('x6 set', 'bf16x6', False, train_step_frame_set, 2), ('fp16 set', 'fp16x3', False, train_step_frame_set, 2),
                                      ('x6 frames', 'bf16x6', False, train_step_frames, 2),
                                      ('fp16 frames', 'fp16x3', False, train_step_frames, 2),
                                      ('fp16 frames 1 lane', 'fp16x3', False, train_step_frames, 1),
                                      ('fp16 frames 1 lane inline', 'fp16x3', True, train_step_frames, 1)):
    cfg.config['convmath'] = math
    pipeline.LANES = lanes
    _hip._wgrad_scope = (lambda acc, *t: _hip._Inline()) if inline else scope
    bucket.zero()
    fn(model, batch, G, [370.0, 1224.0])
    _hip.join_side_stream()
    torch.cuda.synchronize()
    res[name] = {k: p.grad.clone() for k, p in hot}
ref = res['x6 set']
for name in res:
    if name == 'x6 set':
        continue
    bad = [(k, float((res[name][k] - ref[k]).abs().max() / ref[k].abs().max())) for k in ref]
    print(name, ' '.join('%s=%.1e' % (k.replace('head.fusion.', '').replace('.weight', ''), e) for k, e in bad if e > 5e-5), flush=True)
