"""Developer tool: the fusion MLP's weight gradients of a small frame set in fp16x3 -- every linear_wgrad call against float64."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
import numpy as np
import torch
import modules.config as cfg
cfg.config['convmath'] = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg.config['voxelshape'] = [16, 24, 10]
cfg.config['velorange'] = [0.0, -2.4, -3.0, 3.2, 2.4, 1.0]
cfg.config['voxelsize'] = [0.2, 0.2, 0.4]
from modules import _hip, parallel
import test_frames_gpu as T

def golden(name):
    return dict(np.load(os.path.join(REPO, 'tests', 'golden', name + '.npz')))

orig = _hip.linear_wgrad
def spy(x, dz, accumulate_into=None, split=None):
    torch.cuda.synchronize()
    _hip.join_side_stream()
    before = accumulate_into.clone() if accumulate_into is not None else None
    r = orig(x, dz, accumulate_into=accumulate_into, split=split)
    _hip.join_side_stream()
    torch.cuda.synchronize()
    got = (accumulate_into - before) if accumulate_into is not None else r
    ref = dz.double().t() @ x.double()
    ax, az = _hip.amax_of(x), _hip.amax_of(dz)
    print('wgrad x%s dz%s  err %.2e  | amax tag x %s (true %.3e)  dz %s (true %.3e) row-amax dz: median %.2e' % (
        tuple(x.shape), tuple(dz.shape), float((got.double() - ref).abs().max() / ref.abs().max()),
        None if ax is None else '%.3e' % float(ax), float(x.abs().max()), None if az is None else '%.3e' % float(az),
        float(dz.abs().max()), float(dz.abs().amax(1).median())), flush=True)
    return r
_hip.linear_wgrad = spy
from MVXNet import MVXNet
from modules.pipeline import train_step_frame_set
torch.manual_seed(3)
model = MVXNet().to('cuda')
batch, G = T._small_batch(golden, B, False)
for f in range(B):
    nlive = int(batch.n_points[f])
    batch.perms[f, :nlive] = torch.randperm(nlive, generator=torch.Generator().manual_seed(f)).to('cuda')
hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
bucket = parallel.GradBucket([p for _, p in hot])
bucket.zero()
train_step_frame_set(model, batch, G, [370.0, 1224.0])
torch.cuda.synchronize()
