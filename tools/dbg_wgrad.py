import os, sys
import numpy as np, torch
sys.path.insert(0, 'mvxnet-makise_amd'); sys.path.insert(0, 'oracle'); sys.path.insert(0,'tests')
import mvx_oracle as O
from modules import _hip
import modules.config as cfg
from modules.voxelnet import VoxelNet
g = np.load('tests/golden/voxelnet_small.npz')
cfg.config['voxelshape'] = [int(v) for v in g['voxelshape']]
DEV='cuda'
net = VoxelNet()
P = O.strip_prefix(O.make_params(7), 'backbone.')
sd = net.state_dict()
for k in sd:
    if k in P: sd[k] = P[k]
net.load_state_dict(sd); net = net.to(DEV)
caps = {}
orig = _hip.conv3d_wgrad
def spy(x, dz, sdd, pd, split=False):
    caps[(x.shape, dz.shape)] = (x.clone(), dz.clone(), sdd, pd)
    return orig(x, dz, sdd, pd, split=split)
_hip.conv3d_wgrad = spy
cfg.config['convmath'] = 'bf16x3'
x = torch.from_numpy(g['x'])[None].to(DEV); idx = torch.from_numpy(g['idx']).to(DEV)
mid = net.middle(x, idx); (mid[0] * torch.from_numpy(g['G']).to(DEV)).sum().backward()
for key, (xx, dz, sdd, pd) in caps.items():
    a = orig(xx, dz, sdd, pd, split=False); b = orig(xx, dz, sdd, pd, split=True)
    print(key, sdd, pd, 'split vs f32 rel', float((a-b).abs().max()/a.abs().max()), 'absmax x', float(xx.abs().max()), 'dz', float(dz.abs().max()), 'dz nonfinite', int((~torch.isfinite(dz)).sum()))
    # f64 reference on CPU
    xc = xx.cpu().double().permute(3,0,1,2)[None]; dzc = dz.cpu().double().permute(3,0,1,2)[None]
    w = torch.zeros(64, xx.shape[3], 3,3,3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(xc, w, None, (sdd,1,1), (pd,1,1)).backward(dzc)
    print('   f32 vs f64', float((a.cpu()-w.grad).abs().max()/w.grad.abs().max()), ' split vs f64', float((b.cpu()-w.grad).abs().max()/w.grad.abs().max()))
