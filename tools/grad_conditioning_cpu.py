"""How far is ANY fp32 evaluation of the whole model's loss-derived parameter gradients from float64?  The oracle (oracle/
mvx_oracle.py, torch-CPU) run in float32 and in float64 on one full-size S2 frame with the real loss (clsLoss + regLoss,
train.py:146-161): per-parameter max-norm / 2-norm distance of the float32 gradients from the float64 ones.  This is the
yardstick for tests/test_fullsize_gpu.py::test_whole_model_losses_match_oracle_at_full_size (CPU only, ~10 min on 8 cores).
    python tools/grad_conditioning_cpu.py [out.json]"""
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, 'gpurun_out', 'grad_conditioning_cpu.json')
sys.argv = sys.argv[:1]
import mvx_oracle as O  # noqa: E402
from MVXNet import MVXNet  # noqa: E402

P_pts = 20000
pc = O.synth_ring(0, P_pts)
proj = O.lidar2img(pc, O.KITTI_CALIB, np.float32)[:, ::-1]
pts6 = np.ascontiguousarray(np.concatenate([pc, proj], 1), np.float32)
rv, ri, _ = O.group(pts6, O.synth_perm(0, P_pts), O.VELORANGE, O.voxelsize(), 35)
V = rv.shape[0]
fpn = [torch.from_numpy(f) for f in O.synth_fpn(0)]
torch.manual_seed(0)
model = MVXNet()
trainable = [k for k, p in model.named_parameters() if p.requires_grad]
gg = np.random.default_rng(11)
n = 8
gt = torch.tensor(np.stack([gg.uniform(8, 60, n), gg.uniform(-30, 30, n), gg.uniform(-1.8, -0.6, n), gg.uniform(3.4, 4.4, n),
                            gg.uniform(1.5, 1.8, n), gg.uniform(1.4, 1.7, n), gg.choice([0.0, np.pi / 2], n) + gg.normal(0, 0.05, n)], 1),
                  dtype=torch.float32)
anchors = O.create_anchors(176, 200)
rp, rn, rg = O.classify_anchors(O.bbox3d2bev(gt), gt[:, [0, 1]], O.bbox3d2bev(anchors.reshape(176, 200, 2, 7)), O.VELORANGE, 0.45, 0.6)
vox = torch.from_numpy(rv.astype(np.float32))
idx = torch.from_numpy(np.concatenate([np.zeros((V, 1), np.int64), ri.astype(np.int64)], 1))
with torch.no_grad():
    imf = O.feature_mapping(vox, fpn, torch.tensor([370.0, 1224.0]))
grads, losses = {}, {}
for dt in (torch.float32, torch.float64):
    P = {k: v.detach().to(dt).clone() for k, v in model.state_dict().items()}
    for k in trainable:
        P[k].requires_grad_(True)
    imf_t = O.image_feature_fusion(imf.to(dt), P, 'head.fusion.')
    v23 = torch.cat([vox[..., :7].to(dt), imf_t], dim=-1)
    bb = O.strip_prefix(P, 'backbone.')
    mid = O.voxelnet_middle(v23, idx, bb)
    score, reg = O.rpn(mid, bb)
    cls, rl = O.voxel_loss(rp, rn, rg, gt.to(dt), score[0].permute(1, 2, 0), reg[0].permute(1, 2, 0), anchors.to(dt), 2)
    (cls + rl).backward()
    grads[dt] = {k: P[k].grad.double() for k in trainable}
    losses[dt] = (float(cls), float(rl))
    print(dt, losses[dt], flush=True)
    del P, imf_t, v23, mid, score, reg, cls, rl
res = {'losses_f32': losses[torch.float32], 'losses_f64': losses[torch.float64], 'rel_maxnorm': {}, 'rel_2norm': {}}
for k in trainable:
    a, b = grads[torch.float32][k], grads[torch.float64][k]
    res['rel_maxnorm'][k] = float((a - b).abs().max() / b.abs().max())
    res['rel_2norm'][k] = float((a - b).norm() / b.norm())
for k, v in sorted(res['rel_2norm'].items(), key=lambda t: -t[1]):
    print('%-45s maxnorm %.2e  2norm %.2e' % (k, res['rel_maxnorm'][k], v))
with open(out_path, 'w') as fh:
    json.dump(res, fh, indent=1)
