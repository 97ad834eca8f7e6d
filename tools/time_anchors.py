import os, sys, time
sys.argv = sys.argv[:1]
sys.path.insert(0, '/root/repo/mvxnet-makise_amd')
import numpy as np, torch
import modules.config as cfg
from modules import Calc, _hip
from modules.data import Preprocessing as pre
dev = torch.device('cuda')
anchors = pre.createAnchors(cfg.voxelshape[0] // 2, cfg.voxelshape[1] // 2, cfg.velorange, cfg.carsize)
bevs = Calc.bbox3d2bev(anchors.reshape(anchors.shape[:2] + (-1, 7))).to(dev).contiguous()
gg = np.random.default_rng(11); n = 8
gt = np.stack([gg.uniform(8, 60, n), gg.uniform(-30, 30, n), gg.uniform(-1.8, -0.6, n), gg.uniform(3.4, 4.4, n),
               gg.uniform(1.5, 1.8, n), gg.uniform(1.4, 1.7, n), gg.choice([0.0, np.pi / 2], n) + gg.normal(0, 0.05, n)], 1)
gt = torch.tensor(gt, dtype=torch.float32)
gb = Calc.bbox3d2bev(gt)
for _ in range(3):
    pi, ni, gi = Calc.classifyAnchors(gb, gt[:, [0, 1]], bevs, cfg.velorange, 0.45, 0.6)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    pi, ni, gi = Calc.classifyAnchors(gb, gt[:, [0, 1]], bevs, cfg.velorange, 0.45, 0.6)
torch.cuda.synchronize()
print('classifyAnchors: %.3f ms per call (host + GPU), %d positives, %d non-negatives' % ((time.perf_counter() - t0) / 20 * 1e3, len(pi[0]), len(ni[0])))
nls, nws = Calc.anchorCenterCells(gt[:, [0, 1]], bevs.shape, cfg.velorange)
g_dev = gb.float().contiguous().to(dev); nl = nls.to(dev); nw = nws.to(dev)
r = Calc._window_radius(gb, bevs[:2, :2].cpu())
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    _hip.classify_anchors(g_dev, bevs, nl, nw, 0.45, 0.6, r)
e.record(); torch.cuda.synchronize()
print('kernels only: %.3f ms per call, radius %d' % (s.elapsed_time(e) / 20, r))
