import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip
from modules.data import Synthetic as S
import modules.config as cfg
dev = torch.device('cuda'); D, H, W = 10, 352, 400
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
pc = S.synth_ring(0); p6 = np.concatenate([pc, np.zeros((pc.shape[0], 2), np.float32)], 1)
res = _hip.voxelize(torch.from_numpy(p6).to(dev)[None], torch.from_numpy(S.synth_perm(0, pc.shape[0])).to(dev)[None], None, cfg.velorange[:3], cfg.voxelsize, 35, 9)
V = int(res.n_voxels[0]); coords = res.coords[0, :V].contiguous()
P = torch.randn(V, 27 * 64, device=dev); b = torch.randn(64, device=dev)
for name, nv in (('empty', 0), ('ring', V)):
    g, st = _hip.index_grid(coords[:nv], (D, H, W))
    for ws in (True, False):
        t = timeit(lambda: _hip.sparse_conv_output(P, g, (D, H, W), b, 64, 2, 1, want_stats=ws))
        print('%-6s stats=%d  %.1f us' % (name, ws, t))
print('index_grid %.1f us' % timeit(lambda: _hip.index_grid(coords, (D, H, W))))
y = torch.empty(5, H, W, 64, device=dev)
print('fill 180MB %.1f us' % timeit(lambda: y.fill_(1.0)))
print('bn_apply-like copy %.1f us' % timeit(lambda: torch.add(y, 1.0, out=y)))
