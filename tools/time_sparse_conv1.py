"""Where does the input-sparse conv1 forward spend its time?  (developer tool)"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip
from modules.data import Synthetic as S
import modules.config as cfg

dev = torch.device('cuda')
D, H, W = 10, 352, 400
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
pc = S.synth_ring(0)
p6 = np.concatenate([pc, np.zeros((pc.shape[0], 2), np.float32)], 1)
res = _hip.voxelize(torch.from_numpy(p6).to(dev)[None], torch.from_numpy(S.synth_perm(0, pc.shape[0])).to(dev)[None], None,
                    cfg.velorange[:3], cfg.voxelsize, 35, 9)
V = int(res.n_voxels[0]); coords = res.coords[0, :V].contiguous()
feat = torch.randn(V, 128, device=dev)
w = torch.randn(64, 128, 3, 3, 3, device=dev) * 0.02; b = torch.zeros(64, device=dev)
wpk = _hip.conv3d_pack(w, False)
for name, nv in (('empty grid', 0), ('ring frame', V)):
    grid, st, occ = _hip.scatter_voxels(feat[:nv], coords[:nv], (D, H, W), want_occupancy=True)
    t = timeit(lambda: _hip.conv3d_forward(grid, wpk, b, 64, 2, 1, occupancy=occ))
    t2 = timeit(lambda: _hip.conv3d_forward(grid, wpk, b, 64, 2, 1, occupancy=occ, want_stats=False))
    print('%-12s V=%5d  sparse fwd %.1f us   (no stats: %.1f us)' % (name, nv, t, t2))
t = timeit(lambda: _hip.scatter_voxels(feat, coords, (D, H, W), want_occupancy=True))
print('scatter+memset %.1f us' % t)
y = torch.empty(5, H, W, 64, device=dev)
print('copy 180MB (read+write) %.1f us' % timeit(lambda: y.copy_(y)))
iz = coords[:, 3]
for name, mask in (('plane 3 only', iz == 3), ('all but plane 3', iz != 3), ('plane 4..9', iz >= 4)):
    c2 = coords[mask].contiguous(); f2 = feat[mask].contiguous()
    grid, st, occ = _hip.scatter_voxels(f2, c2, (D, H, W), want_occupancy=True)
    _hip.KERNEL_TIMERS = {}
    _hip.SPARSE_QUADS = None
    _hip.conv3d_forward(grid, wpk, b, 64, 2, 1, occupancy=occ)
    q = int(_hip.SPARSE_QUADS)
    _hip.KERNEL_TIMERS = None
    t = timeit(lambda: _hip.conv3d_forward(grid, wpk, b, 64, 2, 1, occupancy=occ))
    print('%-16s V=%5d  %.1f us   executed %.2f GFLOP  active tiles(occ>0) %d' % (name, c2.shape[0], t, q * 8 * 4096 / 1e9, int((occ[0] > 0).sum())))
