"""Batched voxelizer throughput vs batch size (algorithmic HBM bytes / time) -- developer tool."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
import bench
import modules.config as cfg
from modules import _hip
dev = torch.device('cuda')
for B in (1, 4, 16, 64):
    batch = bench.make_batch(list(range(B)), dev, 20000)
    def run():
        return _hip.voxelize(batch.points6, batch.perms, batch.n_points, cfg.velorange[0:3], cfg.voxelsize, cfg.samplenum, 9)
    res = run(); torch.cuda.synchronize()
    V = int(res.n_voxels.sum())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    nbytes = B * 20000 * 28 + V * (35 * 9 * 4 + 36)
    print('frames %3d  voxels %7d  %.3f ms  %.1f us/frame  %.0f GB/s algorithmic' % (B, V, ms, ms * 1e3 / B, nbytes / ms / 1e6))
