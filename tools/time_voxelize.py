"""Voxelizer alone (developer tool): the voxelize_concat call of a 4- or 16-frame batch, repeated; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel times.    python tools/time_voxelize.py [frames]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sys.argv = sys.argv[:1]
import bench  # noqa: E402
import modules.config as cfg  # noqa: E402
from modules import _hip  # noqa: E402

dev = torch.device('cuda')
batch = bench.make_batch(list(range(frames)), dev, 20000, 'S2')
points6, n_points = batch.prepared()
torch.cuda.synchronize()
for it in range(12):
    if it == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    out = _hip.voxelize_concat(points6, batch.perms, n_points, cfg.velorange[0:3], cfg.voxelsize, cfg.samplenum, 9)
torch.cuda.synchronize()
print('%d frames: %.3f ms per call, %d voxels' % (frames, (time.perf_counter() - t0) / 10 * 1e3, int(out[4][-1])))
