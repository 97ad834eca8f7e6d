"""Batched voxelizer throughput vs batch size and workload (algorithmic HBM bytes / time) -- developer tool.
Writes gpurun_out/voxelize_timing.json."""
import json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
import bench
import modules.config as cfg
from modules import _hip
dev = torch.device('cuda')
out = []
for wl in ('S2', 'S1'):
    for B in (1, 4, 16, 64):
        batch = bench.make_batch(list(range(B)), dev, 20000, wl, raw_points=20000)
        points6, n_points = batch.prepared()
        def run():
            return _hip.voxelize_concat(points6, batch.perms, n_points, cfg.velorange[0:3], cfg.voxelsize, cfg.samplenum, 9)
        res = run(); torch.cuda.synchronize()
        V = int(res[4][-1])
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            run()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        nbytes = B * 20000 * 28 + V * (35 * 9 * 4 + 36)
        out.append({'workload': wl, 'frames': B, 'voxels': V, 'ms': ms, 'us_per_frame': ms * 1e3 / B, 'algorithmic_GBps': nbytes / ms / 1e6})
        print('%s frames %3d  voxels %7d  %.3f ms  %.1f us/frame  %.0f GB/s algorithmic' % (wl, B, V, ms, ms * 1e3 / B, nbytes / ms / 1e6), flush=True)
        del batch, points6, res
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(REPO, 'gpurun_out', 'voxelize_timing.json'), 'w'), indent=1)
