"""tests/golden/mvxnet_small_f64.npz: float64 run of the CPU oracle on the inputs of the
reference-generated fixture mvxnet_small.npz.  TEST INFRASTRUCTURE ONLY.

Why: on the tiny fixture grid the reference's own fp32 parameter gradients are 1-6 % away
from exact arithmetic (bias gradients in front of a BatchNorm are pure cancellation), so
tests measure the HIP path and the reference fixture against this float64 yardstick.
Does not need /root/reference.  Usage: python oracle/gen_f64_fixture.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import mvx_oracle as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def main():
    g = np.load(os.path.join(GOLDEN, 'mvxnet_small.npz'))
    dt = torch.float64
    P = {k: v.to(dt).clone().requires_grad_(True) for k, v in O.make_params(7).items()}
    vox = torch.from_numpy(g['voxels'].copy()).to(dt)
    feats = [torch.from_numpy(g[k]).to(dt) for k in ('f0', 'f1', 'f2')]
    v23 = O.mvx_point_features(vox, feats, torch.from_numpy(g['imsize_hw']).to(dt), P)
    mid = O.voxelnet_middle(v23, torch.from_numpy(g['idx']), O.strip_prefix(P, 'backbone.'),
                            [int(v) for v in g['voxelshape']])
    (mid[0] * torch.from_numpy(g['G']).to(dt)).sum().backward()
    np.savez_compressed(os.path.join(GOLDEN, 'mvxnet_small_f64.npz'), mid=mid[0].detach().numpy().astype(np.float32),
                        **{'grad.' + k: v.grad.numpy().astype(np.float32) for k, v in P.items() if v.numel() <= 20000})


def voxelnet_small():
    """float64 run of the VoxelNet-only fixture (voxelnet_small.npz): middle output and gradients."""
    g = np.load(os.path.join(GOLDEN, 'voxelnet_small.npz'))
    dt = torch.float64
    P = {k: v.to(dt).clone().requires_grad_(True) for k, v in O.strip_prefix(O.make_params(7), 'backbone.').items()}
    x = torch.from_numpy(g['x']).to(dt).requires_grad_(True)
    mid = O.voxelnet_middle(x, torch.from_numpy(g['idx']), P, [int(v) for v in g['voxelshape']])
    (mid[0] * torch.from_numpy(g['G']).to(dt)).sum().backward()
    np.savez_compressed(os.path.join(GOLDEN, 'voxelnet_small_f64.npz'), mid=mid[0].detach().numpy().astype(np.float32),
                        grad_x=x.grad.numpy().astype(np.float32),
                        **{'grad.' + k: v.grad.numpy().astype(np.float32) for k, v in P.items()})


if __name__ == '__main__':
    voxelnet_small()
    main()
