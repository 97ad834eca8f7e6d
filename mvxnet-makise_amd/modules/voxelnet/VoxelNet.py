"""VoxelNet backbone with the reference's interface (modules/voxelnet/VoxelNet.py):
``forward(x (1,N,T,23), idx (N,4) long) -> (score (1,2,H/2,W/2), reg (1,14,H/2,W/2))``,
``staticmethod reindex(x, idx)``, attributes ``svfe, fcn, cml, rpn``."""
import torch
from torch import nn

import modules.config as cfg
from modules import _hip
from modules.layers.Blocks import _as_rows
from modules.voxelnet import Pipe


class ReindexFunction(torch.autograd.Function):
    """(N,C) voxel rows -> dense channels-last grid (D,H,W,C); backward gathers the rows."""

    @staticmethod
    def forward(ctx, x, idx, dhw):
        grid, status = _hip.scatter_voxels(x.contiguous(), idx, dhw)
        ctx.save_for_backward(idx)
        ctx.n = x.shape[0]
        return grid

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return _hip.gather_voxels(g.contiguous(), idx, ctx.n), None, None


class VoxelNet(nn.Module):

    def __init__(self):
        super().__init__()
        self.svfe = Pipe.SVFE(cfg.samplenum)
        self.fcn = Pipe.FCN(128, 128)
        self.cml = Pipe.CML()
        self.rpn = Pipe.RPN()

    @staticmethod
    def reindex(x, idx):
        """x (N,128), idx (N,4) = (batch, ix, iy, iz) -> (1,128,D,H,W) zeros elsewhere
        (reference VoxelNet.py:16-22).  The result is a logical NCDHW view of channels-last
        storage, which the CML kernels consume without a copy."""
        d, h, w = cfg.voxelshape[2], cfg.voxelshape[0], cfg.voxelshape[1]
        idx = idx.contiguous()
        grid = ReindexFunction.apply(x, idx, (d, h, w))
        return grid.permute(3, 0, 1, 2)[None]

    def voxel_features(self, x):
        """SVFE -> FCN(128,128) -> max over T: (1,N,T,23) -> (N,128) (VoxelNet.py:27-33)."""
        b, n, t, _ = x.shape
        x = self.svfe(x)
        return Pipe.FCNMaxFunction.apply(_as_rows(x), self.fcn.fc.weight, self.fcn.fc.bias, b * n, t, cfg.eps)

    def middle(self, x, idx):
        """Everything before the RPN: (1,N,T,23), (N,4) -> (1,128,H,W), channel = c*2+d."""
        x = self.voxel_features(x)
        x = self.reindex(x, idx)
        x = self.cml(x)
        return x.reshape((1, -1, cfg.voxelshape[0], cfg.voxelshape[1]))

    def forward(self, x, idx):
        score, reg = self.rpn(self.middle(x, idx))
        return score, reg
