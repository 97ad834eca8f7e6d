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


class BEVFunction(torch.autograd.Function):
    """(1,C,D,H,W) logical / channels-last physical -> (1,C*D,H,W) contiguous: the reshape of
    reference VoxelNet.py:36 as one tiled transposition kernel (and its inverse for the gradient)."""

    @staticmethod
    def forward(ctx, x):
        cl = x[0].permute(1, 2, 3, 0).contiguous()       # no copy for channels-last storage
        ctx.d = cl.shape[0]
        return _hip.cl_to_bev(cl)[None]

    @staticmethod
    def backward(ctx, g):
        return _hip.bev_to_cl(g[0].contiguous(), ctx.d).permute(3, 0, 1, 2)[None]


class VoxelNet(nn.Module):

    def __init__(self):
        super().__init__()
        self.svfe = Pipe.SVFE(cfg.samplenum)
        self.fcn = Pipe.FCN(128, 128)
        self.cml = Pipe.CML()
        self.rpn = Pipe.RPN()
        self.sparse_first_layer = True     # exact input-sparse evaluation of reindex + cml.conv1
        self.restricted_backward = True    # tile-restricted backward inside the CML chain (exact; see middle_cl)

    @staticmethod
    def reindex(x, idx):
        """x (N,128), idx (N,4) = (batch, ix, iy, iz) -> (1,128,D,H,W) zeros elsewhere
        (reference VoxelNet.py:16-22).  The result is a logical NCDHW view of channels-last
        storage, which the CML kernels consume without a copy."""
        d, h, w = cfg.voxelshape[2], cfg.voxelshape[0], cfg.voxelshape[1]
        idx = idx.contiguous()
        grid = ReindexFunction.apply(x, idx, (d, h, w))
        return grid.permute(3, 0, 1, 2).unsqueeze(0)

    def voxel_features(self, x):
        """SVFE -> FCN(128,128) -> max over T: (1,N,T,23) -> (N,128) (VoxelNet.py:27-33)."""
        b, n, t, _ = x.shape
        x = self.svfe(x)
        return Pipe.FCNMaxFunction.apply(_as_rows(x), self.fcn.fc.weight, self.fcn.fc.bias, b * n, t, cfg.eps, None)

    def voxel_features_compact(self, rows, cr):
        """The same on compact rows (n_real + V, 23) -> (V,128); exact inside MVXNet."""
        x = self.svfe.forward_compact(rows, cr)
        return Pipe.FCNMaxFunction.apply(x, self.fcn.fc.weight, self.fcn.fc.bias, cr.V, cr.T, cfg.eps, cr)

    def middle_cl(self, x, idx, compact_rows=None):
        """Everything before the BEV reshape: (1,N,T,23) [or compact rows], (N,4) -> the normalised CML output as a logical
        (1,C,D,H,W) view of channels-last storage."""
        from modules.layers import Blocks
        x = self.voxel_features(x) if compact_rows is None else self.voxel_features_compact(x, compact_rows)
        if self.sparse_first_layer:
            # reindex + cml.conv1 fused on the sparse rows (same numbers as the dense path below).  conv1 -> conv2 -> conv3 is
            # a chain inside this method -- no intermediate activation has another consumer -- so the restricted backward
            # forms (gradients only on the tiles a producer reads + closed-form plane sums) are safe here
            d, h, w = cfg.voxelshape[2], cfg.voxelshape[0], cfg.voxelshape[1]
            old, Blocks.RESTRICTED_BACKWARD = Blocks.RESTRICTED_BACKWARD, (Blocks.RESTRICTED_BACKWARD or self.restricted_backward)
            try:
                x = self.cml.conv1.forward_voxels(x, idx.contiguous(), (d, h, w))
                x = self.cml.conv3(self.cml.conv2(x))
            finally:
                Blocks.RESTRICTED_BACKWARD = old
        else:
            # the grid holds this library's voxel features (outputs of a BatchNorm + max): not a foreign input to conv1
            x = self.cml(_hip.mark_lib(self.reindex(x, idx)))
        return x

    def middle(self, x, idx, compact_rows=None):
        """Everything before the RPN: (1,N,T,23), (N,4) -> (1,128,H,W), channel = c*2+d.
        With ``compact_rows`` x is the compact row matrix instead of the dense tensor."""
        return BEVFunction.apply(self.middle_cl(x, idx, compact_rows))

    def heads(self, x_cl5):
        """RPN on the CML output (logical (1,C,D,H,W), channels-last storage).  On the GPU the (1,C*D,H,W) reshape of
        VoxelNet.py:36 is folded into the first RPN layer (modules/rpn_frames.py reads the planes directly)."""
        _, c, d, h, w = x_cl5.shape
        if self.rpn._hip_ok(x_cl5, h, w):
            planes = x_cl5[0].permute(1, 2, 3, 0)                  # (D,H,W,C): no copy for channels-last storage
            return self.rpn.forward_cl(planes, d, h, w, c)
        return self.rpn.forward_torch(BEVFunction.apply(x_cl5))

    def forward(self, x, idx, compact_rows=None):
        score, reg = self.heads(self.middle_cl(x, idx, compact_rows))
        return score, reg
