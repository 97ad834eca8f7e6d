"""MVXNet top module with the reference's interface (MVXNet.py:13-27):
``forward(voxels (1,N,T,9), imgs, idx (N,4), calibs, imsize) -> (score, reg)``."""
import torch
from torch import nn

from modules.imhead import ImageHead
from modules.voxelnet import VoxelNet

__all__ = ['MVXNet']


def initWeights(m):
    # the reference re-initialises nn.Conv2d only (MVXNet.py:8-11): Linear / Conv3d /
    # ConvTranspose2d keep PyTorch's defaults
    if isinstance(m, nn.Conv2d):
        nn.init.xavier_uniform_(m.weight.data)
        m.bias.data.zero_()


class MVXNet(nn.Module):

    def __init__(self):
        super().__init__()
        self.head = ImageHead()
        self.backbone = VoxelNet()
        self.backbone.apply(initWeights)

    def prepack(self):
        """Pack the dense-conv weights on the CURRENT stream for the arithmetic in use, so frames that
        run on other streams find them ready (modules/pipeline.py)."""
        from modules.layers.Blocks import CRB3d, conv_split_math
        split = conv_split_math()
        for m in self.backbone.cml.modules():
            if isinstance(m, CRB3d):
                m._packer(False, split)
                m._packer(True, split)

    def point_features(self, voxels, imgs, calibs, imsize):
        """(1,N,T,9) -> (1,N,T,23): 7 geometric channels + 16 fused image channels
        (MVXNet.py:25-26).  Padded rows of ``voxels`` are zeroed in place by the head."""
        imfeatures = self.head(imgs, voxels, calibs, imsize)
        return torch.concat([voxels[..., :7], imfeatures], dim=-1)

    def middle(self, voxels, imgs, idx, calibs, imsize, compact=True, prepared=None, status_sink=None):
        """Hot path up to the RPN input.  ``compact=True`` evaluates the fusion MLP and the VFE stack on
        real rows + one padded row per voxel instead of all N*T rows -- identical results because
        inside MVXNet all padded rows of a voxel are identical; ``compact=False`` runs the
        reference's dense formulation."""
        if not compact:
            return self.backbone.middle(self.point_features(voxels, imgs, calibs, imsize), idx)
        from modules.voxelnet.Pipe import CompactInputFunction
        imfeat, cr, vox2d = self.head.forward_compact(imgs, voxels, calibs, imsize, prepared, status_sink)
        rows = CompactInputFunction.apply(imfeat, vox2d, cr)
        return self.backbone.middle(rows, idx, compact_rows=cr)

    def forward(self, voxels, imgs, idx, calibs, imsize, compact=True):
        """(1,N,T,9) voxels, image or FPN maps, (N,4) indices -> (score (1,2,H/2,W/2), reg (1,14,H/2,W/2)) (MVXNet.py:21-27).
        ``compact=True`` (default) evaluates the fusion MLP and the VFE stack on the real rows + one padded row per voxel
        and the first CML layer on the voxel rows: exact for ANY input, because featureMaping zeroes every row whose
        x = y = z = 0 in place (imhead/Pipe.py:54-59), which makes all such rows of a voxel identical from there on
        (SURVEY Q5).  On the GPU the whole forward is then ONE autograd node over the frame-set kernels (modules/whole.py);
        ``compact='modules'`` keeps the same arithmetic on the per-module autograd path (modules.voxelnet / imhead / layers),
        ``compact=False`` is the reference's dense (1,N,T,23) formulation."""
        if not compact:
            return self.backbone(self.point_features(voxels, imgs, calibs, imsize), idx)
        if compact != 'modules':
            from modules import whole
            if whole.supported(self, voxels, idx):
                return whole.forward(self, voxels, imgs, idx, imsize)
        from modules.voxelnet.Pipe import CompactInputFunction
        imfeat, cr, vox2d = self.head.forward_compact(imgs, voxels, calibs, imsize)
        rows = CompactInputFunction.apply(imfeat, vox2d, cr)
        return self.backbone(rows, idx, compact_rows=cr)
