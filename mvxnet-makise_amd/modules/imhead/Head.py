"""ImageHead with the reference's interface (modules/imhead/Head.py): frozen extractor +
trainable fusion MLP; ``forward(img, voxels, calibs, imsize) -> (1, N, T, 16)``."""
import torch
from torch import nn

import modules.config as cfg
from modules import _hip
from .Pipe import ExpandRowsFunction, ImageFeatureExtractor, ImageFeatureFusion, _channels_last_levels


class ImageHead(nn.Module):

    def __init__(self):
        super().__init__()
        self.extractor = ImageFeatureExtractor()
        self.extractor.train(False)
        for p in self.extractor.parameters():
            p.requires_grad = False
        self.fusion = ImageFeatureFusion()

    @staticmethod
    def compact_map(voxels):
        """Enqueue the dense-row -> compact-row map of one frame; returns device tensors only (no host
        sync), so several frames can be prepared before a single read-back of their n_real."""
        v = voxels.squeeze(0) if voxels.dim() == 4 else voxels[0]
        n, t, c = v.shape
        return _hip.row_compact_map(v.view(n * t, c))

    def forward_compact(self, x, voxels, calibs, imsize, prepared=None, status_sink=None):
        """Fusion branch on compact rows: returns (imfeat (n_real+1, 16), CompactRows).  Row n_real is
        the shared padded row.  ``voxels`` (1,N,T,9) is zeroed in place on padded rows like the
        reference (imhead/Pipe.py:54-59).  ``prepared`` = (row_map, rows_sel, n_real:int) from
        ``compact_map`` avoids the host sync here; ``status_sink`` (a list) collects the device status
        word instead of checking it immediately (the reference's assert, imhead/Pipe.py:71)."""
        feats = self.extractor(x)
        v = voxels.squeeze(0) if voxels.dim() == 4 else voxels[0]
        hw = imsize.tolist() if torch.is_tensor(imsize) else list(imsize)
        if not v.is_contiguous():
            raise ValueError('ImageHead needs a contiguous voxel tensor (zeroed in place)')
        n, t, c = v.shape
        rows = n * t
        vox2d = v.view(rows, c)
        if prepared is None:
            row_map, rows_sel, n_real = _hip.row_compact_map(vox2d)
            nr = int(n_real)                               # one host sync, where the reference asserts
        else:
            row_map, rows_sel, nr = prepared
        levels = _channels_last_levels(feats, 0)
        width = levels[0].shape[2] * len(levels)
        compact = torch.empty((nr + 1, width), dtype=torch.float32, device=v.device)
        compact[nr].zero_()                                # the shared padded row (Pipe.py:80)
        status = _hip.feature_sample(vox2d, levels, (float(hw[0]), float(hw[1])), cfg.eps, compact, row_map, rows_sel=rows_sel, n_real=nr)
        row_w = torch.ones((nr + 1,), dtype=torch.float32, device=v.device)
        row_w[nr] = float(rows - nr)
        y = self.fusion.forward_rows(compact, row_w, rows)
        if status_sink is not None:
            status_sink.append(status)
        elif int(status) & 1:
            raise AssertionError('projected point outside the feature map')
        return y, _hip.CompactRows(row_map, rows_sel, nr, n, t), vox2d

    def forward(self, x, voxels, calibs, imsize):
        """``x``: image (1,3,H,W) or the list of FPN maps; ``voxels``: (1,N,T,9), zeroed in place on
        padded rows like the reference.  Returns the dense (1,N,T,16) tensor of the reference."""
        y, cr, _ = self.forward_compact(x, voxels, calibs, imsize)
        dense = ExpandRowsFunction.apply(y, cr.row_map, cr.n_real)
        return dense.view(1, cr.V, cr.T, -1)
