"""Image branch of MVXNet with the reference's names (modules/imhead/Pipe.py): the frozen
extractor wrapper, ``featureMaping`` and ``ImageFeatureFusion`` -- sampling and the fusion MLP
run on HIP kernels.

Compact evaluation: every padded row of a frame enters the fusion MLP as the same
all-zero 768-vector, so ``ImageHead`` evaluates the MLP on the real rows plus ONE shared padded
row whose BatchNorm weight is the number of padded rows, then expands back to (1,N,T,16).
``featureMaping`` / ``ImageFeatureFusion.forward`` keep the reference's dense contracts."""
import torch
from torch import nn

import modules.config as cfg
from modules import _hip
from modules.layers import FCN, CRB2d
from modules.layers.Blocks import fcn_rows


class ImageFeatureExtractor(nn.Module):
    """Frozen torchvision Faster-RCNN-v2 ResNet50-FPN trunk (reference Pipe.py:8-21).  Third-party,
    forward-only and needs downloaded weights, so it is outside the hot path:
    it is built lazily and only if torchvision is importable; FPN maps can always be passed to
    ``ImageHead`` / ``MVXNet`` directly instead of an image."""

    def __init__(self):
        super().__init__()
        self.transform = None
        self.backbone = None

    def _build(self):
        try:
            from torchvision.models.detection.faster_rcnn import (FasterRCNN_ResNet50_FPN_V2_Weights,
                                                                  fasterrcnn_resnet50_fpn_v2)
        except ImportError as e:                      # pragma: no cover - depends on the image
            raise RuntimeError('torchvision is not installed: pass the three FPN maps instead of an '
                               'image (MVXNet.forward(voxels, [f0, f1, f2], ...))') from e
        net = fasterrcnn_resnet50_fpn_v2(weights=FasterRCNN_ResNet50_FPN_V2_Weights.DEFAULT)
        self.transform, self.backbone = net.transform, net.backbone

    def forward(self, x):
        if isinstance(x, (list, tuple)):              # precomputed FPN maps
            return list(x)
        if self.backbone is None:
            self._build()
        x, _ = self.transform(x)
        f = self.backbone(x.tensors)
        return [f['0'], f['1'], f['2']]


def _channels_last_levels(features, i):
    """FPN levels of batch element i as contiguous (H, W, C) maps (no copy if the extractor
    already produced channels_last memory format)."""
    return [f[i].permute(1, 2, 0).contiguous() for f in features]


def featureMaping(voxels, features, calibs, imsize):
    """Dense drop-in of the reference function (Pipe.py:23-82): ``voxels`` = batch of (N,T,C) with
    the projected (row, col) in the last two channels, ``features`` = levels of (batch,C1,H,W),
    ``imsize`` = (h, w) tensor.  Returns a list of (N,T,C1*levels); padded rows (x=y=z=0) are
    zeroed IN PLACE in ``voxels`` and yield zero features, like the reference."""
    hw = imsize.tolist() if torch.is_tensor(imsize) else [float(imsize[0]), float(imsize[1])]
    res = []
    for i in range(len(voxels)):
        v = voxels[i]
        if not v.is_contiguous():
            raise ValueError('featureMaping needs contiguous voxel tensors (zeroed in place)')
        n, t, c = v.shape
        levels = _channels_last_levels(features, i)
        out = torch.empty((n * t, levels[0].shape[2] * len(levels)), dtype=torch.float32, device=v.device)
        status = _hip.feature_sample(v.view(n * t, c), levels, hw, cfg.eps, out)
        if int(status) & 1:                           # the reference's assert (Pipe.py:71), also a sync
            raise AssertionError('projected point outside the feature map')
        res.append(out.view(n, t, -1))
    return res


class ExpandRowsFunction(torch.autograd.Function):
    """compact rows (n_real+1, C) -> dense rows (R, C); padded rows share the last compact row."""

    @staticmethod
    def forward(ctx, compact, row_map, pad_row):
        ctx.save_for_backward(row_map)
        ctx.meta = (pad_row, compact.shape[0])
        return _hip.expand_rows(compact.contiguous(), row_map, pad_row)

    @staticmethod
    def backward(ctx, g):
        (row_map,) = ctx.saved_tensors
        pad_row, n = ctx.meta
        return _hip.expand_rows_backward(g.contiguous(), row_map, pad_row, n), None, None


class ImageFeatureFusion(nn.Module):
    """768 -> 768 -> 128 -> 128 -> 16 -> 16, each affine -> ReLU -> BN (reference Pipe.py:84-104)."""

    def __init__(self):
        super().__init__()
        self.fcn1 = FCN(768, 768)
        self.conv1 = CRB2d(768, 128, 1, 1, 0)
        self.fcn2 = FCN(128, 128)
        self.conv2 = CRB2d(128, 16, 1, 1, 0)
        self.fcn3 = FCN(16, 16)

    def _layers(self):
        return ((self.fcn1.fc.weight, self.fcn1.fc.bias), (self.conv1.conv.weight, self.conv1.conv.bias),
                (self.fcn2.fc.weight, self.fcn2.fc.bias), (self.conv2.conv.weight, self.conv2.conv.bias),
                (self.fcn3.fc.weight, self.fcn3.fc.bias))

    def forward_rows(self, x2d, row_w=None, count=None):
        """rows (R,768) [+ multiplicities] -> (R,16)."""
        if _hip.split_pieces() == 4 and x2d.is_cuda and x2d.is_contiguous() and _hip.amax_of(x2d) is None:
            _hip.tensor_amax(x2d)          # fp16x3: the range of the sampled features, for layer 0 (_hip.foreign_split)
        for i, (w, b) in enumerate(self._layers()):
            x2d = fcn_rows(x2d, w, b, row_w, count, foreign=(i == 0))      # layer 0 reads the sampled image features
        return x2d

    def forward(self, x):
        # dense contract: (batch, N, T, 768) -> (batch, N, T, 16)
        out = self.forward_rows(x.reshape(-1, x.shape[-1]))
        return out.reshape(x.shape[:-1] + (out.shape[-1],))
