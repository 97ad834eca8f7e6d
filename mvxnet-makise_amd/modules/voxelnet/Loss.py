"""VoxelLoss with the reference's interface and arithmetic (modules/voxelnet/Loss.py:6-45), forward and backward in
one HIP call (csrc/loss.hip):

    posLoss = -sum log(score[pi] + eps) / (len(pi) + eps)
    negLoss = (sum_all -log(1 - score + eps) - sum over ni of the same) / (N - len(ni) + eps)
    clsLoss = a * posLoss + b * negLoss                      (a = 1.5, b = 1)
    regLoss = SmoothL1Loss()(reg[pi], targets)               (mean over len(pi) * 7)

``ni`` is the list of NOT-negative anchors that ``classifyAnchors`` returns (positives included); with ``pi is None``
the result is ``(mean(-log(1 - score + eps)), None)``, with an empty ``pi`` ``(clsLoss, None)`` -- as the reference.
The score / regression maps are read through their strides: the permuted views of train.py:132-133 cost no copy.
"""
import numpy as np
import torch
from torch import nn

import modules.config as cfg
from modules import _hip
from modules import Extension as X


def _index_block(idx, dev):
    """Index triple (numpy arrays, lists or tensors) -> contiguous i64 (3, n) on the device."""
    if idx is None:
        return None, 0
    cols = [torch.as_tensor(np.asarray(c) if not isinstance(c, torch.Tensor) else c).long().reshape(-1) for c in idx]
    n = cols[0].shape[0]
    if n == 0:
        return None, 0
    return torch.stack(cols).to(dev).contiguous(), n


class _VoxelLossFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, score, reg, pos, n_pos, neg, n_neg, gi, gts, anchors, A, a, b, eps):
        want = score.requires_grad or (reg is not None and reg.requires_grad)
        losses, dscore, dreg = _hip.voxel_loss(score, reg if n_pos > 0 else None, pos, neg, gi, n_pos, n_neg, gts, anchors,
                                               A, a, b, eps, want_grads=want)
        ctx.save_for_backward(dscore, dreg)
        ctx.has_reg = reg is not None
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_cls, g_reg):
        dscore, dreg = ctx.saved_tensors
        gs = dscore * g_cls if dscore is not None else None
        gr = None
        if ctx.has_reg and dreg is not None:
            gr = dreg * g_reg
        return (gs, gr) + (None,) * 11


class VoxelLoss(nn.Module):

    def __init__(self, a=1.5, b=1, eps=cfg.eps):
        super().__init__()
        self.a, self.b, self.eps = a, b, eps

    def forward(self, pi, ni, gi, gts, score, reg, anchors, anchorsPerLoc):
        if not score.is_cuda:
            raise X.MvxHipError('VoxelLoss runs on the GPU (no CPU fallback)')
        dev = score.device
        pos, n_pos = _index_block(pi, dev)
        neg, n_neg = _index_block(ni, dev)
        regress = pi is not None and n_pos > 0
        gi_d = gts_d = anc = None
        if regress:
            gi_d = torch.as_tensor(np.asarray(gi) if not isinstance(gi, torch.Tensor) else gi).long().to(dev).contiguous()
            gts_d = gts.detach().float().to(dev).contiguous()
            anc = anchors.detach().float().to(dev).contiguous()
        cls, rl = _VoxelLossFunction.apply(score.float(), reg.float() if regress else None, pos, n_pos, neg, n_neg, gi_d,
                                           gts_d, anc, int(anchorsPerLoc), float(self.a), float(self.b), float(self.eps))
        return cls, (rl if regress else None)
