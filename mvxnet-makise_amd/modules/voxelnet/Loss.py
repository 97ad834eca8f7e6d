"""VoxelLoss placeholder with the reference's name (modules/voxelnet/Loss.py).  The loss is
outside the hot path -- plain PyTorch, same arithmetic: positive/negative
classification terms (a = 1.5, b = 1) and SmoothL1 on the 7-dof residuals."""
import torch
from torch import nn


class VoxelLoss(nn.Module):

    def __init__(self, a=1.5, b=1.0, eps=1e-6):
        super().__init__()
        self.a, self.b, self.eps = a, b, eps
        self.smoothl1 = nn.SmoothL1Loss(reduction='sum')

    def forward(self, pi, ni, gi, gts, score, reg, anchors, anchorsPerLoc):
        """score (L,W,2), reg (L,W,14); pi/ni index triples (x, y, anchor) of positive/negative
        anchors, gi the ground-truth id of each positive."""
        if pi is None:
            neg = score.reshape(-1)
            return -self.b * torch.log(1 - neg + self.eps).mean(), None
        pos = score[pi[0], pi[1], pi[2]]
        negs = score[ni[0], ni[1], ni[2]]
        cls = -self.a * torch.log(pos + self.eps).sum() / max(1, pos.numel()) \
              - self.b * torch.log(1 - negs + self.eps).sum() / max(1, negs.numel())
        a = anchors.reshape(anchors.shape[0], anchors.shape[1], anchorsPerLoc, 7)[pi[0], pi[1], pi[2]]
        g = gts[gi]
        d = torch.sqrt(a[:, 3] ** 2 + a[:, 4] ** 2)
        target = torch.stack([(g[:, 0] - a[:, 0]) / d, (g[:, 1] - a[:, 1]) / d, (g[:, 2] - a[:, 2]) / a[:, 5],
                              torch.log(g[:, 3] / a[:, 3]), torch.log(g[:, 4] / a[:, 4]),
                              torch.log(g[:, 5] / a[:, 5]), g[:, 6] - a[:, 6]], dim=1)
        r = reg.reshape(reg.shape[0], reg.shape[1], anchorsPerLoc, 7)[pi[0], pi[1], pi[2]]
        return cls, self.smoothl1(r, target) / max(1, pos.numel())
