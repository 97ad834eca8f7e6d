"""Box geometry and target assignment with the reference's names (modules/Calc.py).

``classifyAnchors`` -- the call train.py:46 makes per frame -- runs on the GPU (csrc/anchors.hip) and returns the
reference's ``(pi, ni, gi)`` with the same members in the same order; the box conversions are a few tensor
operations on a handful of boxes (label preparation) and stay plain torch, as in the reference.
"""
import math
from typing import Sequence, Tuple, Union

import numpy as np
import torch

from modules import _hip
from modules import Extension as X

index3d = Tuple[torch.Tensor, torch.Tensor, torch.Tensor]


def getRotationMatrices(r: torch.Tensor):
    c, s = torch.cos(r).reshape((-1, 1)), torch.sin(r).reshape((-1, 1))
    return torch.concat([c, -s, s, c], dim=1).reshape((-1, 2, 2))


def bbox3d2bev(bbox3ds: torch.Tensor) -> torch.Tensor:
    """(..., 7) xyzlwhr -> BEV corner points (..., 4, 2) (reference Calc.py:15-38: unit-square corners scaled by
    (l, w), multiplied from the right by the rotation matrix, shifted by (x, y))."""
    assert bbox3ds.shape[-1] >= 7
    lead = bbox3ds.shape[:-1]
    b = bbox3ds.reshape((-1, bbox3ds.shape[-1]))
    corners = torch.tensor([[0.5, 0.5], [-0.5, 0.5], [-0.5, -0.5], [0.5, -0.5]], dtype=b.dtype, device=b.device)
    res = corners[None] * b[:, None, [3, 4]]
    res = res @ getRotationMatrices(b[:, 6]) + b[:, None, [0, 1]]
    return res.reshape(lead + (4, 2)) if len(lead) > 0 else res[0]


def bbox3d2corner(bbox3ds: torch.Tensor) -> torch.Tensor:
    """(..., 7) -> the 8 corners (..., 8, 3): top face then bottom face (reference Calc.py:40-62)."""
    assert bbox3ds.shape[-1] >= 7
    lead = bbox3ds.shape[:-1]
    b = bbox3ds.reshape((-1, bbox3ds.shape[-1]))
    bev = bbox3d2bev(b)
    z = b[:, None, 2:3].expand(-1, 4, 1)
    h = b[:, None, 5:6].expand(-1, 4, 1)
    res = torch.concat([torch.concat([bev, z + h], dim=2), torch.concat([bev, z], dim=2)], dim=1)
    return res.reshape(lead + (8, 3)) if len(lead) > 0 else res[0]


def anchorCenterCells(gtCenters: torch.Tensor, anchors_shape, velorange: Sequence[float]):
    """Centre cell of every ground truth, the f32 arithmetic of reference Calc.py:91-94."""
    l = (velorange[3] - velorange[0]) / anchors_shape[0]
    w = (velorange[4] - velorange[1]) / anchors_shape[1]
    nls = ((gtCenters[:, 0] - velorange[0] - l / 2) / l + 0.5).long()
    nws = ((gtCenters[:, 1] - velorange[1] - w / 2) / w + 0.5).long()
    return nls, nws


def _window_radius(gts: torch.Tensor, anchors: torch.Tensor) -> int:
    """Cells around the centre that can reach IoU >= 0.1: the boxes must at least touch, i.e. the centres are closer
    than half the sum of the two diagonals (the centre cell itself may be half a cell off the box centre)."""
    def diag(q):
        return float(torch.linalg.norm(q[..., 0, :] - q[..., 2, :], dim=-1).max())
    cell = min(float((anchors[1, 0, 0] - anchors[0, 0, 0]).abs().max()) if anchors.shape[0] > 1 else math.inf,
               float((anchors[0, 1, 0] - anchors[0, 0, 0]).abs().max()) if anchors.shape[1] > 1 else math.inf)
    if not math.isfinite(cell) or cell <= 0:
        return 55
    return max(2, min(55, int(math.ceil((diag(gts) + diag(anchors[0, 0])) / 2 / cell)) + 2))


def classifyAnchors(gts: torch.Tensor, gtCenters: torch.Tensor, anchors: torch.Tensor, velorange: Sequence[float],
                    negThr: float, posThr: float, device=None) -> Tuple[index3d, index3d, torch.Tensor]:
    """Reference Calc.py:88-96 (-> cpp/voxelutil.cpp:138-316).  gts (G,4,2) BEV corners, gtCenters (G,2), anchors
    (L,W,A,4,2) BEV corners.  Returns ``(pi, ni, gi)``: index triples (LongTensors on the GPU) of the positive and of
    the not-negative anchors and the ground-truth id of every positive, in the reference's order.  One host read
    (the two list lengths)."""
    dev = torch.device(device) if device is not None else (anchors.device if anchors.is_cuda else X.device())
    nls, nws = anchorCenterCells(gtCenters.float().cpu(), anchors.shape, velorange)
    a_dev = anchors if anchors.is_cuda else _anchors_on(anchors, dev)
    g_dev = gts.detach().float().contiguous().to(dev)
    radius = _window_radius(gts.detach().float().cpu(), anchors[:2, :2].detach().float().cpu())
    while True:
        pos, neg, gi, counts, status = _hip.classify_anchors(g_dev, a_dev, nls.to(dev), nws.to(dev), negThr, posThr, radius)
        n_pos, n_neg, st = counts.tolist() + status.tolist()
        if st & 1 and radius < 55:              # a box much larger than estimated: widen the window and replay
            radius = min(55, radius * 2)
            continue
        break
    if st & 4:
        raise X.MvxHipError('classifyAnchors: a ground-truth centre lies outside the anchor grid '
                            '(the reference reads out of bounds there)')
    if st & 1:
        raise X.MvxHipError('classifyAnchors: a ground truth overlaps anchors more than 55 cells from its centre')
    pi = (pos[0, :n_pos], pos[1, :n_pos], pos[2, :n_pos])
    ni = (neg[0, :n_neg], neg[1, :n_neg], neg[2, :n_neg])
    return pi, ni, gi[:n_pos]


def classifyAnchorsFrames(frames, anchors: torch.Tensor, velorange: Sequence[float], negThr: float, posThr: float, device=None):
    """classifyAnchors (train.py:46) for the frames of a step with ONE kernel pass and ONE host read: ``frames`` = per frame
    ``(gts (G,4,2), gtCenters (G,2))`` or None (no box).  Returns per frame ``(pi, ni, gi)`` exactly as classifyAnchors
    would (ground-truth ids local to the frame), or None."""
    dev = torch.device(device) if device is not None else (anchors.device if anchors.is_cuda else X.device())
    live = [k for k, fr in enumerate(frames) if fr is not None and fr[0].shape[0] > 0]
    out = [None] * len(frames)
    if not live:
        return out
    a_dev = anchors if anchors.is_cuda else _anchors_on(anchors, dev)
    gts = torch.cat([frames[k][0].detach().float().cpu().reshape(-1, 4, 2) for k in live])
    cen = torch.cat([frames[k][1].detach().float().cpu().reshape(-1, 2) for k in live])
    off = [0]
    for k in live:
        off.append(off[-1] + frames[k][0].shape[0])
    nls, nws = anchorCenterCells(cen, anchors.shape, velorange)
    radius = _window_radius(gts, anchors[:2, :2].detach().float().cpu())
    g_dev = gts.contiguous().to(dev)
    nls_d, nws_d = nls.to(dev), nws.to(dev)
    # frame sets hold at most MVX_MAX_FRAMES frames: larger batches go in groups
    res = {}
    step = X.MAX_FRAMES
    for lo in range(0, len(live), step):
        ids = live[lo:lo + step]
        sub_off = [o - off[lo] for o in off[lo:lo + len(ids) + 1]]
        sl = slice(off[lo], off[lo + len(ids)])
        while True:
            pos, neg, gi, counts, status = _hip.classify_anchors_frames(g_dev[sl], sub_off, a_dev, nls_d[sl], nws_d[sl], negThr,
                                                                        posThr, radius)
            host = torch.cat([counts.reshape(-1), status]).tolist()          # the one host read of the group
            st = host[-1]
            if st & 1 and radius < 55:              # a box much larger than estimated: widen the window and replay
                radius = min(55, radius * 2)
                continue
            break
        if st & 4:
            raise X.MvxHipError('classifyAnchors: a ground-truth centre lies outside the anchor grid '
                                '(the reference reads out of bounds there)')
        if st & 1:
            raise X.MvxHipError('classifyAnchors: a ground truth overlaps anchors more than 55 cells from its centre')
        for j, k in enumerate(ids):
            n_pos, n_neg = host[2 * j], host[2 * j + 1]
            res[k] = ((pos[j, 0, :n_pos], pos[j, 1, :n_pos], pos[j, 2, :n_pos]),
                      (neg[j, 0, :n_neg], neg[j, 1, :n_neg], neg[j, 2, :n_neg]), gi[j, :n_pos])
    for k in live:
        out[k] = res[k]
    return out


_ANCHOR_CACHE = {}


def _anchors_on(anchors, dev):
    """The anchor grid is a constant of the run (train.py:59-60): keep its device copy."""
    key = (anchors.data_ptr(), tuple(anchors.shape), str(dev))
    hit = _ANCHOR_CACHE.get(key)
    if hit is None or hit[0]() is not anchors:
        import weakref
        hit = (weakref.ref(anchors), anchors.detach().float().contiguous().to(dev))
        _ANCHOR_CACHE.clear()
        _ANCHOR_CACHE[key] = hit
    return hit[1]


def bboxCam2Lidar(camBoxes: Union[torch.Tensor, np.ndarray], c2v: Union[torch.Tensor, np.ndarray], inplace: bool = False):
    """(N,7) camera-frame 'hwlxyzr' -> lidar-frame 'xyzlwhr' (reference Calc.py:206-226)."""
    if not inplace:
        camBoxes = camBoxes.clone() if isinstance(camBoxes, torch.Tensor) else torch.as_tensor(np.array(camBoxes))
    xyz1 = torch.concat([camBoxes[:, 3:6], torch.ones((camBoxes.shape[0], 1))], dim=1).T
    xyz = (c2v @ xyz1).T
    camBoxes[:, 3:6] = camBoxes[:, [2, 1, 0]]
    camBoxes[:, :3] = xyz[:, :3]
    camBoxes[:, 6] = camBoxes[:, 6] - 0.5 * torch.pi
    return camBoxes


def decodeRegression(regmap: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
    """Reference Calc.py:228-236, as written there (the diagonal is taken from columns 0:2)."""
    assert regmap.shape == anchors.shape
    d = torch.sqrt(anchors[..., [0]] ** 2 + anchors[..., [1]] ** 2)
    res = torch.empty(regmap.shape, device=regmap.device)
    res[..., :2] = regmap[..., :2] * d + anchors[..., :2]
    res[..., 2] = regmap[..., 2] * anchors[..., 5] + anchors[..., 2]
    res[..., 3:6] = torch.exp(regmap[..., 3:6]) * anchors[..., 3:6]
    res[..., 6] = regmap[..., 6] + anchors[..., 6]
    return res
