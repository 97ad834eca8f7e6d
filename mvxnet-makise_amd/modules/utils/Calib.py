"""Calibration helpers with the reference's names (modules/utils/Calib.py).  ``lidar2Img`` -- the
one on the hot path (train.py:32,38) -- runs on the GPU; numpy input -> f64 arithmetic and numpy
output, tensor input -> f32 arithmetic and a tensor on the input's device."""
from typing import Union

import numpy as np
import torch

from modules import _hip
from modules import Extension as X
from modules.data.Preprocessing import _calib_products


def lidar2Img(pcd: Union[np.ndarray, torch.Tensor], calib: dict, uncheck=False):
    """(N, 3+C) -> (N', 2) projected (width coord, height coord) = P2 @ R0_rect @ Tr_velo_to_cam @ p
    with the perspective divide (reference Calib.py:47-69).  ``uncheck=False`` drops the points
    behind the camera first."""
    assert pcd.ndim == 2, 'Point cloud should be in (N, 3 + C)'
    is_np = isinstance(pcd, np.ndarray)
    if not is_np and not isinstance(pcd, torch.Tensor):
        raise TypeError('pcd should be ndarray or Tensor')
    dev = X.device() if is_np or not pcd.is_cuda else pcd.device
    src = torch.from_numpy(np.ascontiguousarray(pcd, dtype=np.float32)) if is_np else pcd.detach().float()
    src = src.contiguous().to(dev)
    m, p2 = _calib_products(calib, not is_np)
    uv, z = _hip.lidar2img(src, m, p2, math_f32=not is_np, want_z=not uncheck)
    if not uncheck:
        uv = uv[z > 0]
    if is_np:
        return uv.cpu().numpy().astype(np.float64)
    return uv.to(pcd.device)


def _hom(pcd):
    one = np.ones if isinstance(pcd, np.ndarray) else torch.ones
    cat = np.concatenate if isinstance(pcd, np.ndarray) else torch.cat
    return cat([pcd[:, :3], one((pcd.shape[0], 1), dtype=pcd.dtype)], 1).T


def lidar2P2(pcd, calib):
    """Points in the rectified camera-2 frame (reference Calib.py:5-23; not used by training)."""
    return (calib['P2'] @ calib['R0_rect'] @ calib['Tr_velo_to_cam'] @ _hom(pcd))[:3].T


def p22Lidar(pcd, calib):
    """Inverse of lidar2P2 (reference Calib.py:25-45; not used by training)."""
    inv = np.linalg.inv if isinstance(pcd, np.ndarray) else torch.linalg.inv
    return (inv(calib['Tr_velo_to_cam']) @ inv(calib['R0_rect']) @ inv(calib['P2']) @ _hom(pcd))[:3].T
