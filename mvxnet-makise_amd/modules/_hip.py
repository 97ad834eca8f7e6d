"""Typed torch-tensor wrappers over the C ABI (plumbing only: allocation, pointers, stream)."""
import collections

import torch

from modules import Extension as X

VoxelizeResult = collections.namedtuple('VoxelizeResult', 'voxels coords counts n_voxels status')

_ws_cache = {}


def workspace(nbytes, dev, tag):
    """Grow-only scratch buffer per (device, tag); stream-ordered reuse on the current stream."""
    key = (dev.index, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


def voxelize(pcd, perm, n_points, lo, size, T, out_channels, cap_voxels=None, ext_idx=None):
    """pcd f32 (F,capP,ncol) on the GPU; perm i32 (F,capP) or None; n_points i32 (F,) or None
    (= every frame full).  Returns capacity-sized outputs plus device-side counts."""
    assert pcd.dim() == 3 and pcd.dtype == torch.float32
    F, cap, ncol = pcd.shape
    dev = pcd.device
    if n_points is None:
        n_points = torch.full((F,), cap, dtype=torch.int32, device=dev)
    cap_voxels = cap if cap_voxels is None else int(cap_voxels)
    voxels = torch.empty((F, cap_voxels, T, out_channels), dtype=torch.float32, device=dev)
    coords = torch.empty((F, cap_voxels, 4), dtype=torch.int64, device=dev)
    counts = torch.empty((F, cap_voxels), dtype=torch.int32, device=dev)
    n_vox = torch.empty((F,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    nbytes = X.lib.mvx_voxelize_workspace_bytes(F, cap)
    ws = workspace(nbytes, dev, 'voxelize')
    rc = X.lib.mvx_voxelize(X.ptr(pcd), X.ptr(perm), X.ptr(n_points), X.ptr(ext_idx), F, cap, ncol,
                            float(lo[0]), float(lo[1]), float(lo[2]),
                            float(size[0]), float(size[1]), float(size[2]),
                            int(T), int(out_channels), cap_voxels,
                            X.ptr(voxels), X.ptr(coords), X.ptr(counts), X.ptr(n_vox), X.ptr(status),
                            X.ptr(ws), ws.numel(), X.stream())
    X.check(rc, 'mvx_voxelize')
    return VoxelizeResult(voxels, coords, counts, n_vox, status)
