"""Point-cloud preprocessing on the GPU, with the reference's function names and return
conventions (modules/data/Preprocessing.py).  Every function runs HIP kernels through the
C ABI; numpy arrays are accepted and returned where the reference used numpy.

One deliberate API addition: the reference shuffles ``pcd`` in place with the global numpy
RNG inside ``group``/``group_`` (Preprocessing.py:66,86).  Here the host still draws that
permutation with ``np.random`` (same stream of random numbers) unless ``perm=`` is given, and
the shuffle itself happens on the GPU; ``pcd`` is shuffled in place afterwards so callers
that relied on the side effect still see it.
"""
from typing import List, Sequence, Union

import numpy as np
import torch

from modules import _hip
from modules import Extension as X


def _draw_perm(n: int) -> np.ndarray:
    """The permutation np.random.shuffle would apply to an (n, k) array (same RNG draws)."""
    a = np.arange(n, dtype=np.int32)
    np.random.shuffle(a)
    return a


def _voxelize_numpy(pcd: np.ndarray, rng, size, T: int, channels: int, perm):
    dev = X.device()
    P = pcd.shape[0]
    if perm is None:
        perm = _draw_perm(P)
    perm = np.ascontiguousarray(perm, dtype=np.int32)
    if P == 0:
        return (np.zeros((0, T, channels), np.float32), np.zeros((0, 3), np.int64),
                np.zeros(0, np.int64), perm)
    src = np.ascontiguousarray(pcd, dtype=np.float32)
    res = _hip.voxelize(torch.from_numpy(src).to(dev)[None], torch.from_numpy(perm).to(dev)[None], None,
                        rng[0:3], size, T, channels)
    st = int(res.status)
    if st & 1:
        raise X.MvxHipError('voxel index outside the 21-bit key range (points far outside velorange?)')
    V = int(res.n_voxels[0])
    voxel = res.voxels[0, :V].cpu().numpy()
    idx = res.coords[0, :V, 1:].cpu().numpy()
    cnt = res.counts[0, :V].cpu().numpy().astype(np.int64)
    return voxel, idx, cnt, perm


def group(pcd: np.ndarray, range: List[float], size: List[float], samplesPerVoxel: int, perm=None):
    """9-channel voxelizer (reference ``group``, Preprocessing.py:75-116).

    Returns ``(voxel (V,T,9), indices (V,3))``.  ``voxel`` is float32 -- the reference's float64
    rounded once, i.e. exactly ``torch.Tensor(voxel)`` of train.py:125 -- and ``indices`` holds
    the integer triples (the reference stores the same integers as float64)."""
    voxel, idx, _, perm = _voxelize_numpy(pcd, range, size, samplesPerVoxel, 9, perm)
    if isinstance(pcd, np.ndarray) and pcd.shape[0]:
        pcd[...] = pcd[perm]
    return voxel, idx.astype(np.float64)


def group_(pcd: np.ndarray, range: Sequence[float], size: Sequence[float], samplesPerVoxel: int, perm=None):
    """7-channel voxelizer (reference ``group_``, Preprocessing.py:57-73).
    Returns ``(voxel f32 (V,T,7), indices i64 (V,3))``."""
    voxel, idx, _, perm = _voxelize_numpy(pcd[:, :4], range, size, samplesPerVoxel, 7, perm)
    if isinstance(pcd, np.ndarray) and pcd.shape[0]:
        pcd[...] = pcd[perm]
    return voxel, idx
