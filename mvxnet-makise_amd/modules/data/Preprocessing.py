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


def _calib_products(calib, f32):
    """(R0_rect @ Tr_velo_to_cam, P2) as float64 host arrays.  The 4x4 product is formed on the host
    with the same operator the reference uses (numpy f64 @ for the numpy path, torch f32 @ for the
    tensor path, Preprocessing.py:46), so the kernel starts from identical matrices."""
    if f32:
        r0, tr, p2 = (torch.as_tensor(calib[k]).detach().cpu().float() for k in ('R0_rect', 'Tr_velo_to_cam', 'P2'))
        return (r0 @ tr).double().numpy(), p2.double().numpy()
    r0, tr, p2 = (np.asarray(calib[k], dtype=np.float64) for k in ('R0_rect', 'Tr_velo_to_cam', 'P2'))
    return r0 @ tr, p2


def _crop_generic(pcd, range6, bounds_f32, calib, imsize_wh, math_f32):
    is_np = isinstance(pcd, np.ndarray)
    dev = X.device() if is_np else pcd.device
    src = torch.from_numpy(np.ascontiguousarray(pcd, dtype=np.float32)).to(dev) if is_np else pcd.float().contiguous()
    if src.shape[0] == 0:
        return pcd[:0]
    m, p2 = _calib_products(calib, math_f32) if calib is not None else (None, None)
    out, n_out, _ = _hip.crop_points(src[None], None, range6, bounds_f32, m, p2,
                                     imsize_wh if imsize_wh is not None else (0.0, 0.0), math_f32)
    n = int(n_out[0])
    res = out[0, :n]
    return res.cpu().numpy() if is_np else res


def crop(pcd: np.ndarray, range: Sequence[float]):
    """Keep points with low <= xyz < high (reference Preprocessing.py:12-17; numpy in/out)."""
    return _crop_generic(pcd, list(range), False, None, None, False)


def cropTensor(pcd: torch.Tensor, range: Sequence[float]):
    """Tensor variant (reference Preprocessing.py:19-24): bounds rounded to f32 like torch.Tensor(range)."""
    return _crop_generic(pcd, list(range), True, None, None, False)


def cropToSight(pcd: Union[np.ndarray, torch.Tensor], calib: dict, imsize: Sequence[int]):
    """Keep points in front of the camera that project inside the image; ``imsize`` is (w, h)
    (reference Preprocessing.py:26-55).  numpy input -> f64 arithmetic, tensor input -> f32."""
    return _crop_generic(pcd, None, False, calib, (float(imsize[0]), float(imsize[1])), not isinstance(pcd, np.ndarray))


def cropFused(pcd, range, calib, imsize):
    """crop + cropToSight in ONE compaction pass (same result as cropToSight(crop(pcd)))."""
    return _crop_generic(pcd, list(range), not isinstance(pcd, np.ndarray), calib,
                         (float(imsize[0]), float(imsize[1])), not isinstance(pcd, np.ndarray))


def createAnchors(l, w, range, size):
    """(l, w, 14) anchor grid: two 7-dof anchors (yaw 0 and pi/2) per BEV cell at z = -1
    (reference Preprocessing.py:118-142).  Label preparation, outside the hot path; plain torch."""
    ls, ws = (range[3] - range[0]) / l, (range[4] - range[1]) / w
    x = torch.linspace(range[0] + ls / 2, range[3] - ls / 2, l)
    y = torch.linspace(range[1] + ws / 2, range[4] - ws / 2, w)
    gx, gy = torch.meshgrid(x, y, indexing='ij')
    base = torch.stack([gx, gy, torch.full_like(gx, -1.0)], dim=2)
    dims = torch.Tensor(size).expand(l, w, 3)
    yaw0, yaw1 = torch.zeros((l, w, 1)), torch.full((l, w, 1), torch.pi / 2)
    return torch.concat([base, dims, yaw0, base, dims, yaw1], dim=2)


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
