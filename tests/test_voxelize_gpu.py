"""GPU parity of the HIP voxelizer (C ABI mvx_voxelize) against the reference-generated
fixtures and the CPU oracle.  Integer outputs and the f32 payload must be bit-exact."""
import numpy as np
import pytest
import torch

import mvx_oracle as O

pytestmark = pytest.mark.gpu


def _as_f32(v64):
    return v64.astype(np.float32)


@pytest.mark.parametrize('name,T', [('group_small', 35), ('group_full', 35), ('group_T5', 5)])
def test_group_matches_reference_fixture(golden, name, T):
    from modules.data import Preprocessing as pre
    g = golden(name)
    pcd = g['pcd'].copy()
    voxel, idx = pre.group(pcd, list(g['rng']), list(g['size']), T, perm=g['perm'])
    assert np.array_equal(pcd, g['pcd'][g['perm']])           # in-place shuffle side effect
    assert idx.dtype == np.float64 and np.array_equal(idx, g['uidx'])
    assert voxel.shape == g['voxel'].shape
    assert np.array_equal(voxel, _as_f32(g['voxel']))          # == torch.Tensor(voxel) of train.py:125
    if 'voxel7' in g.files:
        v7, i7 = pre.group_(g['pcd'][:, :4].copy(), g['rng'], g['size'], T, perm=g['perm'])
        assert np.array_equal(i7, g['uidx7'])
        assert np.array_equal(v7, g['voxel7'])


def test_native_group_signature(golden):
    """cpp._group(pcd, idx, T) keeps the reference extension's contract (voxelutil.cpp:325-360)."""
    from modules.Extension import cpp
    g = golden('group_small')
    s = g['pcd'][g['perm']][:, :4].copy()
    idx = O.voxel_index(s[:, :3], g['rng'], g['size'])
    voxel, (x, y, z), cnt = cpp._group(s, idx, 35)
    ref = g['voxel7'].copy()
    ref[..., 3:6] = 0
    assert np.array_equal(voxel, ref)
    assert np.array_equal(np.stack([x, y, z], 1), g['uidx7'])
    assert cnt.dtype == np.int64 and cnt.max() <= 35


@pytest.mark.parametrize('kind', ['uniform', 'ring'])
def test_group_full_size_vs_oracle(kind):
    """BASELINE-size frames (20k points, full grid): bit-exact vs the CPU oracle, plus the
    size-independent properties: every point lands in exactly one voxel, counts <= T."""
    from modules.data import Preprocessing as pre
    pc4 = O.synth_uniform(3) if kind == 'uniform' else O.synth_ring(3)
    proj = O.lidar2img(pc4, O.KITTI_CALIB, np.float32)[:, ::-1]
    pcd = np.concatenate([pc4, proj], axis=1).astype(np.float32)
    perm = O.synth_perm(3, pcd.shape[0])
    size = O.voxelsize()
    ref_v, ref_i, ref_c = O.group(pcd, perm, O.VELORANGE, size, 35)
    voxel, idx = pre.group(pcd.copy(), O.VELORANGE, size, 35, perm=perm)
    assert np.array_equal(idx, ref_i)
    assert np.array_equal(voxel, ref_v.astype(np.float32))
    real = ~np.all(voxel[..., :3] == 0, axis=-1)
    assert real.sum() == ref_c.sum()                            # every kept point is a real row
    assert len(np.unique(idx, axis=0)) == idx.shape[0]          # voxels are unique


def test_batched_frames_and_edge_cases():
    from modules import _hip
    dev = torch.device('cuda')
    size = O.voxelsize()
    frames = [O.synth_uniform(10, 3000), O.synth_uniform(11, 1000), np.zeros((0, 4), np.float32)]
    cap = 3000
    pcd = np.zeros((3, cap, 4), np.float32)
    perm = np.zeros((3, cap), np.int32)
    n = np.array([f.shape[0] for f in frames], np.int32)
    for k, f in enumerate(frames):
        pcd[k, :n[k]] = f
        perm[k, :n[k]] = O.synth_perm(k, int(n[k]))
    res = _hip.voxelize(torch.from_numpy(pcd).to(dev), torch.from_numpy(perm).to(dev),
                        torch.from_numpy(n).to(dev), O.VELORANGE[:3], size, 35, 9)
    nv = res.n_voxels.cpu().numpy()
    assert int(res.status) == 0 and nv[2] == 0
    for k in range(2):
        p6 = np.concatenate([frames[k], np.zeros((n[k], 2), np.float32)], 1)
        ref_v, ref_i, _ = O.group(p6, perm[k, :n[k]], O.VELORANGE, size, 35)
        assert nv[k] == ref_v.shape[0]
        assert np.array_equal(res.coords[k, :nv[k], 1:].cpu().numpy(), ref_i.astype(np.int64))
        assert np.array_equal(res.voxels[k, :nv[k]].cpu().numpy(), ref_v.astype(np.float32))
    # one voxel holding far more than T points (T-cap + >64-member segments)
    one = np.tile(np.array([[10.05, 3.05, -1.0, 0.5]], np.float32), (500, 1))
    one[:, 0] += np.linspace(0, 0.09, 500, dtype=np.float32)
    pm = O.synth_perm(5, 500)
    r = _hip.voxelize(torch.from_numpy(one).to(dev)[None], torch.from_numpy(pm).to(dev)[None], None,
                      O.VELORANGE[:3], size, 35, 9)
    ref_v, ref_i, ref_c = O.group(np.concatenate([one, np.zeros((500, 2), np.float32)], 1), pm,
                                  O.VELORANGE, size, 35)
    V = int(r.n_voxels[0])
    assert V == ref_v.shape[0]
    assert np.array_equal(r.voxels[0, :V].cpu().numpy(), ref_v.astype(np.float32))
    assert np.array_equal(r.counts[0, :V].cpu().numpy(), ref_c.astype(np.int32))
