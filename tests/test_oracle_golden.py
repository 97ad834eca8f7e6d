"""Pins the CPU oracle (oracle/mvx_oracle.py, oracle/group_c.c) against fixtures
produced by running the reference itself (oracle/gen_golden.py).  CPU only."""
import ctypes
import os

import numpy as np
import pytest
import torch

import mvx_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_crop_matches_reference(golden):
    g = golden('crop')
    raw = g['raw']
    assert np.array_equal(O.crop(raw, g['velorange']), g['crop'])
    assert np.array_equal(O.crop(raw, g['velorange'], bounds_f32=True), g['crop_tensor'])
    c2 = O.crop_to_sight(g['crop'], O.KITTI_CALIB, g['imsize_wh'])
    assert np.array_equal(c2, g['crop_to_sight'])
    c2t = O.crop_to_sight(g['crop'], O.KITTI_CALIB, g['imsize_wh'], dtype=np.float32)
    assert np.array_equal(c2t, g['crop_to_sight_tensor'])


def test_lidar2img_matches_reference(golden):
    g = golden('lidar2img')
    p32 = O.lidar2img(g['pcd'], O.KITTI_CALIB, np.float32)
    p64 = O.lidar2img(g['pcd'], O.KITTI_CALIB, np.float64)
    np.testing.assert_allclose(p32, g['proj_f32'], rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(p64, g['proj_f64'], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize('name,T', [('group_small', 35), ('group_full', 35), ('group_T5', 5)])
def test_group_bit_exact(golden, name, T):
    g = golden(name)
    voxel, uidx, cnt = O.group(g['pcd'], g['perm'], g['rng'], g['size'], T)
    assert voxel.dtype == np.float64 and voxel.shape == g['voxel'].shape
    assert np.array_equal(uidx, g['uidx'])                    # integer indices: bit-exact
    assert np.array_equal(voxel, g['voxel'])                  # f64 payload: bit-exact too
    if 'voxel7' in g.files:
        v7, u7, _ = O.group7(g['pcd'][:, :4].copy(), g['perm'], g['rng'], g['size'], T)
        assert np.array_equal(u7, g['uidx7'])
        np.testing.assert_allclose(v7, g['voxel7'], rtol=0, atol=2e-6)


@pytest.mark.parametrize('name,T', [('group_small', 35), ('group_full', 35), ('group_T5', 5)])
def test_group_c_restatement_bit_exact(golden, name, T):
    lib = ctypes.CDLL(os.path.join(REPO, 'oracle', 'liboracle_c.so'))
    lib.oracle_group9.restype = ctypes.c_int64
    g = golden(name)
    pcd = np.ascontiguousarray(g['pcd'], np.float32)
    perm = np.ascontiguousarray(g['perm'], np.int32)
    P = pcd.shape[0]
    voxel = np.empty((P, T, 9), np.float64)
    uidx = np.empty((P, 3), np.float64)
    cnt = np.empty(P, np.int64)
    rng = np.ascontiguousarray(g['rng'], np.float64)
    size = np.ascontiguousarray(g['size'], np.float64)
    V = lib.oracle_group9(pcd.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(pcd.shape[1]),
                          perm.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(P),
                          rng.ctypes.data_as(ctypes.c_void_p), size.ctypes.data_as(ctypes.c_void_p),
                          ctypes.c_int32(T), voxel.ctypes.data_as(ctypes.c_void_p),
                          uidx.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p))
    assert V == g['voxel'].shape[0]
    assert np.array_equal(uidx[:V], g['uidx'])
    assert np.array_equal(voxel[:V], g['voxel'])


def test_group_edge_cases():
    rng, size = O.VELORANGE, O.voxelsize()
    v, u, c = O.group(np.zeros((0, 6), np.float32), np.zeros(0, np.int32), rng, size, 35)
    assert v.shape == (0, 35, 9) and u.shape == (0, 3)
    one = np.array([[1.0, 2.0, 0.5, 0.3, 10.0, 20.0]] * 50, np.float32)     # 50 identical points
    v, u, c = O.group(one, np.arange(50, dtype=np.int32), rng, size, 35)
    assert v.shape == (1, 35, 9) and c[0] == 35
    assert np.all(v[0, :, 3:6] == 0)


def test_feature_mapping_matches_reference(golden):
    g = golden('feature_mapping')
    vox = torch.from_numpy(g['voxels_in'].copy())
    feats = [torch.from_numpy(g[k]) for k in ('f0', 'f1', 'f2')]
    out = O.feature_mapping(vox, feats, torch.from_numpy(g['imsize_hw']))
    assert np.array_equal(vox.numpy(), g['voxels_after'])     # in-place zeroing side effect
    np.testing.assert_allclose(out.numpy(), g['out'], rtol=1e-6, atol=1e-6)


def test_vfe_stack_matches_reference(golden):
    g = golden('vfe')
    P = O.strip_prefix(O.make_params(7), 'backbone.')
    x = torch.from_numpy(g['x'])
    w, b = P['svfe.vfe1.fcn.fc.weight'], P['svfe.vfe1.fcn.fc.bias']
    np.testing.assert_allclose(O.fcn(x, w, b).numpy(), g['fcn_out'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(O.vfe(x, w, b).numpy(), g['vfe_out'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(O.svfe(x, P).numpy(), g['svfe_out'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(O.voxel_features(x, P).numpy(), g['head_out'], rtol=1e-4, atol=1e-4)


def test_voxelnet_middle_and_grads_match_reference(golden):
    g = golden('voxelnet_small')
    P = {k: v.clone().requires_grad_(True) for k, v in O.strip_prefix(O.make_params(7), 'backbone.').items()}
    x = torch.from_numpy(g['x']).requires_grad_(True)
    idx = torch.from_numpy(g['idx'])
    shape = [int(v) for v in g['voxelshape']]
    feat = O.voxel_features(x, P)
    np.testing.assert_allclose(feat.detach().numpy(), g['feat'], rtol=1e-4, atol=1e-4)
    mid = O.voxelnet_middle(x, idx, P, shape)
    np.testing.assert_allclose(mid[0].detach().numpy(), g['mid'], rtol=1e-4, atol=1e-4)
    (mid[0] * torch.from_numpy(g['G'])).sum().backward()
    for k, v in P.items():
        ref = g['grad.' + k]
        scale = max(1e-6, float(np.abs(ref).max()))
        assert float(np.abs(v.grad.numpy() - ref).max()) / scale < 2e-3, k
    ref = g['grad_x']
    assert float(np.abs(x.grad.numpy() - ref).max()) / float(np.abs(ref).max()) < 2e-3


def test_fusion_matches_reference(golden):
    g = golden('fusion')
    P = {k: v.clone().requires_grad_(True) for k, v in O.strip_prefix(O.make_params(7), 'head.fusion.').items()}
    x = torch.from_numpy(g['x']).requires_grad_(True)
    y = O.image_feature_fusion(x, P)
    np.testing.assert_allclose(y.detach().numpy(), g['out'], rtol=1e-4, atol=1e-4)
    (y * torch.from_numpy(g['G'])).sum().backward()
    for k, v in P.items():
        if 'grad.' + k in g.files:
            ref = g['grad.' + k]
            scale = max(1e-6, float(np.abs(ref).max()))
            assert float(np.abs(v.grad.numpy() - ref).max()) / scale < 5e-3, k
        else:
            ref = g['gradslice.' + k]
            got = v.grad.numpy().reshape(v.shape[0], -1)[:8, :64]
            scale = max(1e-6, float(np.abs(ref).max()))
            assert float(np.abs(got - ref).max()) / scale < 5e-3, k


def test_mvxnet_front_matches_reference(golden):
    g = golden('mvxnet_small')
    P = O.make_params(7)
    vox = torch.from_numpy(g['voxels'].copy())
    feats = [torch.from_numpy(g[k]) for k in ('f0', 'f1', 'f2')]
    v23 = O.mvx_point_features(vox, feats, torch.from_numpy(g['imsize_hw']), P)
    np.testing.assert_allclose(v23.numpy(), g['v23'], rtol=1e-4, atol=2e-4)
    bp = O.strip_prefix(P, 'backbone.')
    feat = O.voxel_features(v23, bp)
    np.testing.assert_allclose(feat.numpy(), g['feat'], rtol=1e-4, atol=5e-4)
    shape = [int(v) for v in g['voxelshape']]
    mid = O.voxelnet_middle(v23, torch.from_numpy(g['idx']), bp, shape)
    np.testing.assert_allclose(mid[0].numpy(), g['mid'], rtol=1e-3, atol=2e-3)


def test_rpn_matches_reference(golden):
    """oracle.rpn against the score / regression maps the reference's VoxelNet produced on the small grid."""
    g = golden('voxelnet_small')
    P = O.rpn_params(golden('rpn_shapes'))
    score, reg = O.rpn(torch.from_numpy(g['mid'])[None], P)
    assert float((score[0] - torch.from_numpy(g['score'])).abs().max()) < 2e-5
    assert float((reg[0] - torch.from_numpy(g['reg'])).abs().max() / np.abs(g['reg']).max()) < 2e-5
