"""crop / cropTensor / cropToSight / lidar2Img on the GPU against the reference fixtures."""
import numpy as np
import pytest
import torch

import mvx_oracle as O

pytestmark = pytest.mark.gpu


def test_crop_and_crop_to_sight_match_reference(golden):
    from modules.data import Preprocessing as pre
    g = golden('crop')
    raw = g['raw']
    c1 = pre.crop(raw.copy(), list(g['velorange']))
    assert np.array_equal(c1, g['crop'])                       # order-preserving, bit-exact rows
    ct = pre.cropTensor(torch.from_numpy(raw.copy()).cuda(), list(g['velorange']))
    assert np.array_equal(ct.cpu().numpy(), g['crop_tensor'])
    c2 = pre.cropToSight(c1.copy(), O.KITTI_CALIB, list(g['imsize_wh']))
    assert np.array_equal(c2, g['crop_to_sight'])
    calib32 = {k: torch.tensor(v, dtype=torch.float32) for k, v in O.KITTI_CALIB.items()}
    c2t = pre.cropToSight(torch.from_numpy(c1.copy()).cuda(), calib32, list(g['imsize_wh']))
    assert np.array_equal(c2t.cpu().numpy(), g['crop_to_sight_tensor'])
    fused = pre.cropFused(raw.copy(), list(g['velorange']), O.KITTI_CALIB, list(g['imsize_wh']))
    assert np.array_equal(fused, g['crop_to_sight'])


def test_crop_full_size_properties():
    """120k-point raw cloud (cropdata.py-sized): idempotence and agreement with the oracle."""
    from modules.data import Preprocessing as pre
    raw = O.synth_raw(1)
    a = pre.cropFused(raw.copy(), O.VELORANGE, O.KITTI_CALIB, (1224, 370))
    ref = O.crop_to_sight(O.crop(raw, O.VELORANGE), O.KITTI_CALIB, (1224, 370))
    assert np.array_equal(a, ref)
    assert np.array_equal(pre.cropFused(a.copy(), O.VELORANGE, O.KITTI_CALIB, (1224, 370)), a)
    assert pre.crop(np.zeros((0, 4), np.float32), O.VELORANGE).shape == (0, 4)


def test_lidar2img_matches_reference(golden):
    from modules.utils import lidar2Img
    g = golden('lidar2img')
    calib32 = {k: torch.tensor(v, dtype=torch.float32) for k, v in O.KITTI_CALIB.items()}
    p32 = lidar2Img(torch.from_numpy(g['pcd']), calib32, True)
    np.testing.assert_allclose(p32.numpy(), g['proj_f32'], rtol=2e-5, atol=2e-3)
    p64 = lidar2Img(g['pcd'].copy(), O.KITTI_CALIB, True)
    np.testing.assert_allclose(p64, g['proj_f64'], rtol=1e-6, atol=1e-4)   # returned through f32
    assert lidar2Img(g['pcd'].copy(), O.KITTI_CALIB, False).shape[0] <= g['pcd'].shape[0]
