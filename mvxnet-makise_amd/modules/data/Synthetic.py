"""Synthetic KITTI-shaped inputs: there is no dataset on the GPU box.

S1 "uniform": 20,000 points uniform in the crop range (worst case, ~1 point per voxel).
S2 "ring": 64-beam spinning-lidar model over flat ground with box occluders, cropped to the
range and the camera frustum, subsampled to 20,000 points (KITTI-like occupancy).
Seeds: 1000+frame (points), 2000+frame (shuffle), 3000+frame (FPN maps).  Host-side numpy:
data generation is not part of the hot path.
"""
import numpy as np

VELORANGE = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0]


def _crop(pcd, rng):
    lo, hi = np.asarray(rng[:3], np.float64), np.asarray(rng[3:], np.float64)
    roi = pcd[:, :3].astype(np.float64)
    return pcd[np.all((lo <= roi) & (roi < hi), axis=1)]


def _crop_to_sight(pcd, calib, imsize_wh):
    pts = np.ones((4, pcd.shape[0]), np.float64)
    pts[:3] = pcd[:, :3].T
    cam = (calib['R0_rect'] @ calib['Tr_velo_to_cam']) @ pts
    img = calib['P2'] @ cam
    with np.errstate(divide='ignore', invalid='ignore'):
        u, v = img[0] / img[2], img[1] / img[2]
    lim = np.asarray(imsize_wh, np.float64) - 1e-3
    return pcd[(cam[2] > 0) & (u >= 0) & (v >= 0) & (u < lim[0]) & (v < lim[1])]


KITTI_CALIB = {
    # public KITTI 2011_09_26 object calibration, padded to 4x4 as
    # modules/data/Load.py:24-41 does (values parsed as float32 there).
    'P2': np.array([[721.5377, 0.0, 609.5593, 44.85728],
                    [0.0, 721.5377, 172.854, 0.2163791],
                    [0.0, 0.0, 1.0, 0.002745884],
                    [0.0, 0.0, 0.0, 1.0]], dtype=np.float32).astype(np.float64),
    'R0_rect': np.array([[0.9999239, 0.00983776, -0.007445048, 0.0],
                         [-0.009869795, 0.9999421, -0.004278459, 0.0],
                         [0.007402527, 0.004351614, 0.9999631, 0.0],
                         [0.0, 0.0, 0.0, 1.0]], dtype=np.float32).astype(np.float64),
    'Tr_velo_to_cam': np.array([[0.007533745, -0.9999714, -0.000616602, -0.004069766],
                                [0.01480249, 0.0007280733, -0.9998902, -0.07631618],
                                [0.9998621, 0.00752379, 0.01480755, -0.2717806],
                                [0.0, 0.0, 0.0, 1.0]], dtype=np.float32).astype(np.float64),
}


def synth_uniform(frame_id: int, P: int = 20000, rng=VELORANGE) -> np.ndarray:
    """S1 'uniform' frame: (P,4) f32 inside the crop range."""
    g = np.random.default_rng(1000 + frame_id)
    lo = np.asarray(rng[:3], np.float32)
    hi = np.asarray(rng[3:], np.float32)
    xyz = (g.random((P, 3)) * (hi - lo) + lo).astype(np.float32)
    xyz = np.minimum(xyz, np.nextafter(hi, -np.inf, dtype=np.float32))
    xyz = np.maximum(xyz, lo)
    r = g.random((P, 1)).astype(np.float32)
    return np.concatenate([xyz, r], axis=1)


def synth_uniform_in_sight(frame_id: int, P: int = 20000, rng=VELORANGE, calib=None, imsize_wh=(1224, 370)) -> np.ndarray:
    """S1 'uniform' for runs WITH image fusion: uniform in the crop range, restricted to the camera frustum (every point
    must project into the image, modules/imhead/Pipe.py:71), P points; still about one point per voxel."""
    calib = KITTI_CALIB if calib is None else calib
    g = np.random.default_rng(1000 + frame_id)
    lo = np.asarray(rng[:3], np.float32)
    hi = np.asarray(rng[3:], np.float32)
    parts, have = [], 0
    while have < P:
        xyz = (g.random((4 * P, 3)) * (hi - lo) + lo).astype(np.float32)
        xyz = np.maximum(np.minimum(xyz, np.nextafter(hi, -np.inf, dtype=np.float32)), lo)
        pc = np.concatenate([xyz, g.random((4 * P, 1)).astype(np.float32)], axis=1)
        pc = _crop_to_sight(_crop(pc, rng), calib, imsize_wh)
        parts.append(pc)
        have += pc.shape[0]
    return np.ascontiguousarray(np.concatenate(parts, 0)[:P])


def synth_raw_around(kept: np.ndarray, frame_id: int, total: int = 120000, rng=VELORANGE, calib=None,
                     imsize_wh=(1224, 370)) -> np.ndarray:
    """A raw (un-cropped) cloud of `total` points whose crop + cropToSight is exactly `kept`, in the same order: the kept
    points interleaved with points that fail the range crop or the frustum test (what cropdata.py / Load.py read from
    disk before cropping)."""
    calib = KITTI_CALIB if calib is None else calib
    g = np.random.default_rng(4000 + frame_id)
    need = total - kept.shape[0]
    assert need >= 0
    if need == 0:
        return np.ascontiguousarray(kept[:, :4], np.float32)
    rejected, have = [], 0
    while have < need:
        cand = np.stack([g.uniform(-80, 80, 2 * need + 16), g.uniform(-80, 80, 2 * need + 16), g.uniform(-4, 3, 2 * need + 16),
                         g.random(2 * need + 16)], axis=1).astype(np.float32)
        lo, hi = np.asarray(rng[:3], np.float64), np.asarray(rng[3:], np.float64)
        roi = cand[:, :3].astype(np.float64)
        in_range = np.all((lo <= roi) & (roi < hi), axis=1)
        bad = cand[~in_range]
        inr = cand[in_range]
        if inr.shape[0]:
            ok = _crop_to_sight(inr, calib, imsize_wh)
            # in-range points outside the frustum: everything in `inr` that is not in `ok` (rows are unique with probability 1)
            keep_rows = {r.tobytes() for r in ok}
            outside = np.array([r for r in inr if r.tobytes() not in keep_rows], np.float32).reshape(-1, 4)
            bad = np.concatenate([bad, outside], 0)
        rejected.append(bad)
        have += bad.shape[0]
    rej = np.concatenate(rejected, 0)[:need]
    slot = np.zeros(total, bool)
    slot[np.sort(g.choice(total, kept.shape[0], replace=False))] = True
    out = np.empty((total, 4), np.float32)
    out[slot] = kept[:, :4]
    out[~slot] = rej
    return out


def synth_raw(frame_id: int, P: int = 120000) -> np.ndarray:
    """Raw un-cropped cloud for the crop/cropToSight config."""
    g = np.random.default_rng(1000 + frame_id)
    x = g.uniform(-80, 80, P)
    y = g.uniform(-80, 80, P)
    z = g.uniform(-4, 3, P)
    r = g.random(P)
    return np.stack([x, y, z, r], axis=1).astype(np.float32)


def synth_ring(frame_id: int, P: int = 20000, rng=VELORANGE, calib=None,
               imsize_wh=(1224, 370)) -> np.ndarray:
    """S2 'ring' frame: 64-beam spinning-lidar model over flat ground with box
    occluders, cropped to range + camera frustum, subsampled to P points."""
    calib = KITTI_CALIB if calib is None else calib
    g = np.random.default_rng(1000 + frame_id)
    elev = np.deg2rad(np.linspace(-24.8, 2.0, 64))
    azim = np.deg2rad(np.arange(-45.0, 45.0, 0.09))
    pts = []
    nbox = 12
    bc = np.stack([g.uniform(5, 60, nbox), g.uniform(-20, 20, nbox)], 1)
    bs = np.stack([g.uniform(1.5, 4.5, nbox), g.uniform(1.5, 2.5, nbox), g.uniform(1.4, 2.5, nbox)], 1)
    h = 1.73
    for rep in range(4):                                 # several sweeps with jitter -> enough points
        az = (azim + g.normal(0, 2e-4, azim.size))[None, :]
        el = (elev + g.normal(0, 2e-4, elev.size))[:, None]
        dx, dy, dz = np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el) * np.ones_like(az)
        with np.errstate(divide='ignore'):
            t = np.where(dz < 0, -h / dz, np.inf)
        t = np.minimum(t, 120.0)
        for k in range(nbox):                            # ray/AABB slab test
            lo = np.array([bc[k, 0] - bs[k, 0] / 2, bc[k, 1] - bs[k, 1] / 2, -h])
            hi = np.array([bc[k, 0] + bs[k, 0] / 2, bc[k, 1] + bs[k, 1] / 2, -h + bs[k, 2]])
            with np.errstate(divide='ignore', invalid='ignore'):
                t1 = np.stack([lo[0] / dx, lo[1] / dy, lo[2] / dz])
                t2 = np.stack([hi[0] / dx, hi[1] / dy, hi[2] / dz])
            tn = np.nanmax(np.minimum(t1, t2), axis=0)
            tf = np.nanmin(np.maximum(t1, t2), axis=0)
            hit = (tn <= tf) & (tn > 0)
            t = np.where(hit & (tn < t), tn, t)
        t = t * (1 + g.normal(0, 2e-3, t.shape))
        ok = np.isfinite(t) & (t < 119.0)
        p = np.stack([(t * dx)[ok], (t * dy)[ok], (t * dz)[ok], g.random(int(ok.sum()))], 1)
        pts.append(p.astype(np.float32))
    pcd = np.concatenate(pts, 0)
    pcd = _crop(pcd, rng)
    pcd = _crop_to_sight(pcd, calib, imsize_wh)
    if pcd.shape[0] >= P:
        sel = np.sort(g.choice(pcd.shape[0], P, replace=False))
        pcd = pcd[sel]
    return np.ascontiguousarray(pcd)


def synth_perm(frame_id: int, P: int) -> np.ndarray:
    return np.random.default_rng(2000 + frame_id).permutation(P).astype(np.int32)


def synth_fpn(frame_id: int, shapes=((256, 104, 336), (256, 52, 168), (256, 26, 84))):
    g = np.random.default_rng(3000 + frame_id)
    return [g.standard_normal(s, dtype=np.float32) for s in shapes]


def write_kitti_tree(root: str, frame_ids, points: int = 20000, raw_points: int = 120000, cars: int = 6, seed: int = 0):
    """A KITTI-format directory tree with synthetic frames, for the reader / training-loop tests (there is no dataset on
    the GPU box): training/{velodyne, velodyne_croped, label_2, calib, image_2} and ImageSets/train.txt, in the layout
    modules/data/Load.py:17-20 and cropdata.py:21-28 expect.  Labels hold `cars` random 'Car' boxes (camera frame, KITTI
    column order) plus one 'Pedestrian' row that the reader must skip."""
    import os
    from PIL import Image
    t = os.path.join(root, 'training')
    for d in ('velodyne', 'velodyne_croped', 'label_2', 'calib', 'image_2'):
        os.makedirs(os.path.join(t, d), exist_ok=True)
    os.makedirs(os.path.join(root, 'ImageSets'), exist_ok=True)
    names = []
    v2c = KITTI_CALIB['Tr_velo_to_cam']
    for fid in frame_ids:
        name = '%06d' % fid
        names.append(name)
        pc = synth_ring(fid, points)
        pc.tofile(os.path.join(t, 'velodyne_croped', name + '.bin'))
        synth_raw_around(pc, fid, raw_points).tofile(os.path.join(t, 'velodyne', name + '.bin'))
        g = np.random.default_rng(seed * 1000 + fid)
        img = g.integers(0, 256, (375, 1242, 3), dtype=np.uint8)
        Image.fromarray(img).save(os.path.join(t, 'image_2', name + '.png'))
        with open(os.path.join(t, 'calib', name + '.txt'), 'w') as f:
            p2 = ' '.join('%.6e' % v for v in KITTI_CALIB['P2'][:3].reshape(-1))
            r0 = ' '.join('%.6e' % v for v in KITTI_CALIB['R0_rect'][:3, :3].reshape(-1))
            tr = ' '.join('%.6e' % v for v in v2c[:3].reshape(-1))
            f.write('P0: %s\nP1: %s\nP2: %s\nP3: %s\nR0_rect: %s\nTr_velo_to_cam: %s\nTr_imu_to_velo: %s\n' % (p2, p2, p2, p2, r0, tr, tr))
        with open(os.path.join(t, 'label_2', name + '.txt'), 'w') as f:
            for k in range(cars):
                # a car in the lidar frame, written in the camera frame (inverse of modules/Calc.py bboxCam2Lidar)
                x, y, z = g.uniform(8, 60), g.uniform(-20, 20), g.uniform(-1.8, -1.2)
                l, w, h = g.uniform(3.4, 4.4), g.uniform(1.5, 1.8), g.uniform(1.4, 1.7)
                yaw = g.choice([0.0, np.pi / 2]) + g.normal(0, 0.05)
                cam = v2c @ np.array([x, y, z, 1.0])
                ry = yaw + 0.5 * np.pi
                f.write('Car 0.00 0 0.00 100.00 100.00 200.00 200.00 %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n'
                        % (h, w, l, cam[0], cam[1], cam[2], ry))
            f.write('Pedestrian 0.00 0 0.00 10.00 10.00 20.00 20.00 1.70 0.60 0.80 1.00 1.50 10.00 0.10\n')
    with open(os.path.join(root, 'ImageSets', 'train.txt'), 'w') as f:
        f.write('\n'.join(names) + '\n')
    return names
