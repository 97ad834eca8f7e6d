"""KITTI reader with the reference's names and return conventions (modules/data/Load.py:24-95).

``createDataset`` returns, per frame, ``(velo (N,4) f32 numpy, img (h,w,3) uint8 BGR numpy, bbox2d, bbox3d, bev, calib)``
with the boxes of class 'Car' in LiDAR coordinates (``None`` x3 when the frame has none in range) and ``calib`` a dict of
4x4 float32 tensors -- exactly what ``train.py:26-49`` consumes.  Differences from the reference, all on purpose:
  * images are decoded with PIL (OpenCV is not part of this stack) and converted to OpenCV's BGR channel order;
  * ``needCrop=True`` runs crop + cropToSight on the GPU (modules/data/Preprocessing.py);
  * the data root is an argument (the reference reads it from ``sys.argv`` at import time).
"""
import os
from typing import Dict, List, Tuple

import numpy as np
import torch

import modules.config as cfg
import modules.data.Preprocessing as pre
from modules import Calc

dataroot = getattr(cfg, 'dataroot', '../mmdetection3d-master/data/kitti')


def _roots(root):
    t = os.path.join(root, 'training')
    return (os.path.join(t, 'velodyne_croped'), os.path.join(t, 'velodyne'), os.path.join(t, 'label_2'),
            os.path.join(t, 'calib'), os.path.join(t, 'image_2'))


def readCalib(path) -> dict:
    """Lines 5 / 2 / 4 of a KITTI calib file -> 4x4 ``Tr_velo_to_cam``, ``P2``, ``R0_rect`` (values parsed as float32,
    stored in float64 arrays like the reference's np.concatenate / np.zeros results; Load.py:24-41)."""
    calib = {}
    with open(path, 'r') as f:
        lines = f.read().splitlines()
    for row, shape in ((5, (3, 4)), (2, (3, 4)), (4, (3, 3))):
        tok = lines[row].split(' ')
        m = np.zeros((4, 4))
        m[:shape[0], :shape[1]] = np.array(tok[1:]).astype('float32').reshape(shape)
        m[3, 3] = 1
        calib[tok[0][:-1]] = m
    return calib


def _read_image(path):
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])          # cv2.imread order (BGR)


def _read_car_labels(path):
    """Columns 4..14 of the 'Car' rows of a label_2 file: bbox2d (4) + h w l x y z ry (Load.py:66-67)."""
    rows = []
    with open(path, 'r') as f:
        for line in f:
            tok = line.split(' ')
            if tok and tok[0] == 'Car':
                rows.append([float(v) for v in tok[4:15]])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 11)


def createDataset(splitSet: List[str], needCrop=False, root=None) -> \
        List[Tuple[np.ndarray, np.ndarray, torch.Tensor, torch.Tensor, torch.Tensor, Dict[str, torch.Tensor]]]:
    """Read KITTI frames ``splitSet`` (e.g. ['000000', ...]) from ``root`` (reference Load.py:43-95)."""
    root = dataroot if root is None else root
    veloroot, rawroot, labelroot, calibroot, imroot = _roots(root)
    range_min, range_max = torch.Tensor(cfg.velorange[:3]), torch.Tensor(cfg.velorange[3:])
    imsize = cfg.imsize[::-1]                              # (w, h)
    dataset = []
    for s in splitSet:
        velo_path = os.path.join(rawroot if needCrop else veloroot, s + '.bin')
        velo = np.fromfile(velo_path, dtype='float32').reshape((-1, 4))
        if needCrop:
            velo = pre.crop(velo, cfg.velorange)
        img = _read_image(os.path.join(imroot, s + '.png'))[:imsize[1], :imsize[0]]
        labels = _read_car_labels(os.path.join(labelroot, s + '.txt'))
        calib = readCalib(os.path.join(calibroot, s + '.txt'))
        if needCrop:
            velo = pre.cropToSight(velo, calib, imsize)
        calib = {k: torch.Tensor(v) for k, v in calib.items()}
        if len(labels) == 0:
            dataset.append((velo, img, None, None, None, calib))
            continue
        c2v = torch.linalg.inv(calib['Tr_velo_to_cam'])
        labels = torch.Tensor(labels)
        labels[:, 4:] = Calc.bboxCam2Lidar(labels[:, 4:], c2v, True)
        in_range = torch.all(labels[:, 4:7] < range_max[None], dim=1) & torch.all(labels[:, 4:7] >= range_min[None], dim=1)
        labels = labels[in_range].contiguous()
        if len(labels) == 0:
            dataset.append((velo, img, None, None, None, calib))
            continue
        dataset.append((velo, img, labels[:, :4], labels[:, 4:], Calc.bbox3d2bev(labels[:, 4:]), calib))
    return dataset
