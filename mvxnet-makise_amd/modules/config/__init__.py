"""Configuration access with the reference's surface (modules/config/__init__.py:1-16):
``import modules.config as cfg; cfg.<key>`` reads ``config[<key>]`` at call time, so
``cfg.config['voxelshape'] = [...]`` re-targets every module, like the reference.

Differences on purpose: ``config.yml`` is looked up in the CWD first (reference behaviour)
and then next to this package; ``sys.argv`` is parsed leniently so importing under pytest or
torchrun does not abort."""
import os

from .Config import config
from .Parser import options, args

dataroot = args[0] if len(args) > 0 else '../mmdetection3d-master/data/kitti'
veloroot = os.path.join(dataroot, 'training/velodyne_croped')
labelroot = os.path.join(dataroot, 'training/label_2')
calibroot = os.path.join(dataroot, 'training/calib')
imroot = os.path.join(dataroot, 'training/image_2')
trainInfoPath = os.path.join(dataroot, 'ImageSets/train.txt')
testInfoPath = os.path.join(dataroot, 'ImageSets/val.txt')


def __getattr__(name):
    try:
        return config[name]
    except KeyError:
        raise AttributeError(name) from None
