"""Loads config.yml and derives voxelsize / eps / dtype (reference: modules/config/Config.py:4-13)."""
import os

import torch
import yaml

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _find():
    for cand in ('./config.yml', os.path.join(_PKG_ROOT, 'config.yml')):
        if os.path.exists(cand):
            return cand
    raise FileNotFoundError('config.yml not found in the CWD or in %s' % _PKG_ROOT)


with open(_find(), 'r') as _f:
    config = yaml.load(_f, yaml.Loader)

r, s = config['velorange'], config['voxelshape']
# python-float (f64) true division: the voxel size must be exactly the reference's doubles
config['voxelsize'] = [(r[k + 3] - r[k]) / s[k] for k in range(3)]
config['eps'] = 1e-3 if config['half'] else 1e-6
config['dtype'] = torch.float16 if config['half'] else torch.float32
# MVX_CONVMATH overrides `convmath` of config.yml (f32 | fp16x3 | bf16x6 | bf16x3): lets one test / bench run be repeated in another
# arithmetic without editing the file
if os.environ.get('MVX_CONVMATH'):
    config['convmath'] = os.environ['MVX_CONVMATH']
