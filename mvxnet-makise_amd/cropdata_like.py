"""The reference's one-off crop pass (cropdata.py:21-67) on this package's kernels: for every frame of a KITTI tree read
``training/velodyne/<id>.bin`` and ``training/calib/<id>.txt``, keep the points inside ``velorange`` and inside the camera
frustum, write ``training/velodyne_croped/<id>.bin``.

    python cropdata_like.py <dataroot> [numpy|torch] [--synthetic N] [--frames K]

mode ``numpy`` (default, cropdata.py:30-31,64): float64 bounds and float64 projection arithmetic -- bit-identical output to
the reference's numpy path; mode ``torch`` (:32-34): float32.  Both run on the GPU (crop + cropToSight fused in one
compaction pass, modules/data/Preprocessing.py cropFused); ``torch-cuda`` is accepted as an alias of ``torch``.
BASELINE.json config 1 is this script on 8 synthetic frames (``--synthetic 8``).
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('dataroot')
    ap.add_argument('mode', nargs='?', default='numpy', choices=['numpy', 'torch', 'torch-cuda'])
    ap.add_argument('--synthetic', type=int, default=0, help='write a synthetic KITTI tree with this many frames first')
    ap.add_argument('--frames', type=int, default=0, help='process only the first K frames of training/velodyne')
    ap.add_argument('--quiet', action='store_true')
    a = ap.parse_args(argv)
    sys.argv = sys.argv[:1]              # modules.config parses argv at import (reference modules/config/Parser.py:12)
    import torch
    import modules.config as cfg
    from modules.data import Load, Preprocessing as pre
    if a.synthetic:
        from modules.data import Synthetic
        Synthetic.write_kitti_tree(a.dataroot, list(range(a.synthetic)))
    velo_dir = os.path.join(a.dataroot, 'training/velodyne')
    calib_dir = os.path.join(a.dataroot, 'training/calib')
    out_dir = os.path.join(a.dataroot, 'training/velodyne_croped')
    os.makedirs(out_dir, exist_ok=True)
    names = sorted(f[:-4] for f in os.listdir(velo_dir) if f.endswith('.bin'))
    if a.frames:
        names = names[:a.frames]
    imsize = cfg.imsize[::-1]                                       # (w, h): cropdata.py:19
    t0, kept = time.perf_counter(), 0
    for i, s in enumerate(names):
        velo = np.fromfile(os.path.join(velo_dir, s + '.bin'), dtype='float32').reshape((-1, 4))
        calib = Load.readCalib(os.path.join(calib_dir, s + '.txt'))
        if a.mode == 'numpy':
            out = pre.cropFused(velo, cfg.velorange, calib, imsize)                  # f64 arithmetic, numpy in / numpy out
        else:
            ct = {k: torch.Tensor(v).cuda() for k, v in calib.items()}
            out = pre.cropFused(torch.Tensor(velo).cuda(), cfg.velorange, ct, imsize).cpu().numpy()
        out.tofile(os.path.join(out_dir, s + '.bin'))
        kept += out.shape[0]
        if not a.quiet:
            print('\rProcessing: %d/%d' % (i + 1, len(names)), end='')
    dt = time.perf_counter() - t0
    if not a.quiet:
        print('\n%d frames, %d points kept, %.3f s (%.1f frames/s incl. file I/O)' % (len(names), kept, dt, len(names) / max(dt, 1e-9)))
    return len(names), kept, dt


if __name__ == '__main__':
    main()
