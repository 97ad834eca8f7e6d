"""Frames/s of the training loop itself (train_like.py --mode fast, the train.py-shaped driver) on a synthetic KITTI tree,
file I/O included: createDataset (velodyne .bin + image .png + label + calib per frame, read into RAM like the reference,
train.py:53-57) and then one epoch of B-frame steps, with the prefetch thread (modules/data/Prefetch.py) and with every
batch prepared inside its step.  On the GPU box:  python tools/time_train_loop.py [frames] [frames_per_step]
-> gpurun_out/train_loop_timing.json"""
import json
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sys.argv = sys.argv[:1]
import numpy as np  # noqa: E402
import train_like  # noqa: E402

root = tempfile.mkdtemp(prefix='mvx_kitti_')
res = {'frames': n, 'frames_per_step': B, 'points': 20000}
first = True
for label, extra in (('prefetch', ['--prefetch']), ('prefetch_prio0', ['--prefetch', '--prefetch-priority', '0']),
                     ('prefetch_depth4', ['--prefetch', '--prefetch-depth', '4']), ('no_prefetch', [])):
    args = train_like.parse_args([root, '-n', '2', '--mode', 'fast', '--frames', str(B), '--points', '20000', '--quiet',
                                  '--checkpoints', os.path.join(root, 'ck_' + label)] + extra + (['--synthetic', str(n)] if first else []))
    first = False
    np.random.seed(0)
    r = train_like.train(args)
    res[label] = {'createDataset_s': r['dataset_s'], 'createDataset_frames_per_s': r['dataset_frames'] / r['dataset_s'],
                  'epochs': r['loop'],                                  # epoch 1 includes warm-up (allocator, first launches)
                  'loop_frames_per_s_epoch2': r['loop'][-1]['frames_per_s'],
                  'end_to_end_frames_per_s_epoch2': r['loop'][-1]['frames'] / (r['loop'][-1]['seconds'] + r['dataset_s'])}
print(json.dumps(res))
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
with open(os.path.join(REPO, 'gpurun_out', 'train_loop_timing.json'), 'w') as fh:
    json.dump(res, fh, indent=1)
