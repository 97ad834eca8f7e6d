"""BatchNorm backward (mvx_bn_relu_backward_frames: reduce + apply passes) and apply alone, per tensor shape of the hot and
full steps, on an otherwise idle GPU: time per call and algorithmic GB/s (5 tensor passes backward, 2 forward).
    python tools/time_bn.py  ->  gpurun_out/bn_timing.json"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
sys.argv = sys.argv[:1]
from modules import _hip  # noqa: E402
from modules import Extension as X  # noqa: E402

dev = torch.device('cuda')
F = 4
shapes = [('fusion L1 rows', 80000, 768), ('fusion L2/L3 rows', 80000, 128), ('fusion L4/L5 rows', 80000, 16),
          ('VFE1 rows', 100000, 16), ('VFE2 rows', 100000, 64), ('FCN rows', 100000, 128),
          ('CML conv3 out grid', F * 2 * 352 * 400, 64),
          ('RPN blk1 176x200x128', F * 176 * 200, 128), ('RPN deconv1 176x200x256', F * 176 * 200, 256),
          ('RPN blk2 88x100x128', F * 88 * 100, 128), ('RPN blk3 44x50x256', F * 44 * 50, 256),
          ('RPN deconv2 rows 88x100x4x256', F * 88 * 100 * 4, 256), ('RPN deconv3 rows 44x50x16x256', F * 44 * 50 * 16, 256)]
desc = X.FramesDesc.make([0] * (F + 1), [0] * (F + 1), 1)
out = []
for name, rows, C in shapes:
    g = torch.randn((rows, C), device=dev)
    y = torch.randn((rows, C), device=dev)
    mi = torch.stack([torch.zeros(F, C), torch.ones(F, C)], 1).to(dev).contiguous()
    dz = torch.empty_like(y)
    db = torch.zeros(C, device=dev)
    nscr = X.lib.mvx_bn_backward_scratch_bytes_frames(C, F) // 8
    scratch = torch.zeros(nscr, dtype=torch.float64, device=dev)
    o = torch.empty_like(y)

    def bwd():
        scratch.zero_()
        X.check(X.lib.mvx_bn_relu_backward_frames(X.ptr(g), X.ptr(y), X.ptr(mi), 1.0, X.ptr(dz), X.ptr(db), X.ptr(scratch), None,
                                                  rows, C, _hip.FLAG_ACCUMULATE | _hip.FLAG_PREZEROED, desc.ref(), X.ROWS_GRID,
                                                  None, X.stream()), 'bn_bwd')

    def fwd():
        X.check(X.lib.mvx_bn_apply_frames(X.ptr(y), X.ptr(mi), X.ptr(o), rows, C, desc.ref(), X.ROWS_GRID, X.stream()), 'bn_apply')
    rec = {'shape': name, 'rows': rows, 'channels': C, 'tensor_MB': rows * C * 4 / 1e6}
    for tag, fn, passes in (('backward', bwd, 5), ('apply', fwd, 2)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        rec[tag + '_ms'] = ms
        rec[tag + '_GBps'] = passes * rows * C * 4 / (ms * 1e-3) / 1e9
    out.append(rec)
    print('%-34s %7.1f MB  backward %.3f ms (%5.0f GB/s)  apply %.3f ms (%5.0f GB/s)'
          % (name, rec['tensor_MB'], rec['backward_ms'], rec['backward_GBps'], rec['apply_ms'], rec['apply_GBps']))
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(REPO, 'gpurun_out', 'bn_timing.json'), 'w'), indent=1)
