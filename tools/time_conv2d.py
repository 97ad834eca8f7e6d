"""Isolated timings of the 2-D convolution entry points (forward, dgrad, wgrad) at the RPN's layer shapes, one stream, nothing
else on the GPU (developer tool).  usage: python tools/time_conv2d.py [frames] [narrow-unit limit, MVX_TUNE_GATHER_NARROW_MAX_UNITS]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NARROW = int(sys.argv[2]) if len(sys.argv) > 2 else None
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip  # noqa: E402
from modules import rpn_frames as rf  # noqa: E402
import modules.config as cfg  # noqa: E402

dev = torch.device('cuda')
_hip.ASYNC_WGRAD = False
if NARROW is not None:
    from modules import Extension as X
    X.check(X.lib.mvx_tuning_set(2, NARROW), 'mvx_tuning_set')
    print('narrow-unit limit', NARROW)
SHAPES = [('blk1 s1 128>128 @176x200', 176, 200, 128, 128, 0), ('blk2 s1 128>128 @88x100', 88, 100, 128, 128, 0),
          ('blk3 s1 256>256 @44x50', 44, 50, 256, 256, 0), ('deconv1 128>256 @176x200', 176, 200, 128, 256, 0),
          ('blk1 s2 512>128 @176x200 (4 taps)', 176, 200, 512, 128, rf.TAPS2), ('blk2 s2 512>128 @88x100 (4 taps)', 88, 100, 512, 128, rf.TAPS2),
          ('blk3 s2 512>256 @44x50 (4 taps)', 44, 50, 512, 256, rf.TAPS2)]


def clock(fn, n=20):
    """Wall clock around n back-to-back calls (the weight-gradient kernels run on the library's side stream, which an
    event pair on the current stream would not see)."""
    import time
    for _ in range(2):
        fn()
    _hip.join_side_stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    _hip.join_side_stream()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('%-36s %8s %8s %8s   (ms | TFLOP/s of the counted taps, F=%d)' % ('layer', 'fwd', 'dgrad', 'wgrad', F))
for name, h, w, cin, cout, flags in SHAPES:
    nt = 4 if flags else 9
    x = torch.randn((F, h, w, cin), device=dev)
    wt = torch.randn((cout, cin, 3, 3), device=dev) * 0.03
    b = torch.zeros((cout,), device=dev)
    wpk = _hip.conv3d_pack(wt, False)
    wpd = _hip.conv3d_pack(wt, True)
    dz = torch.randn((F, h, w, cout), device=dev)
    fl = 2.0 * F * h * w * cin * cout * nt
    t_f = clock(lambda: rf._conv(x, wpk, b, F, h, w, cin, cout, flags, cfg.eps))
    t_d = clock(lambda: rf._dgrad(dz, wpd, F, h, w, cin, cout, flags))
    t_w = clock(lambda: rf._wgrad(x, dz, F, h, w, cin, cout, flags))
    print('%-36s %8.3f %8.3f %8.3f   %6.1f %6.1f %6.1f' % (name, t_f, t_d, t_w, fl / t_f / 1e9, fl / t_d / 1e9, fl / t_w / 1e9))
