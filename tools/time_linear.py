"""Isolated timing of the row GEMM (mvx_linear_forward with ReLU + BatchNorm statistics) and its weight gradient at the
fusion MLP's layer shapes (developer tool).  usage: python tools/time_linear.py [rows]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 79700
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip  # noqa: E402

dev = torch.device('cuda')


def clock(fn, n=20):
    for _ in range(3):
        fn()
    _hip.join_side_stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    _hip.join_side_stream()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('%-22s %9s %9s %9s   TFLOP/s' % ('layer (rows=%d)' % rows, 'fwd ms', 'dgrad ms', 'wgrad ms'))
for K, N in ((768, 768), (768, 128), (128, 128), (128, 16), (23, 32), (64, 128)):
    x = torch.randn((rows, K), device=dev)
    w = torch.randn((N, K), device=dev) * 0.03
    b = torch.zeros((N,), device=dev)
    dz = torch.randn((rows, N), device=dev)
    fl = 2.0 * rows * K * N
    t_f = clock(lambda: _hip.linear_forward(x, w, b, relu=True, want_stats=True))
    t_d = clock(lambda: _hip.linear_forward(dz, w, None, relu=False, want_stats=False, w_transposed=True))
    t_w = clock(lambda: _hip.linear_wgrad(x, dz))
    print('%4d -> %-14d %9.3f %9.3f %9.3f   %6.1f %6.1f %6.1f' % (K, N, t_f, t_d, t_w, fl / t_f / 1e9, fl / t_d / 1e9, fl / t_w / 1e9))
