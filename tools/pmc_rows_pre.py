"""fcn1-shaped row GEMMs (79,700 x 768 -> 768) on pre-cut operands for `rocprofv3 --pmc` passes -- developer tool.
usage: python tools/pmc_rows_pre.py [bf16x6|fp16x3]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip  # noqa: E402
from modules import Extension as X  # noqa: E402
dev = torch.device('cuda')
R, K, N = 79700, 768, 768
code = {'bf16x6': 3, 'fp16x3': 4}[mode]
flags = _hip.split_flags(code, True)
x = torch.randn((R, K), device=dev)
w = torch.randn((N, K), device=dev) * 0.03
dz = torch.randn((R, N), device=dev)
b = torch.zeros(N, device=dev)


def planes_of(t, scale=1.0):
    rows, k = t.shape
    p = torch.empty((X.lib.mvx_split_planes_bytes(rows, k, flags) // 2,), dtype=torch.int16, device=t.device)
    X.check(X.lib.mvx_split_rows(X.ptr(t), k, rows, k, X.ptr(p), flags, scale, X.stream()), 'mvx_split_rows')
    return p


xp, wp, zp = planes_of(x), planes_of(w, 256.0 if code == 4 else 1.0), planes_of(dz)
y = torch.empty((R, N), device=dev)
st = torch.zeros((_hip.STATS_REPLICAS, 2, N), dtype=torch.float64, device=dev)
ws_b = X.lib.mvx_linear_wgrad_pre_workspace_bytes(R, K, N)
wsp = torch.empty((ws_b // 4,), device=dev)
dw = torch.empty((N, K), device=dev)
for _ in range(3):
    X.check(X.lib.mvx_linear_forward_pre_frames(X.ptr(xp), X.ptr(wp), X.ptr(b), X.ptr(y), N, X.ptr(st), None, R, K, N,
                                                flags | _hip.FLAG_RELU, 1.0, None, 0.0, None, None, 0, X.stream()), 'fwd')
    X.check(X.lib.mvx_linear_wgrad_pre(X.ptr(xp), X.ptr(zp), X.ptr(dw), R, K, N, flags, 1.0, X.ptr(wsp), ws_b, X.stream()), 'wgrad')
    _hip.linear_forward(x, w, b, relu=True, want_stats=True, split=code)
    _hip.linear_wgrad(x, dz, split=code)
torch.cuda.synchronize()
