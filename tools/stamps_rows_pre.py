"""s_memtime stamps of one workgroup of rowgemm_fwd_pre (bf16x6, ping-pong loop): where an interval's cycles go (developer tool)."""
import ctypes
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip  # noqa: E402
from modules import Extension as X  # noqa: E402
dev = torch.device('cuda')
R, K, N = 79700, 768, 768
flags = _hip.split_flags(3, True)
x = torch.randn((R, K), device=dev)
w = torch.randn((N, K), device=dev) * 0.03


def planes_of(t):
    rows, k = t.shape
    p = torch.empty((X.lib.mvx_split_planes_bytes(rows, k, flags) // 2,), dtype=torch.int16, device=t.device)
    X.check(X.lib.mvx_split_rows(X.ptr(t), k, rows, k, X.ptr(p), flags, 1.0, X.stream()), 'mvx_split_rows')
    return p


xp, wp = planes_of(x), planes_of(w)
y = torch.empty((R, N), device=dev)
fn = ctypes.CDLL(os.path.join(REPO, 'mvxnet-makise_amd', 'lib', 'libmvx_hip.so')).mvx_debug_rowgemm_fwd_pre_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
stats = torch.zeros((_hip.STATS_REPLICAS, 2, N), dtype=torch.float64, device=dev)
for blk in (8, 500, 900):
    st = torch.zeros((8, 512), dtype=torch.int64, device=dev)
    for _ in range(3):
        assert fn(xp.data_ptr(), wp.data_ptr(), y.data_ptr(), R, K, N, st.data_ptr(), blk, X.stream(), stats.data_ptr()) == 0
    torch.cuda.synchronize()
    t = st.cpu().numpy().astype(np.int64)
    NS, U = 6, K // 16
    print('block', blk, 'cycles per wave, first stamp -> last:', (t[:, NS * U - 1] - t[:, 0]).tolist(), ' ideal (U + 1) x 48 x 32 =', (U + 1) * 1536)
    x = t[0, 480:490]
    print('  wave 0: entry -> loop done %d cyc, bias+relu %d, stats %d, stores issued %d; whole kernel %d cyc in %.2f us = %.3f GHz' % (x[2] - x[0], x[4] - x[2], x[6] - x[4], x[8] - x[6], x[8] - x[0], (x[9] - x[1]) / 100.0, (x[8] - x[0]) / ((x[9] - x[1]) * 10.0)))
    names = ['reads', 'dma(+g1 vmcnt)', 'bar1', 'lgkm', 'mfma(+g0 vmcnt)', 'bar2']
    for wv in range(8):
        seg = np.zeros(NS)
        for u in range(4, U - 4):
            base = NS * u
            for k in range(NS):
                seg[k] += t[wv, base + k] - t[wv, base + k - 1]
        seg /= (U - 8)
        print(' wave %d (group %d): ' % (wv, wv >> 2) + ', '.join('%s %.0f' % (n, v) for n, v in zip(names, seg)) + '  | per k-step %.0f' % seg.sum())
