"""Times the K = 128 row layers on both kernels (developer tool, GPU box): python tools/time_k128.py [rows]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'mvxnet-makise_amd'))
import torch
from modules import _hip
from modules import Extension as X

rows_list = [int(a) for a in sys.argv[1:]] or [100000, 320000, 80000]
dev = 'cuda'
for rows in rows_list:
    for N in (128, 768):
        x = torch.randn((rows, 128), device=dev)
        w = torch.randn((N, 128), device=dev) * 0.05
        b = torch.randn((N,), device=dev) * 0.1
        y = torch.empty((rows, N), device=dev)
        st = torch.zeros((1, _hip.STATS_REPLICAS, 2, N), dtype=torch.float64, device=dev)
        cnt = torch.zeros((1,), dtype=torch.float64, device=dev)
        mi = torch.empty((1, 2, N), device=dev)
        desc = X.FramesDesc.make([0, 100], [0, rows], 35)
        for split in (3, 4):
            flags = _hip.split_flags(split, True) | _hip.FLAG_RELU
            for on in (0, 1):
                X.check(X.lib.mvx_tuning_set(3, on), 'tune')
                for stats in (True, False):
                    def go():
                        if stats:
                            st.zero_()
                            X.check(X.lib.mvx_linear_forward_bn_frames(X.ptr(x), 128, X.ptr(w), 128, 0, X.ptr(b), X.ptr(y), N, X.ptr(st), None,
                                                                       rows, 128, N, flags, X.ptr(cnt), 1e-6, X.ptr(mi), desc.ref(), X.ROWS_REAL,
                                                                       X.stream()), 'fwd')
                        else:
                            X.check(X.lib.mvx_linear_forward(X.ptr(x), 128, X.ptr(w), 128, 0, None, X.ptr(y), N, None, None, rows, 128, N,
                                                             flags & ~_hip.FLAG_RELU, None, 0, X.stream()), 'fwd')
                    for _ in range(3):
                        go()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        go()
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 20
                    fl = 2.0 * rows * 128 * N * (6 if split == 3 else 3)
                    print('rows %7d N %3d split %d k128 %d stats %d: %.4f ms  %.0f TFLOP/s executed  %.2f TB/s' % (
                        rows, N, split, on, stats, ms, fl / ms / 1e9, (rows * 128 * 4 + rows * N * 4) / ms / 1e9), flush=True)
X.check(X.lib.mvx_tuning_set(3, 1), 'tune')
