"""Row GEMMs on pre-cut operands (csrc/rowgemm_pre.hip) against the in-kernel-cut kernels (csrc/linear_split.hip) and float64:
accuracy, bit-equality of the forward, isolated timing (developer tool).  usage: python tools/time_rows_pre.py [frames] [modes]"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
want = sys.argv[2].split(',') if len(sys.argv) > 2 else ['bf16x6']
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip  # noqa: E402
from modules import Extension as X  # noqa: E402

dev = torch.device('cuda')
CODE = {'bf16x6': 3, 'fp16x3': 4}
MFMAS = {'bf16x6': 6, 'fp16x3': 3}


def clock(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def planes_of(t, flags, scale=1.0):
    rows, k = t.shape
    p = torch.empty((X.lib.mvx_split_planes_bytes(rows, k, flags) // 2,), dtype=torch.int16, device=t.device)
    X.check(X.lib.mvx_split_rows(X.ptr(t), k, rows, k, X.ptr(p), flags, scale, X.stream()), 'mvx_split_rows')
    return p


g = torch.Generator(device='cpu').manual_seed(0)
rows = 19925 * frames
for K, N in ((768, 768), (128, 768), (768, 256)):
    x = torch.randn((rows, K), generator=g).to(dev)
    w = (torch.randn((N, K), generator=g) * 0.03).to(dev)
    b = (torch.randn((N,), generator=g) * 0.1).to(dev)
    dz = torch.randn((rows, N), generator=g).to(dev)
    fl = 2.0 * rows * K * N
    sub = slice(0, 2048)
    ref_y = torch.relu(x[sub].double() @ w.double().t() + b.double())
    ref_w = dz.double().t() @ x.double()
    for mode in want:
        code = CODE[mode]
        flags = _hip.split_flags(code, True)
        rec = {'layer': '%d -> %d' % (K, N), 'rows': rows, 'mode': mode}
        xs, ws, zs = 1.0, (256.0 if code == 4 else 1.0), 1.0
        xp, wp, zp = planes_of(x, flags, xs), planes_of(w, flags, ws), planes_of(dz, flags, zs)
        t_split = clock(lambda: planes_of(x, flags, xs))
        rec['split_x_ms'] = round(t_split, 4)
        y = torch.empty((rows, N), device=dev)
        st = torch.zeros((_hip.STATS_REPLICAS, 2, N), dtype=torch.float64, device=dev)

        def fwd(extra=0):
            X.check(X.lib.mvx_linear_forward_pre_frames(X.ptr(xp), X.ptr(wp), X.ptr(b), X.ptr(y), N, X.ptr(st), None, rows, K, N,
                                                        flags | _hip.FLAG_RELU | extra, 1.0 / (xs * ws), None, 0.0, None, None, 0, X.stream()),
                    'mvx_linear_forward_pre_frames')
        fwd()
        y_old, st_old = _hip.linear_forward(x, w, b, relu=True, want_stats=True, split=code)
        rec['fwd_vs_f64'] = rel(y[sub], ref_y)
        rec['fwd_equals_in_kernel_cut'] = bool(torch.equal(y, y_old))
        rec['fwd_max_diff_vs_in_kernel_cut'] = float((y - y_old).abs().max())
        rec['bn_sum_vs_old'] = rel(st.sum(0), st_old.sum(0))
        t = clock(fwd)
        rec['fwd_ms'], rec['fwd_exec_tflops'] = round(t, 4), round(MFMAS[mode] * fl / t / 1e9, 1)
        t = clock(lambda: _hip.linear_forward(x, w, b, relu=True, want_stats=True, split=code))
        rec['fwd_old_ms'] = round(t, 4)
        if K % 256 == 0 and N % 256 == 0:
            ws_b = X.lib.mvx_linear_wgrad_pre_workspace_bytes(rows, K, N)
            wsp = torch.empty((ws_b // 4,), device=dev)
            dw = torch.empty((N, K), device=dev)
            for order in (0, 4096):
                def wg():
                    X.check(X.lib.mvx_linear_wgrad_pre(X.ptr(xp), X.ptr(zp), X.ptr(dw), rows, K, N, flags | order, 1.0 / (xs * zs),
                                                       X.ptr(wsp), ws_b, X.stream()), 'mvx_linear_wgrad_pre')
                wg()
                rec['wgrad_vs_f64_order%d' % (order // 4096)] = rel(dw, ref_w)
                t = clock(wg)
                rec['wgrad_ms_order%d' % (order // 4096)] = round(t, 4)
                rec['wgrad_exec_tflops_order%d' % (order // 4096)] = round(MFMAS[mode] * fl / t / 1e9, 1)
            t = clock(lambda: _hip.linear_wgrad(x, dz, split=code))
            rec['wgrad_old_ms'] = round(t, 4)
        print(json.dumps(rec), flush=True)
    del x, dz
