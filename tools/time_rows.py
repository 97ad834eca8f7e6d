"""Isolated timing + accuracy of the row GEMMs (forward with ReLU + BatchNorm sums, weight gradient) at the fusion MLP's shapes
in each arithmetic (developer tool).  usage: python tools/time_rows.py [frames] [modes, comma separated]"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
want = sys.argv[2].split(',') if len(sys.argv) > 2 else ['f32', 'bf16x6', 'fp16x3']
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip  # noqa: E402

dev = torch.device('cuda')
CODE = {'f32': 0, 'bf16x3': 2, 'bf16x6': 3, 'fp16x3': 4}
MFMAS = {'f32': 1, 'bf16x3': 3, 'bf16x6': 6, 'fp16x3': 3}


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


g = torch.Generator(device='cpu').manual_seed(0)
rows = 19925 * frames
for K, N in ((768, 768), (768, 128), (128, 768), (128, 128), (1728, 128)):
    x = torch.randn((rows, K), generator=g).to(dev)
    w = (torch.randn((N, K), generator=g) * 0.03).to(dev)
    b = torch.zeros((N,), device=dev)
    dz = torch.randn((rows, N), generator=g).to(dev)
    fl = 2.0 * rows * K * N
    sub = slice(0, 2048)
    ref_y = torch.relu(x[sub].double() @ w.double().t())
    ref_w = dz.double().t() @ x.double()
    for mode in want:
        np_ = CODE[mode]
        _hip.tensor_amax(dz)
        _hip.tensor_amax(x)
        rec = {'layer': '%d -> %d' % (K, N), 'rows': rows, 'mode': mode}
        y, st = _hip.linear_forward(x, w, b, relu=True, want_stats=True, split=np_, foreign=True)
        rec['fwd_vs_f64'] = rel(y[sub], ref_y)
        s_ref = torch.relu(x.double() @ w.double().t()).sum(0)
        rec['bn_sum_vs_f64'] = rel(st.sum(0)[0], s_ref)
        dw = _hip.linear_wgrad(x, dz, split=np_)
        rec['wgrad_vs_f64'] = rel(dw, ref_w)
        t = clock(lambda: _hip.linear_forward(x, w, b, relu=True, want_stats=True, split=np_, foreign=True))
        rec['fwd_ms'], rec['fwd_exec_tflops'] = round(t, 4), round(MFMAS[mode] * fl / t / 1e9, 1)
        t = clock(lambda: _hip.linear_wgrad(x, dz, split=np_))
        rec['wgrad_ms'], rec['wgrad_exec_tflops'] = round(t, 4), round(MFMAS[mode] * fl / t / 1e9, 1)
        print(json.dumps(rec), flush=True)
    del x, dz
