"""Isolated timing of the MFMA kernels in the three arithmetics -- exact f32 (v_mfma_f32_32x32x2_f32), bf16x3 (two bf16 pieces,
three MFMAs per product) and bf16x6 (three pieces, six MFMAs, fp32-grade) -- at the shapes of the hot step: the CML's conv2 /
conv3 forward, input gradient and weight gradient on a dense 4-frame grid and the fusion MLP's row GEMMs (developer tool).
Also prints each result's distance from a float64 evaluation of a sub-block, so that speed and accuracy stand side by side.

usage: python tools/time_split.py [frames]      -> one JSON document on stdout"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip  # noqa: E402
from modules import Extension as X  # noqa: E402

dev = torch.device('cuda')
# 'fp16x3' runs with the weights scaled by WS = 2^10 (and the result scaled back): the low fp16 piece of a 0.04-sized weight is
# subnormal otherwise; 'fp16x3 unscaled' shows what that costs
MODES = (('f32', 0, 1.0), ('bf16x3', 2, 1.0), ('bf16x6', 3, 1.0), ('fp16x3', 4, 1024.0), ('fp16x3 unscaled', 4, 1.0))


def clock(fn, n=10):
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


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


out = {'frames': frames, 'conv': [], 'rows': []}
H, W = 352, 400
g = torch.Generator(device='cpu').manual_seed(0)
for name, cin, cout, din, sd, pd in (('conv2', 64, 64, 5, 1, 0), ('conv3', 64, 64, 3, 2, 1), ('rpn 128->128 @176x200', 128, 128, 1, 1, 1)):
    if name.startswith('rpn'):
        h, w_ = 176, 200
    else:
        h, w_ = H, W
    dout = _hip.conv_out_depth(din, sd, pd)
    # frames stacked along depth are only available through the *_frames entries; a dense single "frame" with frames x planes
    # would connect planes across frames, so time ONE frame's launch `frames` times larger in-plane instead: use depth as is
    x = torch.randn((din, h, w_, cin), generator=g).to(dev)
    wt = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.04).to(dev)
    b = torch.zeros(cout, device=dev)
    dz = torch.randn((dout, h, w_, cout), generator=g).to(dev)
    fl_f = _hip.conv_flops(dout, din, h, w_, cin, cout, sd, pd)
    fl_d = _hip.conv_flops(din, dout, h, w_, cout, cin, sd, pd, True)
    ref = {}
    # float64 forward of a 24 x 24 corner (CPU)
    CR = 24
    xc = x[:, :CR + 1, :CR + 1].permute(3, 0, 1, 2)[None].double().cpu()
    y64 = torch.nn.functional.conv3d(xc, wt.double().cpu(), None, (sd, 1, 1), (pd, 1, 1))[0].permute(1, 2, 3, 0)[:, :CR, :CR]
    for mode, np_, ws in MODES:
        wf, wd = _hip.conv3d_pack(wt * ws, False, split=np_), _hip.conv3d_pack(wt * ws, True, split=np_)
        rec = {'layer': name, 'mode': mode}
        y, _ = _hip.conv3d_forward(x, wf, b, cout, sd, pd, relu=False, want_stats=False, split=np_)
        dx = _hip.conv3d_dgrad(dz, wd, din, cin, sd, pd, split=np_)
        y, dx = y / ws, dx / ws
        rec['fwd_vs_f64'] = rel(y[:, :CR, :CR].cpu(), y64)
        dw = _hip.conv3d_wgrad(x, dz, sd, pd, split=np_) if cout == 64 else None
        if mode == 'f32':
            ref = {'y': y.clone(), 'dx': dx.clone(), 'dw': dw.clone() if dw is not None else None}
        else:
            rec['fwd_vs_f32'] = rel(y, ref['y'])
            rec['dgrad_vs_f32'] = rel(dx, ref['dx'])
            if dw is not None:
                rec['wgrad_vs_f32'] = rel(dw, ref['dw'])
        t = clock(lambda: _hip.conv3d_forward(x, wf, b, cout, sd, pd, relu=True, want_stats=True, split=np_))
        rec['fwd_ms'], rec['fwd_tflops'] = t, fl_f / t / 1e9
        t = clock(lambda: _hip.conv3d_dgrad(dz, wd, din, cin, sd, pd, split=np_))
        rec['dgrad_ms'], rec['dgrad_tflops'] = t, fl_d / t / 1e9
        if cout == 64:
            t = clock(lambda: _hip.conv3d_wgrad(x, dz, sd, pd, split=np_))
            rec['wgrad_ms'], rec['wgrad_tflops'] = t, fl_f / t / 1e9
        out['conv'].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
    del x, dz

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
    for mode, np_, ws in MODES:
        rec = {'layer': '%d -> %d' % (K, N), 'rows': rows, 'mode': mode}
        y, _ = _hip.linear_forward(x, w * ws, b, relu=True, want_stats=True, split=np_)
        rec['fwd_vs_f64'] = rel(y[sub] / ws, ref_y)
        dw = _hip.linear_wgrad(x, dz, split=np_)
        rec['wgrad_vs_f64'] = rel(dw, ref_w)
        t = clock(lambda: _hip.linear_forward(x, w, b, relu=True, want_stats=True, split=np_))
        rec['fwd_ms'], rec['fwd_tflops'] = t, fl / t / 1e9
        t = clock(lambda: _hip.linear_wgrad(x, dz, split=np_))
        rec['wgrad_ms'], rec['wgrad_tflops'] = t, fl / t / 1e9
        out['rows'].append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
    del x, dz
print(json.dumps(out))
