"""Times the conv3d kernels at the full KITTI grid (10x352x400) -- developer tool."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip

dev = torch.device('cuda')
H, W = 352, 400
LAYERS = [(128, 64, 10, 2, 1), (64, 64, 5, 1, 0), (64, 64, 3, 2, 1)]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


tot = 0.0
for cin, cout, din, sd, pd in LAYERS:
    dout = _hip.conv_out_depth(din, sd, pd)
    x = torch.randn((din, H, W, cin), device=dev)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.02
    b = torch.zeros(cout, device=dev)
    wpk = _hip.conv3d_pack(w, False)
    wpd = _hip.conv3d_pack(w, True)
    dz = torch.randn((dout, H, W, cout), device=dev)
    flop = 2.0 * dout * H * W * 27 * cin * cout
    t_f = timeit(lambda: _hip.conv3d_forward(x, wpk, b, cout, sd, pd))
    t_d = timeit(lambda: _hip.conv3d_dgrad(dz, wpd, din, cin, sd, pd))
    t_w = timeit(lambda: _hip.conv3d_wgrad(x, dz, sd, pd))
    ws = _hip.conv3d_pack(w, False, split=True); wsd = _hip.conv3d_pack(w, True, split=True)
    t_fs = timeit(lambda: _hip.conv3d_forward(x, ws, b, cout, sd, pd, split=True))
    t_ds = timeit(lambda: _hip.conv3d_dgrad(dz, wsd, din, cin, sd, pd, split=True))
    t_ws = timeit(lambda: _hip.conv3d_wgrad(x, dz, sd, pd, split=True))
    print('   bf16x3: fwd %.2f ms (%.1f TF eff)  dgrad %.2f ms (%.1f TF eff)  wgrad %.2f ms (%.1f TF eff)' % (t_fs, flop / t_fs / 1e9, t_ds, flop / t_ds / 1e9, t_ws, flop / t_ws / 1e9))
    tot += t_f + t_d + t_w
    print('conv %3d->%2d D%2d: fwd %.2f ms (%.1f TF)  dgrad %.2f ms (%.1f TF)  wgrad %.2f ms (%.1f TF)' % (
        cin, cout, din, t_f, flop / t_f / 1e9, t_d, flop / t_d / 1e9, t_w, flop / t_w / 1e9))
print('total fwd+bwd %.2f ms' % tot)
