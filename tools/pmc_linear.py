"""fcn1-shaped row GEMM launches (20,000 x 768 -> 768) for `rocprofv3 --pmc` passes / timing -- developer tool."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
from modules import _hip
dev = torch.device('cuda')
R, K, N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 768, 768
x = torch.randn((R, K), device=dev)
w = torch.randn((N, K), device=dev) * 0.03
b = torch.zeros(N, device=dev)
dz = torch.randn((R, N), device=dev)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
fl = 2.0 * R * K * N
for name, fn in (('fwd', lambda: _hip.linear_forward(x, w, b, relu=True, want_stats=True)),
                 ('dgrad', lambda: _hip.linear_forward(dz, w, None, relu=False, want_stats=False, w_transposed=True)),
                 ('wgrad', lambda: _hip.linear_wgrad(x, dz))):
    ms = t(fn)
    print('%-6s %.3f ms  %.1f TFLOP/s' % (name, ms, fl / ms / 1e9))
